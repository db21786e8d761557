"""ctypes binding of libgort_amd.so (include/gort_amd.h).

Python here is plumbing for tests and bench.py: the product is the C-ABI shared library
and the `gortt` executable.  Device buffers come from torch (`tensor.data_ptr()`); no
numerics happen in Python.  Importing this module never touches the oracle.
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GORT_AMD_LIB") or os.path.join(PKG, "libgort_amd.so")     # override: A/B builds
GORTT_BIN = os.path.join(PKG, "bin", "gortt")
D = C.c_double
NTH, NLAYERS, NBANDS, COEF_STRIDE = 91, 15, 2101, 16

OK, EINVAL, ERANGE, ENODEVICE, EIO, ENOMEM = 0, -1, -2, -3, -4, -5


class GortError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gort_amd error %d: %s" % (code, msg))
        self.code = code


class Canopy(C.Structure):
    """struct gort_canopy"""
    _fields_ = [
        ("r", D), ("b", D), ("h1", D), ("h2", D), ("lambda_", D), ("favd", D),
        ("beta", D), ("use_user_beta", C.c_int32), ("use_user_fd", C.c_int32),
        ("fd_user", D), ("use_q08", C.c_int32), ("reserved0", C.c_int32),
        ("ell", D), ("rr", D), ("rrr", D), ("h", D), ("k", D), ("elai", D), ("tau", D),
        ("z1", D), ("z2", D), ("lv", D),
        ("favd_p", D), ("tau_p", D), ("lv_p", D), ("z1_p", D), ("z2_p", D), ("h1_p", D), ("h2_p", D),
        ("dz", D), ("ds", D), ("dz_p", D), ("dth", D),
        ("height_p", D * NLAYERS), ("theta", D * NTH), ("theta_p", D * NTH),
        ("p_n0", D * NTH), ("epgap", D * NTH), ("k_open", D), ("k_openep", D),
    ]


class LeafSoil(C.Structure):
    """struct gort_leaf_soil"""
    _fields_ = [
        ("N", D), ("Cab", D), ("Car", D), ("Anth", D), ("Cbrown", D), ("Cw", D), ("Cm", D),
        ("rsl", D * 4), ("use_alb_leaf", C.c_int32), ("use_alb_soil", C.c_int32),
        ("alb_leaf", D), ("alb_soil", D),
    ]


class Grid(C.Structure):
    """struct gort_grid"""
    _fields_ = [
        ("sza0", D), ("dsza", D), ("nsza", C.c_int32), ("pad0", C.c_int32),
        ("vza0", D), ("dvza", D), ("nvza", C.c_int32), ("pad1", C.c_int32),
        ("phi0", D), ("dphi", D), ("nphi", C.c_int32), ("pad2", C.c_int32),
    ]


class PipeChunk(C.Structure):
    """struct gort_pipe_chunk"""
    _fields_ = [("n", C.c_long), ("angles", C.POINTER(D)), ("rsurf", C.POINTER(D)), ("scomp", C.POINTER(D)),
                ("K", C.POINTER(D)), ("energy", C.POINTER(D)), ("energy_index", C.POINTER(C.c_uint32)), ("energy_rows", C.c_long)]


PIPE_SCOMP, PIPE_ENERGY, PIPE_ENERGY_ONLY, PIPE_ENERGY_INDEXED = 1, 2, 4, 8

DECLARED_SYMBOLS = [
    "gort_last_error", "gort_version", "gort_canopy_defaults", "gort_leaf_soil_defaults",
    "gort_canopy_newstyle", "gort_canopy_set_lai", "gort_canopy_init", "gort_price_soil",
    "gort_prospect_d", "gort_spectra", "gort_gauleg", "gort_format_f6", "gort_format_f6_row", "gort_lut_format", "gort_lut_read",
    "gort_device_count", "gort_dev_malloc", "gort_dev_free", "gort_memcpy_h2d", "gort_memcpy_d2h",
    "gort_lut_alloc", "gort_lut_free",
    "gort_rccl_unique_id", "gort_rccl_comm_init_rank", "gort_rccl_comm_init_all", "gort_rccl_comm_destroy", "gort_lut_allgather",
    "gort_gap_probabilities", "gort_gap_probabilities_dev", "gort_gap_cache_stats", "gort_gap_cache_clear",
    "gort_canopy_check_geometry",
    "gort_canopy_key", "gort_lut_cache_store", "gort_lut_cache_load",
    "gort_engine_create", "gort_engine_destroy", "gort_engine_stream", "gort_engine_synchronize",
    "gort_engine_set_canopy", "gort_engine_set_spectra", "gort_engine_nw",
    "gort_engine_n_members", "gort_engine_set_members", "gort_engine_set_members_leaf", "gort_engine_reserve_members", "gort_engine_get_member",
    "gort_rsurf_members_grid_dev", "gort_rsurf_members_stream", "gort_rsurf_members_stream_dev",
    "gort_rsurf_stream", "gort_rsurf_stream_dev", "gort_rsurf_grid_dev",
    "gort_energy_stream", "gort_energy_stream_dev", "gort_energy_members_dev",
    "gort_energy_stream_indexed", "gort_energy_stream_indexed_dev",
    "gort_host_malloc", "gort_host_free", "gort_set_device", "gort_get_device",
    "gort_pipe_create", "gort_pipe_acquire", "gort_pipe_submit", "gort_pipe_wait", "gort_pipe_release", "gort_pipe_destroy",
]

#: include/gort_amd_tuning.h: measurement and tuning hooks, not part of the drop-in boundary
TUNING_SYMBOLS = [
    "gort_engine_last_expand_ms", "gort_engine_last_stream_ms", "gort_engine_time_streams", "gort_engine_stream_form",
    "gort_engine_xcd_mapping", "gort_engine_xcd_weights", "gort_engine_set_xcd_weights", "gort_engine_store_pattern_gbs",
    "gort_engine_probe_store_pattern", "gort_selftest_index_math", "gort_engine_set_lut_slack_gib",
    "gort_engine_time_expand", "gort_engine_energy_beside_grids",
]

_lib = None


def lib():
    """Load libgort_amd.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s missing: run `python -m gort_amd.build` (hipcc, gfx950)" % LIB_PATH)
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64.  If torch is going
        # to be used for device buffers it must be loaded first so that libgort_amd.so binds to the
        # same runtime (two runtimes in one process leave the second without devices).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.gort_last_error.restype = C.c_char_p
        L.gort_version.restype = C.c_char_p
        L.gort_lut_format.restype = C.c_long
        L.gort_engine_stream.restype = C.c_void_p
        L.gort_engine_last_expand_ms.restype = D
        L.gort_dev_malloc.restype = C.c_void_p
        L.gort_dev_malloc.argtypes = [C.c_size_t]
        L.gort_dev_free.argtypes = [C.c_void_p]
        L.gort_lut_alloc.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
        L.gort_lut_free.argtypes = [C.c_void_p]
        L.gort_lut_free.restype = None
        L.gort_engine_probe_store_pattern.restype = C.c_double
        L.gort_engine_probe_store_pattern.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gort_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gort_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gort_engine_create.argtypes = [C.POINTER(C.c_void_p)]
        for name in ("gort_engine_destroy", "gort_engine_synchronize", "gort_engine_stream", "gort_engine_nw",
                     "gort_engine_last_expand_ms", "gort_engine_xcd_mapping"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.gort_host_malloc.restype = C.c_void_p
        L.gort_host_malloc.argtypes = [C.c_size_t]
        L.gort_host_free.argtypes = [C.c_void_p]
        L.gort_set_device.argtypes = [C.c_int]
        L.gort_pipe_create.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_uint, C.POINTER(C.c_void_p)]
        L.gort_pipe_acquire.argtypes = [C.c_void_p, C.POINTER(C.POINTER(D))]
        L.gort_pipe_submit.argtypes = [C.c_void_p, C.c_long]
        L.gort_pipe_wait.argtypes = [C.c_void_p, C.POINTER(PipeChunk)]
        L.gort_pipe_release.argtypes = [C.c_void_p]
        L.gort_pipe_destroy.argtypes = [C.c_void_p]
        L.gort_pipe_destroy.restype = None
        L.gort_engine_stream_form.argtypes = [C.c_void_p]
        L.gort_engine_time_streams.argtypes = [C.c_void_p, C.c_int]
        L.gort_engine_last_stream_ms.argtypes = [C.c_void_p]
        L.gort_engine_last_stream_ms.restype = D
        L.gort_engine_xcd_weights.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.gort_engine_set_xcd_weights.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.gort_engine_store_pattern_gbs.argtypes = [C.c_void_p]
        L.gort_engine_store_pattern_gbs.restype = D
        L.gort_engine_set_lut_slack_gib.argtypes = [C.c_void_p, C.c_int]
        L.gort_engine_set_canopy.argtypes = [C.c_void_p, C.POINTER(Canopy)]
        L.gort_engine_set_spectra.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gort_rsurf_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
        L.gort_rsurf_stream_dev.argtypes = L.gort_rsurf_stream.argtypes
        L.gort_rsurf_grid_dev.argtypes = [C.c_void_p, C.POINTER(Grid), C.c_long, C.c_long, C.c_void_p]
        L.gort_rsurf_members_grid_dev.argtypes = [C.c_void_p, C.POINTER(Grid), C.c_int, C.c_int, C.c_void_p]
        L.gort_rsurf_members_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.gort_rsurf_members_stream_dev.argtypes = L.gort_rsurf_members_stream.argtypes
        L.gort_engine_n_members.argtypes = [C.c_void_p]
        L.gort_engine_set_members.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.gort_engine_reserve_members.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.gort_rccl_unique_id.argtypes = [C.c_void_p]
        L.gort_rccl_comm_init_rank.argtypes = [C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.gort_rccl_comm_init_all.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gort_rccl_comm_destroy.argtypes = [C.c_void_p]
        L.gort_lut_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.gort_engine_set_members_leaf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_int]
        L.gort_engine_get_member.argtypes = [C.c_void_p, C.c_int, C.POINTER(Canopy), C.c_void_p, C.c_void_p,
                                             C.c_void_p]
        L.gort_energy_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        L.gort_energy_stream_dev.argtypes = L.gort_energy_stream.argtypes
        L.gort_energy_members_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
        L.gort_energy_stream_indexed.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.POINTER(C.c_long)]
        L.gort_energy_stream_indexed_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.gort_gap_probabilities.argtypes = [C.c_void_p, C.c_int]
        L.gort_gap_cache_stats.argtypes = [C.POINTER(C.c_long)] * 3
        L.gort_gap_cache_stats.restype = None
        L.gort_gap_cache_clear.argtypes = []
        L.gort_gap_cache_clear.restype = None
        L.gort_canopy_key.argtypes = [C.POINTER(Canopy)]
        L.gort_canopy_key.restype = C.c_uint64
        L.gort_lut_cache_store.argtypes = [C.c_char_p, C.POINTER(Canopy)]
        L.gort_lut_cache_load.argtypes = [C.c_char_p, C.POINTER(Canopy)]
        L.gort_gap_probabilities_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.gort_canopy_newstyle.argtypes = [C.POINTER(Canopy), C.c_float, C.c_float, C.c_float]
        L.gort_canopy_set_lai.argtypes = [C.POINTER(Canopy), C.c_float]
        L.gort_prospect_d.argtypes = [D] * 7 + [C.c_void_p]
        L.gort_gauleg.argtypes = [D, D, C.c_void_p, C.c_void_p, C.c_int]
        L.gort_lut_read.argtypes = [C.c_char_p, C.POINTER(Canopy)]
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        raise GortError(rc, lib().gort_last_error().decode())


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    """host numpy array or torch tensor -> void*"""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    if isinstance(a, (DeviceBuffer, LutBuffer)):
        return C.c_void_p(a.ptr)
    if isinstance(a, C.c_void_p):
        return a
    return a.ctypes.data_as(C.c_void_p)


class DeviceBuffer:
    """Raw HBM allocation through the C ABI (gort_dev_malloc) for torch-free callers."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().gort_dev_malloc(self.nbytes)
        if not self.ptr:
            raise GortError(ENOMEM, lib().gort_last_error().decode())
        self.shape = (self.nbytes // 8,)

    def to_numpy(self, count=None, offset=0):
        count = (self.nbytes // 8 - offset) if count is None else count
        out = np.empty(count, dtype=np.float64)
        _check(lib().gort_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + 8 * offset), 8 * count))
        return out

    def free(self):
        if self.ptr:
            lib().gort_dev_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


RCCL_ID_BYTES = 128


def rccl_unique_id():
    """128 opaque bytes (ncclGetUniqueId): rank 0 draws them, the host's bootstrap hands them to the other ranks."""
    buf = C.create_string_buffer(RCCL_ID_BYTES)
    _check(lib().gort_rccl_unique_id(buf))
    return bytes(buf.raw)


class RcclComm:
    """An RCCL communicator made through the C ABI (gort_rccl_comm_init_rank on the CURRENT device)."""

    def __init__(self, world, unique_id, rank):
        assert len(unique_id) == RCCL_ID_BYTES
        h = C.c_void_p()
        _check(lib().gort_rccl_comm_init_rank(int(world), C.create_string_buffer(unique_id, RCCL_ID_BYTES), int(rank), C.byref(h)))
        self.h, self.world, self.rank = h, int(world), int(rank)

    def destroy(self):
        if self.h:
            _check(lib().gort_rccl_comm_destroy(self.h))
            self.h = None


class LutPlacement(C.Structure):
    _fields_ = [("draws", C.c_int32), ("picked", C.c_int32), ("probe_gbs", D * 64), ("accept_gbs", D), ("shifted", C.c_int32),
                ("rescans", C.c_int32), ("slack_bytes", C.c_uint64)]


class LutBuffer:
    """Device memory from gort_lut_alloc: float64, `doubles` elements, placement measured by the engine (include/gort_amd.h).
    Speaks __cuda_array_interface__, so `torch.as_tensor(buf, device="cuda")` (or `.tensor()`) gives a zero-copy view for
    torch.distributed collectives; the view keeps the buffer alive."""

    def __init__(self, ptr, nbytes, placement, window=None):
        self.ptr, self.nbytes = int(ptr), int(nbytes)
        self.shape = (self.nbytes // 8,)
        self.placement = {"draws": int(placement.draws), "picked": int(placement.picked),
                          "probe_gbs": [float(placement.probe_gbs[i]) for i in range(placement.draws)],
                          "accept_gbs": float(placement.accept_gbs), "shifted": bool(placement.shifted),
                          "rescans": int(placement.rescans), "slack_bytes": int(placement.slack_bytes)}
        self.window = window                 # (offset_doubles, doubles) this process writes, or None = everything

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 2, "strides": None}

    def tensor(self, shape=None):
        import torch
        t = torch.as_tensor(self, device="cuda")
        if t.data_ptr() != self.ptr:
            raise GortError(EINVAL, "LutBuffer.tensor(): torch copied the buffer instead of viewing it")
        t._gort_owner = self                 # the view must not outlive the allocation
        return t.view(*shape) if shape else t

    def at(self, offset_doubles):
        """void* `offset_doubles` into the buffer (for the *_dev entry points)."""
        return C.c_void_p(self.ptr + 8 * int(offset_doubles))

    def to_numpy(self, count=None, offset=0):
        count = (self.nbytes // 8 - offset) if count is None else count
        out = np.empty(count, dtype=np.float64)
        _check(lib().gort_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + 8 * offset), 8 * count))
        return out

    def free(self):
        if self.ptr:
            lib().gort_lut_free(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """float64 numpy array over pinned host memory (gort_host_malloc): DMA target of the host entry points."""

    def __init__(self, shape):
        self.shape = tuple(int(x) for x in (shape if hasattr(shape, "__len__") else (shape,)))
        n = int(np.prod(self.shape)) if self.shape else 1
        self.ptr = lib().gort_host_malloc(8 * max(n, 1))
        if not self.ptr:
            raise GortError(ENOMEM, lib().gort_last_error().decode())
        buf = (D * n).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.float64, count=n).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().gort_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Pipe:
    """gort_pipe: chunks of an angle stream in flight (see include/gort_amd.h)."""

    def __init__(self, engine, max_lines, depth=3, flags=0):
        h = C.c_void_p()
        _check(lib().gort_pipe_create(engine.h, max_lines, depth, flags, C.byref(h)))
        self.h, self.nw, self.max_lines = h, engine.nw, max_lines

    def acquire(self):
        p = C.POINTER(D)()
        _check(lib().gort_pipe_acquire(self.h, C.byref(p)))
        return np.ctypeslib.as_array(p, shape=(self.max_lines, 4))

    def submit(self, n):
        _check(lib().gort_pipe_submit(self.h, n))

    def wait(self):
        """dict of numpy views into the slot's pinned buffers, valid until release()."""
        c = PipeChunk()
        _check(lib().gort_pipe_wait(self.h, C.byref(c)))
        n, nw = c.n, self.nw
        v = lambda p, shape: np.ctypeslib.as_array(p, shape=shape) if p and n > 0 else None
        rows = c.energy_rows
        return {"n": n, "angles": v(c.angles, (n, 4)), "rsurf": v(c.rsurf, (n, nw)), "scomp": v(c.scomp, (n, nw, 4)),
                "K": v(c.K, (n, 4)), "energy": v(c.energy, (rows, nw, 3)) if rows > 0 else None,
                "energy_index": v(c.energy_index, (n,)), "energy_rows": rows}

    def release(self):
        _check(lib().gort_pipe_release(self.h))

    def close(self):
        if self.h:
            lib().gort_pipe_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------ host side
def make_canopy(lai=None, newstyle=None, favd=None, r=None, b=None, h1=None, h2=None, lam=None,
                beta=None, diffuse=None, q08=False):
    """Canopy record the way the gortt flags build it (gortt.c:1026-1131), initialised."""
    L = lib()
    c = Canopy()
    L.gort_canopy_defaults(C.byref(c))
    for name, v in (("favd", favd), ("r", r), ("b", b), ("h1", h1), ("h2", h2), ("lambda_", lam)):
        if v is not None:
            setattr(c, name, v)
    if newstyle is not None:
        L.gort_canopy_newstyle(C.byref(c), *[C.c_float(x) for x in newstyle])
    if lai is not None:
        L.gort_canopy_set_lai(C.byref(c), C.c_float(lai))
    if beta is not None:
        c.use_user_beta, c.beta = 1, beta
    if diffuse is not None:
        c.use_user_fd, c.fd_user = 1, 1.0 - diffuse
    c.use_q08 = 1 if q08 else 0
    _check(L.gort_canopy_init(C.byref(c)))
    return c


def leaf_soil(prospect=None, rsl=None, alb_leaf=None, alb_soil=None):
    s = LeafSoil()
    lib().gort_leaf_soil_defaults(C.byref(s))
    for k, v in (prospect or {}).items():
        setattr(s, k, v)
    if rsl is not None:
        for i in range(4):
            s.rsl[i] = rsl[i]
    if alb_leaf is not None:
        s.use_alb_leaf, s.alb_leaf = 1, alb_leaf
    if alb_soil is not None:
        s.use_alb_soil, s.alb_soil = 1, alb_soil
    return s


def spectra(wl, ls=None):
    wl = _f64(wl)
    ls = ls or leaf_soil()
    rs, rl, tl = np.zeros(wl.size), np.zeros(wl.size), np.zeros(wl.size)
    _check(lib().gort_spectra(C.byref(ls), _ptr(wl), wl.size, _ptr(rs), _ptr(rl), _ptr(tl)))
    return rs, rl, tl


def prospect_d(N=1.2, Cab=30., Car=10., Anth=1.0, Cbrown=0.0, Cw=0.015, Cm=0.009):
    RT = np.zeros(2 * NBANDS)
    _check(lib().gort_prospect_d(N, Cab, Car, Anth, Cbrown, Cw, Cm, _ptr(RT)))
    return RT


def gauleg(n=32):
    x, w = np.zeros(n), np.zeros(n)
    lib().gort_gauleg(-1.0, 1.0, _ptr(x), _ptr(w), n)
    return x, w


def lut_text(c):
    buf = C.create_string_buffer(1 << 15)
    n = lib().gort_lut_format(C.byref(c), buf, len(buf))
    if n < 0:
        _check(int(n))
    return buf.raw[:n].decode()


def lut_read(path, c):
    _check(lib().gort_lut_read(path.encode(), C.byref(c)))


def device_count():
    return lib().gort_device_count()


def set_device(device):
    """The calling thread's device (include/gort_amd.h gort_set_device): engines, pipes and LUT buffers created afterwards
    belong to it.  A process with one rank per GPU calls this (beside torch.cuda.set_device) before it creates its engine."""
    _check(lib().gort_set_device(int(device)))


def get_device():
    d = lib().gort_get_device()
    if d < 0:
        _check(d)
    return d


# ---------------------------------------------------------------- device side
def gap_probabilities(members):
    """In place, for one Canopy or a list of them (one workgroup per member)."""
    single = isinstance(members, Canopy)
    arr = (Canopy * (1 if single else len(members)))(*([members] if single else members))
    _check(lib().gort_gap_probabilities(arr, len(arr)))
    if single:
        C.memmove(C.byref(members), C.byref(arr[0]), C.sizeof(Canopy))
        return members
    for i, m in enumerate(members):
        C.memmove(C.byref(m), C.byref(arr[i]), C.sizeof(Canopy))
    return members


def gap_cache_stats():
    """(hits, misses, entries) of the in-process cache of gap tables per crown geometry."""
    h, m, n = C.c_long(), C.c_long(), C.c_long()
    lib().gort_gap_cache_stats(C.byref(h), C.byref(m), C.byref(n))
    return h.value, m.value, n.value


def gap_cache_clear():
    lib().gort_gap_cache_clear()


def canopy_key(canopy):
    """64-bit key of the crown geometry the gap tables depend on (r, b, h1, h2, lambda, favd, q08)."""
    return int(lib().gort_canopy_key(C.byref(canopy)))


def lut_cache_store(directory, canopy):
    _check(lib().gort_lut_cache_store(os.fsencode(directory), C.byref(canopy)))


def lut_cache_load(directory, canopy):
    """True: the canopy's gap tables were filled in from <directory>; False: no valid entry for this geometry."""
    rc = lib().gort_lut_cache_load(os.fsencode(directory), C.byref(canopy))
    if rc < 0:
        _check(rc)
    return rc == 0


class Engine:
    def __init__(self):
        h = C.c_void_p()
        _check(lib().gort_engine_create(C.byref(h)))
        self.h = h
        self.nw = 0

    def close(self):
        if self.h:
            lib().gort_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_canopy(self, c):
        _check(lib().gort_engine_set_canopy(self.h, C.byref(c)))

    def set_spectra(self, rsoil, rleaf, tleaf):
        rs, rl, tl = _f64(rsoil), _f64(rleaf), _f64(tleaf)
        _check(lib().gort_engine_set_spectra(self.h, rs.size, _ptr(rs), _ptr(rl), _ptr(tl)))
        self.nw = rs.size

    # ---- ensembles -------------------------------------------------------------------------------
    def set_members(self, members, spectra, compute_gaps=False):
        """members: list of Canopy; spectra: [n][3][nw] (rsoil, rleaf, tleaf per member)."""
        arr = (Canopy * len(members))(*members)
        sp = _f64(spectra)
        assert sp.shape[0] == len(members) and sp.shape[1] == 3
        _check(lib().gort_engine_set_members(self.h, arr, len(members), int(compute_gaps), sp.shape[2], _ptr(sp)))
        self.nw = sp.shape[2]

    def set_members_leaf(self, members, leaf, wl, compute_gaps=False):
        """Spectra of every member computed on the device from its LeafSoil parameters.  members / leaf: lists of Canopy /
        LeafSoil, or the C arrays of member_arrays() (a driver that cycles keeps its ensemble in those: marshalling a thousand
        records out of Python objects costs a millisecond per call)."""
        arr, larr = members, leaf
        if not isinstance(arr, C.Array):
            arr, larr = member_arrays(members, leaf)
        assert len(larr) == len(arr)
        members, leaf = arr, larr
        w = _f64(wl)
        _check(lib().gort_engine_set_members_leaf(self.h, arr, larr, len(members), int(compute_gaps), _ptr(w), w.size))
        self.nw = w.size

    def reserve_members(self, n_members, nw):
        """Allocate everything the member setters need for n_members x nw bands now (gort_engine_reserve_members)."""
        _check(lib().gort_engine_reserve_members(self.h, int(n_members), int(nw)))

    def get_member(self, m):
        c = Canopy()
        rs, rl, tl = np.zeros(self.nw), np.zeros(self.nw), np.zeros(self.nw)
        _check(lib().gort_engine_get_member(self.h, m, C.byref(c), _ptr(rs), _ptr(rl), _ptr(tl)))
        return c, rs, rl, tl

    def rsurf_members_grid_dev(self, grid, member_begin, member_end, lut_t):
        _check(lib().gort_rsurf_members_grid_dev(self.h, C.byref(grid), member_begin, member_end, _ptr(lut_t)))

    def rsurf_members_stream(self, angles_deg, member_begin=0, member_end=None):
        """The same angle lines for ensemble members [member_begin, member_end): rsurf[member][line][band]."""
        ang = _f64(angles_deg).reshape(-1, 4)
        if member_end is None:
            member_end = lib().gort_engine_n_members(self.h)
        out = np.empty((member_end - member_begin, ang.shape[0], self.nw))
        _check(lib().gort_rsurf_members_stream(self.h, _ptr(ang), ang.shape[0], member_begin, member_end, _ptr(out)))
        return out

    def synchronize(self):
        _check(lib().gort_engine_synchronize(self.h))

    def rsurf_stream(self, angles_deg, want_scomp=False, want_K=True, out=None):
        """out: optional PinnedArray.array of shape (nA, nw) - results then arrive by ONE DMA at the PCIe rate."""
        ang = _f64(angles_deg).reshape(-1, 4)
        nA = ang.shape[0]
        if out is None:
            out = np.zeros((nA, self.nw))
        assert out.shape == (nA, self.nw) and out.dtype == np.float64 and out.flags.c_contiguous
        sc = np.zeros((nA, self.nw, 4)) if want_scomp else None
        K = np.zeros((nA, 4)) if want_K else None
        _check(lib().gort_rsurf_stream(self.h, _ptr(ang), nA, _ptr(out), _ptr(sc), _ptr(K)))
        return out, sc, K

    def rsurf_stream_dev(self, angles_t, rsurf_t, scomp_t=None, K_t=None):
        _check(lib().gort_rsurf_stream_dev(self.h, _ptr(angles_t), angles_t.shape[0], _ptr(rsurf_t),
                                           _ptr(scomp_t), _ptr(K_t)))

    def stream_form(self):
        """'narrow' | 'flat' | 'lines': which kernel family expanded the last stream call (include/gort_amd_tuning.h)."""
        m = lib().gort_engine_stream_form(self.h)
        if m < 0:
            _check(m)
        return {0: "narrow", 1: "flat", 2: "lines"}[m]

    def time_expand(self, on=True):
        """The two events around every LUT expansion launch that last_expand_ms() reads (gort_engine_time_expand)."""
        _check(lib().gort_engine_time_expand(self.h, 1 if on else 0))

    def energy_beside_grids(self, on=True):
        """energy_members_dev on a stream of its own, so that a table asked for before the LUT chunks runs under them."""
        _check(lib().gort_engine_energy_beside_grids(self.h, 1 if on else 0))

    def time_streams(self, on=True):
        """Record the two events last_stream_ms() reads around the expansion stage of every stream call (6 us per call)."""
        _check(lib().gort_engine_time_streams(self.h, 1 if on else 0))

    def last_stream_ms(self):
        return lib().gort_engine_last_stream_ms(self.h)

    def lut_allgather(self, buf, rows_per_rank, row_doubles, comm):
        """gort_lut_allgather: the in-place RCCL all-gather of a gatherable LUT buffer (api.LutBuffer or a device pointer) on
        the engine's stream; rank `comm.rank` has written rows [rank * rows_per_rank, (rank + 1) * rows_per_rank)."""
        ptr = buf.ptr if hasattr(buf, "ptr") else int(buf)
        _check(lib().gort_lut_allgather(self.h, C.c_void_p(ptr), int(rows_per_rank), int(row_doubles) * 8, comm.rank, comm.world, comm.h))

    def lut_alloc(self, doubles, window=None, max_draws=3):
        """gort_lut_alloc: `doubles` float64 of HBM for a LUT; window = (offset, count) in doubles of the part this
        process writes (a rank's slab of a gatherable LUT), default everything.  max_draws > 1: the placement is
        measured (a scan inside one allocation with slack for a small window, else up to 3 allocations).  Returns a
        LutBuffer."""
        off, cnt = window if window is not None else (0, 0)
        out, info = C.c_void_p(), LutPlacement()
        _check(lib().gort_lut_alloc(self.h, C.c_size_t(8 * int(doubles)), C.c_size_t(8 * int(off)), C.c_size_t(8 * int(cnt)),
                                    int(max_draws), C.byref(out), C.byref(info)))
        return LutBuffer(out.value, 8 * int(doubles), info, window)

    def set_lut_slack_gib(self, gib):
        """Cap (GiB) of the slack lut_alloc keeps beside a placed buffer of this engine (default: GORT_LUT_SLACK_GIB or 48)."""
        _check(lib().gort_engine_set_lut_slack_gib(self.h, int(gib)))

    def probe_store_pattern(self, ptr, nbytes):
        """GB/s of the LUT kernel's bare store pattern over device memory the caller owns (contents destroyed)."""
        g = lib().gort_engine_probe_store_pattern(self.h, _ptr(ptr), C.c_size_t(int(nbytes)))
        if g < 0:
            _check(int(g))
        return g

    def rsurf_grid_dev(self, grid, row_begin, row_end, lut_t):
        _check(lib().gort_rsurf_grid_dev(self.h, C.byref(grid), row_begin, row_end, _ptr(lut_t)))

    def last_expand_ms(self):
        return lib().gort_engine_last_expand_ms(self.h)

    def xcd_weights(self):
        """(weights in 32nds per XCD, calibrated?) of the static XCD mapping."""
        w = (C.c_int * 8)()
        rc = lib().gort_engine_xcd_weights(self.h, w)
        return list(w), bool(rc == 1)

    def set_xcd_weights(self, weights):
        """Eight weights 8..32, or None to calibrate again on the next big LUT slab."""
        w = (C.c_int * 8)(*weights) if weights is not None else None
        _check(lib().gort_engine_set_xcd_weights(self.h, w))

    def store_pattern_gbs(self):
        """GB/s of the bare store pattern during the XCD calibration pass (0 before it ran)."""
        return lib().gort_engine_store_pattern_gbs(self.h)

    def xcd_mapping(self):
        """'static' where workgroup dispatch was probed to be round-robin over the XCDs, else 'slots'."""
        m = lib().gort_engine_xcd_mapping(self.h)
        if m < 0:
            _check(m)
        return {1: "static", 2: "slots"}[m]

    def energy_stream(self, angles_deg):
        ang = _f64(angles_deg).reshape(-1, 4)
        out = np.zeros((ang.shape[0], self.nw, 3))
        _check(lib().gort_energy_stream(self.h, _ptr(ang), ang.shape[0], _ptr(out)))
        return out

    def energy_stream_indexed(self, angles_deg, rows_cap=None):
        """(rows[n_rows][nw][3], index[nA]): the distinct albedo rows of the stream in order of first appearance and each
        line's row (gort_energy_stream_indexed); rows_cap: room offered, default one row per line."""
        ang = _f64(angles_deg).reshape(-1, 4)
        cap = ang.shape[0] if rows_cap is None else int(rows_cap)
        rows = np.zeros((cap, self.nw, 3))
        index = np.zeros(ang.shape[0], dtype=np.uint32)
        n_rows = C.c_long()
        _check(lib().gort_energy_stream_indexed(self.h, _ptr(ang), ang.shape[0], _ptr(rows), cap, _ptr(index), C.byref(n_rows)))
        return rows[:n_rows.value], index

    def energy_stream_indexed_dev(self, angles_t, rows_t, index_t, n_rows_t):
        """rows_t[rows_cap][nw][3] float64, index_t[nA] and n_rows_t[1] 32-bit on the device; asynchronous."""
        _check(lib().gort_energy_stream_indexed_dev(self.h, _ptr(angles_t), angles_t.shape[0], _ptr(rows_t), rows_t.shape[0],
                                                    _ptr(index_t), _ptr(n_rows_t)))

    def energy_stream_dev(self, angles_t, energy_t):
        _check(lib().gort_energy_stream_dev(self.h, _ptr(angles_t), angles_t.shape[0], _ptr(energy_t)))

    def energy_members_dev(self, angles_t, member_begin, member_end, energy_t):
        """energy_t[member][nA][nw][3] for the ensemble members [member_begin, member_end)."""
        _check(lib().gort_energy_members_dev(self.h, _ptr(angles_t), angles_t.shape[0], member_begin, member_end,
                                             _ptr(energy_t)))


def member_arrays(members, leaf):
    """(Canopy[n], LeafSoil[n]) as contiguous C arrays: what gort_engine_set_members_leaf takes."""
    assert len(leaf) == len(members)
    return (Canopy * len(members))(*members), (LeafSoil * len(leaf))(*leaf)


def hemisphere_grid(nsza=91, nvza=91, nphi=361):
    """Integer-degree full-hemisphere grid of SURVEY.md 8(d) C3."""
    g = Grid()
    g.sza0, g.dsza, g.nsza = 0.0, 1.0, nsza
    g.vza0, g.dvza, g.nvza = 0.0, 1.0, nvza
    g.phi0, g.dphi, g.nphi = 0.0, 1.0, nphi
    return g
