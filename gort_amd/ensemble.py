"""Ensemble-facing host layer over the member-batched kernels.

The reference README names the use - inverting canopy parameters from multi-angle reflectance with an ensemble
filter - but holds no code for it: each ensemble member is one forward run of `gortt` with that member's flags
(-HB -BR -PCC -LAI for the crown geometry, the PROSPECT-D / Price parameters for the spectra).  What is built
here is therefore exactly that and nothing beyond it: a state vector -> the `gortt` inputs of a member, all
members' observation vectors in one launch pair (gort_rsurf_members_stream), and finite-difference Jacobians as
an ensemble of perturbed members.  Parity is parity of the forward runs (tests/test_gpu_parity.py).
"""
import numpy as np

from . import api
from .shard import all_gather_lut, row_slab

#: state vector layout; the first four are the `gortt` new-style crown flags (gortt.c:1086-1131), the rest
#: PROSPECT-D leaf parameters (PROSPECT-D/prospect_DB.f90:72-191) and the first Price soil coefficient
#: (gortt_price_soil, gortt.c:1286-1328)
STATE = ("HB", "BR", "PCC", "LAI", "N", "Cab", "Car", "Cw", "Cm", "rsl1")
DEFAULT = dict(HB=2.0, BR=2.0, PCC=0.6, LAI=4.0, N=1.2, Cab=30.0, Car=10.0, Cw=0.015, Cm=0.009, rsl1=0.2)


def f32(x):
    """The reference parses flag values with atof into float variables: a member is what gortt would see."""
    return float(np.float32(x))


def member_inputs(state):
    """state (dict or sequence in STATE order) -> (Canopy, LeafSoil) as `gortt -HB .. -BR .. -PCC .. -LAI ..`
    with PROSPECT-D / Price parameters would build them."""
    s = dict(DEFAULT)
    s.update(state if isinstance(state, dict) else dict(zip(STATE, state)))
    canopy = api.make_canopy(newstyle=(f32(s["HB"]), f32(s["BR"]), f32(s["PCC"])), lai=f32(s["LAI"]))
    leaf = api.leaf_soil(prospect=dict(N=s["N"], Cab=s["Cab"], Car=s["Car"], Cw=s["Cw"], Cm=s["Cm"]),
                         rsl=(s["rsl1"], 0.1, 0.03726, -0.002426))
    return canopy, leaf


class Ensemble:
    """N members resident on one GPU: gap probabilities, spectra and band tables of all of them are (re)computed
    on the device by `set_states`, `observe` evaluates the same sun/view geometries for every member."""

    def __init__(self, wavelengths, engine=None):
        self.wl = np.ascontiguousarray(wavelengths, dtype=np.float64)
        self.eng = engine or api.Engine()
        self.n = 0

    def set_states(self, states):
        states = [dict(zip(STATE, s)) if not isinstance(s, dict) else s for s in states]
        pairs = [member_inputs(s) for s in states]
        self.eng.set_members_leaf([p[0] for p in pairs], [p[1] for p in pairs], self.wl, compute_gaps=True)
        self.n = len(pairs)
        return self

    def observe(self, angles_deg, member_begin=0, member_end=None):
        """rsurf[member][line][band] for lines of (vza, vaa, sza, saa) in degrees."""
        return self.eng.rsurf_members_stream(angles_deg, member_begin, self.n if member_end is None else member_end)

    def albedo(self, angles_deg, member_begin=0, member_end=None):
        """energy[member][line][band][3] = albedo, vegetation and soil absorption (gortt -energy per member)."""
        import torch
        ang = torch.as_tensor(np.ascontiguousarray(angles_deg, dtype=np.float64).reshape(-1, 4), device="cuda")
        m1 = self.n if member_end is None else member_end
        out = torch.empty((m1 - member_begin, ang.shape[0], self.wl.size, 3), dtype=torch.float64, device="cuda")
        self.eng.energy_members_dev(ang, member_begin, m1, out)
        self.eng.synchronize()
        return out.cpu().numpy()

    def close(self):
        self.eng.close()


def jacobian(wavelengths, state, angles_deg, rel_step=1e-3, params=STATE, engine=None):
    """Central finite-difference Jacobian d rsurf[line][band] / d state[p] as ONE ensemble of 2 P + 1 members.
    Returns (rsurf0[line][band], J[p][line][band], steps[p]).  Steps are relative to the parameter's value and
    are applied before the float32 rounding of the crown flags, so they must stay well above 1e-7 relative."""
    base = dict(DEFAULT)
    base.update(state if isinstance(state, dict) else dict(zip(STATE, state)))
    members, steps = [base], []
    for p in params:
        h = abs(base[p]) * rel_step
        lo, hi = dict(base), dict(base)
        lo[p], hi[p] = base[p] - h, base[p] + h
        if p in ("HB", "BR", "PCC", "LAI"):              # what the member really gets after the float32 flag parse
            h = 0.5 * (f32(hi[p]) - f32(lo[p]))
        members += [lo, hi]
        steps.append(h)
    ens = Ensemble(wavelengths, engine).set_states(members)
    r = ens.observe(angles_deg)
    if engine is None:
        ens.close()
    J = np.stack([(r[2 + 2 * k] - r[1 + 2 * k]) / (2.0 * steps[k]) for k in range(len(params))])
    return r[0], J, np.array(steps)


# ------------------------------------------------------------------------------------------------ BASELINE config 5
def draw_c5_members(n, seed=12345):
    """The ensemble of BASELINE.json config 5 as SURVEY.md 8(d) fixes it: numpy default_rng(seed); per member
    HB~U(1,3), BR~U(1,3.5), PCC~U(0.2,0.8), LAI~U(0.5,6) (through float32, as the CLI parses them), Cab~U(10,60),
    Cw~U(0.005,0.03), Cm~U(0.002,0.015), N~U(1,2.5), rsl1~U(0.05,0.4).  Returns (canopies, leaf_soil records)."""
    rng = np.random.default_rng(seed)
    canopies, leaf = [], []
    for _ in range(n):
        hb, br, pcc, lai = rng.uniform(1, 3), rng.uniform(1, 3.5), rng.uniform(0.2, 0.8), rng.uniform(0.5, 6)
        cab, cw, cm = rng.uniform(10, 60), rng.uniform(0.005, 0.03), rng.uniform(0.002, 0.015)
        N, rsl1 = rng.uniform(1, 2.5), rng.uniform(0.05, 0.4)
        canopies.append(api.make_canopy(newstyle=(f32(hb), f32(br), f32(pcc)), lai=f32(lai)))
        leaf.append(api.leaf_soil(prospect=dict(N=N, Cab=cab, Cw=cw, Cm=cm), rsl=(rsl1, 0.1, 0.03726, -0.002426)))
    return canopies, leaf


def c5_grid():
    """Per member: sun zenith 30 deg, view zenith 0..90, relative azimuth 0..360 in integer degrees (32 851 tuples)."""
    g = api.Grid()
    g.sza0, g.dsza, g.nsza = 30.0, 1.0, 1
    g.vza0, g.dvza, g.nvza = 0.0, 1.0, 91
    g.phi0, g.dphi, g.nphi = 0.0, 1.0, 361
    return g


def gather_member_tables(local, n_members, group=None):
    """The one exchange step of a member-sharded ensemble: every rank holds the reduced product of ITS members
    (members in gort_amd.shard.row_slab order), local[members_local][...]; returns table[n_members][...] on every
    rank by one all-gather (RCCL over xGMI with the `nccl` backend; 50 KB per member for the albedo table, while
    each member's 552 MB LUT stays on the GPU that made it)."""
    tail = tuple(local.shape[1:])
    flat = local.reshape(local.shape[0], int(np.prod(tail)) if tail else 1)      # explicit: a rank may hold no member
    return all_gather_lut(flat, n_members, group).reshape((n_members,) + tail)


def sharded_albedo_table(n_members, wavelengths, rank, world, sun_zenith=30.0, group=None, lut_chunk=0, seed=12345,
                         gather_on_cpu=False, barrier=None, lut_slack_gib=None, warmup_cycles=0):
    """Config 5 on `world` ranks, one GPU each: rank r draws the whole ensemble (same seed everywhere), keeps the
    members row_slab(r, world, n) on its GPU - gap probabilities, PROSPECT-D/Price, band tables on the device,
    optionally every member's hemisphere LUT in chunks of `lut_chunk` members (into a gort_lut_alloc buffer) - and all
    ranks end with the ensemble's albedo/fAPAR table energy[n_members][nw][3] after ONE all-gather (RCCL over xGMI with
    the `nccl` backend).  Returns (table as numpy, timings dict: setup_s = gap + spectra + band tables, lut_s and
    lut_chunk_ms per chunk, energy_s, gather_s, total_s from the first setup call to the gathered table; rank-local).
    gather_on_cpu: exchange through host memory (gloo rehearsals on one GPU).  barrier: called before the clock
    starts and around the gather, so that the all-gather's time is not another rank's lateness."""
    import time
    import torch
    wl = np.ascontiguousarray(wavelengths, dtype=np.float64)
    canopies, leaf = draw_c5_members(n_members, seed)
    m0, m1 = row_slab(rank, world, n_members)
    eng = api.Engine()
    lut = None
    if m1 > m0:
        # capacity, like the engine itself, before the clock (a filter allocates once and cycles many times): the member
        # buffers, and the ONE chunk buffer the LUTs pass through - placed by the C ABI's allocator (a 14 GB slab lies on a
        # fast stretch of HBM about half the time)
        eng.reserve_members(m1 - m0, wl.size)
        # the ensemble as C arrays, like the engine's buffers before the clock: a driver that cycles keeps it so (marshalling a
        # thousand records out of Python objects cost 0.6 ms of every cycle)
        member_arrays = api.member_arrays(canopies[m0:m1], leaf[m0:m1])
        if lut_chunk:
            g = c5_grid()
            chunk = min(lut_chunk, m1 - m0)
            per_member = g.nvza * g.nphi * wl.size
            # [r6] WHERE the chunk buffer lies decides 10-13 % of the LUTs' time (the same 40 chunks into a plain allocation 84-86
            # ms, now and then 74.6; into the best of three draws 74.3-74.6, now and then not; into a window placed by the
            # allocator's scan 74.3-74.6 five times of five: profiles/r06/c5_placement.log).  So the chunks go into the first
            # half of a buffer twice their size, which gort_lut_alloc places by its scan through slack -
            # 14 GB used of the ~76 GB held while the ensemble is evaluated
            if lut_slack_gib is not None:
                eng.set_lut_slack_gib(lut_slack_gib)
            lut = eng.lut_alloc(2 * chunk * per_member, window=(0, chunk * per_member), max_draws=5)
    sync = barrier or (lambda: None)
    energy = torch.empty((m1 - m0, 1, wl.size, 3), dtype=torch.float64, device="cuda")
    sun = torch.tensor([[0.0, 0.0, float(sun_zenith), 0.0]], dtype=torch.float64, device="cuda")

    def cycle():
        """One cycle of a filter's forward model on this rank: members -> gap probabilities, spectra, band tables -> the members'
        LUTs (chunks into the one buffer) -> the albedo table -> the all-gather.  Returns (table on every rank, timings)."""
        t = {"members": [m0, m1]}
        torch.cuda.synchronize()
        sync()
        t_start = t0 = time.perf_counter()
        if m1 > m0:
            eng.set_members_leaf(member_arrays[0], member_arrays[1], wl, compute_gaps=True)
            eng.synchronize()
        t["setup_s"] = time.perf_counter() - t0
        t["lut_s"], t["lut_chunk_ms"], t["lut_samples"] = 0.0, [], 0
        if lut is not None:
            eng.last_expand_ms()
            t0 = time.perf_counter()
            # The LUTs are a product that is consumed on the device and dropped (what leaves is the reduced table): the chunks
            # go into ONE buffer back to back, no host wait between them.  With records of <= 64 MB per chunk (25 members) the
            # engine runs geometry and sun table of chunk i+1 on its second stream under the expansion of chunk i.
            n_chunks = 0
            for a in range(0, m1 - m0, chunk):
                eng.rsurf_members_grid_dev(g, a, min(m1 - m0, a + chunk), lut)
                n_chunks += 1
            eng.synchronize()
            t["lut_s"] = time.perf_counter() - t0
            t["lut_chunk_ms"] = [t["lut_s"] * 1e3 / n_chunks] * n_chunks          # mean: the chunks are not timed one by one any more
            t["lut_kernel_ms"] = eng.last_expand_ms()                            # mean duration of the expansion kernel (HIP events)
            t["lut_samples"] = (m1 - m0) * per_member
            t["lut_alloc"] = lut.placement
        t0 = time.perf_counter()
        if m1 > m0:
            torch.cuda.synchronize()                 # torch fills on its stream, the engine reads on its own
            eng.energy_members_dev(sun, 0, m1 - m0, energy)
            eng.synchronize()
        torch.cuda.synchronize()
        t["energy_s"] = time.perf_counter() - t0
        sync()
        t0 = time.perf_counter()
        local = energy.view(m1 - m0, wl.size, 3)
        if world > 1:
            full = gather_member_tables(local.cpu() if gather_on_cpu else local, n_members, group)
        else:
            full = local
        torch.cuda.synchronize()
        sync()
        t["gather_s"] = time.perf_counter() - t0
        t["gather_bytes_received"] = int((n_members - (m1 - m0)) * wl.size * 3 * 8) if world > 1 else 0
        t["total_s"] = time.perf_counter() - t_start
        return full, t

    # a filter cycles many times: `warmup_cycles` whole cycles run untimed first (the first one meets what a process meets once -
    # the chunks' record and sun-table buffers, a size class's XCD calibration, code objects: 2 ms of the 82), the next is the
    # one whose times are returned
    for _ in range(max(0, int(warmup_cycles))):
        cycle()
    full, t = cycle()
    t["warmup_cycles"] = max(0, int(warmup_cycles))
    if lut is not None:
        lut.free()
    out = full.cpu().numpy()
    eng.close()
    return out, t
