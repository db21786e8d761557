/*
 * gort_amd.h -- C ABI of libgort_amd.so: MI355X-native GORT BRDF/albedo engine.
 *
 * Drop-in boundary for the hot path of tquaife/gort's `gortt` (Ni et al. 1999 GORT
 * model).  The reference has no library API - its seam is the set of void functions
 * that main() calls on caller-owned structs (reference include/gortt.h:217-219,
 * 251-252).  Each entry point below names the reference interface it replaces
 * (file:line under the reference tree).  Plain pointers and sizes only; no C++ or
 * torch types.  All floating point data is IEEE double unless stated.
 *
 * Conventions
 *   - functions return 0 on success, a negative GORT_E* code otherwise;
 *     gort_last_error() gives the message (thread-local).
 *   - `*_dev` arguments are DEVICE pointers (HIP, current device); everything else
 *     is host memory.  `stream` is a hipStream_t passed as void* (NULL = default).
 *   - angle tuples are the four columns of a gortt stdin line, in DEGREES:
 *     view zenith, view azimuth, sun zenith, sun azimuth (gortt.c:234).
 *   - there is NO CPU fallback: device entry points fail with GORT_ENODEVICE when
 *     no HIP device is usable.
 *   - measurement and tuning hooks (kernel timers, XCD duty weights, kernel-form overrides)
 *     are NOT part of this interface: include/gort_amd_tuning.h.
 */
#ifndef GORT_AMD_H
#define GORT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GORT_NLAYERS   15      /* gortt.c:78  */
#define GORT_NTH       91      /* gortt.c:714 */
#define GORT_MAXCROWNS 30      /* gortt.c:89  */
#define GORT_NH_ES     20      /* gortt.c:90  */
#define GORT_NPOINTS   32      /* gortt.c:93  */
#define GORT_NBANDS    2101    /* gortt.h:31  */
#define GORT_COEF_STRIDE 16    /* doubles per angle record of the stream path (internal buffers only) */

enum {
    GORT_OK = 0,
    GORT_EINVAL = -1,      /* bad argument */
    GORT_ERANGE = -2,      /* wavelength outside 400..2500 nm (gortt.c:1299-1302,1350-1353) */
    GORT_ENODEVICE = -3,   /* no usable HIP device / HIP error */
    GORT_EIO = -4,         /* LUT file cannot be opened (gortt.c:132-135) */
    GORT_ENOMEM = -5
};

/* ---- canopy: the members of the reference's gortt_parameters (gortt.h:123-212) that
 * the live path reads.  One POD record; the same bytes are used on host and device. ---- */
typedef struct gort_canopy {
    /* inputs: flag values (gortt.c:1026-1131) */
    double r, b, h1, h2, lambda, favd;
    double beta;      int32_t use_user_beta;  int32_t use_user_fd;   /* -beta / -diffuse */
    double fd_user;                                                  /* 1 - (-diffuse arg) */
    int32_t use_q08;  int32_t reserved0;                             /* -q08_pn_kopen */
    /* derived by gort_canopy_init (replaces gortt_init_params, gortt.c:632-868) */
    double ell, rr, rrr, h, k, elai, tau, z1, z2, lv;
    double favd_p, tau_p, lv_p, z1_p, z2_p, h1_p, h2_p;
    double dz, ds, dz_p, dth;
    double height_p[GORT_NLAYERS];
    double theta[GORT_NTH], theta_p[GORT_NTH];
    /* gap-probability products: exactly what `gortt -W` writes (gortt.c:123-128) */
    double p_n0[GORT_NTH];        /* p_n0[0][t]                                   */
    double epgap[GORT_NTH];       /* epgap[0][t]; [90] is 0 in the reference      */
    double k_open, k_openep;      /* k_open[0], k_openep[0]                       */
} gort_canopy;

/* ---- leaf/soil parameters (gortt_spectra, gortt.h:96-114; defaults gortt.c:38-59) ---- */
typedef struct gort_leaf_soil {
    double N, Cab, Car, Anth, Cbrown, Cw, Cm;      /* PROSPECT-D */
    double rsl[4];                                 /* Price soil EOF weights */
    int32_t use_alb_leaf, use_alb_soil;            /* -alb_leaf / -alb_soil */
    double alb_leaf, alb_soil;
} gort_leaf_soil;

typedef struct gort_engine gort_engine;      /* opaque, see below */

const char *gort_last_error(void);
const char *gort_version(void);

/* ===================== host-side precompute (no GPU needed) ===================== */

/* defaults of main(): gortt.c:67-72 (canopy), gortt.c:38-59 (leaf/soil) */
void gort_canopy_defaults(gort_canopy *c);
void gort_leaf_soil_defaults(gort_leaf_soil *s);
/* -HB/-BR/-PCC "new style" crown geometry, values as C floats: gortt.c:1014,1117-1125 */
void gort_canopy_newstyle(gort_canopy *c, float hb, float br, float pcc);
/* -LAI (float), applied after new-style: gortt.c:1127-1131 */
void gort_canopy_set_lai(gort_canopy *c, float lai);
/* derived scalars + zenith/height tables: replaces gortt_init_params, gortt.c:632-868 */
int  gort_canopy_init(gort_canopy *c);
/* GORT_EINVAL for a crown whose gap probabilities the reference cannot compute either (it fails in an allocation,
 * loops for ever or reports negative volumes: zero / negative / non-finite radii, h2 <= h1); called by every entry
 * point that computes gap probabilities, before anything reaches the device */
int  gort_canopy_check_geometry(const gort_canopy *c);

/* Price soil reflectance: replaces gortt_price_soil, gortt.c:1286-1328 */
int  gort_price_soil(const double *wl_nm, int nw, const double rsl[4], double *rsoil);
/* PROSPECT-D on the native 400..2500 @1 nm grid: replaces prospect_DB_,
 * PROSPECT-D/prospect_DB.f90:72-191 (+ tav_abs.f90).  RT[0..2100]=R, RT[2101..4201]=T */
int  gort_prospect_d(double N, double Cab, double Car, double Anth, double Cbrown,
                     double Cw, double Cm, double *RT);
/* all three spectra at arbitrary wavelengths incl. the -alb_* overrides: replaces
 * gortt_price_soil + gortt_prospect_interface, gortt.c:224-227,1331-1374 */
int  gort_spectra(const gort_leaf_soil *s, const double *wl_nm, int nw,
                  double *rsoil, double *rleaf, double *tleaf);
/* Gauss-Legendre nodes as the reference computes them: replaces gauleg, gortt_albedo.c:141-199 */
void gort_gauleg(double x1, double x2, double *x, double *w, int n);

/* Fast formatter of the output rows: writes exactly the bytes of printf("%f", v) (correctly rounded,
 * ties to even, glibc style), except that every NaN is written as "-nan" - which is what the
 * reference prints for its NaNs on x86 (gortt.c:310-324).  dst needs 352 bytes; returns the length. */
int  gort_format_f6(double v, char *dst);
/* n values, each followed by one space (the way a gortt row prints them, gortt.c:314-324); returns the bytes
 * written or a negative code.  cap >= 24 n is always enough for values below 4e9 in magnitude. */
long gort_format_f6_row(const double *v, long n, char *dst, size_t cap);

/* probability LUT, text format of `gortt -W` / `gortt -P file`: gortt.c:123-146.
 * gort_lut_format writes into buf (needs <= 16 KiB), returns bytes written or <0. */
long gort_lut_format(const gort_canopy *c, char *buf, size_t cap);
int  gort_lut_read(const char *path, gort_canopy *c);

/* Gap-table cache keyed on crown geometry - generalises `-W` / `-P file` (gortt.c:123-146), where the user has to
 * remember which file belongs to which crown geometry ("must be run with identical crown geometry").  The tables
 * depend on r, b, h1, h2, lambda, favd and on -q08_pn_kopen only (gortt_pn_kopen.c:7-129, 1144-1200): the key is
 * a 64-bit hash of those bit patterns.
 * Files: <dir>/gap-<16 hex digits>.lut = the rows of `-W` as C99 hex floats (exact, unlike "%0.40f", which flushes
 * the horizon values to zero), closed by a line "# gort-gap-lut 1 <r> <b> <h1> <h2> <lambda> <favd> <q08> <p_n0[90]>
 * <epgap[90]> <checksum of the tables>" that is checked on load (a damaged or foreign file is a miss).  fscanf("%d %lf %lf") stops at that line, so a cache file is also a valid `-P` file for the
 * reference and for this library.  Written to a temporary name and renamed: readers never see half a file.
 * gort_lut_cache_load: GORT_OK = tables filled in, 1 = no (valid) entry, <0 = error. */
uint64_t gort_canopy_key(const gort_canopy *c);
int  gort_lut_cache_store(const char *dir, const gort_canopy *c);
int  gort_lut_cache_load(const char *dir, gort_canopy *c);

/* ============================== device entry points ============================== */

int  gort_device_count(void);
/* Device of the calling thread (hipSetDevice / hipGetDevice).  Engines, pipes and buffers belong to the device
 * that was current when they were created; a thread that drives several devices selects the owner's device
 * before calling into it (gort_pipe_* do that themselves).  gort_get_device returns the ordinal or <0. */
int  gort_set_device(int device);
int  gort_get_device(void);

/* Plain device-memory helpers for callers that do not bring their own allocator (the `gortt`
 * host, C drivers).  New surface, not reference surface.  gort_dev_malloc returns NULL on failure. */
void *gort_dev_malloc(size_t bytes);
void  gort_dev_free(void *p_dev);
int   gort_memcpy_h2d(void *dst_dev, const void *src, size_t bytes);
int   gort_memcpy_d2h(void *dst, const void *src_dev, size_t bytes);

/* Device memory for a LUT with a MEASURED placement.  Where a buffer of several GB lies physically decides up to 15 % of
 * the rate at which the LUT kernel can write it (the same buffer is fast or slow for its whole life),
 * and a multi-GPU step ends with its slowest rank.  gort_lut_alloc measures candidates with the LUT kernel's store
 * pattern (no arithmetic, ~1 ms per 6 GB) over the window [win_offset, win_offset + win_bytes) - what this process
 * will write: the whole buffer (win_bytes = 0), or a rank's slab of a gatherable LUT - and keeps the fastest:
 *   - window at most half the buffer: ONE allocation with up to 48 GiB of slack, the buffer placed in 1-GiB steps
 *     inside it (the rate of a 6-25 GB window is a comb over its position: plateaus of 7.2-7.3 TB/s every 8-48 GiB,
 *     6.1-6.3 TB/s between them; a scan finds a plateau, a few random draws mostly do not).  Where the scan is flat
 *     (the allocation lies inside one physical extent) it is repeated, twice at most, on a new allocation made behind a
 *     blocker of a few GiB.  The slack stays allocated until gort_lut_free; the pointer returned may be interior.
 *   - else: up to min(max_draws, 5) separate allocations, alive together as far as the device keeps 64 GiB free beside
 *     them, the rest freed (on some boxes every second 50 GB allocation runs the LUT kernel 11 % slower than the others,
 *     for its whole life; the probe tells them apart: profiles/r04/placement_select.log).  Whether that is worth holding
 *     several buffers for a moment is the caller's call: bench.py asks for three and reports, beside its headline, the
 *     probe and the kernel's time on three plain allocations (`per_draw`) and the same steps on a plain first allocation
 *     (`first_draw`): DESIGN.md 5.1.
 * It stops early at 0.985 x the best rate this engine has measured for the size class.  Windows below 1 GiB and
 * max_draws = 1 are plain allocations.  Contents are undefined.  Release with gort_lut_free.
 * New surface (the reference keeps ONE row of nw doubles, malloc in main(): gortt.c:188). */
#define GORT_LUT_MAX_DRAWS 64
typedef struct gort_lut_placement {
    int32_t draws;                               /* candidates measured */
    int32_t picked;                              /* index of the one kept */
    double probe_gbs[GORT_LUT_MAX_DRAWS];        /* store-pattern rate over the window per candidate, GB/s (0 = not probed) */
    double accept_gbs;                           /* early-stop rate used for this call (0 = no history yet) */
    int32_t shifted;                             /* 1: candidates were placements inside ONE allocation (1-GiB steps) */
    int32_t rescans;                             /* scans repeated on a new allocation because the first found no plateau (0..2) */
    uint64_t slack_bytes;                        /* bytes allocated beyond `bytes` that stay allocated with the buffer (the room the
                                                  * scan moved it through; capped by GORT_LUT_SLACK_GIB, default 48) */
} gort_lut_placement;
int   gort_lut_alloc(gort_engine *e, size_t bytes, size_t win_offset, size_t win_bytes, int max_draws,
                     void **lut_dev, gort_lut_placement *info);
void  gort_lut_free(void *lut_dev);

/* ---- multi-GPU: the one exchange step of the path (an RCCL all-gather over xGMI reassembles the LUT) ----
 * One process per GPU (or one process driving several devices): every rank allocates the GATHERABLE LUT - world x
 * rows_per_rank rows, rows_per_rank = ceil(rows / world) - with gort_lut_alloc(window = its own rows), computes into its
 * window with gort_rsurf_grid_dev, and ONE in-place ncclAllGather fills the other windows: no receive buffer, no copy.
 * gort_lut_allgather enqueues it on the engine's stream behind the kernels (asynchronous; gort_engine_synchronize waits).
 * The communicator is RCCL's (ncclComm_t as void*): the caller's own, or one made here - rank 0 draws a unique id, the
 * host's bootstrap (MPI, a socket, torch.distributed: 128 opaque bytes) hands it to the other ranks, each calls
 * gort_rccl_comm_init_rank with its device current; gort_rccl_comm_init_all serves one process with n devices.  librccl
 * is bound at run time (the copy already in the process, if any).  The reference has no counterpart: it is one process
 * (README.md:30-36); this is new surface. */
#define GORT_RCCL_ID_BYTES 128
int   gort_rccl_unique_id(unsigned char *id /* [GORT_RCCL_ID_BYTES] */);
int   gort_rccl_comm_init_rank(int world, const unsigned char *id, int rank, void **comm);
int   gort_rccl_comm_init_all(int n_devices, const int *devices /* NULL = 0..n-1 */, void **comms /* [n_devices] */);
int   gort_rccl_comm_destroy(void *comm);
int   gort_lut_allgather(gort_engine *e, void *lut_dev, size_t rows_per_rank, size_t row_bytes, int rank, int world, void *comm);

/* Pinned (page-locked) host memory: buffers from here travel over PCIe by DMA at the link rate, and
 * gort_rsurf_stream / gort_energy_stream copy straight into them; results written into ordinary pageable memory
 * are staged through pinned chunks and copied once more by the host.  gort_host_malloc returns NULL on failure. */
void *gort_host_malloc(size_t bytes);
void  gort_host_free(void *p);

/* Pn/EPgap/KOpen for a batch of canopies (one workgroup per member).  Fills
 * p_n0/epgap/k_open/k_openep of every record in place; honours use_q08.
 * Replaces gortt_gap_probabilities (gortt_pn_kopen.c:7-129) and
 * gortt_gap_probabilities_Q08 (gortt_pn_kopen.c:1144-1200).
 * Each canopy must have been through gort_canopy_init. */
int  gort_gap_probabilities(gort_canopy *members, int n_members);
/* gort_gap_probabilities remembers the tables of the last GORT_GAP_CACHE (default 4096, 0 = off) distinct crown
 * geometries of this process (key: gort_canopy_key + the geometry itself): a member whose geometry has been seen
 * gets its tables from there, only the others go to the device.  Thread-safe. */
void gort_gap_cache_stats(long *hits, long *misses, long *entries);
void gort_gap_cache_clear(void);
int  gort_gap_probabilities_dev(gort_canopy *members_dev, int n_members, void *stream);

/* Opaque engine: owns a HIP stream, the device copy of one canopy, the spectra and
 * the wavelength-only tables derived from them.  NOT thread-safe: one thread at a time per engine (the
 * reference's functions are not re-entrant on their structs either); use one engine per thread,
 * or a gort_pipe, whose producer and consumer sides may live on two threads. */
int  gort_engine_create(gort_engine **out);
void gort_engine_destroy(gort_engine *e);
void *gort_engine_stream(gort_engine *e);           /* hipStream_t */
int  gort_engine_synchronize(gort_engine *e);
/* canopy must carry gap tables (gort_gap_probabilities or gort_lut_read) */
int  gort_engine_set_canopy(gort_engine *e, const gort_canopy *c);
int  gort_engine_set_spectra(gort_engine *e, int nw, const double *rsoil,
                             const double *rleaf, const double *tleaf);
int  gort_engine_nw(const gort_engine *e);

/* Ensembles (BASELINE.json config 5: N canopy/leaf parameter members = N independent forward runs of
 * the reference, README.md:8-9).  The engine then holds n_members canopies with their own spectra; the
 * single-canopy calls above and the stream/energy entry points address member 0.
 *   compute_gaps != 0: run the Pn/EPgap/KOpen kernel for all members on the device copy (one workgroup
 *                      per member) instead of expecting the tables in `members`.
 *   gort_engine_set_members:      spectra[n_members][3][nw] = rsoil, rleaf, tleaf per member (host)
 *   gort_engine_set_members_leaf: spectra computed ON THE DEVICE from each member's PROSPECT-D / Price
 *                                 parameters, one thread per (member, band); replaces n_members runs of
 *                                 gortt_price_soil + gortt_prospect_interface (gortt.c:224-227)
 *   gort_engine_get_member:       read a member back (any pointer may be NULL) */
int  gort_engine_n_members(const gort_engine *e);
int  gort_engine_set_members(gort_engine *e, const gort_canopy *members, int n_members, int compute_gaps,
                             int nw, const double *spectra);
int  gort_engine_set_members_leaf(gort_engine *e, const gort_canopy *members, const gort_leaf_soil *leaf,
                                  int n_members, int compute_gaps, const double *wl_nm, int nw);
/* Capacity for n_members x nw bands: device buffers, pinned staging and spectral tables are allocated now, so that the
 * member setters cost their copies and kernels only (an ensemble filter re-submits its members every cycle; the first
 * call of a process otherwise pays ~240 MB of allocation).  New surface; optional.  Call it BEFORE the setters: a call that
 * has to grow a buffer discards the canopies / spectra the engine held, and the entry points then fail with "no canopy set"
 * / "no spectra set" until they are set again. */
int  gort_engine_reserve_members(gort_engine *e, int n_members, int nw);
int  gort_engine_get_member(gort_engine *e, int member, gort_canopy *canopy, double *rsoil, double *rleaf,
                            double *tleaf);

/* BRDF for a stream of angle lines.  Replaces, per line, the angle normalisation of
 * main() (gortt.c:240-291), gortt_set_zenith_dependant_probabilities (gortt.c:872-915)
 * and gortt_rsurf (gortt.c:385-578).
 *   angles[nA][4]   degrees
 *   rsurf[nA][nw]
 *   scomp[nA][nw][4]  C,G,T,Z (-prnspec), may be NULL
 *   K[nA][4]          Kc,Kg,Kt,Kz (-prnprop), may be NULL
 * gort_rsurf_stream_dev with rsurf = scomp = NULL and K given: the viewed proportions alone, which need a canopy but
 * no spectra (the reference prints them for a header without wavelengths, `N 0`: gortt.c:424-449 run in front of
 * the wavelength loop). */
int  gort_rsurf_stream(gort_engine *e, const double *angles, long nA,
                       double *rsurf, double *scomp, double *K);
int  gort_rsurf_stream_dev(gort_engine *e, const double *angles_dev, long nA,
                           double *rsurf_dev, double *scomp_dev, double *K_dev);

/* Regular-grid LUT: every (sun zenith, view zenith, relative azimuth) node in integer
 * steps, equivalent to streaming the lines "vza phi sza 0" (SURVEY.md 8d, C3):
 *   sza = sza0 + i*dsza (i<nsza), vza = vza0 + j*dvza (j<nvza), phi = phi0 + l*dphi (l<nphi)
 *   lut_dev[nsza][nvza][nphi][nw], wavelength fastest.
 * [row_begin,row_end) selects (i,j) rows in flattened order i*nvza+j, so that ranks of a
 * multi-GPU job fill disjoint slabs; lut_dev points at the first selected row.
 * A LUT of any band count is formed with the five-term regrouping of gortt.c:484-557 (the sun enters a sample
 * through five numbers per (sun zenith, band)); the stream entry points group the same arithmetic around two other terms.
 * A LUT therefore equals the stream of its nodes to rounding (1e-13 relative), both within 1e-9 of the reference. */
typedef struct gort_grid {
    double sza0, dsza; int32_t nsza;  int32_t pad0;
    double vza0, dvza; int32_t nvza;  int32_t pad1;
    double phi0, dphi; int32_t nphi;  int32_t pad2;
} gort_grid;
int  gort_rsurf_grid_dev(gort_engine *e, const gort_grid *g, long row_begin, long row_end,
                         double *lut_dev);
/* Same grid for ensemble members [member_begin, member_end): lut_dev[member][nsza][nvza][nphi][nw],
 * one launch sequence for all of them, whatever the band count (a MODIS-style ensemble of 7 bands - the use the
 * reference's README.md:8-9 names - as well as the full spectrum). */
int  gort_rsurf_members_grid_dev(gort_engine *e, const gort_grid *g, int member_begin, int member_end,
                                 double *lut_dev);
/* The same nA angle lines for ensemble members [member_begin, member_end): rsurf[member][nA][nw] - the
 * observation vector of every member of an ensemble filter in one launch pair (the
 * reference has no such code, each member equals a forward run of gortt with that member's flags). */
int  gort_rsurf_members_stream(gort_engine *e, const double *angles, long nA, int member_begin, int member_end,
                               double *rsurf);
int  gort_rsurf_members_stream_dev(gort_engine *e, const double *angles_dev, long nA, int member_begin,
                                   int member_end, double *rsurf_dev);

/* Spectral albedo, vegetation and soil absorption per angle line.  Replaces
 * gortt_energy/gortt_albedo (gortt_albedo.c:7-138): 32x16 Gauss-Legendre nodes over the
 * viewing hemisphere.  energy[nA][nw][3] = albedo, favegt, fasoil (print order, gortt.c:323-324) */
int  gort_energy_stream(gort_engine *e, const double *angles, long nA, double *energy);
int  gort_energy_stream_dev(gort_engine *e, const double *angles_dev, long nA, double *energy_dev);
/* The same without the copies.  What gortt_energy computes for a line depends on the line's SUN direction only - the view
 * angles are the quadrature nodes (gortt_albedo.c:62-138 overwrites g->vza and g->vaa; main() calls it once per line all the
 * same, gortt.c:321-327) - so a stream holds as many DISTINCT rows as it has distinct normalised (sun zenith, sun azimuth)
 * pairs: 91 for a million lines of a 1-degree sun grid.  The indexed form hands out each of them once:
 *   rows[n_rows][nw][3]   the distinct rows, in the order in which their sun directions first appear in the stream
 *   index[nA]             line a's row is rows[index[a]] - bit for bit the row gort_energy_stream writes for line a
 *   n_rows                their number
 * rows_cap = the room in `rows`, in rows (nA is always enough).  Host form: if the stream has more distinct rows than
 * rows_cap nothing is evaluated, *n_rows says how many there are, index is filled in, and the call fails with GORT_ERANGE.
 * Device form (asynchronous on the engine's stream like every *_dev entry point): rows beyond rows_cap are not evaluated,
 * *n_rows_dev holds the full count - compare it with rows_cap after synchronising.  New surface: the reference prints the
 * 3 nw numbers again for every line (gortt.c:323-324). */
int  gort_energy_stream_indexed(gort_engine *e, const double *angles, long nA, double *rows, long rows_cap,
                                uint32_t *index, long *n_rows);
int  gort_energy_stream_indexed_dev(gort_engine *e, const double *angles_dev, long nA, double *rows_dev, long rows_cap,
                                    uint32_t *index_dev, uint32_t *n_rows_dev);
/* the same nA angle lines for ensemble members [member_begin, member_end): energy_dev[member][nA][nw][3] -
 * the reduced per-member product (albedo, fAPAR) an ensemble driver exchanges between GPUs */
int  gort_energy_members_dev(gort_engine *e, const double *angles_dev, long nA, int member_begin,
                             int member_end, double *energy_dev);

/* ---- chunks of an angle stream in flight (what `gortt` runs on) ----
 * Replaces the read-evaluate-print loop of main(), gortt.c:232-329, by a pipeline of `depth` slots: while the
 * kernels of chunk i run, the angles of chunk i+1 are copied in and the results of chunk i-1 are copied out, all
 * on streams of their own and ordered by events; the host is free to parse and format meanwhile.
 *   gort_pipe_create   slots of max_lines lines each for the engine's current band count (set canopy and spectra
 *                      first; do not change the band count while the pipe lives).  flags: GORT_PIPE_SCOMP (component
 *                      spectra, -prnspec), GORT_PIPE_ENERGY (albedo/fAPAR, -energy), GORT_PIPE_ENERGY_ONLY;
 *                      GORT_PIPE_ENERGY_INDEXED beside either of the last two: a chunk's albedo rows arrive in the indexed
 *                      form of gort_energy_stream_indexed - `energy` holds the chunk's energy_rows DISTINCT rows
 *                      [energy_rows][nw][3] and energy_index[n] says which one is a line's - so that rows that are copies
 *                      are neither written, nor copied over PCIe, nor formatted again by the consumer (gort_pipe_submit
 *                      then waits for the chunk's angles to arrive and its sun directions to be counted: tens of us).
 *                      Host buffers are pinned.
 *   gort_pipe_acquire  blocks until a slot is free; *angles = its pinned input buffer [max_lines][4] (degrees)
 *   gort_pipe_submit   n lines of the acquired slot: copy in, kernels, copy out are queued; returns at once
 *   gort_pipe_wait     blocks until the OLDEST submitted chunk has arrived on the host; its buffers stay valid
 *                      until gort_pipe_release (rsurf[n][nw], scomp[n][nw][4] or NULL, K[n][4], energy[energy_rows][nw][3] or NULL)
 *   acquire/submit belong to one thread, wait/release to one (possibly another) thread. */
#define GORT_PIPE_SCOMP  1u
#define GORT_PIPE_ENERGY 2u
#define GORT_PIPE_ENERGY_ONLY 4u     /* albedo/fAPAR without rsurf and K (those pointers of a chunk are NULL) */
#define GORT_PIPE_ENERGY_INDEXED 8u  /* energy = the distinct rows, energy_index[n] = each line's row */
typedef struct gort_pipe gort_pipe;
typedef struct gort_pipe_chunk {
    long n;
    const double *angles, *rsurf, *scomp, *K, *energy;
    const uint32_t *energy_index;    /* GORT_PIPE_ENERGY_INDEXED: [n], else NULL */
    long energy_rows;                /* rows in `energy`: n without GORT_PIPE_ENERGY_INDEXED (0 without energy) */
} gort_pipe_chunk;
int  gort_pipe_create(gort_engine *e, long max_lines, int depth, unsigned flags, gort_pipe **out);
int  gort_pipe_acquire(gort_pipe *p, double **angles);
int  gort_pipe_submit(gort_pipe *p, long n);
int  gort_pipe_wait(gort_pipe *p, gort_pipe_chunk *out);
int  gort_pipe_release(gort_pipe *p);
void gort_pipe_destroy(gort_pipe *p);

#ifdef __cplusplus
}
#endif
#endif /* GORT_AMD_H */
