// gort_internal.h -- shared between the host translation units and the HIP kernels.
#ifndef GORT_INTERNAL_H
#define GORT_INTERNAL_H

#include "gort_amd.h"
#include "gort_amd_tuning.h"

#include <cstdlib>

namespace gort {

int fail(int code, const char *fmt, ...);

// ---- A/B switches ----
// Environment variables that select another kernel form, panel shape or mapping with THE SAME RESULT exist in the MEASURING
// build only (python -m gort_amd.build --ab -> libgort_amd_ab.so, -DGORT_AB; so does the stamps build): the tests marked
// `ab` run there and hold every variant to the bits of the default.  The product library reads none of them - ab_env() is
// a null pointer at compile time and the alternative template instantiations are not compiled in.
#ifdef GORT_AB
inline const char *ab_env(const char *name) { return getenv(name); }
#else
inline const char *ab_env(const char *) { return nullptr; }
#endif

// ---- wavelength-only table L[k][nw] (SoA, k-major so that lanes read consecutive bands) ----
enum LambdaSlot {
    L_GAMMA = 0,   // sqrt(1-omega)                         gortt.c:470
    L_OMEGA,       // rleaf+tleaf                           gortt.c:469
    L_RFF,         // R_inf_ff                              gortt_brdf.c:574
    L_TFF,         // T_inf_ff                              gortt_brdf.c:492
    L_tFF,         // t_ff                                  gortt_brdf.c:401-403
    L_PFF,         // p_ff (= CfC)                          gortt_brdf.c:510-512
    L_RS,          // rsoil
    L_MGK,         // rs/(1-rs*p_ff) * (t'_ff - k_open)     gortt.c:519-525
    L_ZF,          // (t'_ff - k_openep) * rs               gortt.c:492
    L_TF,          // t'_ff * MGK                           gortt.c:545-547
    L_B,           // (1-omega)*omega*(1-g)                 gortt.c:504-506
    L_NSLOT
};

// ---- per-angle record written by the geometry kernel (GORT_COEF_STRIDE doubles) ----
// rsurf = aC*C0 + aB*B + aZ*Z + aG*G + aT*T   (exact regrouping of gortt.c:484-557)
enum CoefSlot {
    A_C = 0,       // Kc
    A_B,           // Kc*fd*kuusk/(2 cos(sza') cos(vza'))
    A_Z,           // Kc*fd*k_openep*K'z + Kz
    A_G,           // Kc*fd*k_openep*K'g + Kg
    A_T,           // Kt
    S_FD,          // fd
    S_MU,          // cos(sza')
    S_T0,          // exp(-k*elai*sec(sza'))                gortt_brdf.c:534
    S_TP0,         // Pn0(sza)+EPgap(sza)                   gortt_brdf.c:447
    S_EPS,         // EPgap(sza)
    S_PN0,         // Pn0(sza)                              (albedo, gortt_albedo.c:37)
    C_FDA,         // fd*kuusk/(2 cos cos)     (component spectra only)
    C_KPZ,         // fd*k_openep*K'z
    C_KPG,         // fd*k_openep*K'g
    C_PAD0,
    C_PAD1
};
static_assert(C_PAD1 + 1 == GORT_COEF_STRIDE, "record size");

// ---- launchers implemented in the .hip files; all asynchronous on `stream` ----
// (void* stream is a hipStream_t)
int launch_gap_probabilities(gort_canopy *members_dev, int n_members, void *stream);
// member-batched: canopies[n], spectra[n][3][nw] (rsoil, rleaf, tleaf) -> L[n][L_NSLOT][nw], followed by the first member's
// StreamBand table [nw][12] (gort_device.h; stream_band_table() finds it): lambda_table_doubles() doubles in all
inline size_t stream_band_table_offset(int nw, int n_members)            // in doubles, even: the table's records are 16-B aligned
{
    return ((size_t)L_NSLOT * n_members * (size_t)nw + 1) & ~(size_t)1;
}
constexpr int STREAM_BAND_TABLE_DOUBLES = 12;      // per band: sizeof(StreamBand) / sizeof(double) (gort_device.h asserts it)
inline size_t lambda_table_doubles(int nw, int n_members) { return stream_band_table_offset(nw, n_members) + (size_t)STREAM_BAND_TABLE_DOUBLES * nw; }
inline const double *stream_band_table(const double *L_dev, int nw, int n_members)
{
    return L_dev + stream_band_table_offset(nw, n_members);
}
int launch_lambda_table(const gort_canopy *canopies_dev, int n_members, int nw, const double *spectra_dev,
                        double *L_dev, void *stream);
// leaf/soil spectra of ensemble members on the device (gort_spectra.hip); tables from gort_host.cpp
int launch_member_spectra(const gort_leaf_soil *leaf_dev, int n_members, int nw, const double *wl_dev,
                          const float *coef_dev, const double *t12_dev, const double *talf_dev,
                          const double *eof_dev, double *spectra_dev, void *stream);
const float *prospect_coeff_table();      // [7][2101]
const double *price_eof_table();          // [4][421]
void interface_transmissivity_tables(const double **t12, const double **talf);   // [2101] each
// ---- geometry (gort_geometry.hip).  n_members > 1: one launch for all members (blockIdx.z), canopy_dev[m],
// records coef_dev[m][nA][GORT_COEF_STRIDE], outputs member-major.
// layout 0: classic records (CoefSlot); 1: the stream family's LineTerms (gort_device.h), what the stream
// expansions of large streams read (stream_is_large)
int launch_geometry_stream(const gort_canopy *canopy_dev, int n_members, const double *angles_dev, long nA,
                           double *coef_dev, double *K_dev, int layout, void *stream, bool proportions_wanted);
// proportions_wanted: the records feed component spectra or printed proportions - every line near the horizon takes the
// reference's route; false: reflectances only, lines typed at exactly +-90 degrees skip it (gort_geometry.h)
// streams of few bands without component spectra: geometry and samples in ONE launch, no records (same bits as the
// two-kernel path: the same functions on the same record, kept in registers)
bool stream_fuses(int nw, bool want_scomp);
int launch_geometry_stream_fused(const gort_canopy *canopy_dev, int n_members, const double *L_dev, int nw,
                                 const double *angles_dev, long nA, double *rsurf_dev, double *K_dev, void *stream);
// compact: 8 doubles per node (A_C..A_T + pad) for the LUT kernel; else full GORT_COEF_STRIDE records
int launch_geometry_grid(const gort_canopy *canopy_dev, const gort_grid &g, long row_begin, long row_end,
                         double *coef_dev, bool compact, void *stream);
// grids of a few bands: samples formed in the geometry kernel itself, rsurf_dev[rows][nphi][nw], no records.  Up to 8 bands the
// (sun zenith, band) terms are formed in the kernel; beyond, sun_dev[q - q_begin][5][nw] (launch_sun_table) holds them for the
// sun rows q = member * nsza + isza of the launch
int launch_geometry_grid_fused(const gort_canopy *canopy_dev, const double *L_dev, int nw, const gort_grid &g,
                               long row_begin, long row_end, double *rsurf_dev, void *stream, const double *sun_dev = nullptr,
                               int q_begin = 0);
// per-XCD slot counters of the flat expansion kernels: 8 ints, one per 128-B line (XCD_SLOT_PITCH ints apart) so
// that the eight XCDs' atomics do not serialise on one line; the caller zeroes XCD_SLOT_BYTES on the stream
constexpr int XCD_SLOT_PITCH = 32;
constexpr size_t XCD_SLOT_BYTES = sizeof(int) * 8 * XCD_SLOT_PITCH;
// static XCD mapping of the flat kernels, 16 bytes that stay in SGPRs (a wave of a short panel cannot afford a
// table lookup in its prologue): XCD x (dispatch slot blockIdx & 7) works in w[x] of every 32 of its workgroups
// and owns the q * w[x] logical blocks from q * (w[0] + ... + w[x-1]); w[x] is byte x of w8, 8..32; weights of
// 32 = all XCDs alike (see xcd_logical_block in gort_brdf.hip)
struct XcdDuty {
    unsigned long long w8;
    long q;
};
// exact division of n < 2^31 by a divisor fixed per launch: n / d == (n * mul) >> (31 + sh)
struct FastDiv {
    unsigned mul, sh;
};
// Measure how fast each XCD writes (the store pattern of expand_flat_kernel over `slab`, whose contents are
// destroyed) and derive the duty weights that make all XCDs finish together; synchronises `stream`.
// *pattern_gbs (may be null): the rate of that bare store pattern with equal XCD shares, GB/s
int calibrate_xcd_weights(void *stream, double *slab, long n_doubles, int weights[8], double *pattern_gbs);
// host-side check of fast_div and of the duty mapping (0 = ok, else the failing source line)
int selftest_index_math();
// round_robin = 1 if workgroups b, b+8, ... of a launch share an XCD (then the static XCD mapping is exact);
// expand_wants_xcd_slots: does the flat expansion need the slot counters (GORT_EXPAND_XCD or the probe says so)
int probe_xcd_dispatch(void *stream, int *round_robin);
bool expand_wants_xcd_slots(bool dispatch_round_robin);
// ---- stream expansion (gort_stream_expand.hip)
// coef_dev: stream records (GORT_COEF_STRIDE doubles each) with ONE readable pad record in front and
// expand_stream_tail_pad_records() behind the last line; xcd_slots_dev: XCD_SLOT_BYTES zeroed on `stream` before the
// call, or nullptr for the static XCD mapping.
// Large streams without component spectra (stream_is_large: >= 128 bands and >= 4M samples) have their records in layout 1
// (the thirteen LineTerms) and are expanded by expand_flat_stream_kernel (aligned flat panels).
// grid_form: the "lines" are the nodes of a few-band LUT (classic records): narrow kernels, LUT family's
// five-term sample, so that every LUT path writes the same bits.
long expand_stream_tail_pad_records(int nw, long nA);
bool stream_is_large(int nw, long nA, bool want_scomp);
// band_table_dev: stream_band_table() of the engine's L (wide streams only; may be null for grid_form)
int launch_expand_stream(const gort_canopy *canopy_dev, const double *L_dev, const double *band_table_dev, int nw,
                         const double *coef_dev, long nA, double *rsurf_dev, double *scomp_dev, int *xcd_slots_dev, void *stream,
                         bool grid_form);
// the nodes of a few-band LUT (9 ... 127 bands) of n_members members, member-major classic records coef_dev[m][lines][16]
// -> lut_dev[m][lines][nw] with the LUT family's five-term sample, one thread per sample; canopies_dev / L_dev at the first member
int launch_expand_grid_members(const gort_canopy *canopies_dev, const double *L_dev, int nw, const double *coef_dev,
                               long lines_per_member, int n_members, double *lut_dev, void *stream);
// ---- streams of 17 ... ~250 bands without component spectra (gort_stream_lines.hip): geometry and samples in one kernel,
// lanes = lines, rows leave LDS as whole 128-B lines whatever the band count; band_table_dev as above
bool stream_takes_lines_kernel(int nw, long nA, bool want_scomp);
bool members_stream_takes_lines_kernel(int nw, long lines_per_member, int n_members);
// n_members > 1: the same lines for canopy_dev[m] with band_table_dev[m][nw][12], rows rsurf_dev[m][nA][nw] (K_dev null)
int launch_stream_lines(const gort_canopy *canopy_dev, int n_members, const double *band_table_dev, int nw, const double *angles_dev,
                        long nA, double *rsurf_dev, double *K_dev, void *stream);
// StreamBand tables of n_members members from their band tables L_dev[m][L_NSLOT][nw]: bands_dev[m][nw][12]
int launch_member_stream_bands(const double *L_dev, int n_members, int nw, double *bands_dev, void *stream);
// the same nA angle lines for n_members members: coef_dev[n][nA][GORT_COEF_STRIDE] scratch, rsurf_dev[n][nA][nw]
int launch_members_stream(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw,
                          const double *angles_dev, long nA, double *coef_dev, double *rsurf_dev, void *stream);
// sun rows q = member * nsza + isza, q in [q_begin, q_end): sun_dev[q - q_begin][5][nw]
int launch_sun_table(const gort_canopy *canopies_dev, const double *L_dev, int nw, const gort_grid &g,
                     int q_begin, int q_end, double *sun_dev, void *stream);
// coef_dev for launch_expand_grid: compact records, with ONE readable pad record in front of coef_dev
// and expand_grid_tail_pad_records() readable records behind the last angle
long expand_grid_tail_pad_records(int nw, long n_total);
// LUTs of 33 ... 127 bands through the aligned-chunk form (expand_flat_few_kernel): which grids take it, the readable records it
// wants in front of node 0 and behind the last node, and its launcher (sun_dev: n_sun rows from isza_base on)
bool grid_takes_few_flat_kernel(int nw, long n_total);
void expand_grid_few_pad_records(int nw, long n_total, long *front, long *tail);
int launch_expand_grid_few(const double *sun_dev, int isza_base, int n_sun, const double *coef_dev, int nw, int nvza, int nphi,
                           long row_begin, long row_end, double *lut_dev, int *xcd_slots_dev, const int *xcd_weights, void *stream);
// xcd_slots_dev: XCD_SLOT_BYTES zeroed on `stream` before the call (per-XCD slot counters), or nullptr for the static
// XCD mapping, for which xcd_weights[8] (32nds, nullptr = all 32) are the XCDs' duty weights
int launch_expand_grid(const double *sun_dev, int isza_base, const double *coef_dev, int nw, int nvza, int nphi,
                       long row_begin, long row_end, double *lut_dev, int *xcd_slots_dev, const int *xcd_weights,
                       void *stream);
// energy_dev[n_members][nA][nw][3]; members are canopies_dev[0..n) with L_dev[m][L_NSLOT][nw].
// ws_dev: energy_dedup_workspace(nA) bytes of device scratch (0 for few lines) - lines with the same normalised sun
// direction are evaluated once and their row is copied; nullptr = every line evaluated (same bits).
// xcd_round_robin: workgroups b, b + 8, ... share an XCD (probe_xcd_dispatch): the row copy then gives every XCD one
// contiguous run of its panels
size_t energy_dedup_workspace(long nA);
// ---- the sun-direction table by itself, and the INDEXED form of the output (gort_energy_stream_indexed*, GORT_PIPE_ENERGY_INDEXED)
// ws_dev: energy_table_workspace(nA) bytes (any nA >= 1).  launch_energy_table leaves in it rep[line], the list of the lines
// whose sun direction appears for the first time (in line order), and idx[line] = the place of line's row in that list;
// *n_rows_out_dev (may be null) = the length of the list.  launch_energy_rows evaluates the list into
// rows_dev[member][rows_cap][nw][3] (places beyond rows_cap are skipped; n_rows_known = the length if the host knows it, else -1).
size_t energy_table_workspace(long nA);
int launch_energy_table(const double *angles_dev, long nA, void *ws_dev, unsigned *n_rows_out_dev, void *stream);
const unsigned *energy_table_index(const void *ws_dev, long nA);      // idx[nA]
const unsigned *energy_table_count(const void *ws_dev, long nA);      // the length of the list
int launch_energy_rows(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw, const double *angles_dev,
                       long nA, const double *nodes_dev, double *rows_dev, long rows_cap, const void *ws_dev, long n_rows_known,
                       void *stream);
int launch_energy(const gort_canopy *canopies_dev, int n_members, const double *L_dev, int nw,
                  const double *angles_dev, long nA, const double *nodes_dev, double *energy_dev, void *ws_dev,
                  bool xcd_round_robin, void *stream);

}  // namespace gort

// the two halves of gort_energy_stream_indexed_dev for gort_pipe (gort_api.hip): the table of a chunk's lines on a stream of
// the caller's choice into a workspace of its own (energy_table_workspace bytes), then - the caller has read the number of
// rows - the rows on the engine's stream
int gort_engine_energy_table(gort_engine *e, const double *angles_dev, long nA, void *ws_dev, uint32_t *n_rows_dev, void *stream);
int gort_engine_energy_rows(gort_engine *e, const double *angles_dev, long nA, const void *ws_dev, long n_rows, double *rows_dev);
#endif
