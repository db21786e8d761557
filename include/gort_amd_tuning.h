/*
 * gort_amd_tuning.h -- measurement and tuning hooks of libgort_amd.so.
 *
 * NOT part of the drop-in boundary (include/gort_amd.h): nothing here changes a result, and no caller of the
 * reference's path needs any of it.  bench.py reads the kernel timers; tests use the form overrides to compare
 * kernels that must write the same bits; the XCD functions expose what the LUT kernel measured about the part.
 * The environment switches of the same category exist in the measuring build only (DESIGN.md, section "Knobs").
 */
#ifndef GORT_AMD_TUNING_H
#define GORT_AMD_TUNING_H

#include "gort_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel timers (HIP events on the engine's own stream) ---- */
/* average duration (ms) of the LUT expansion kernel (expand_flat_kernel) over the launches since the last call
 * (up to 512 are kept); <0 if none */
double gort_engine_last_expand_ms(gort_engine *e);
/* duration (ms) of the expansion stage of the last gort_rsurf_stream[_dev] call; <0 if none, or if the engine was not asked
 * to time its stream calls: the two events around the stage cost every call 6 us on the device's queue (an empty kernel:
 * launch + synchronisation 12.1 us, with an event in front and behind 18.2, tools/probes/clock_probe.hip) - a third of a
 * short call like BASELINE config 2 - so they are recorded only after gort_engine_time_streams(e, 1) */
int    gort_engine_time_streams(gort_engine *e, int on);
/* the two events around every LUT expansion launch (gort_engine_last_expand_ms) on / off; on when the engine is created.  A
 * caller that queues many launches back to back (the chunks of an ensemble's LUTs) and does not read the timer saves their
 * share of the gap between two launches */
int    gort_engine_time_expand(gort_engine *e, int on);
/* ---- ensembles: the albedo / fAPAR table beside the LUT chunks ----
 * on: gort_energy_members_dev queues its kernels on a stream of its own (behind everything the engine's stream holds at the
 * time of the call) instead of the engine's stream, so that a table asked for BEFORE the LUT chunks of the same members is
 * evaluated under them (fp64-issue bound beside HBM-write bound); gort_engine_synchronize waits for it.  No result changes. */
int    gort_engine_energy_beside_grids(gort_engine *e, int on);
double gort_engine_last_stream_ms(gort_engine *e);

/* ---- which kernel family expanded the last gort_rsurf_stream[_dev] call ----
 * 0 = a narrow-stream kernel (per sample / band-major / fused with the geometry for up to 16 bands), 1 = the aligned
 * flat-panel kernel (expand_flat_stream_kernel: >= 128 bands, >= 4M samples, records in between), 2 = the fused line kernel
 * (stream_lines_kernel: 17 ... 255 bands, and up to 600 bands that are not a multiple of 128; >= 256K samples; geometry and
 * samples in one launch).  All of them write the same
 * bits; tests use this to know which one they have compared. */
int  gort_engine_stream_form(gort_engine *e);

/* ---- XCDs ---- */
/* how the flat expansion kernels map workgroups to XCD-contiguous output ranges on this device: 1 = static
 * (workgroup dispatch probed to be round-robin over the XCDs), 2 = per-XCD slot counters; <0 = a GORT_E* code */
int  gort_engine_xcd_mapping(gort_engine *e);
/* duty weights of the eight XCDs in 32nds (static mapping): the XCDs of a part do not write equally fast, and
 * the slower ones get a smaller share of the LUT slab.  Measured with one pass of the bare store pattern over a
 * buffer whose contents are about to be overwritten anyway: a fresh gort_lut_alloc buffer, or the output slab of a
 * grid call of >= 1 GiB right before the call writes it (only bytes the call itself will write); once per engine
 * and again when the slab's size class (power of two) changes.  Returns 1 once calibrated or set, else 0. */
int  gort_engine_xcd_weights(const gort_engine *e, int weights[8]);
/* set the weights (each 8..32) instead of calibrating; NULL = forget them and calibrate on the next big slab */
int  gort_engine_set_xcd_weights(gort_engine *e, const int weights[8]);
/* GB/s of the calibration pass: the LUT kernel's store pattern without any arithmetic, equal XCD shares; 0 before */
double gort_engine_store_pattern_gbs(const gort_engine *e);
/* the placement probe of gort_lut_alloc on memory the caller owns: GB/s of the LUT kernel's bare store pattern over
 * [dev, dev + bytes) (best of two passes after a first one that touches the pages; the contents are destroyed);
 * 0 for regions below ~0.8 GB (the pattern needs 64 panels), < 0 = a GORT_E* code */
double gort_engine_probe_store_pattern(gort_engine *e, void *dev, size_t bytes);
/* the cap (GiB, 0..63) of the slack gort_lut_alloc keeps beside a placed buffer of this engine (include/gort_amd.h): the
 * engine starts with GORT_LUT_SLACK_GIB from the environment (default 48, read once when it is created) */
int  gort_engine_set_lut_slack_gib(gort_engine *e, int gib);
/* host-only self-test of the flat kernels' index arithmetic (multiply-shift divisions, XCD duty mapping as a
 * bijection); 0 = ok.  Needs no GPU. */
int  gort_selftest_index_math(void);

#ifdef __cplusplus
}
#endif
#endif /* GORT_AMD_TUNING_H */
