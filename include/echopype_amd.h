/* echopype_amd -- C ABI of the MI355X (gfx950) hot path of echopype
 *
 *   calibrate.compute_Sv / compute_TS  ->  clean.remove_background_noise  ->  commongrid.compute_MVBS
 *
 * The reference (OSOceanAcoustics/echopype, pure Python) has no FFI for this path; its boundary is
 * a set of Python functions on xarray Datasets.  This header is the C-ABI those
 * functions bind to in the drop-in (echopype_amd/, ctypes; see INTEGRATION.md for the stub a
 * reference maintainer would add).  Each entry point cites the reference code it replaces
 * (paths relative to /root/reference/echopype).
 *
 * Conventions
 *   - every function returns an int status (EPA_OK = 0); epa_last_error() gives the message of the
 *     last failure on the calling thread.  Nothing throws, nothing aborts.
 *   - all array pointers are DEVICE pointers (HBM) unless the name ends in _host.  Arrays are
 *     C-contiguous with the reference's dimension order (channel, ping_time, range_sample[, beam]).
 *   - the library never allocates or frees result memory; the caller owns every buffer.  The only
 *     allocations are the explicit epa_malloc/epa_free helpers offered to callers without a device
 *     allocator of their own.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are asynchronous
 *     with respect to the host; they are re-entrant and keep no global mutable state.
 *   - NaN in -> NaN out; numerical edge cases never raise (SURVEY.md 8b "Error conventions").
 */
#ifndef ECHOPYPE_AMD_H
#define ECHOPYPE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EPA_VERSION 104 /* 0.1.4: epa_apply_masks */

typedef void* epa_stream_t;

enum epa_status { EPA_OK = 0, EPA_EINVAL = 1, EPA_EHIP = 2, EPA_ENOMEM = 3, EPA_EUNSUPPORTED = 4 };
enum epa_dtype { EPA_F32 = 0, EPA_F64 = 1 };
enum epa_cal_type { EPA_CAL_SV = 0, EPA_CAL_TS = 1 };
enum epa_sonar { EPA_SONAR_EK60 = 0, EPA_SONAR_EK80 = 1 };
/* how a per-ping parameter is laid out: one value, one per channel, one per (channel, ping),
 * or (gain / sa_correction only) a per-channel pulse-length table to be looked up per ping */
enum epa_param_mode { EPA_PM_SCALAR = 0, EPA_PM_CHANNEL = 1, EPA_PM_CHANNEL_PING = 2, EPA_PM_PULSE_TABLE = 3 };

/* ---- per-(channel, ping) coefficient row consumed by the power-sample kernels ------------------
 *   echo_range R(s) = fl(fl(s * ra) * rb) + r0     EK: ra = sample_interval, rb = sound_speed/2, r0 = 0
 *                                                  -- the reference's own operation order
 *                                                  (range.py:138), so echo_range is bit-identical and
 *                                                  bin membership can never differ by a rounding;
 *                                                  AZFP: ra = 1, rb = c/4*2N/f, r0 per range.py:81-89
 *   R'(s)          = R(s) - shift                  TVG range shift (range.py:176-199)
 *   out(s)         = g * raw(s) + n * log10(R') + alpha2 * R' + A           n = 20 (Sv) | 40 (TS)
 * with the spreading term evaluated in its separable form: R' = k * (s - d), k = ra * rb,
 * d = (shift - r0) / k, so n*log10(R') = n*log10(k) + n*log10(s - d).  d is (nearly always) the
 * same for every ping of a file -- exactly 2 for EK60, tau/(2*sample_interval) for EK80, -r0/k for
 * AZFP -- which lets the kernels evaluate log10(s - d) once per range column instead of once per
 * sample.  Slot A0 = A + n * log10(k) (so a row is specific to Sv or TS).
 */
#define EPA_NCOEF 8
enum epa_coef_slot { EPA_CF_RA = 0, EPA_CF_RB = 1, EPA_CF_R0 = 2, EPA_CF_SHIFT = 3, EPA_CF_ALPHA2 = 4,
                     EPA_CF_A0 = 5, EPA_CF_G = 6, EPA_CF_D = 7 };

/* flags of epa_sv_power / epa_sv_mvbs_fused */
#define EPA_FLAG_GUARD_POS 1u  /* R' <= 0 -> NaN (calibrate_ek.py:107); AZFP has no such guard  */
#define EPA_FLAG_MASK_RANGE 2u /* echo_range is NaN where the raw sample is NaN (range.py:143-148) */

/* flags of the binned reductions */
#define EPA_BIN_SKIPNA 1u       /* nanmean (skip NaN values) vs mean (NaN poisons the bin)            */
#define EPA_BIN_CLOSED_RIGHT 2u /* intervals (a, b] instead of [a, b)  (commongrid/utils.py:283-302) */
#define EPA_BIN_RANGE_AS_STORED 4u /* epa_mvbs with coefficient rows in place of the range array: round the range to
                                    * dtype before binning, i.e. bin exactly as on the echo_range array epa_sv_power
                                    * would have written (a no-op for EPA_F64)                                        */

/* ---- runtime ------------------------------------------------------------------------------------ */
int epa_version(void);
/* Hex digest (sha256, 32 characters) of the sources and compiler flags this library was built from
 * (echopype_amd/build.py:source_digest).  The binding recomputes it from the files beside it at import time and refuses
 * a library built from other sources: the shipped binary is the shipped code.  No reference counterpart. */
const char* epa_source_digest(void);
const char* epa_last_error(void);
/* Launch trace, test infrastructure: mode 1 clears the calling thread's trace and switches tracing on, mode 0 clears and
 * switches it off, any other mode leaves it as it is.  Returns the "kernel;kernel;..." names of the launches this thread
 * made since (2 KiB kept; the string is valid until the next call on the thread).  Lets a parity test assert WHICH
 * kernel served a call -- the reference has no counterpart. */
const char* epa_launch_trace(int mode);
/* 1 when the last epa_sv_mvbs_fused / epa_sv_noise_fused call on the calling thread was served by the kernel that leaves range_stats_out
 * {nanmin, nanmax, NaN count}; 0 when the generic kernel served it (it leaves the maximum only and writes NaN count -1).
 * The same fact as the -1 on the device, known on the host without waiting for the kernel. */
int epa_last_range_stats_filled(void);
/* Every distinct kernel name this PROCESS has launched so far, "name;name;..." (sorted; valid until the next call): the
 * test suite's kernel coverage (tests/conftest.py writes it out at the end of a GPU session). */
const char* epa_launch_seen(void);
int epa_device_count(int* n);
int epa_set_device(int dev);
int epa_device_name(int dev, char* buf, size_t len);
int epa_malloc(void** ptr, size_t bytes);
int epa_free(void* ptr);
int epa_memset(void* ptr, int value, size_t bytes, epa_stream_t stream);
int epa_memcpy_h2d(void* dst, const void* src_host, size_t bytes, epa_stream_t stream);
int epa_memcpy_d2h(void* dst_host, const void* src, size_t bytes, epa_stream_t stream);
int epa_stream_synchronize(epa_stream_t stream);
/* HIP-event timing on `stream` (used by bench.py for the roofline figure): record start/stop
 * around a region, then read the elapsed milliseconds.  Handles are opaque. */
int epa_timer_create(void** timer);
int epa_timer_destroy(void* timer);
int epa_timer_start(void* timer, epa_stream_t stream);
int epa_timer_stop(void* timer, epa_stream_t stream);
int epa_timer_elapsed_ms(void* timer, float* ms); /* synchronises on the stop event */

/* ---- K0: coefficient table for EK60 / EK80 power samples ------------------------------------------
 * Replaces the per-(channel, ping) arithmetic of
 *   calibrate/range.py:138 (k = sample_interval*sound_speed/2), :176-199 (TVG shift),
 *   calibrate/cal_params.py:261-324 (pulse-length table lookup of gain / sa_correction),
 *   calibrate/calibrate_ek.py:98,154-162 (CSv) and :176-181 (CSp).
 * sample_interval, tau_nominal, transmit_power: [C*P].  sound_speed / absorption / gain / sa:
 * pointer + epa_param_mode.  With EPA_PM_PULSE_TABLE, `gain`/`sa` are [C*K] tables matched against
 * pulse_length [C*K] by argmin_k |tau_nominal - pulse_length| (first minimum; NaN tau -> NaN).
 * psi (equivalent_beam_angle): pointer + epa_param_mode (scalar, [C] or [C*P]: calibrate_ek.py:154-162 broadcasts any
 * (channel, ping_time) cal parameter into CSv; Sv only).  f_nominal: [C].  tau_eff: [C] (EPA_PM_CHANNEL) or [C*P]
 * (EPA_PM_CHANNEL_PING: an EK80 file with several filter_time intervals has one effective pulse length per (channel,
 * interval), calibrate/api.py:125-197).  gpt: [C] bytes or NULL (EK80: channels with GPT transceivers).
 * coef out: [C*P*EPA_NCOEF].
 */
int epa_power_coef_ek(int C, int P, const double* sample_interval, const double* tau_nominal,
                      const double* transmit_power, const double* sound_speed, int ss_mode,
                      const double* absorption, int abs_mode, const double* gain, int gain_mode,
                      const double* sa_correction, int sa_mode, const double* pulse_length, int K,
                      const double* psi, int psi_mode, const double* f_nominal, const double* tau_eff,
                      int tau_eff_mode, const uint8_t* gpt, int sonar, int cal_type, double* coef, epa_stream_t stream);

/* The pulse-length table lookup on its own (calibrate/cal_params.py:261-324 get_vend_cal_params_power): out[c,p] =
 * table[c, argmin_k |tau_nominal[c,p] - pulse_length[c,k]|] (first minimum; NaN tau -> NaN).  What compute_Sv attaches to
 * its output as gain_correction / sa_correction (channel, ping_time) when the per-ping parameters live in HBM.
 * tau_nominal, out: f64 [C*P]; pulse_length, table: f64 [C*K]. */
int epa_pulse_table_lookup(const double* tau_nominal, const double* pulse_length, const double* table, int C, int P,
                           int K, double* out, epa_stream_t stream);

/* ---- K1: fused power-sample calibration (EK60, EK80 CW power, AZFP) ---------------------------------
 * Replaces calibrate/range.py:98-201 + calibrate/calibrate_ek.py:104-110,154-184 (EK) and
 * calibrate/range.py:69-95 + calibrate/calibrate_azfp.py:64-97 (AZFP): ~15 whole-array passes
 * become one.  raw: f32 [C*P*S] (backscatter_r).  out: Sv or TS, [C*P*S] of out_dtype.
 * range_out: echo_range [C*P*S] of out_dtype; may be NULL.
 */
int epa_sv_power(const float* raw, const double* coef, int C, int P, int S, int cal_type,
                 unsigned flags, void* out, void* range_out, int out_dtype, epa_stream_t stream);
/* The same with {nanmin, nanmax, NaN count} of the echo_range written as a by-product (range_stats_out f64 [3],
 * as epa_nanminmax would give): what compute_MVBS (api.py:108-110), compute_NASC and the masks ask of the range
 * variable next, without another sweep of it.  workspace: f64 [EPA_SV_POWER_STATS_WS_DOUBLES(S)]. */
#define EPA_SV_POWER_STATS_WS_DOUBLES(S) (3 * ((size_t)(S) / 256 + 8192 + 1024))
int epa_sv_power_stats(const float* raw, const double* coef, int C, int P, int S, int cal_type,
                       unsigned flags, void* out, void* range_out, int out_dtype, double* workspace,
                       double* range_stats_out, epa_stream_t stream);
/* range_out may be NULL there (S even for F64 / S % 4 == 0 for F32, 16-byte aligned buffers; EPA_EUNSUPPORTED
 * otherwise): the statistics are taken of the echo_range that WOULD be written and the array itself -- 8 of the 20
 * bytes per sample the pass moves -- is left to this entry for whoever asks for it later (the reference's echo_range is
 * just as lazy when the echodata is dask-backed; range.py:98-157 with the NaN mask of :143-148 under
 * EPA_FLAG_MASK_RANGE, then raw is read).  epa_mvbs / epa_noise_estimate / epa_noise_apply take the same coefficient
 * rows in place of the array. */
int epa_range_power(const float* raw, const double* coef, int C, int P, int S, unsigned flags, void* range_out,
                    int out_dtype, epa_stream_t stream);

/* ---- time-bin CSR for the binned reductions ----------------------------------------------------------
 * Replaces the pandas-resample bin assignment of commongrid/api.py:118-128.  ping_time: int64 ns,
 * sorted ascending, [P].  Bin b covers [t0 + b*dt, t0 + (b+1)*dt) (or (..] when closed right).
 * bin_start out: int32 [n_bins + 1]; pings bin_start[b] .. bin_start[b+1]-1 belong to bin b.
 */
int epa_time_bin_offsets(const int64_t* ping_time, int P, int64_t t0, int64_t dt, int n_bins,
                         unsigned flags, int32_t* bin_start, epa_stream_t stream);

/* ---- K1+K5: fused compute_Sv -> compute_MVBS -----------------------------------------------------------
 * One pass over the raw power: writes Sv (optional) and the MVBS grid.  Replaces K1's references
 * plus commongrid/utils.py:592,614-627 (flox group-by nanmean in the linear domain) and :92.
 *   bin_start  : int32 [n_tbins+1] (epa_time_bin_offsets);  ping_perm: int32 [.] or NULL (identity)
 *   range bins : edges e_i = i * range_bin, i = 0..n_rbins (np.arange(0, max+bin, bin), api.py:115)
 *   sv_out     : [C*P*S] of dtype or NULL;  range_out as in epa_sv_power or NULL
 *   mvbs_out   : f64/f32 [C*n_tbins*n_rbins] in dB (fill_value where a bin is empty)
 *   sum_out/cnt_out : optional raw linear sums (dtype) / counts (u32) per bin, same shape, for
 *                 cross-shard merges; may be NULL
 *   range_max_out : optional f64 [1]: nanmax(echo_range) as a by-product (what api.py:108-110 needs
 *                 to size the range grid: call with a conservative n_rbins, then trim); NULL if not
 *                 wanted.
 *   range_stats_out : optional f64 [3] (needs range_max_out): {nanmin, nanmax, NaN count} of the echo_range array
 *                 epa_range_power would write (rounded to dtype) -- what the next compute_MVBS / add_depth on
 *                 the same dataset asks of its range variable; {NaN, NaN, -1} when the configuration is served
 *                 by the generic kernel, which leaves the maximum only.
 */
int epa_sv_mvbs_fused(const float* raw, const double* coef, int C, int P, int S, int cal_type,
                      unsigned cal_flags, const int32_t* bin_start, const int32_t* ping_perm,
                      int n_tbins, double range_bin, int n_rbins, unsigned bin_flags,
                      double fill_value, void* sv_out, void* range_out, void* mvbs_out,
                      void* sum_out, uint32_t* cnt_out, double* range_max_out, double* range_stats_out,
                      int dtype, epa_stream_t stream);

/* Same pass fed with the instrument's own int16 power samples (SURVEY 8f "next" row 4; replaces the
 * ingest arithmetic of convert/parse_base.py:24,302 -- float32(int16) * float32(10*log10(2)/256) --
 * and the NaN padding of short pings, pad_shorter_ping): raw int16 [C*P*S], n_valid int32 [C*P] =
 * recorded length of each ping (samples at or beyond it are padding = NaN).  2 B/sample of input
 * instead of 4.  Default configuration only (R' <= 0 guard, echo_range masked by padding, skipna,
 * left-closed bins, sorted pings, S % 4 == 0); other arguments as epa_sv_mvbs_fused.
 */
int epa_sv_mvbs_fused_i16(const int16_t* raw, const int32_t* n_valid, const double* coef, int C, int P,
                          int S, int cal_type, const int32_t* bin_start, int n_tbins, double range_bin,
                          int n_rbins, double fill_value, void* sv_out, void* mvbs_out, void* sum_out,
                          uint32_t* cnt_out, double* range_max_out, int dtype, epa_stream_t stream);

/* The same pass binned on ``depth`` -- consolidate.add_depth between compute_Sv and compute_MVBS(range_var="depth")
 * (SURVEY 8f "next" row 1).  Replaces consolidate/api.py:221 (depth = transducer_depth + orientation * echo_range *
 * echo_range_scaling: one multiplication and one addition per sample, each rounded, in dtype) fused into K1+K5, so the
 * three reference calls cost the 12 B/sample of the two: the depth array is an affine function of the coefficient
 * rows and is written only when asked for (depth_out, +8 B/sample).
 *   depth_scale, depth_offset : f64 [C*P] -- orientation * echo_range_scaling and transducer_depth per (channel, ping)
 *   depth_out       : [C*P*S] of dtype or NULL (needs sv_out); NaN where the raw sample is, like the echo_range
 *   depth_stats_out : f64 [64], 128-byte aligned: {nanmin, nanmax, NaN count} of the depth array in slots 0..2 (the rest
 *                     is scratch) -- what sizes the range grid (commongrid/api.py:108-115): call with a conservative
 *                     n_rbins, then trim
 * Default configuration only (R' <= 0 guard, echo_range masked by NaN input, skipna, left-closed bins, sorted pings,
 * S % 4 == 0, grid within LDS): EPA_EUNSUPPORTED otherwise.  Other arguments as epa_sv_mvbs_fused.
 */
int epa_sv_mvbs_fused_depth(const float* raw, const double* coef, const double* depth_scale,
                            const double* depth_offset, int C, int P, int S, int cal_type, const int32_t* bin_start,
                            int n_tbins, double range_bin, int n_rbins, double fill_value, void* sv_out,
                            void* depth_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out, double* depth_stats_out,
                            int dtype, epa_stream_t stream);

/* ---- K5: compute_MVBS on an existing Sv dataset -----------------------------------------------------------
 * Replaces commongrid/utils.py:504-628 (+ :92).  sv: [C*P*S] of dtype.  Range coordinate either
 * as a full array `range` ([C*P*S], f64 when dtype F64 else f32; echo_range or depth) or, when
 * `range` is NULL, as the affine form of `coef` (R = r0 + s*k, valid everywhere).
 */
int epa_mvbs(const void* sv, const void* range, const double* coef, int C, int P, int S,
             const int32_t* bin_start, const int32_t* ping_perm, int n_tbins, double range_bin,
             int n_rbins, unsigned bin_flags, double fill_value, void* mvbs_out, void* sum_out,
             uint32_t* cnt_out, int dtype, epa_stream_t stream);

/* Self-test hook: out[i] = 10^(u[i]/10) evaluated with the table-driven f64 routine the fused kernel
 * uses for the linear-domain average (_log2lin, utils/compute.py:14-27); u, out f64 [n]. */
int epa_selftest_lin_from_db(const double* u, double* out, size_t n, epa_stream_t stream);
/* Same for the table-driven f64 log10 used by _lin2log and the transmission-loss terms. */
int epa_selftest_log10(const double* x, double* out, size_t n, epa_stream_t stream);
/* The call-free variant of that log10 (every special case folded into selects) the chain kernels use. */
int epa_selftest_log10_inline(const double* x, double* out, size_t n, epa_stream_t stream);

/* Finalise merged partial sums: out = 10*log10(sum/cnt), fill_value where cnt == 0. */
int epa_mvbs_finalize(const void* sum, const uint32_t* cnt, size_t n, double fill_value, void* out,
                      int dtype, epa_stream_t stream);

/* ---- cross-shard edge exchange (no reference counterpart: the reference is single-process) -------------
 * A ping-sharded dataset sums an MVBS time bin over ALL its pings (commongrid/utils.py:614-627) and takes a background-
 * noise block's mean before the minimum over range blocks (clean/api.py:402-411): the raw linear (sum, count) rows of
 * the bins / blocks cut by a shard or tile edge are exchanged.  One step = epa_edge_pack -> ONE all-reduce(SUM) of
 * `buf` (torch.distributed: RCCL over xGMI, the buffer stays in HBM) -> epa_edge_gather.
 * buf: f64 [n_slots * 2 * C * R], slot layout (slot, {sum, count}, channel, range bin).
 * epa_edge_pack: HOST tables of n_rows device rows: element (c, r) of row i at sum_rows[i][c * chan_stride[i] + r]
 * (sum_dtype EPA_F32 / EPA_F64) and cnt_rows[i][...] (u32), written to slot slots[i]; zero_first clears the whole
 * buffer before (slots of absent edges must be 0 for the SUM).
 * epa_edge_gather: for local shared edge e, the sum of the slots group_slots[group_off[e] .. group_off[e+1]) (device
 * int32 tables built once per plan), added in table order -- the same order on every rank, so every holder of a bin
 * reads bit-identical totals.  Outputs (each optional): totals_out f64 [n_edges * 2 * C * R] (sum and count planes, what
 * epa_noise_finalize takes); sum_out [n_edges * C * R] of sum_dtype + cnt_out u32 [n_edges * C * R] (what
 * epa_mvbs_finalize takes).
 * epa_edge_finalize_mvbs: gather + 10 log10(sum / count) (fill_value where the count is 0; the arithmetic of
 * epa_mvbs_finalize on the totals rounded to dtype) of the listed edges (HOST tables: edges[i] = index into group_off,
 * dst_rows[i] = device row with element (c, r) at [c * chan_stride[i] + r] of dtype) -- the owner of a cut time bin
 * writes the bin's row of its MVBS array in one launch. */
int epa_edge_pack(const void* const* sum_rows, const uint32_t* const* cnt_rows, const long long* chan_stride,
                  const int* slots, int n_rows, int sum_dtype, int C, int R, int n_slots, int zero_first, double* buf,
                  epa_stream_t stream);
int epa_edge_gather(const double* buf, int n_slots, const int32_t* group_off, const int32_t* group_slots, int n_edges,
                    int C, int R, double* totals_out, void* sum_out, uint32_t* cnt_out, int sum_dtype,
                    epa_stream_t stream);
int epa_edge_finalize_mvbs(const double* buf, int n_slots, const int32_t* group_off, const int32_t* group_slots,
                           const int* edges, void* const* dst_rows, const long long* chan_stride, int n_rows, int C,
                           int R, double fill_value, int dtype, epa_stream_t stream);
/* nanmax(echo_range) of a shard before its all-reduce(MAX) over the ranks (commongrid/api.py:108-115 takes the maximum of
 * the WHOLE dataset): NaN (a shard without a valid range) -> -inf, in place, n device doubles. */
int epa_edge_prepare_max(double* values, int n, epa_stream_t stream);

/* ---- depth = offset[c,p] + scale[c,p] * echo_range -----------------------------------------------------------
 * The array pass of consolidate.add_depth (consolidate/api.py:226: transducer_depth +
 * orientation * echo_range * cos(tilt)); SURVEY 8f "next" row 1.  x, out: [C*P*S] of dtype;
 * scale, offset: f64 [C*P].
 */
int epa_affine_rows(const void* x, const double* scale, const double* offset, int C, int P, int S,
                    void* out, int dtype, epa_stream_t stream);
/* The same with (i) echo_range either as the array `range` or -- range NULL -- evaluated from the power-sample
 * coefficient rows `coef`, NaN where mask_raw (float [C*P*S], optional) is NaN: an echo_range left out of the sample
 * pass (epa_sv_power_stats with range_out = NULL) never has to be written for add_depth; (ii) {nanmin, nanmax, NaN
 * count} of the depth as a by-product (stats_out f64 [3], workspace f64 [EPA_DEPTH_ROWS_WS_DOUBLES]; both optional):
 * what compute_MVBS(range_var="depth") asks of the variable next (commongrid/api.py:108-110). */
#define EPA_DEPTH_ROWS_WS_DOUBLES (3 * 16384)
int epa_depth_rows(const void* range, const double* coef, const float* mask_raw, const double* scale,
                   const double* offset, int C, int P, int S, void* out, int dtype, double* workspace,
                   double* stats_out, epa_stream_t stream);

/* ---- NaN-skipping min/max of a device array ---------------------------------------------------------------
 * Replaces the reductions the reference forces with ds_Sv[range_var].max(skipna=True)
 * (commongrid/api.py:108-110) and the actual_range attributes (clean/utils.py:392-395,
 * commongrid/api.py:252-255).  x: [n] of dtype; workspace: f64 [3072]; out: f64 [3] = {min, max,
 * number of NaN elements} (min = max = NaN when no element is non-NaN).  The NaN count serves the
 * "coordinate array contain NaNs" warning of commongrid/utils.py:595-608 without another sweep.
 */
int epa_nanminmax(const void* x, size_t n, int dtype, double* workspace, double* out,
                  epa_stream_t stream);

/* ---- K5': compute_MVBS_index_binning ------------------------------------------------------------------------
 * Replaces commongrid/api.py:217-222 (coarsen(ping_num, range_sample_num, "pad").mean(skipna) in the
 * linear domain) and :232-238 (echo_range block min).  Outputs [C * ceil(P/ping_num) *
 * ceil(S/range_sample_num)].  range / range_min_out may be NULL.
 */
int epa_mvbs_index(const void* sv, const void* range, int C, int P, int S, int ping_num,
                   int range_sample_num, void* mvbs_out, void* range_min_out, int dtype,
                   epa_stream_t stream);

/* ---- K6: background-noise estimate (De Robertis & Higginbottom 2007) -------------------------------------------
 * Replaces clean/api.py:397-422: TL = 20log10(max(R,1) [NaN->1]) + 2*alpha*R; block mean of
 * 10^((Sv-TL)/10) over ping_num x range_sample_num (NaN-padded, NaN-skipping) -> dB -> min over
 * range blocks -> optional clamp to noise_max (NaN when noise_max is NaN = not given).
 * alpha2: 2*alpha per (c,p) [C*P] f64.
 * ping_phase (0 .. ping_num-1): the arrays hold a ping shard of a longer dataset whose first ping sits
 * `ping_phase` pings into a block (global ping index of the shard's first ping modulo ping_num; 0 for a whole
 * dataset): local ping p belongs to block (p + ping_phase) / ping_num.
 * noise_out: f64 [C * ceil((P + ping_phase) / ping_num)].
 * edge_sum_out / edge_cnt_out (optional, both or neither; f64 / u32 [2 * C * ceil(S/range_sample_num)], zeroed by the
 * caller): the raw linear (sum, count) per range block of the shard's FIRST (slot 0) and LAST (slot 1, written only
 * when it is another block) ping block -- blocks a shard edge may cut; the owner merges them across shards before the
 * mean and the min, then epa_noise_finalize.
 */
int epa_noise_estimate(const void* sv, const void* range, const double* coef, const double* alpha2,
                       int C, int P, int S, int ping_num, int range_sample_num, int ping_phase, double noise_max,
                       double* noise_out, double* edge_sum_out, uint32_t* edge_cnt_out, int dtype,
                       epa_stream_t stream);

/* Merged (sum, count) rows of noise blocks -> noise value per row (clean/api.py:402-422: mean -> dB -> min over the
 * range blocks -> clamp).  sum, cnt: f64 [rows * n_rblocks] (the counts travel through an all-reduce with the sums);
 * noise_out: f64 [rows]. */
int epa_noise_finalize(const double* sum, const double* cnt, int rows, int n_rblocks, double noise_max,
                       double* noise_out, epa_stream_t stream);

/* ---- K7: noise removal -------------------------------------------------------------------------------------------
 * Replaces clean/api.py:425-430 (ffill upsample + TL) and :485-487.  Writes Sv_noise and
 * Sv_corrected ([C*P*S] of dtype; either may be NULL).  snr_threshold in dB.  ping_phase as in
 * epa_noise_estimate (noise holds ceil((P + ping_phase) / ping_num) blocks per channel).  minmax_out (f64 [4],
 * optional): NaN-skipping {min, max} of Sv_noise and of Sv_corrected as a by-product (the actual_range
 * attributes of clean/utils.py:392-395 without two more sweeps).
 */
int epa_noise_apply(const void* sv, const void* range, const double* coef, const double* alpha2,
                    const double* noise, int C, int P, int S, int ping_num, int ping_phase, double snr_threshold,
                    void* sv_noise_out, void* sv_corrected_out, double* minmax_out, int dtype,
                    epa_stream_t stream);
/* The same on an Sv whose echo_range was left out of the sample pass (epa_sv_power_stats with range_out = NULL): the
 * range is evaluated from the power-sample coefficient rows and -- mask_raw, float [C*P*S], optional -- set to NaN
 * where the raw sample is NaN, as the echo_range array would be (range.py:143-148; Sv_noise is NaN there, api.py:425-430).
 * 4 B/sample of raw instead of 8 (or 4) of echo_range, and the array is never written. */
int epa_noise_apply_rows(const void* sv, const double* coef, const float* mask_raw, const double* alpha2,
                         const double* noise, int C, int P, int S, int ping_num, int ping_phase,
                         double snr_threshold, void* sv_noise_out, void* sv_corrected_out, double* minmax_out,
                         int dtype, epa_stream_t stream);

/* ---- K3+K4: EK80 complex samples (CW complex and BB pulse compression) ---------------------------------------------
 * Replaces calibrate/ek80_complex.py:285-369 (compress_pulse: matched filter with the transmit
 * replica per ping and sector), :372-391 (norm factor), calibrate_ek.py:483-490 (received power
 * from the sector mean) and :571-638 (Sv / TS chain).
 *   re, im   : backscatter_r / backscatter_i as stored by echopype: f64 (in_dtype F64) or f32,
 *              [C*P*S*B], beam innermost, NaN-padded
 *   replica  : float2-interleaved complex f32 [sum(taps)], conj NOT applied, per channel at
 *              replica_off[c] .. replica_off[c+1] (int32 [C+1], device); both NULL for CW (no
 *              pulse compression).  max_taps = longest replica (sizes the LDS tile).
 *   ccoef    : per-(c,p) rows of EPA_NCCOEF doubles, see enum below.  PSCALE excludes the
 *              1/||tx||^4 of the pulse-compression normalisation: the kernel computes ||tx||^2
 *              itself (wavefront shuffle reduction over the LDS-resident replica)
 *   out      : Sv/TS [C*P*S] of out_dtype; range_out, prx_out (same dtype) optional.  A sample whose beam-0 real
 *              part is NaN is NaN in out and range_out whatever its other sectors hold: the reference calibrates
 *              with the masked echo_range (range.py:143-148, calibrate_ek.py:571-576); prx_out is unaffected.
 *              F64 output accumulates the matched filter in f64, F32 in f32
 */
#define EPA_NCCOEF 8
/* echo_range R = fl(fl(s*RA)*RB) (RA = sample_interval, RB = sound_speed/2: range.py:138 order);
 * R' = R - SHIFT (<= 0 -> NaN); out = 10log10(prx) + n*log10(R') + ALPHA2*R' + A;
 * prx = PSCALE * |sector mean of the (pulse-compressed, normalised) samples|^2  (<= 0 -> NaN) */
enum epa_ccoef_slot { EPA_CC_RA = 0, EPA_CC_RB = 1, EPA_CC_SHIFT = 2, EPA_CC_ALPHA2 = 3, EPA_CC_A = 4,
                      EPA_CC_PSCALE = 5, EPA_CC_RSV0 = 6, EPA_CC_RSV1 = 7 };
/* The rows above built on the device from the per-(channel, ping) parameters (the EK80 counterpart of
 * epa_power_coef_ek; replaces the (channel, ping_time) arithmetic of calibrate_ek.py:483-490, 507-530, 583-638 and
 * range.py:180-199).  params / modes: HOST arrays of EPA_CCP_COUNT DEVICE pointers (f64) and their epa_param_mode
 * (SCALAR, CHANNEL [C] or CHANNEL_PING [C*P]), indexed by epa_ccoef_param; entries a mode does not use may be NULL
 * (the angle offsets / beamwidths without bb; sa_correction with bb or TS; psi with TS).  tau_eff: f64 [C] or [C*P] (device;
 * tau_eff_mode EPA_PM_CHANNEL / EPA_PM_CHANNEL_PING as for epa_power_coef_ek);
 * gpt: u8 [C] or NULL.  bb != 0: broadband (gain is compensated by B(theta, phi), no sa_correction). */
enum epa_ccoef_param { EPA_CCP_SAMPLE_INTERVAL = 0, EPA_CCP_TAU_NOMINAL, EPA_CCP_TRANSMIT_POWER, EPA_CCP_SOUND_SPEED,
                       EPA_CCP_ABSORPTION, EPA_CCP_GAIN, EPA_CCP_FREQ_CENTER, EPA_CCP_PSI, EPA_CCP_SA_CORRECTION,
                       EPA_CCP_Z_ER, EPA_CCP_Z_ET, EPA_CCP_ANGLE_OFFSET_ALONGSHIP, EPA_CCP_ANGLE_OFFSET_ATHWARTSHIP,
                       EPA_CCP_BEAMWIDTH_ALONGSHIP, EPA_CCP_BEAMWIDTH_ATHWARTSHIP, EPA_CCP_COUNT };
int epa_complex_coef_ek80(int C, int P, const double* const* params, const int* modes, const double* tau_eff,
                          int tau_eff_mode, const uint8_t* gpt, int B, int bb, int cal_type, double* ccoef,
                          epa_stream_t stream);

int epa_sv_complex(const void* re, const void* im, int in_dtype, const float* replica,
                   const int32_t* replica_off, int max_taps, const double* ccoef, int C, int P,
                   int S, int B, int cal_type, void* out, void* range_out, void* prx_out,
                   int out_dtype, epa_stream_t stream);
/* The same with one replica per (channel, filter interval) instead of one per channel -- an EK80 file with several
 * filter_time entries, which the reference calibrates slice by slice and merges (calibrate/api.py:125-197): ONE launch
 * over the whole (channel, ping_time) grid.  replica_off: int32 [n_replicas + 1]; replica_id: int32 [C*P], the replica
 * of every ping, -1 for a ping no interval covers (give it a NaN coefficient row: its output and range are NaN, the
 * outer join's fill). */
int epa_sv_complex_indexed(const void* re, const void* im, int in_dtype, const float* replica,
                           const int32_t* replica_off, const int32_t* replica_id, int n_replicas, int max_taps,
                           const double* ccoef, int C, int P, int S, int B, int cal_type, void* out, void* range_out,
                           void* prx_out, int out_dtype, epa_stream_t stream);

/* Same result for long replicas through an LDS-resident 2048-point FFT per tile (the matched filter as a
 * circular correlation; scipy.signal.convolve's method="auto" makes the same switch in the reference,
 * ek80_complex.py:310-313): ~11x fewer flops than the direct form at 177 taps, the kernel becomes HBM-bound
 * like CW.  Replicas of 1 .. EPA_EK80_NFFT/2 taps (EPA_EUNSUPPORTED beyond: use epa_sv_complex).
 *   fft_dtype  : arithmetic of the transform.  EPA_F64: errors ~1e-16 of the strongest echo of the 2048-sample
 *                tile (the reference convolves in complex128 and stores complex64, ek80_complex.py:304-313).
 *                EPA_F32: complex64 butterflies, errors ~3e-7 of the tile's strongest echo -- for float32 output.
 *   workspace  : f64 [EPA_EK80_FFT_WS_DOUBLES(C, P, S)] (twiddles; ||tx||^2, span of the non-zero taps and
 *                conj(FFT(tx))/N per channel; range-statistics slots; one bit per tile marking tiles with a
 *                partly-NaN sample, which a second launch redoes sector by sector; the time-varied gain
 *                n log10(R') + 2 alpha R' of each channel's first ping by sample index, read instead of computed
 *                by every ping with the same range numbers; the logarithm's lookup table; rebuilt by every call)
 *   range_stats_out : optional f64 [3] = {nanmin, nanmax, NaN count} of the echo_range (written to range_out, or --
 *                range_out NULL -- left to epa_range_complex for whoever reads the array later), a by-product of the
 *                same pass (what compute_MVBS, commongrid/api.py:108-110, asks next) */
#define EPA_EK80_NFFT 2048
#define EPA_EK80_FFT_WS_DOUBLES(C, P, S)                                              \
  (768 + 4 * (size_t)(C) + 3 * (size_t)(C) * EPA_EK80_NFFT + 3 * 1024 + 2 +          \
   ((size_t)(C) * (size_t)(P) * ((size_t)(S) / (EPA_EK80_NFFT / 2 + 1) + 1) + 63) / 64 + \
   (size_t)(C) * ((size_t)(S) + 4) + 256)
int epa_sv_complex_fft(const void* re, const void* im, int in_dtype, const float* replica,
                       const int32_t* replica_off, int max_taps, const double* ccoef, int C, int P,
                       int S, int B, int cal_type, void* out, void* range_out, void* prx_out,
                       int out_dtype, int fft_dtype, double* workspace, double* range_stats_out,
                       epa_stream_t stream);
/* ... with one replica per (channel, filter interval), as epa_sv_complex_indexed.  workspace: f64
 * [EPA_EK80_FFT_WS_DOUBLES(max(C, n_replicas), P, S)]. */
int epa_sv_complex_fft_indexed(const void* re, const void* im, int in_dtype, const float* replica,
                               const int32_t* replica_off, const int32_t* replica_id, int n_replicas, int max_taps,
                               const double* ccoef, int C, int P, int S, int B, int cal_type, void* out,
                               void* range_out, void* prx_out, int out_dtype, int fft_dtype, double* workspace,
                               double* range_stats_out, epa_stream_t stream);

/* epa_sv_complex on CW samples (no replica) with {nanmin, nanmax, NaN count} of the echo_range as a by-product
 * (range_stats_out f64 [3]; merged through 1024 slots of f64 atomics in workspace, f64
 * [EPA_SV_COMPLEX_CW_STATS_WS_DOUBLES]); range_out may be NULL: the statistics are then those of the array
 * epa_range_complex would write.  Other arguments as epa_sv_complex. */
#define EPA_SV_COMPLEX_CW_STATS_WS_DOUBLES 3072
int epa_sv_complex_cw_stats(const void* re, const void* im, int in_dtype, const double* ccoef, int C, int P, int S,
                            int B, int cal_type, void* out, void* range_out, void* prx_out, int out_dtype,
                            double* workspace, double* range_stats_out, epa_stream_t stream);

/* echo_range of complex samples alone (range.py:98-157: (range_sample * sample_interval) * sound_speed / 2, NaN where
 * the real part of sector 0 is, :143-148) -- what epa_sv_complex / epa_sv_complex_fft write as range_out, for a caller
 * that left it out of the sample pass.  re as there ([C*P*S*B], in_dtype), range_out [C*P*S] of out_dtype. */
int epa_range_complex(const void* re, int in_dtype, const double* ccoef, int C, int P, int S, int B,
                      void* range_out, int out_dtype, epa_stream_t stream);

/* ==== SURVEY 8f "next" row 2: Ryan et al. (2015) noise masks + apply_mask ==============================
 * Masks are uint8 [C*P*S] (1 = True) in the (channel, ping_time, range_sample) layout of Sv.        */

/* Depth-bin smoothing used by mask_impulse_noise: every sample takes 10*log10 of the NaN-skipping
 * linear mean of its depth bin within its own ping.  Replaces clean/utils.py:244-304
 * (index_binning_downsample_upsample_along_depth: coarsen(range_sample=nper, "pad").mean + ffill;
 * range == NULL, bin b = s / nper -- call per channel, nper is channel specific) and :173-241
 * (downsample_upsample_along_depth: flox nanmean over bins np.arange(r0, max + bin, bin) closed on
 * the left, np.digitize up-sampling; range != NULL, nbins = len(edges) - 1).
 * sv, range, up_out: [C*P*S] of dtype.  (A ping with more bins than the LDS accumulators hold, ~13 000, is processed
 * in segments of bins.) */
int epa_range_bin_smooth(const void* sv, const void* range, int C, int P, int S, int nper, double r0,
                         double bin, int nbins, void* up_out, int dtype, epa_stream_t stream);

/* Two-sided ping comparison (clean/utils.py:307-323 echopy_impulse_noise_mask):
 * mask = (up[p] - up[p+n] > thr) & (up[p] - up[p-n] > thr), NaN differences count as +inf. */
int epa_impulse_mask(const void* up, int C, int P, int S, int num_side_pings, double threshold,
                     uint8_t* mask_out, int dtype, epa_stream_t stream);

/* Pooled Sv of mask_transient_noise, index-binning variant (clean/utils.py:109-170: dask_image
 * generic_filter over a (2*num_side_pings+1) x (2*num_side_samples+1) window of the linear Sv,
 * mode="reflect", restricted to range_sample >= first_sample; NaN above) and the mask
 * Sv - pooled > threshold (clean/api.py:166).  func: EPA_POOL_NANMEAN (separable box sums; needs
 * ws_sum f64 [C*P*S] and ws_cnt int32 [C*P*S]) or EPA_POOL_NANMEDIAN (exact; the window carried from ping to
 * ping with its histogram when it is at most 256 samples wide and fits LDS, a radix selection per output
 * sample otherwise; workspaces unused).
 * pooled_out ([C*P*S] of dtype) and mask_out may each be NULL. */
enum epa_pool_func { EPA_POOL_NANMEAN = 0, EPA_POOL_NANMEDIAN = 1 };
int epa_pool_sv(const void* sv, int C, int P, int S, int first_sample, int num_side_pings,
                int num_side_samples, int func, double threshold, void* pooled_out, uint8_t* mask_out,
                double* ws_sum, int32_t* ws_cnt, int dtype, epa_stream_t stream);

/* Attenuated-signal mask (clean/utils.py:326-372 echopy_attenuated_signal_mask, per channel):
 * per ping, up / lw = argmin |range - limit| of that ping (first NaN if any, as np.argmin); when
 * p-n >= 0, p+n <= P-1 and Sv[p, up:lw] is not all NaN, the whole ping is masked if
 * 10log10(nanmedian lin Sv[p, up:lw]) - 10log10(nanmedian lin Sv[p-n:p+n, up:lw]) < threshold.
 * S % 4 == 0, S >= 16 and a 4-byte aligned mask_out take the carried-block route (mask_out's rows hold the
 * per-ping limits and medians between its two kernels), anything else two selections per ping. */
int epa_attenuated_mask(const void* sv, const void* range, int C, int P, int S, double upper_limit,
                        double lower_limit, int num_side_pings, double threshold, uint8_t* mask_out,
                        int dtype, epa_stream_t stream);

/* out[i] = mask[i % mask_period] ? src[i] : fill  (mask/api.py:428-432 xr.where(final_mask, Sv,
 * fill_value)); fill = fill_array[i % fill_period] when fill_array != NULL, else fill_value.
 * mask_period = P*S broadcasts a channel-less mask over the channels. */
int epa_apply_mask(const void* src, const uint8_t* mask, size_t n, size_t mask_period,
                   double fill_value, const void* fill_array, size_t fill_period, void* out,
                   int dtype, epa_stream_t stream);

/* out[i] = (masks[0][i % p0] & ... & masks[n_masks-1][..]) ? src[i] : fill -- mask/api.py:402-432 in ONE sweep for a LIST
 * of masks (np.logical_and.reduce over the broadcast masks, :405-408, then xr.where, :428-432): 1 <= n_masks <= 4 (AND
 * further ones with epa_mask_and first); ``masks`` / ``mask_periods`` are HOST arrays of device pointers / periods.
 * minmax_out != NULL: f64 [2] on the device = NaN-skipping {min, max} of what was written (the variable's actual_range,
 * mask/api.py:434-444 via clean/utils.py:392-395), with workspace = f64 [EPA_APPLY_MASKS_WS_DOUBLES] on the device. */
#define EPA_APPLY_MASKS_WS_DOUBLES 131072
int epa_apply_masks(const void* src, const uint8_t* const* masks, const size_t* mask_periods, int n_masks, size_t n,
                    double fill_value, const void* fill_array, size_t fill_period, void* out, double* workspace,
                    double* minmax_out, int dtype, epa_stream_t stream);

/* out = a & b[i % b_period]  (mask/api.py:405-408, np.logical_and.reduce over broadcast masks). */
int epa_mask_and(const uint8_t* a, const uint8_t* b, size_t n, size_t b_period, uint8_t* out,
                 epa_stream_t stream);

/* Mean range step per channel, np.nanmean(np.diff(range, axis=2), axis=(1, 2)) (clean/utils.py:131,
 * :258: sizes the per-channel index bins).  workspace: f64 [2*C*P]; out: f64 [C] (NaN: no finite step). */
int epa_range_step_mean(const void* range, int C, int P, int S, int dtype, double* workspace,
                        double* out, epa_stream_t stream);

/* Flat index of the first element that is not <= limit (NaN counts); n when there is none.
 * np.argmin(range <= exclude_above) of clean/utils.py:143.  x: [n] of dtype; out: uint64 [1]. */
int epa_first_not_le(const void* x, size_t n, double limit, int dtype, uint64_t* out,
                     epa_stream_t stream);

/* Row check of a range variable for the value-window pooling below: nvalid_out int32 [C*P] = index of
 * the first NaN of each (c,p) row (S if none); violations_out int32 [1] = number of rows that are not
 * non-decreasing over their valid prefix or hold a number after their first NaN. */
int epa_range_rows_check(const void* range, int C, int P, int S, int dtype, int32_t* nvalid_out,
                         int32_t* violations_out, epa_stream_t stream);

/* Pooled Sv of mask_transient_noise, value-window variant (clean/utils.py:29-106 pool_Sv, a triple
 * Python loop in the reference): for the sample at depth d of ping p, func over the linear Sv of pings
 * p-n..p+n (clipped to the data) with range in [d - depth_bin, d + depth_bin]; NaN where
 * d - bin < range_min, d + bin > range_max, d - bin < exclude_above, p - n < 0 or p + n > P.
 * range rows must pass epa_range_rows_check (nvalid from it).  func / threshold / outputs as epa_pool_sv. */
#define EPA_POOL_VALUE_WS_BYTES(C, P, S) \
  ((size_t)(C) * (P) * (S) * 40 + (size_t)(C) * (S) * 8 + (size_t)(C) * 8 + (size_t)(C) * (P))
/* ws (optional, nanmean only): EPA_POOL_VALUE_WS_BYTES bytes, 8-byte aligned.  With it a channel whose pings all share
 * one range vector (checked on the device) gets per-row interval sums (blocks of 16 samples, no subtraction) + a
 * sliding sum down every column, O(1) per sample; any other channel per-row running sums kept in double-double,
 * O(pings) per sample.  Without it every window is summed value by value.  Same results to rounding. */
/* ws for func = nanmedian (optional): EPA_POOL_VALUE_MEDIAN_WS_BYTES bytes, 4-byte aligned.  With it the channels whose
 * pings share one range vector carry their window from ping to ping (a histogram of the window kept up to date, the
 * median's bin ranked exactly): O(window columns) per sample instead of O(window).  Same results. */
#define EPA_POOL_VALUE_MEDIAN_WS_BYTES(C, S) (((size_t)(C) * (S) * 2 + (size_t)(C) * 2) * 4)
int epa_pool_sv_value(const void* sv, const void* range, const int32_t* nvalid, int C, int P, int S,
                      double depth_bin, int num_side_pings, double exclude_above, double range_min,
                      double range_max, int func, double threshold, void* pooled_out, uint8_t* mask_out,
                      void* ws, int dtype, epa_stream_t stream);

/* ==== SURVEY 8f "next" row 3: compute_NASC ==============================================================
 * Array pass of commongrid/utils.py:97-205 (compute_raw_NASC): NASC = sv_mean * h_mean * 4*pi*1852^2 with
 * sv_mean the (nan)mean of the linear Sv per (channel, distance bin, depth bin) and h_mean the nansum of
 * diff(depth) (labelled by its lower sample) per cell divided by the pings of the distance bin.
 *   sv, depth : [C*P*S] of dtype, pings ordered by cumulative distance
 *   bin_start : int32 [n_dbins+1], pings bin_start[d] .. bin_start[d+1]-1 belong to distance bin d
 *   depth bins: edges i*range_bin, i = 0..n_rbins (np.arange(0, max + bin, bin), api.py:351)
 *   bin_flags : EPA_BIN_SKIPNA, EPA_BIN_CLOSED_RIGHT (depth bins; the distance side is in bin_start)
 *   workspace : 24 * C*n_dbins*n_rbins bytes (zeroed by the call); a depth grid too fine for the LDS accumulators
 *               (> ~6500 bins) is accumulated straight into it with atomics
 *   nasc_out  : [C*n_dbins*n_rbins] of dtype; sv_mean_out / h_mean_out likewise, optional */
/* Along-track step of every ping for compute_NASC's distance bins: step_m_out[i] = WGS-84 geodesic length in metres
 * from ping i to ping i + 1 (Vincenty's inverse formula; NaN for the last ping and wherever a position is NaN), the
 * quantity commongrid/utils.py:208-231 takes from geopy.distance.distance ping by ping.  lat / lon: f64 [P], degrees. */
int epa_geodesic_steps(const double* lat, const double* lon, int P, double* step_m_out, epa_stream_t stream);
int epa_nasc(const void* sv, const void* depth, int C, int P, int S, const int32_t* bin_start, int n_dbins,
             double range_bin, int n_rbins, unsigned bin_flags, void* workspace, void* nasc_out,
             void* sv_mean_out, void* h_mean_out, int dtype, epa_stream_t stream);

/* ==== the whole north-star chain in two passes ==========================================================
 * compute_Sv -> remove_background_noise -> compute_MVBS as four separate calls moves 84 B/sample in
 * fp64 (K1 20, K6 16, K7 32, K5 16); these two entry points do the same arithmetic in 12 + 16..24.
 *
 * Pass 1 = K1 + K6: Sv (and optionally echo_range) written once, the noise estimate of
 * clean/api.py:397-422 accumulated from the Sv values still in registers (transmission loss from the
 * coefficient rows).  Arguments as epa_sv_power and epa_noise_estimate.  sv_out may be NULL;
 * range_max_out (f64 [1], optional) = nanmax(echo_range), which sizes the range grid of pass 2;
 * range_stats_out (f64 [3], optional, needs range_max_out) = {nanmin, nanmax, NaN count} of the echo_range as
 * epa_sv_mvbs_fused leaves them ({NaN, NaN, -1} when the generic kernel serves the configuration). */
/* A ping shard of a longer file: ping_phase = (global index of the shard's first ping) % ping_num, i.e. local
 * ping p belongs to noise block (p + ping_phase) / ping_num and noise_out has ceil((P + ping_phase) / ping_num) columns;
 * edge_sum_out f64 [2][C][Sb] / edge_cnt_out u32 [2][C][Sb] (optional, together; zero them first) receive the raw linear
 * (sum, count) per range block of the shard's FIRST and LAST ping block -- what a block cut by a shard edge contributes to
 * the mean over all its pings (clean/api.py:402-411); a shard with a single block fills slot 0 only.  As
 * epa_noise_estimate. */
int epa_sv_noise_fused(const float* raw, const double* coef, const double* alpha2, int C, int P, int S,
                       int cal_type, unsigned cal_flags, int ping_num, int range_sample_num, int ping_phase,
                       double noise_max, void* sv_out, void* range_out, double* noise_out, double* edge_sum_out,
                       uint32_t* edge_cnt_out, double* range_max_out, double* range_stats_out, int dtype,
                       epa_stream_t stream);

/* Pass 2 = K7 + K5: reads Sv once, applies clean/api.py:425-430,485-487 with the per-ping-block noise
 * of pass 1 and bins the CORRECTED Sv (commongrid/utils.py:592-627) in the same sweep.  Arguments as
 * epa_noise_apply (sv, range | coef, alpha2, noise, ping_num, snr_threshold) and epa_mvbs (bins,
 * outputs); sv_noise_out / sv_corrected_out ([C*P*S] of dtype) may each be NULL. */
/* (With `coef` instead of `range`, Sv_noise is finite at NaN-padded samples, where the reference's
 * echo_range -- and therefore its Sv_noise -- is NaN; Sv_corrected and the MVBS are unaffected.) */
int epa_denoise_mvbs(const void* sv, const void* range, const double* coef, const double* alpha2,
                     const double* noise, int C, int P, int S, int ping_num, double snr_threshold,
                     const int32_t* bin_start, const int32_t* ping_perm, int n_tbins, double range_bin,
                     int n_rbins, unsigned bin_flags, double fill_value, void* sv_noise_out,
                     void* sv_corrected_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                     epa_stream_t stream);

/* Pass 2 fed with the RAW power again instead of Sv (4 B/sample read instead of 8, and the NaN padding
 * that masks echo_range is seen directly): Sv is recomputed in registers as in epa_sv_power, then as
 * epa_denoise_mvbs.  With pass 1 the chain costs 4 + 8 (Sv) and 4 + 8 (Sv_corrected) [+ 8 Sv_noise]
 * = 24..32 B/sample.  range_out optional as in epa_sv_power.  minmax_out (f64 [4], optional): NaN-skipping
 * {min, max} of Sv_noise and {min, max} of Sv_corrected as a by-product (the actual_range attributes of
 * clean/utils.py:392-395, otherwise one more sweep of each array). */
/* ping_phase as in epa_sv_noise_fused: the noise of local ping p is noise[c][(p + ping_phase) / ping_num]. */
int epa_sv_denoise_mvbs(const float* raw, const double* coef, const double* alpha2, const double* noise,
                        int C, int P, int S, int cal_type, unsigned cal_flags, int ping_num, int ping_phase,
                        double snr_threshold, const int32_t* bin_start, const int32_t* ping_perm,
                        int n_tbins, double range_bin, int n_rbins, unsigned bin_flags, double fill_value,
                        void* sv_noise_out, void* sv_corrected_out, void* range_out, void* mvbs_out,
                        void* sum_out, uint32_t* cnt_out, double* minmax_out, int dtype, epa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ECHOPYPE_AMD_H */
