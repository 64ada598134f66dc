// Internal helpers shared by the HIP translation units of libechopype_amd.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <atomic>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "echopype_amd.h"

namespace epa {

void set_error(const char* fmt, ...);

#define EPA_CHECK_ARG(cond, ...)       \
  do {                                 \
    if (!(cond)) {                     \
      ::epa::set_error(__VA_ARGS__);   \
      return EPA_EINVAL;               \
    }                                  \
  } while (0)

#define EPA_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::epa::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                       __LINE__);                                                        \
      return EPA_EHIP;                                                                   \
    }                                                                                    \
  } while (0)

// Launch trace (epa_launch_trace in echopype_amd.h): while it is on, every kernel launch notes its name -- the tests
// assert WHICH kernel served a call (a specialised kernel silently declining a shape otherwise passes every
// "fast == generic" comparison as generic == generic).
extern std::atomic<bool> g_trace_on;
void note_launch(const char* what);
// epa_last_range_stats_filled (echopype_amd.h): did the last fused call on this thread leave the range statistics?
void note_range_stats_filled(int filled);

// every distinct kernel name launched by this process (epa_launch_seen): the test suite's kernel coverage
void note_seen(const char* what);

inline int check_launch(const char* what) {
  if (g_trace_on.load(std::memory_order_relaxed)) note_launch(what);
  note_seen(what);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return EPA_EHIP;
  }
  return EPA_OK;
}

constexpr int kBlock = 256;  // 4 wavefronts of 64

// ---- device math, templated on the compute type ------------------------------------------------
template <typename T>
struct M;
template <>
struct M<double> {
  static __device__ __forceinline__ double log10(double x) { return ::log10(x); }
  static __device__ __forceinline__ double exp10(double x) { return ::exp10(x); }
  static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
};
template <>
struct M<float> {
  static __device__ __forceinline__ float log10(float x) { return ::log10f(x); }
  static __device__ __forceinline__ float exp10(float x) { return ::exp10f(x); }
  static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
};

// Per-(channel, ping) coefficient row (EPA_NCOEF doubles, 64 B): two 32-B halves so that a
// wave-uniform row read is two scalar s_load_dwordx8.
struct CoefRow {
  double ra, rb, r0, shift, alpha2, A0, g, d;
};
static_assert(sizeof(CoefRow) == EPA_NCOEF * sizeof(double), "coef row layout");

// Workgroups are dealt to the 8 XCDs round-robin by their linear id.  With this remap of a 1-D work index the
// workgroups of one XCD walk ONE contiguous eighth of the index range instead of every eighth item: on the streaming
// probe (scripts/probes/hbm_mix_probe.hip, 4 B read + 8 B written per sample by workgroups that each walk a long run)
// that is 6.1 instead of 5.5 TB/s.  Items past the last multiple of 8 keep their index.
__device__ __forceinline__ int xcd_contiguous(int x, int n) {
  const int per = n >> 3;
  return x < per * 8 ? (x & 7) * per + (x >> 3) : x;
}
// EPA_XCD_MAP=0 turns the remap off (development knob, read once)
inline bool xcd_map_enabled() {
  static const bool on = [] {
    const char* e = getenv("EPA_XCD_MAP");
    return !(e && e[0] == '0');
  }();
  return on;
}

// Uniform-edge bin index: edges e_i = i*bin (np.arange(0, stop, bin) evaluates 0 + i*bin in
// double), membership decided against those exact edge values (SURVEY A.6 caveat (i)).
// Returns -1 for NaN, out of range.
__device__ __forceinline__ int range_bin_index(double x, double bin, double inv_bin, int nbins,
                                               bool closed_right) {
  if (!(x == x)) return -1;
  double t = x * inv_bin;
  // clamp before the int conversion (inf / huge values)
  if (!(t > -2.0)) return -1;
  if (t > (double)nbins + 2.0) return -1;
  int i;
  if (!closed_right) {
    i = (int)floor(t);
    // fix-up against the true edges
    if (x < (double)i * bin) --i;
    else if (x >= (double)(i + 1) * bin) ++i;
  } else {
    i = (int)ceil(t) - 1;
    if (x <= (double)i * bin) --i;
    else if (x > (double)(i + 1) * bin) ++i;
  }
  return (i >= 0 && i < nbins) ? i : -1;
}

// echo_range of sample s in the reference's operation order (range.py:138: (s*si)*c/2)
__device__ __forceinline__ double row_range(const CoefRow& r, int s) {
  return ((double)s * r.ra) * r.rb + r.r0;
}

// depth of a sample (consolidate/api.py:221: transducer_depth + orientation * echo_range * echo_range_scaling): the
// product is rounded, then the sum -- NumPy's two operations, never one contracted fma -- in the array's own type, so
// that the depth array, and whatever is binned on it without writing it, hold the reference's bits.
// (contraction switched off for the two statements: hipcc's default -ffp-contract=fast turns even __dadd_rn(o, __dmul_rn(a, r))
// into one v_fma_f64)
__device__ __forceinline__ double depth_of(double scale, double offset, double range) {
#pragma clang fp contract(off)
  const double prod = scale * range;
  return offset + prod;
}
__device__ __forceinline__ float depth_of(float scale, float offset, float range) {
#pragma clang fp contract(off)
  const float prod = scale * range;
  return offset + prod;
}

}  // namespace epa

// reduce_util.hip: echo_range / depth from the coefficient rows as one-piece workgroups (-1: the shape is not served)
int epa_rows_piece_launch(const float* mask_raw, const void* x, const double* coef, const double* scale,
                          const double* offset, long long rows, int S, void* out, int dtype, double* workspace,
                          double* stats_out, hipStream_t st);
// reduce_util.hip: {min, max, NaN count} partials (3 doubles per workgroup) -> out[3]
int epa_minmax_final(const double* part, int nparts, double* out, hipStream_t st);
