// Table-driven fp64 10^(u/10) and log10 for the dB <-> linear conversions of the hot path
// (utils/compute.py:14-42 _log2lin / _lin2log, used by every reduction and by the noise removal).
//
// ocml's fp64 exp10 / log10 cost ~35 / ~109 instructions (double-double arithmetic); the per-sample
// kernels are otherwise a handful of FMAs, so these two functions decide whether they are HBM-bound.
// Both routines keep 1-ulp-class accuracy (max relative error 4e-16 vs 80-bit references, checked by
// epa_selftest_lin_from_db / epa_selftest_log10 in tests/test_gpu_kernels.py) using small LDS tables
// that every workgroup builds once with the full-precision ocml functions:
//   exp : 256 x f64   2^(j/256)
//   log : 128 x (f64, f64)  (1/c_j, log10(c_j) [- log10 2 for c_j > sqrt 2]),  c_j = 1 + (j+1/2)/128
// float versions map to the hardware v_exp_f32 / v_log_f32 through ocml and need no table.
#pragma once
#include "epa_internal.h"

namespace epa {

constexpr int kExpTabN = 256;
constexpr int kLogTabN = 128;
constexpr size_t kExpTabBytes = kExpTabN * sizeof(double);
constexpr size_t kLogTabBytes = kLogTabN * 2 * sizeof(double);
constexpr size_t kMathTabBytes = kExpTabBytes + kLogTabBytes;  // 4 KiB

struct MathTabs {
  const double* exp2_tab;  // [256]
  const double2* log_tab;  // [128] (inv, log10c)
};

// Build both tables in LDS at `base` (16-byte aligned, kMathTabBytes).  All threads of a
// 256-thread workgroup must call it; the caller synchronises before first use.
__device__ __forceinline__ MathTabs build_math_tabs(unsigned char* base) {
  double* e = reinterpret_cast<double*>(base);
  double2* l = reinterpret_cast<double2*>(base + kExpTabBytes);
  const int t = threadIdx.x;
  if (t < kExpTabN) e[t] = ::exp2((double)t * (1.0 / kExpTabN));
  if (t < kLogTabN) {
    const double c = 1.0 + ((double)t + 0.5) * (1.0 / kLogTabN);
    const double inv = 1.0 / c;
    double lg = -::log10(inv);            // log10 of the value 1/inv the reduction really divides by
    if (c > 1.4142135623730951) lg -= 0.30102999566398120;  // fold one factor 2 into the exponent
    l[t] = make_double2(inv, lg);
  }
  return MathTabs{e, l};
}

// 10^(u/10), fp64.
__device__ __forceinline__ double lin_from_db(double u, const double* __restrict__ tab) {
  constexpr double K256_HI = 85.04135922911648;       // 256*log2(10)/10 rounded to double
  constexpr double K256_LO = -4.272771985668806e-15;  // 256*log2(10)/10 - K256_HI
  constexpr double Z = 0.0027076061740622863;         // ln(2)/256
  const double t = u * K256_HI;
  const double m = __builtin_rint(t);
  double r = fma(u, K256_HI, -m);
  r = fma(u, K256_LO, r);
  const double z = r * Z;
  double p = fma(z, 1.0 / 24.0, 1.0 / 6.0);  // |z| <= ln(2)/512: the first term left out, z^5/120, is < 4e-17 relative
  p = fma(p, z, 0.5);
  p = fma(p, z, 1.0);
  p = fma(p, z, 1.0);
  const double mc = fmin(fmax(m, -300000.0), 300000.0);
  const int mi = (int)mc;
  const double v = ldexp(p * tab[mi & 255], mi >> 8);
  // non-finite arguments: NaN -> NaN, +inf -> +inf, -inf -> 0 (as exp10)
  return (fabs(t) < __builtin_inf()) ? v : (t < 0.0 ? 0.0 : t);
}
__device__ __forceinline__ float lin_from_db(float u, const double*) { return ::exp10f(u * 0.1f); }

// The same value with the non-finite cases out of the instruction stream: NaN falls through the arithmetic, +-inf
// (never a calibrated sample) takes a branch -- six selects / compares fewer per call in VALU-bound loops.
__device__ __forceinline__ double lin_from_db_lean(double u, const double* __restrict__ tab) {
  constexpr double K256_HI = 85.04135922911648, K256_LO = -4.272771985668806e-15, Z = 0.0027076061740622863;
  const double t = u * K256_HI;
  const double m = __builtin_rint(t);
  double r = fma(u, K256_HI, -m);
  r = fma(u, K256_LO, r);
  const double z = r * Z;
  double p = fma(z, 1.0 / 24.0, 1.0 / 6.0);  // |z| <= ln(2)/512: the first term left out, z^5/120, is < 4e-17 relative
  p = fma(p, z, 0.5);
  p = fma(p, z, 1.0);
  p = fma(p, z, 1.0);
  // v_cvt_i32_f64 saturates and turns NaN into 0: no clamp -- the table index is masked, a saturated exponent makes
  // ldexp return inf / 0 as it should, and with a NaN argument p is NaN, hence the result
  const int mi = __double2int_rz(m);
  double v = ldexp(p * tab[mi & 255], mi >> 8);
  if (__builtin_expect(__builtin_isinf(t), 0)) v = t < 0.0 ? 0.0 : t;
  return v;
}
__device__ __forceinline__ float lin_from_db_lean(float u, const double*) { return ::exp10f(u * 0.1f); }

__device__ __noinline__ double log10_special(double x) { return ::log10(x); }

// log10(x), fp64.  x <= 0, subnormal, inf and NaN take the (out-of-line) ocml path.
__device__ __forceinline__ double fast_log10(double x, const double2* __restrict__ tab) {
  constexpr double LOG10_2 = 0.30102999566398120;
  // log1p(r)/ln(10) = r*(c1 + r*(c2 + ... )), |r| <= 2^-8 (+ rounding of 1/c): truncation < 2e-19
  constexpr double C1 = 0.43429448190325182765, C2 = -0.21714724095162591383,
                   C3 = 0.14476482730108394255, C4 = -0.10857362047581295691,
                   C5 = 0.086858896380650365530, C6 = -0.072382413650541971275;
  const unsigned long long bits = __double_as_longlong(x);
  const unsigned ex = (unsigned)(bits >> 52) & 0x7ffu;
  const bool special = (long long)bits <= 0 || ex == 0u || ex == 0x7ffu;  // <= 0, subnormal, inf/NaN
  if (__builtin_expect(special, 0)) return log10_special(x);
  const int j = (int)((bits >> 45) & 127ull);  // top 7 mantissa bits
  const double m = __longlong_as_double((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  const double2 t = tab[j];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, C6, C5);
  p = fma(p, r, C4);
  p = fma(p, r, C3);
  p = fma(p, r, C2);
  p = fma(p, r, C1);
  const double ef = (double)((int)ex - 1023 + (j >= 53 ? 1 : 0));  // c_53 = 1.418 > sqrt 2 > c_52
  const double res = fma(ef, LOG10_2, t.y) + r * p;
  return x == 1.0 ? 0.0 : res;  // exact at 1 (the R < 1 m clamp of the noise TL lands here)
}
__device__ __forceinline__ float fast_log10(float x, const double2*) { return ::log10f(x); }

// The same log10 without the out-of-line call: every special case folded into selects, so that a hot loop
// holding many live SGPRs does not have to spill them around a call it (almost) never makes.
//   subnormal -> scaled by 2^54 first; +inf / NaN -> x; 0 -> -inf; x < 0 -> NaN.
// POSITIVE = the caller guarantees x > 0 or NaN (saves the last two selects); EXACT_AT_1 = return exactly 0
// for x == 1 (otherwise within 1e-17 of it).
template <bool POSITIVE, bool EXACT_AT_1 = true>
__device__ __forceinline__ double fast_log10_inl(double x, const double2* __restrict__ tab) {
  constexpr double LOG10_2 = 0.30102999566398120;
  constexpr double C1 = 0.43429448190325182765, C2 = -0.21714724095162591383,
                   C3 = 0.14476482730108394255, C4 = -0.10857362047581295691,
                   C5 = 0.086858896380650365530, C6 = -0.072382413650541971275;
  const bool sub = ((unsigned)__double2hiint(x) & 0x7ff00000u) == 0u;  // exponent field 0: subnormal or 0
  const double xs = sub ? x * 0x1p54 : x;
  const unsigned long long bits = __double_as_longlong(xs);
  const unsigned ex = (unsigned)(bits >> 52) & 0x7ffu;
  const int j = (int)((bits >> 45) & 127ull);
  const double m = __longlong_as_double((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  const double2 t = tab[j];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, C6, C5);
  p = fma(p, r, C4);
  p = fma(p, r, C3);
  p = fma(p, r, C2);
  p = fma(p, r, C1);
  const double ef = (double)((int)ex - (sub ? 1023 + 54 : 1023) + (j >= 53 ? 1 : 0));
  double res = fma(ef, LOG10_2, t.y) + r * p;
  if (EXACT_AT_1) res = x == 1.0 ? 0.0 : res;
  res = ex == 0x7ffu ? x : res;
  if (!POSITIVE) {
    res = x == 0.0 ? -__builtin_inf() : res;
    res = x < 0.0 ? __builtin_nan("") : res;
  }
  return res;
}
template <bool POSITIVE, bool EXACT_AT_1 = true>
__device__ __forceinline__ float fast_log10_inl(float x, const double2*) { return ::log10f(x); }

// log10 of a positive NORMAL number without any select; zero, subnormals, +inf and NaN take a branch to the general
// routine (the argument must not be negative: callers select on x > 0 themselves).  Within 1e-17 of 0 at x == 1.
__device__ __forceinline__ double fast_log10_lean(double x, const double2* __restrict__ tab) {
  constexpr double LOG10_2 = 0.30102999566398120;
  constexpr double C1 = 0.43429448190325182765, C2 = -0.21714724095162591383,
                   C3 = 0.14476482730108394255, C4 = -0.10857362047581295691,
                   C5 = 0.086858896380650365530, C6 = -0.072382413650541971275;
  const unsigned long long bits = __double_as_longlong(x);
  const unsigned ex = (unsigned)(bits >> 52) & 0x7ffu;
  const int j = (int)((bits >> 45) & 127ull);
  const double m = __longlong_as_double((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  const double2 t = tab[j];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, C6, C5);
  p = fma(p, r, C4);
  p = fma(p, r, C3);
  p = fma(p, r, C2);
  p = fma(p, r, C1);
  const double ef = (double)((int)ex - 1023 + (j >= 53 ? 1 : 0));
  double res = fma(ef, LOG10_2, t.y) + r * p;
  if (__builtin_expect(ex == 0u || ex == 0x7ffu, 0)) res = fast_log10_inl<true, false>(x, tab);
  return res;
}
__device__ __forceinline__ float fast_log10_lean(float x, const double2*) { return ::log10f(x); }

// The polynomial coefficients of fast_log10_lean held in SGPR pairs.  In straight-line code (no loop to hoist them
// out of) the compiler otherwise re-creates each 64-bit literal with two v_mov_b32 at every use -- ten VALU moves per
// log; an opaque scalar copy made once per kernel turns every Horner step into one v_fma_f64 with an SGPR operand.
struct LogCoef {
  double c1, c2, c3, c4, c5, c6, log10_2;
};
__device__ __forceinline__ double opaque_scalar(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ LogCoef make_log_coef() {
  return LogCoef{opaque_scalar(0.43429448190325182765),  opaque_scalar(-0.21714724095162591383),
                 opaque_scalar(0.14476482730108394255),  opaque_scalar(-0.10857362047581295691),
                 opaque_scalar(0.086858896380650365530), opaque_scalar(-0.072382413650541971275),
                 opaque_scalar(0.30102999566398120)};
}
__device__ __forceinline__ double fast_log10_lean(double x, const double2* __restrict__ tab, const LogCoef& K) {
  const unsigned long long bits = __double_as_longlong(x);
  const unsigned ex = (unsigned)(bits >> 52) & 0x7ffu;
  const int j = (int)((bits >> 45) & 127ull);
  const double m = __longlong_as_double((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
  const double2 t = tab[j];
  const double r = fma(m, t.x, -1.0);
  double p = fma(r, K.c6, K.c5);
  p = fma(p, r, K.c4);
  p = fma(p, r, K.c3);
  p = fma(p, r, K.c2);
  p = fma(p, r, K.c1);
  const double ef = (double)((int)ex - 1023 + (j >= 53 ? 1 : 0));
  double res = fma(ef, K.log10_2, t.y) + r * p;
  if (__builtin_expect(ex == 0u || ex == 0x7ffu, 0)) res = fast_log10_inl<true, false>(x, tab);
  return res;
}
__device__ __forceinline__ float fast_log10_lean(float x, const double2*, const LogCoef&) { return ::log10f(x); }

}  // namespace epa
