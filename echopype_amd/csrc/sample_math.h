// Per-sample device math shared by the calibration and reduction kernels.
#pragma once
#include "epa_internal.h"

namespace epa {

// Row constants converted once per (channel, ping) row to the compute type.
//   R  = (s*ra)*rb + r0                echo_range, reference operation order, always in double
//                                      (it also decides bin membership)
//   R' = R - shift                     TVG range; guard R' <= 0 -> NaN decided on this value
//   out = g*raw + n*log10(R') + alpha2*R' + A
//       = g*raw + n*log10(s - d) + alpha2*R' + A0          A0 = A + n*log10(k), d = (shift-r0)/k
// n*log10(s - d) is per range column (ColumnLog): no transcendental per sample.
template <typename T>
struct RowK {
  double ra, rb, r0, shift, d;
  T alpha2, A0, g;
  __device__ __forceinline__ explicit RowK(const CoefRow& r)
      : ra(r.ra), rb(r.rb), r0(r.r0), shift(r.shift), d(r.d), alpha2((T)r.alpha2), A0((T)r.A0),
        g((T)r.g) {}
  __device__ __forceinline__ double range(int s) const { return ((double)s * ra) * rb + r0; }
};

template <typename T>
__device__ __noinline__ T log10_noinline(T x) {
  return M<T>::log10(x);
}

// Per-lane cache of n*log10(s - d) for the VEC range samples the lane owns; refreshed only when
// the row's d differs from the cached one (EK60: d == 2 for every ping -> computed once).
template <typename T, int VEC>
struct ColumnLog {
  double d;
  bool have;
  T nL[VEC];  // n * log10(s - d); NaN / -inf where s - d <= 0
  __device__ __forceinline__ ColumnLog() : d(0.0), have(false) {}
  __device__ __forceinline__ void update(double dnew, int s0, T nspread) {
    if (have && dnew == d) return;
    d = dnew;
    have = (dnew == dnew);
    // out-of-line log10 (by value: no scratch): runs once per column in the common case
    for (int j = 0; j < VEC; ++j) nL[j] = nspread * log10_noinline<T>((T)((double)(s0 + j) - dnew));
  }
};

// R' > 0 although s - d <= 0: R' is the rounding residue of R - shift (only possible within an
// ulp of R' == 0).  Reproduce the reference's value for it; A0 carries +n*log10(k), undo that.
template <typename T>
__device__ __noinline__ T residue_spread(T rt, T nspread, double k) {
  return nspread * (log10_noinline<T>(rt) - log10_noinline<T>((T)k));
}

// One power sample -> Sv/TS (calibrate_ek.py:104-110,165-171,184; calibrate_azfp.py:64-97).
// `range` = echo_range R (before NaN masking).  The guard and the absorption term use R' exactly as
// the reference computes it; when R' > 0 but the separable log is not finite-positive-consistent
// (s - d <= 0: R' is pure rounding residue of R - shift), log10(R') is evaluated directly so that
// even that residue matches.
template <typename T>
__device__ __forceinline__ T cal_power_sample(float raw, int s, const RowK<T>& r, T nspread, T nL,
                                              bool guard_pos, double range) {
  const double rtd = range - r.shift;
  const T rt = (T)rtd;
  T spread = nL;
  if (guard_pos) {
    if (!(rtd > 0.0)) {
      spread = M<T>::nan();
    } else if (!(nL > -(T)__builtin_inf())) {
      spread = residue_spread<T>(rt, nspread, r.ra * r.rb);
    }
  }
  return fma(r.g, (T)raw, spread) + fma(r.alpha2, rt, r.A0);
}

// vector load/store helpers: VEC consecutive elements, 16-B accesses where the type allows
template <int VEC>
struct RawVec;
template <>
struct RawVec<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
template <>
struct RawVec<2> {
  float v[2];
  __device__ __forceinline__ void load(const float* p) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  }
};
template <>
struct RawVec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p) { v[0] = *p; }
};

// Streaming (non-temporal) stores for outputs that are written once and not re-read by the kernel.
typedef double epa_d2 __attribute__((ext_vector_type(2)));
typedef float epa_f2 __attribute__((ext_vector_type(2)));
typedef float epa_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt2(double* p, double a, double b) {
  epa_d2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<epa_d2*>(p));
}
__device__ __forceinline__ void store_nt2(float* p, float a, float b) {
  epa_f2 v = {a, b};
  __builtin_nontemporal_store(v, reinterpret_cast<epa_f2*>(p));
}

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const T (&v)[VEC]);
// all bulk outputs of the path are written once and never re-read by the writing kernel: streaming
// (non-temporal) stores, measured +2...6 % on the write-heavy kernels
template <>
__device__ __forceinline__ void store_vec<double, 4>(double* p, const double (&v)[4]) {
  // two 16-B halves at a 32-B lane stride: plain stores (nt on half lines measured slower)
  reinterpret_cast<double2*>(p)[0] = make_double2(v[0], v[1]);
  reinterpret_cast<double2*>(p)[1] = make_double2(v[2], v[3]);
}
template <>
__device__ __forceinline__ void store_vec<float, 4>(float* p, const float (&v)[4]) {
  epa_f4 t = {v[0], v[1], v[2], v[3]};
  __builtin_nontemporal_store(t, reinterpret_cast<epa_f4*>(p));
}
template <>
__device__ __forceinline__ void store_vec<double, 2>(double* p, const double (&v)[2]) {
  store_nt2(p, v[0], v[1]);
}
template <>
__device__ __forceinline__ void store_vec<float, 2>(float* p, const float (&v)[2]) {
  store_nt2(p, v[0], v[1]);
}
template <>
__device__ __forceinline__ void store_vec<double, 1>(double* p, const double (&v)[1]) { p[0] = v[0]; }
template <>
__device__ __forceinline__ void store_vec<float, 1>(float* p, const float (&v)[1]) { p[0] = v[0]; }

// Lane -> sample mapping of the streaming kernels inside a 1024-sample chunk (256 lanes x 4
// samples), chosen so that EVERY access of a wavefront is one contiguous run of 16 B per lane:
//   f32 outputs: 4 consecutive samples per lane (16-B load, 16-B store)            NSEG=1, LEN=4
//   f64 outputs: two pairs per lane, {2l, 2l+1} and {128+2l, 128+2l+1} within the wave's 256
//                samples (8-B loads of the f32 input, 16-B stores / loads of f64)    NSEG=2, LEN=2
template <typename T>
struct LaneMap;
template <>
struct LaneMap<float> {
  static constexpr int NSEG = 1, LEN = 4;
  static __device__ __forceinline__ int first(int chunk0, int) { return chunk0 + (int)threadIdx.x * 4; }
};
template <>
struct LaneMap<double> {
  static constexpr int NSEG = 2, LEN = 2;
  static __device__ __forceinline__ int first(int chunk0, int seg) {
    return chunk0 + ((int)threadIdx.x >> 6) * 256 + seg * 128 + 2 * ((int)threadIdx.x & 63);
  }
};

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, T (&v)[VEC]);
template <>
__device__ __forceinline__ void load_vec<double, 4>(const double* p, double (&v)[4]) {
  const double2 a = reinterpret_cast<const double2*>(p)[0];
  const double2 b = reinterpret_cast<const double2*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
template <>
__device__ __forceinline__ void load_vec<float, 4>(const float* p, float (&v)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
template <>
__device__ __forceinline__ void load_vec<double, 2>(const double* p, double (&v)[2]) {
  const double2 a = *reinterpret_cast<const double2*>(p);
  v[0] = a.x; v[1] = a.y;
}
template <>
__device__ __forceinline__ void load_vec<float, 2>(const float* p, float (&v)[2]) {
  const float2 a = *reinterpret_cast<const float2*>(p);
  v[0] = a.x; v[1] = a.y;
}
template <>
__device__ __forceinline__ void load_vec<double, 1>(const double* p, double (&v)[1]) { v[0] = p[0]; }
template <>
__device__ __forceinline__ void load_vec<float, 1>(const float* p, float (&v)[1]) { v[0] = p[0]; }

}  // namespace epa
