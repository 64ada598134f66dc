// K1+K5 headline kernel: fused compute_Sv -> compute_MVBS for power samples, one pass over HBM.
//
// Specialisation of block_reduce.hip for the hot configuration (sorted pings, one workgroup per
// (channel, ping-bin), range grid in LDS, no echo_range output); everything else goes through the
// generic kernel there.  Same arithmetic, leaner instruction stream: the coefficient row is read
// with scalar loads, the rare paths (log refresh, rounding-residue, bin change) are out of the
// straight-line body, and all per-sample NaN handling is predicated instead of branched.
//
// Per sample (f64): echo_range in the reference's operation order (2 mul + add), R' (sub), Sv
// (2 fma + add), exp10 (the only transcendental), two compares for the bin test, one add.
// Reference lines replaced: see sv_power.hip and block_reduce.hip.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>
#include <type_traits>

#include "fast_math.h"
#include "sample_math.h"

namespace epa_fused {

constexpr int VEC = 4;
constexpr int kChunk = epa::kBlock * VEC;

struct Args {
  int P, S, n_tbins, n_rbins;
  double range_bin, inv_range_bin;
  double nspread, fill_value;
  unsigned guard, mask_range, skipna, closed_right;
  unsigned cnt_off, tab_off;
  unsigned long long* rmax_key;  // optional: max valid echo_range as an order-preserving u64 key
  unsigned long long* rstat;     // optional, with rmax_key: {min valid echo_range as a key, number of NaN echo_range values}
  int xcd_map;  // time bins dealt to the XCDs in contiguous eighths (epa::xcd_contiguous)
  int flagged_only;  // mvbs_of_sv_rows_kernel after the fixed-bin kernel: only the time bins that one left (kLeftToRows)
  // binned on ``depth`` instead of the echo_range (consolidate/api.py:221 between compute_Sv and compute_MVBS):
  // depth[c,p,s] = doffset[c,p] + dscale[c,p] * echo_range[c,p,s], [C*P] doubles each; NULL: binned on the echo_range
  const double* dscale;
  const double* doffset;
  void* depth_out;  // optional [C*P*S] of T: the depth array written by the same pass (NaN where the raw sample is)
};

// order-preserving map double -> u64 (so that atomicMax on the key is a max on the double)
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long b = __double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <typename T>
__device__ __noinline__ T log10_slow(T x) {
  return epa::M<T>::log10(x);
}

template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
  unsafeAtomicAdd(p, v);
}

// Lane-private state of one range column (kept across the pings of a time bin).
template <typename T>
struct Column {
  double sra;       // fl(s * ra): first factor of the range product (range.py:138 order)
  double blo, bhi;  // edges of the range bin the column currently sits in (empty: blo > bhi)
  T nL;             // n * log10(s - d)
  T acc_sum;
  int acc_rb;
  uint32_t acc_cnt;
  __device__ __forceinline__ void init() {
    sra = 0.0; blo = 1.0; bhi = 0.0; nL = epa::M<T>::nan(); acc_sum = (T)0; acc_rb = -1; acc_cnt = 0u;
  }
  // ``n_clean``: the pings of this chunk the whole wavefront took on the lean path so far (a scalar: every column of
  // every lane got one valid value from each of them).  A column counts only what the general path added, relative to
  // the scalar's value when the column last changed its bin: values in the bin = acc_cnt + n_clean (mod 2^32).
  __device__ __forceinline__ void flush(T* lsum, uint32_t* lcnt, uint32_t n_clean) {
    const uint32_t n = acc_cnt + n_clean;
    if (acc_rb >= 0 && n > 0u) {
      lds_add(lsum + acc_rb, acc_sum);
      atomicAdd(lcnt + acc_rb, n);
    }
  }
  // the column has left the range bin it sat in: find the new one (``valid`` false: a NaN raw sample, parked outside)
  __device__ __forceinline__ void rebin(double x, bool valid, double bin, double inv_bin, int n_rbins, T* lsum,
                                        uint32_t* lcnt, uint32_t n_clean) {
    const int rb = valid ? epa::range_bin_index(x, bin, inv_bin, n_rbins, false) : -1;
    if (rb != acc_rb) {
      flush(lsum, lcnt, n_clean);
      acc_rb = rb;
      acc_sum = (T)0;
      acc_cnt = 0u - n_clean;
    }
    blo = rb >= 0 ? (double)rb * bin : 1.0;
    bhi = rb >= 0 ? (double)(rb + 1) * bin : 0.0;
  }
};

// 10^(u/10) for the BIN sums of the lean path: the argument reduction in one word and a third-degree polynomial on the
// 256-entry table -- 2e-13 relative (z^4/24 with |z| <= ln(2)/512, plus |u| * 1.2e-17 for the dropped low word of
// 256 log2(10)/10), five instructions fewer than lin_from_db_lean's 4e-16.  A mean of such values in dB is off by less
// than 1e-12 dB; the Sv array never goes through it.  Finite arguments only (the lean path's precondition).
__device__ __forceinline__ double lin_bins(double u, const double* __restrict__ tab) {
  constexpr double K256 = 85.04135922911648, Z = 0.0027076061740622863;
  constexpr double C1 = Z, C2 = Z * Z / 2.0, C3 = Z * Z * Z / 6.0;
  const double t = u * K256;
  const double m = __builtin_rint(t);
  const double r = fma(u, K256, -m);
  double p = fma(r, C3, C2);
  p = fma(p, r, C1);
  p = fma(p, r, 1.0);
  const int mi = __double2int_rz(m);
  return ldexp(p * tab[mi & 255], mi >> 8);
}
__device__ __forceinline__ float lin_bins(float u, const double*) { return ::exp10f(u * 0.1f); }

// v_max that returns the operand that is a number (IEEE maxNum), without the canonicalising copy fmax() adds
__device__ __forceinline__ double vmax_num(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax_num(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin_num(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// One sample: echo_range, R', Sv, linear value, bin membership, private accumulation.
// Compile-time flags of the hot instantiation = the EK default (R' <= 0 guard, range masked by NaN
// input, skipna, left-closed bins); any other combination runs the generic kernel (block_reduce.hip).
// The kernel is as much VALU- as HBM-bound (without the Sv store it takes 3/4 of its time): every instruction of
// this function counts.  r0v / A0v: the ping's r0 and A0 in vector registers (a VALU instruction takes one scalar
// operand; the compiler otherwise copies the second one per sample).
template <typename T, bool STATS>
__device__ __forceinline__ T process_sample(Column<T>& c, float raw, const epa::CoefRow& r, double r0v, T g, T a2,
                                            T A0v, T nL, T nspread, double bin, double inv_bin, int n_rbins,
                                            const double* tab, T* lsum, uint32_t* lcnt, double& xmax, double& xmin,
                                            unsigned& nnan, uint32_t n_clean) {
  const T NaN = epa::M<T>::nan();
  const double x = fma(c.sra, r.rb, r0v);  // echo_range = (s*ra)*rb [+0]
  if (STATS) {
    // {min, max, NaN count} of the echo_range, NaN exactly where the raw sample is: x + 0 * raw is the range or NaN,
    // and v_max_f64 / v_min_f64 return the operand that is a number
    const double xq = fma((double)raw, 0.0, x);
    xmax = vmax_num(xmax, xq);
    xmin = vmin_num(xmin, xq);
    nnan += (unsigned)__builtin_popcountll(__ballot(raw != raw));  // (per wavefront, in a scalar register)
  }
  const double rtd = x - r.shift;
  const T rt = (T)rtd;
  const bool pos = rtd > 0.0;
  T s1 = fma(g, (T)raw, nL);
  if (pos & !(nL > -(T)__builtin_inf()))  // rounding residue of R - shift (rare)
    s1 = fma(g, (T)raw, nspread * (log10_slow<T>(rt) - log10_slow<T>((T)(r.ra * r.rb))));
  s1 = pos ? s1 : NaN;
  const T sv = s1 + fma(a2, rt, A0v);
  const T v = epa::lin_from_db_lean(sv, tab);  // (+-inf through a rare branch)
  // still inside the bin of the previous ping?  (A NaN raw sample makes Sv and v NaN: it is never accumulated, whatever
  // bin the column sits in; the slow path below still parks such a column outside the grid.)
  const bool same = (x >= c.blo) & (x < c.bhi);
  if (!same) c.rebin(x, raw == raw, bin, inv_bin, n_rbins, lsum, lcnt, n_clean);
  // (a column outside the grid, acc_rb < 0, accumulates too: flush() drops it.)  v >= 0 or NaN: max(v, 0) adds nothing
  // for a NaN
  c.acc_sum += vmax_num(v, (T)0);
  c.acc_cnt += v == v ? 1u : 0u;
  return sv;
}

// ---- one PAIR of samples of a ping, lean unless the wavefront found the ping not clean ------------------------------------
// ``clean`` (a scalar, established per ping by the kernel): every raw sample of the wavefront is a finite number, every
// cached n log10(s - d) is finite and R' > 0 for every lane.  Then none of process_sample's per-sample NaN handling is
// needed -- no guard select, no rounding-residue test, no max(v, 0), no count (the ping is counted once, in a scalar
// register), no inf test around the exponential, {min, max} of the echo_range taken once per ping by the caller.  A ping
// that is not clean (NaN padding, the first samples of a row, a NaN row) runs the SAME body with the general forms
// patched in by wave-uniform branches: one instruction stream, the registers of one.  The Sv arithmetic is
// process_sample's, operation for operation, on either side of the branches: the same bits.
// DEPTH (1: bin on depth, 2: and write it): the coordinate that is binned and whose {min, max, NaN count} are taken is
// depth = dof + dsc * echo_range, evaluated in T on the echo_range rounded to T -- the value the depth array holds
// (epa::depth_of; the array is written only for DEPTH == 2) -- instead of the echo_range itself.
template <typename T, bool STATS, bool WRITE_SV, int DEPTH>
__device__ __forceinline__ void process_pair(Column<T>& c0, Column<T>& c1, float2 in, bool clean, const epa::CoefRow& r,
                                             double r0v, T g, T a2, T A0v, T nL0, T nL1, T nspread, double bin,
                                             double inv_bin, int n_rbins, const double* tab, T* lsum, uint32_t* lcnt,
                                             T* __restrict__ sv_dst, double& xmax, double& xmin, unsigned& nnan,
                                             uint32_t n_clean, double& xfirst, double& xlast, T dsc, T dof,
                                             T* __restrict__ depth_dst, double sra0, double sra1) {
  const double e0 = fma(sra0, r.rb, r0v), e1 = fma(sra1, r.rb, r0v);  // echo_range = (s*ra)*rb [+0]
  const double rtd0 = e0 - r.shift, rtd1 = e1 - r.shift;
  T d0 = (T)0, d1 = (T)0;
  if (DEPTH) {
    d0 = epa::depth_of(dsc, dof, (T)e0);
    d1 = epa::depth_of(dsc, dof, (T)e1);
  }
  const double x0 = DEPTH ? (double)d0 : e0, x1 = DEPTH ? (double)d1 : e1;  // the binning coordinate
  const T rt0 = (T)rtd0, rt1 = (T)rtd1;
  T s10 = fma(g, (T)in.x, nL0), s11 = fma(g, (T)in.y, nL1);
  if (!clean) {  // (scalar) R' <= 0 guard, rounding residue of R - shift, {min, max, NaN count} per sample
    const T NaN = epa::M<T>::nan();
    const bool pos0 = rtd0 > 0.0, pos1 = rtd1 > 0.0;
    if (pos0 & !(nL0 > -(T)__builtin_inf()))
      s10 = fma(g, (T)in.x, nspread * (log10_slow<T>(rt0) - log10_slow<T>((T)(r.ra * r.rb))));
    if (pos1 & !(nL1 > -(T)__builtin_inf()))
      s11 = fma(g, (T)in.y, nspread * (log10_slow<T>(rt1) - log10_slow<T>((T)(r.ra * r.rb))));
    s10 = pos0 ? s10 : NaN;
    s11 = pos1 ? s11 : NaN;
    if (STATS) {  // x + 0 * raw is the range or NaN; v_max_f64 / v_min_f64 return the operand that is a number
      const double xq0 = fma((double)in.x, 0.0, x0), xq1 = fma((double)in.y, 0.0, x1);
      xmax = vmax_num(vmax_num(xmax, xq0), xq1);
      xmin = vmin_num(vmin_num(xmin, xq0), xq1);
      // (a NaN coordinate: the raw sample's NaN, or a NaN coefficient / depth row)
      nnan += (unsigned)__builtin_popcountll(__ballot(xq0 != xq0)) + (unsigned)__builtin_popcountll(__ballot(xq1 != xq1));
    }
    if (DEPTH == 2) {  // the depth array is NaN where the echo_range is: where the raw sample is
      d0 = fma((T)in.x, (T)0, d0);
      d1 = fma((T)in.y, (T)0, d1);
    }
  }
  if (DEPTH == 2) epa::store_nt2(depth_dst, d0, d1);
  const T sv0 = s10 + fma(a2, rt0, A0v), sv1 = s11 + fma(a2, rt1, A0v);
  if (WRITE_SV) {
#ifndef EPA_PLAIN_STORES  // streaming (nt) stores: +2 % at 4 G samples, Sv is never re-read here
    epa::store_nt2(sv_dst, sv0, sv1);
#else
    const T o[2] = {sv0, sv1};
    epa::store_vec<T, 2>(sv_dst, o);
#endif
  }
  T v0 = lin_bins(sv0, tab), v1 = lin_bins(sv1, tab);
  if (!clean) {  // (scalar) +-inf goes through lin_bins as NaN: 10^(+inf) = +inf, 10^(-inf) = 0
    if (__builtin_expect(__builtin_isinf(sv0), 0)) v0 = sv0 < (T)0 ? (T)0 : sv0;
    if (__builtin_expect(__builtin_isinf(sv1), 0)) v1 = sv1 < (T)0 ? (T)0 : sv1;
  }
  // still inside the bin of the previous ping?  (A NaN raw sample makes Sv and v NaN: it is never accumulated, whatever
  // bin the column sits in; rebin still parks such a column outside the grid.)
  if (!((x0 >= c0.blo) & (x0 < c0.bhi))) c0.rebin(x0, in.x == in.x, bin, inv_bin, n_rbins, lsum, lcnt, n_clean);
  if (!((x1 >= c1.blo) & (x1 < c1.bhi))) c1.rebin(x1, in.y == in.y, bin, inv_bin, n_rbins, lsum, lcnt, n_clean);
  if (clean) {
    c0.acc_sum += v0;
    c1.acc_sum += v1;
  } else {  // v >= 0 or NaN: max(v, 0) adds nothing for a NaN (a column outside the grid accumulates too: flush drops it)
    c0.acc_sum += vmax_num(v0, (T)0);
    c1.acc_sum += vmax_num(v1, (T)0);
    c0.acc_cnt += v0 == v0 ? 1u : 0u;
    c1.acc_cnt += v1 == v1 ? 1u : 0u;
  }
  xfirst = x0;
  xlast = x1;
}

// raw sample that is NaN or +-inf (v_cmp_class_f32: signalling / quiet NaN, -inf, +inf)
__device__ __forceinline__ bool not_finite(float x) { return __builtin_amdgcn_classf(x, 0x207); }

// Raw-sample sources.  float: backscatter_r as the converter stores it (f32, NaN-padded,
// convert/parse_base.py:302).  int16_t: the instrument's own power samples:
// value = float32(int16) * float32(10*log10(2)/256) exactly as parse_base.py:24,302 computes it, and
// samples at or beyond the ping's recorded length n_valid[c,p] are the NaN padding
// (parse_base.py pad_shorter_ping) -- 2 B/sample of input traffic instead of 4.
template <typename RawT>
struct PingLoad;
template <>
struct PingLoad<float> {
  float2 a, b;
  __device__ __forceinline__ PingLoad() : a(make_float2(0.f, 0.f)), b(make_float2(0.f, 0.f)) {}
  __device__ __forceinline__ void issue(const float* __restrict__ row, int sA, int sB, bool hasB, int, int) {
    a = *reinterpret_cast<const float2*>(row + sA);
    if (hasB) b = *reinterpret_cast<const float2*>(row + sB);
  }
  __device__ __forceinline__ void resolve(float2& A, float2& B, int, int, int, bool) const {
    A = a;
    B = b;
  }
};
template <>
struct PingLoad<int16_t> {
  // The float path's ownership with the float path's two requests per lane, 4 bytes each: word A = samples
  // {sA, sA+1}, word B = {sB, sB+1}; one SDWA conversion + one multiplication per sample.  Measured on
  // 4 x 500 000 x 2000 (round 4): 9.10 ms -- the float source's 8.98 -- against 10.6 ms for one 8-byte load per lane
  // redistributed by four ds_bpermute, and 11.3 ms with the padding test on every sample (round 3).
  unsigned a, b;
  __device__ __forceinline__ PingLoad() : a(0u), b(0u) {}
  __device__ __forceinline__ void issue(const int16_t* __restrict__ row, int sA, int sB, bool hasB, int, int) {
    a = *reinterpret_cast<const unsigned*>(row + sA);
    if (hasB) b = *reinterpret_cast<const unsigned*>(row + sB);
  }
  static __device__ __forceinline__ float2 unpack(unsigned w, int s, int n_valid) {
    constexpr float kIndex2Power = 0.011758984205624266f;  // float32(10*log10(2)/256)
    const float nanv = __builtin_nanf("");
    const float x = (float)(short)(w & 0xffffu), y = (float)(short)(w >> 16);
    return make_float2(s < n_valid ? x * kIndex2Power : nanv, s + 1 < n_valid ? y * kIndex2Power : nanv);
  }
  static __device__ __forceinline__ float2 unpack_all(unsigned w) {  // every sample of the pair is a recorded one
    constexpr float kIndex2Power = 0.011758984205624266f;
    return make_float2((float)(short)(w & 0xffffu) * kIndex2Power, (float)(short)(w >> 16) * kIndex2Power);
  }
  // ``full`` (a scalar): every sample of the wavefront's 256 is a recorded one -- the usual case, a ping recorded at
  // full length -- and no lane tests for padding (round 6: the per-lane test, and the finite-value test the kernel
  // then repeated on values that cannot be anything else, made the int16 source 15-18 % SLOWER than the float one
  // wherever the Sv store did not hide it: fp32 out 7.3 against 6.2 ms, bins only 6.4 against 5.5 per 4 G samples)
  __device__ __forceinline__ void resolve(float2& A, float2& B, int sA, int sB, int nv, bool full) const {
    if (full) {
      A = unpack_all(a);
      B = unpack_all(b);
    } else {
      A = unpack(a, sA, nv);
      B = unpack(b, sB, nv);
    }
  }
};

// Lane -> sample mapping inside a 1024-sample chunk (4 waves x 256 samples): lane l of wave w owns
// the two PAIRS {base + 2l, +1} and {base + 128 + 2l, +1}, base = chunk0 + 256 w.  Every load
// (8 B/lane) and every store (16 B/lane f64) of a wave is then one contiguous 512 B / 1 KiB
// segment -- the 4-consecutive-samples-per-lane layout makes each f64 store instruction touch
// only half of every 128-B line.
#ifndef EPA_FUSED_MIN_WAVES
#define EPA_FUSED_MIN_WAVES 1
#endif
#ifndef EPA_FUSED_LEAN  // development knob: 0 = every ping takes the general per-sample path (the round-4 kernel)
#define EPA_FUSED_LEAN 1
#endif
// a wave-uniform double, pinned to scalar registers
__device__ __forceinline__ double uniform(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// (the exponent field of a double held in scalar registers: integer instructions only)
__device__ __forceinline__ bool finite_bits(double v) { return (__double2hiint(v) & 0x7ff00000) != 0x7ff00000; }

template <typename T, typename RawT, bool WRITE_SV, bool RMAX, int DEPTH>
__global__ __launch_bounds__(epa::kBlock, EPA_FUSED_MIN_WAVES > 4 ? EPA_FUSED_MIN_WAVES : 4) void fused_sv_mvbs_kernel(
    const RawT* __restrict__ raw, const int32_t* __restrict__ n_valid,
    const epa::CoefRow* __restrict__ coef,
    const int32_t* __restrict__ bin_start, T* __restrict__ sv_out, T* __restrict__ mvbs_out,
    T* __restrict__ sum_out, uint32_t* __restrict__ cnt_out, const double* __restrict__ dscale,
    const double* __restrict__ doffset, Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);  // synchronised below
  const double* tab = mt.exp2_tab;
  // A column's cached n log10(s - d) lives in LDS, every lane its own four entries (written and read by the same lane:
  // no barrier), instead of eight registers: four wavefronts per SIMD without scratch.  The float instances without
  // the statistics have the registers to spare (measured round 5: 6.4-6.5 ms with the LDS reads, 6.0-6.3 without).
  constexpr bool NL_LDS = RMAX || sizeof(T) == 8;
  __shared__ T col_nL[NL_LDS ? kChunk : 1];
  // ... and, for the variants that also carry the coordinate's {min, max, NaN count}, a column's fl(s * ra) likewise
  // (eight more registers: round 6 found those variants at 128 VGPRs + 32 B of scratch per lane, and a scratch reload
  // inside the ping loop waits for every store in flight: 2.12 instead of 1.94 ms per 0.8 G samples)
  constexpr bool SRA_LDS = (RMAX || std::is_same<RawT, int16_t>::value) && sizeof(T) == 8;  // (the int16 source: likewise)
  __shared__ double col_sra[SRA_LDS ? kChunk : 1];
  __shared__ unsigned long long wg_keys[2];  // RMAX: the workgroup's {max key, min key} and NaN count (see the end)
  __shared__ unsigned wg_nnan;
  if (RMAX && threadIdx.x == 0) {
    wg_keys[0] = 0ull;
    wg_keys[1] = ~0ull;
    wg_nnan = 0u;
  }

  const int c = blockIdx.y, tb = a.xcd_map ? epa::xcd_contiguous(blockIdx.x, a.n_tbins) : (int)blockIdx.x;
  const int S = a.S, n_rbins = a.n_rbins;
  // tb == n_tbins: the pings that belong to NO time bin (NaT, or on the first edge of
  // right-closed bins) still get their Sv; two segments, nothing is accumulated.
  const bool extra = tb == a.n_tbins;
  if (extra && !WRITE_SV) return;
  const int nseg = extra ? 2 : 1;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  __syncthreads();

  const T nspread = (T)a.nspread;
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const RawT* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  const int32_t* __restrict__ nv_c = n_valid ? n_valid + (size_t)c * a.P : nullptr;
  T* __restrict__ sv_c = WRITE_SV ? sv_out + (size_t)c * a.P * S : nullptr;
  T* __restrict__ dp_c = DEPTH == 2 ? reinterpret_cast<T*>(a.depth_out) + (size_t)c * a.P * S : nullptr;
  // (kernel parameters, not members of ``a``: read-only + restrict is what makes the per-ping reads scalar loads)
  const double* __restrict__ dsc_c = DEPTH ? dscale + (size_t)c * a.P : nullptr;
  const double* __restrict__ dof_c = DEPTH ? doffset + (size_t)c * a.P : nullptr;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (wave: a scalar)
  double xmax = -__builtin_inf(), xmin = __builtin_inf();
  unsigned nnan = 0u;

  for (int seg = 0; seg < nseg; ++seg) {
  const int pb = extra ? (seg == 0 ? 0 : bin_start[a.n_tbins]) : bin_start[tb];
  const int pe = extra ? (seg == 0 ? bin_start[0] : a.P) : bin_start[tb + 1];
  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane;  // first sample of pair A
    const int sB = sA + 128;                        // first sample of pair B
    if (sA >= S) continue;
    const bool hasB = sB < S;
    Column<T> col[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) col[j].init();
    const int eA = wave * 256 + 2 * lane;  // the lane's entries of col_nL: eA, eA + 1, eA + 128, eA + 129
    if (NL_LDS) {
      col_nL[eA] = col_nL[eA + 1] = col_nL[eA + 128] = col_nL[eA + 129] = epa::M<T>::nan();
    }
    // (compared as BITS, in scalar registers: a float compare of two scalars is two vector instructions per ping; the
    //  initial pattern is a NaN payload no row holds)
    long long dcur = 0x7ff8dead00000001ll, racur = 0x7ff8dead00000002ll;
    uint32_t n_clean = 0u;  // (scalar) pings of this chunk the wavefront took on the lean path
    bool plain = false;     // (scalar) every cached n log10(s - d) of the wavefront is a finite number
    // software prefetch: the raw samples and the coefficient row of ping p+1 are requested before
    // ping p is processed, so their latency hides behind ~150 instructions of arithmetic (+8 %)
    const int s4 = chunk0 + wave * 256 + 4 * lane;  // (a source that loads quads: unused by the present ones)
    PingLoad<RawT> nxt;
    epa::CoefRow nxtR = rowp0[pb < pe ? pb : 0];
    double nxtDs = 1.0, nxtDo = 0.0;  // (scalar) the ping's depth scale and offset
    if (DEPTH) {
      nxtDs = uniform(dsc_c[pb < pe ? pb : 0]);
      nxtDo = uniform(dof_c[pb < pe ? pb : 0]);
    }
    int nxtNv = nv_c && pb < pe ? nv_c[pb] : S;  // (the recorded length of the next ping, requested a ping ahead like its row:
    if (pb < pe) nxt.issue(raw_c + (size_t)pb * S, sA, sB, hasB, s4, S);  //  read where it is used it stalled every ping)
    for (int p = pb; p < pe; ++p) {
      const epa::CoefRow r = nxtR;
      const double curDs = nxtDs, curDo = nxtDo;
      const size_t row_off = (size_t)p * S;
      const int nv = nxtNv;
      float2 inA, inB;
      // (scalar) int16 source: no padding inside this wavefront's samples of the ping, hence every value finite
      const bool full = nv_c != nullptr && nv >= chunk0 + wave * 256 + 256;
      nxt.resolve(inA, inB, sA, sB, nv, full);
      if (p + 1 < pe) {
        nxtR = rowp0[p + 1];
        if (nv_c) nxtNv = nv_c[p + 1];
        if (DEPTH) {
          nxtDs = uniform(dsc_c[p + 1]);
          nxtDo = uniform(dof_c[p + 1]);
        }
        nxt.issue(raw_c + row_off + S, sA, sB, hasB, s4, S);
      }
      if (!((__double_as_longlong(r.d) == dcur) & (__double_as_longlong(r.ra) == racur))) {  // uniform; once per column
        dcur = __double_as_longlong(r.d);                     // for a file with constant tau / sample_interval
        racur = __double_as_longlong(r.ra);
        bool fin = true;
        for (int j = 0; j < VEC; ++j) {
          const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
          const T nl = nspread * log10_slow<T>((T)(sj - r.d));
          if (NL_LDS) col_nL[eA + (j < 2 ? 0 : 128) + (j & 1)] = nl;  // (see col_nL)
          else col[j].nL = nl;
          if (SRA_LDS) col_sra[eA + (j < 2 ? 0 : 128) + (j & 1)] = sj * r.ra;
          else col[j].sra = sj * r.ra;
          fin = fin & ((j >= 2 && !hasB) | (fabs(nl) < (T)__builtin_inf()));
        }
        plain = __ballot(!fin) == 0ull;
      }
      const T g = (T)r.g, a2 = (T)r.alpha2;
      T A0 = (T)r.A0;
      double r0v = r.r0;
      asm volatile("" : "+v"(A0), "+v"(r0v));  // one copy per ping into vector registers, not one per sample
      // Is the ping CLEAN for this wavefront -- every raw sample finite, R' > 0 at the lane's first column (then at all
      // of them: the range grows with the sample number when ra, rb > 0; a NaN row fails the test), the cached logs
      // finite?  Then process_pair takes its lean form (about half the instructions).
      bool clean = false;
#if EPA_FUSED_LEAN
      {
        const double xa = fma(SRA_LDS ? col_sra[eA] : col[0].sra, r.rb, r0v);
        bool bad = !(xa - r.shift > 0.0);
        if (!full) bad = bad | not_finite(inA.x) | not_finite(inA.y) | not_finite(inB.x) | not_finite(inB.y);
        // (scalar) ra, rb > 0, and the row's other coefficients are numbers: a NaN / inf gain, absorption or constant
        // makes Sv NaN / inf with finite raw samples -- the general forms skip such values, the lean ones would add them
        bool kpos = (__double2hiint(r.ra) > 0) & (__double2hiint(r.rb) > 0) & finite_bits(r.g) & finite_bits(r.A0) &
                    finite_bits(r.alpha2);
        if (DEPTH) kpos = kpos & finite_bits(curDs) & finite_bits(curDo);  // (a NaN depth row: counted per sample)
        clean = plain & kpos & (__ballot(bad) == 0ull);
      }
#endif
      const T dsc = (T)curDs, dof = (T)curDo;
      double xf, xl, xdummy;
      process_pair<T, RMAX, WRITE_SV, DEPTH>(col[0], col[1], inA, clean, r, r0v, g, a2, A0, NL_LDS ? col_nL[eA] : col[0].nL,
                                             NL_LDS ? col_nL[eA + 1] : col[1].nL, nspread, bin, inv_bin, n_rbins, tab, lsum,
                                             lcnt, WRITE_SV ? sv_c + row_off + sA : nullptr, xmax, xmin, nnan, n_clean, xf,
                                             xl, dsc, dof, DEPTH == 2 ? dp_c + row_off + sA : nullptr,
                                             SRA_LDS ? col_sra[eA] : col[0].sra, SRA_LDS ? col_sra[eA + 1] : col[1].sra);
      if (hasB)
        process_pair<T, RMAX, WRITE_SV, DEPTH>(col[2], col[3], inB, clean, r, r0v, g, a2, A0,
                                               NL_LDS ? col_nL[eA + 128] : col[2].nL, NL_LDS ? col_nL[eA + 129] : col[3].nL,
                                               nspread, bin, inv_bin, n_rbins, tab, lsum, lcnt,
                                               WRITE_SV ? sv_c + row_off + sB : nullptr, xmax, xmin, nnan, n_clean, xdummy,
                                               xl, dsc, dof, DEPTH == 2 ? dp_c + row_off + sB : nullptr,
                                               SRA_LDS ? col_sra[eA + 128] : col[2].sra,
                                               SRA_LDS ? col_sra[eA + 129] : col[3].sra);
      if (clean) {  // (scalar)
        if (RMAX) {  // no NaN among the wavefront's samples: the lane's smallest / largest range of the ping
          if (DEPTH) {  // (depth falls with the sample number for an upward-looking transducer: either end is either)
            xmin = vmin_num(vmin_num(xmin, xf), xl);
            xmax = vmax_num(vmax_num(xmax, xf), xl);
          } else {
            xmin = vmin_num(xmin, xf);
            xmax = vmax_num(xmax, xl);
          }
        }
        ++n_clean;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) col[j].flush(lsum, lcnt, n_clean);
  }
  }
  if (RMAX) {
    // {max, min, NaN count} of the binned coordinate seen by this workgroup: wavefront reduction, then the workgroup's
    // four wavefronts meet in LDS, then ONE lane sends at most three atomics WITHOUT a return value -- nobody waits for
    // them.  Measured round 6 (4 x 100 000 x 2000 fp64, ms per launch): one atomic per wavefront and key 1.96-2.12
    // (2.5 with the three words in one 128-byte line: the L2 serialises a line's atomics); per workgroup, filtered by a
    // read of the current keys 2.05 (the read is what a workgroup then waits for); per workgroup, unconditional 1.82 =
    // the kernel without the statistics.  The words live in different 128-byte lines where the caller's layout allows.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_down(xmax, o, 64));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmin = fmin(xmin, __shfl_down(xmin, o, 64));  // (nnan is the wavefront's already)
    if (lane == 0) {
      if (xmax > -__builtin_inf()) atomicMax(&wg_keys[0], ordered_key(xmax));
      if (xmin < __builtin_inf()) atomicMin(&wg_keys[1], ordered_key(xmin));
      if (nnan > 0u) atomicAdd(&wg_nnan, nnan);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long kmax = wg_keys[0], kmin = wg_keys[1];
      if (kmax != 0ull) atomicMax(a.rmax_key, kmax);  // (key 0 = nothing seen)
      if (a.rstat) {  // the rest of {nanmin, nanmax, NaN count}: what compute_MVBS asks of its range variable
        if (kmin != ~0ull) atomicMin(a.rstat, kmin);  // (key ~0 = nothing seen)
        if (wg_nnan > 0u) atomicAdd(a.rstat + 1, (unsigned long long)wg_nnan);
      }
    }
  }
  if (extra) return;
  __syncthreads();
  const size_t cell0 = ((size_t)c * a.n_tbins + tb) * n_rbins;
  T* out = mvbs_out + cell0;
  T* gsum = sum_out ? sum_out + cell0 : nullptr;
  uint32_t* gcnt = cnt_out ? cnt_out + cell0 : nullptr;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    const uint32_t n = lcnt[i];
    const T s = lsum[i];
    out[i] = n > 0u ? (T)10 * epa::M<T>::log10(s / (T)n) : (T)a.fill_value;
    if (gsum) gsum[i] = s;
    if (gcnt) gcnt[i] = n;
  }
}

template <typename T, typename RawT>
int launch(Args& a, const RawT* raw, const int32_t* n_valid, const double* coef, const int32_t* bin_start,
           void* sv_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out, int C, size_t lds_bytes,
           hipStream_t st) {
  const dim3 grid((unsigned)a.n_tbins + 1u, (unsigned)C);  // +1: pings outside every time bin
  a.tab_off = (unsigned)((lds_bytes + 15) & ~(size_t)15);
  lds_bytes = a.tab_off + epa::kMathTabBytes;
  a.xcd_map = epa::xcd_map_enabled() ? 1 : 0;
#define EPA_FL(W, R) EPA_FLD(W, R, 0)
#define EPA_FLD(W, R, D)                                                                       \
  do {                                                                                         \
    auto kern = fused_sv_mvbs_kernel<T, RawT, W, R, D>;                                        \
    if (lds_bytes > 64 * 1024)                                                                 \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                        (int)lds_bytes));                                      \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds_bytes, st, raw, n_valid,             \
                       reinterpret_cast<const epa::CoefRow*>(coef), bin_start, (T*)sv_out,     \
                       (T*)mvbs_out, (T*)sum_out, cnt_out, a.dscale, a.doffset, a);            \
  } while (0)
  if (a.dscale) {  // binned on depth (float samples, always with the statistics of the depth)
    if constexpr (std::is_same<RawT, float>::value) {
      if (a.depth_out) EPA_FLD(true, true, 2);
      else if (sv_out) EPA_FLD(true, true, 1);
      else EPA_FLD(false, true, 1);
    } else {
      epa::set_error("epa_sv_mvbs_fused_depth: float power samples only");
      return EPA_EUNSUPPORTED;
    }
  } else if (a.rmax_key) {
    if (sv_out) EPA_FL(true, true); else EPA_FL(false, true);
  } else {
    if (sv_out) EPA_FL(true, false); else EPA_FL(false, false);
  }
#undef EPA_FLD
#undef EPA_FL
  return epa::check_launch("fused_sv_mvbs_kernel");
}

__global__ __launch_bounds__(epa::kBlock) void selftest_lin_kernel(const double* __restrict__ u,
                                                                   double* __restrict__ out,
                                                                   size_t n) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = epa::lin_from_db(u[i], mt.exp2_tab);
}

__global__ __launch_bounds__(epa::kBlock) void selftest_log_kernel(const double* __restrict__ x,
                                                                   double* __restrict__ out,
                                                                   size_t n) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = epa::fast_log10(x[i], mt.log_tab);
}

__global__ __launch_bounds__(epa::kBlock) void selftest_log_inl_kernel(const double* __restrict__ x,
                                                                       double* __restrict__ out, size_t n) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = epa::fast_log10_inl<false>(x[i], mt.log_tab);
}

// ---- the same reduction on an EXISTING Sv whose echo_range is given by the coefficient rows ------------------------------
// compute_MVBS(ds_Sv) after compute_Sv left echo_range lazy (epa_mvbs with coef, no range array): Sv is read (8 or 4
// B/sample, nothing written but the MVBS), the range of a sample is evaluated as in the fused kernel and -- AS_STORED --
// rounded to T first, i.e. binned exactly as the echo_range array would be.  Same lane map, same software prefetch, same
// lane-private accumulation over the pings of a time bin; a NaN Sv is skipped (skipna), a NaN coefficient row bins
// nothing.
template <typename T>
struct SvColumn {
  double sra, blo, bhi;
  T acc_sum;
  int acc_rb;
  uint32_t acc_cnt;
  __device__ __forceinline__ void init() { sra = 0.0; blo = 1.0; bhi = 0.0; acc_sum = (T)0; acc_rb = -1; acc_cnt = 0u; }
  __device__ __forceinline__ void flush(T* lsum, uint32_t* lcnt) {
    if (acc_rb >= 0 && acc_cnt > 0u) {
      lds_add(lsum + acc_rb, acc_sum);
      atomicAdd(lcnt + acc_rb, acc_cnt);
    }
  }
};

template <typename T, bool AS_STORED>
__device__ __forceinline__ void bin_sv_sample(SvColumn<T>& c, T sv, const epa::CoefRow& r, double bin, double inv_bin,
                                              int n_rbins, const double* tab, T* lsum, uint32_t* lcnt) {
  const double x0 = c.sra * r.rb + r.r0;
  const double x = AS_STORED ? (double)(T)x0 : x0;
  const T v = epa::lin_from_db(sv, tab);
  const bool same = (x >= c.blo) & (x < c.bhi);
  if (!same) {
    const int rb = (x == x) ? epa::range_bin_index(x, bin, inv_bin, n_rbins, false) : -1;
    if (rb != c.acc_rb) {
      c.flush(lsum, lcnt);
      c.acc_rb = rb;
      c.acc_sum = (T)0;
      c.acc_cnt = 0u;
    }
    c.blo = rb >= 0 ? (double)rb * bin : 1.0;
    c.bhi = rb >= 0 ? (double)(rb + 1) * bin : 0.0;
  }
  const bool take = (c.acc_rb >= 0) & (v == v);
  c.acc_sum += take ? v : (T)0;
  c.acc_cnt += take ? 1u : 0u;
}

// ---- the time bins whose pings put (almost) every range column into ONE range bin ------------------------------------
// The range of column s at ping p is fl(fl(s ra) rb_p) + r0: with ra and r0 shared by the pings of a time bin it is
// monotone in rb_p, so a column whose range at the bin's smallest and largest rb falls into the same range bin stays
// there for every ping.  Such columns need neither the range nor the two edge compares per sample -- the sweep is a
// plain streaming sum of 10^(Sv/10) into a fixed bin (about 25 VALU instructions per sample, under 64 VGPRs: twice the
// wavefronts of the per-sample form in flight).  The few columns near a range-bin edge are redone with the wavefront's
// lanes spread over the pings.  A time bin whose pings differ in ra or r0, or longer than kFixedPings, is marked in its
// first MVBS cell (a NaN payload no computation produces) and left to mvbs_of_sv_rows_kernel, launched right after.
constexpr int kFixedPings = 512;
template <typename T>
struct LeftMark;
template <>
struct LeftMark<double> {
  static constexpr unsigned long long kBits = 0x7ff8dead0c0ffee2ull;
};
template <>
struct LeftMark<float> {
  static constexpr unsigned kBits = 0x7fcdead2u;
};
__device__ __forceinline__ bool left_to_rows(const double* cell) {
  return *reinterpret_cast<const unsigned long long*>(cell) == LeftMark<double>::kBits;
}
__device__ __forceinline__ bool left_to_rows(const float* cell) {
  return *reinterpret_cast<const unsigned*>(cell) == LeftMark<float>::kBits;
}
__device__ __forceinline__ void mark_left(double* cell) {
  *reinterpret_cast<unsigned long long*>(cell) = LeftMark<double>::kBits;
}
__device__ __forceinline__ void mark_left(float* cell) { *reinterpret_cast<unsigned*>(cell) = LeftMark<float>::kBits; }

// (wavefronts per SIMD: measured at 4 x 500 000 x 2000 -- fp64 6.30 ms at 8 (64 VGPRs, six dwords of scratch), 5.80 at 6,
//  the per-sample form 6.01; fp32 3.08 at 8, 3.17 at 6, the per-sample form 3.39)
template <typename T, bool AS_STORED>
__global__ __launch_bounds__(epa::kBlock, sizeof(T) == 8 ? 6 : 8) void mvbs_of_sv_fixed_kernel(
    const T* __restrict__ sv, const epa::CoefRow* __restrict__ coef, const int32_t* __restrict__ bin_start,
    T* __restrict__ mvbs_out, T* __restrict__ sum_out, uint32_t* __restrict__ cnt_out, Args a) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);  // synchronised below
  const double* tab = mt.exp2_tab;
  __shared__ int differs;
  __shared__ unsigned long long rb_lo_key, rb_hi_key;
  const int c = blockIdx.y, tb = a.xcd_map ? epa::xcd_contiguous(blockIdx.x, a.n_tbins) : (int)blockIdx.x;
  const int S = a.S, n_rbins = a.n_rbins;
  const int pb = bin_start[tb], pe = bin_start[tb + 1], np = pe - pb;
  const size_t cell0 = ((size_t)c * a.n_tbins + tb) * n_rbins;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  if (threadIdx.x == 0) {
    differs = np > kFixedPings ? 1 : 0;
    rb_lo_key = ~0ull;
    rb_hi_key = 0ull;
  }
  __syncthreads();
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const epa::CoefRow r = rowp0[np > 0 ? pb : 0];
  for (int i = threadIdx.x; i < min(np, kFixedPings); i += epa::kBlock) {
    const epa::CoefRow ri = rowp0[pb + i];
    if (!((ri.ra == r.ra) & (ri.r0 == r.r0) & (ri.rb == ri.rb) & (ri.ra > 0.0))) differs = 1;  // (a NaN row, too)
    atomicMin(&rb_lo_key, ordered_key(ri.rb));
    atomicMax(&rb_hi_key, ordered_key(ri.rb));
  }
  __syncthreads();
  if (differs) {
    if (threadIdx.x == 0) mark_left(mvbs_out + cell0);
    return;
  }
  auto unkey = [](unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
  };
  const double rb_lo = unkey(rb_lo_key), rb_hi = unkey(rb_hi_key);
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const T* __restrict__ sv_c = sv + (size_t)c * a.P * S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto range_of = [&](int sx, double rb) {
    const double x0 = ((double)sx * r.ra) * rb + r.r0;
    return AS_STORED ? (double)(T)x0 : x0;
  };
  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane;  // first sample of pair A
    const int sB = sA + 128;                        // first sample of pair B
    if (sA >= S) continue;
    const bool hasB = sB < S;
    int rbin[VEC];
    T acc_sum[VEC];
    uint32_t acc_cnt[VEC];
    unsigned fixed = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int sx = (j < 2 ? sA : sB) + (j & 1);
      const int b_lo = epa::range_bin_index(range_of(sx, rb_lo), bin, inv_bin, n_rbins, false);
      const int b_hi = epa::range_bin_index(range_of(sx, rb_hi), bin, inv_bin, n_rbins, false);
      rbin[j] = b_lo;
      if (b_lo == b_hi) fixed |= 1u << j;  // (out of the grid at both ends: fixed, nothing is taken)
      acc_sum[j] = (T)0;
      acc_cnt[j] = 0u;
    }
    // two pings in flight per lane
    T2 nA0 = {(T)0, (T)0}, nB0 = nA0, nA1 = nA0, nB1 = nA0;
    if (np > 0) {
      nA0 = *reinterpret_cast<const T2*>(sv_c + (size_t)pb * S + sA);
      if (hasB) nB0 = *reinterpret_cast<const T2*>(sv_c + (size_t)pb * S + sB);
    }
    if (np > 1) {
      nA1 = *reinterpret_cast<const T2*>(sv_c + (size_t)(pb + 1) * S + sA);
      if (hasB) nB1 = *reinterpret_cast<const T2*>(sv_c + (size_t)(pb + 1) * S + sB);
    }
    for (int p = pb; p < pe; ++p) {
      const T2 inA = nA0, inB = nB0;
      nA0 = nA1;
      nB0 = nB1;
      if (p + 2 < pe) {
        const T* nx = sv_c + (size_t)(p + 2) * S;
        nA1 = *reinterpret_cast<const T2*>(nx + sA);
        if (hasB) nB1 = *reinterpret_cast<const T2*>(nx + sB);
      }
      const T in[VEC] = {inA.x, inA.y, inB.x, inB.y};
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (j >= 2 && !hasB) break;
        // every column accumulates (v >= 0 or NaN: max(v, 0) adds nothing for a NaN); the loose ones and those outside
        // the grid are dropped below -- nothing per sample depends on the column's kind
        const T v = epa::lin_from_db_lean(in[j], tab);  // (+-inf through a rare branch)
        acc_sum[j] += vmax_num(v, (T)0);
        acc_cnt[j] += v == v ? 1u : 0u;
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (((fixed >> j) & 1u) != 0u && rbin[j] >= 0 && acc_cnt[j] > 0u) {
        lds_add(lsum + rbin[j], acc_sum[j]);
        atomicAdd(lcnt + rbin[j], acc_cnt[j]);
      }
    }
    // the columns near a range-bin edge: one at a time, the lanes spread over the pings
    const unsigned loose = (~fixed) & (hasB ? 0xfu : 0x3u);
    if (__ballot(loose != 0u) != 0ull) {
      // the lanes still here are 0 .. nact - 1 (sA grows with the lane): in the row's last, partial wavefront only
      // they exist to share the pings, so the stride is their number, not 64
      const int nact = (int)__popcll(__ballot(true));
#pragma unroll 1
      for (int j = 0; j < VEC; ++j) {
        unsigned long long todo = __ballot(((loose >> j) & 1u) != 0u);
        while (todo != 0ull) {  // (wave-uniform)
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1ull;
          const int sx = chunk0 + wave * 256 + 2 * src + (j < 2 ? 0 : 128) + (j & 1);
          for (int p = pb + lane; p < pe; p += nact) {
            const T v = epa::lin_from_db_lean(sv_c[(size_t)p * S + sx], tab);
            const double x = range_of(sx, rowp0[p].rb);
            const int rb = epa::range_bin_index(x, bin, inv_bin, n_rbins, false);
            if ((rb >= 0) & (v == v)) {
              lds_add(lsum + rb, v);
              atomicAdd(lcnt + rb, 1u);
            }
          }
        }
      }
    }
  }
  __syncthreads();
  T* out = mvbs_out + cell0;
  T* gsum = sum_out ? sum_out + cell0 : nullptr;
  uint32_t* gcnt = cnt_out ? cnt_out + cell0 : nullptr;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    const uint32_t n = lcnt[i];
    const T s = lsum[i];
    out[i] = n > 0u ? (T)10 * epa::M<T>::log10(s / (T)n) : (T)a.fill_value;
    if (gsum) gsum[i] = s;
    if (gcnt) gcnt[i] = n;
  }
}

template <typename T, bool AS_STORED>
__global__ __launch_bounds__(epa::kBlock, EPA_FUSED_MIN_WAVES) void mvbs_of_sv_rows_kernel(
    const T* __restrict__ sv, const epa::CoefRow* __restrict__ coef, const int32_t* __restrict__ bin_start,
    T* __restrict__ mvbs_out, T* __restrict__ sum_out, uint32_t* __restrict__ cnt_out, Args a) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);  // synchronised below
  const double* tab = mt.exp2_tab;
  const int c = blockIdx.y, tb = a.xcd_map ? epa::xcd_contiguous(blockIdx.x, a.n_tbins) : (int)blockIdx.x;
  const int S = a.S, n_rbins = a.n_rbins;
  if (a.flagged_only && !left_to_rows(mvbs_out + ((size_t)c * a.n_tbins + tb) * n_rbins)) return;  // (uniform)
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  __syncthreads();
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const T* __restrict__ sv_c = sv + (size_t)c * a.P * S;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pb = bin_start[tb], pe = bin_start[tb + 1];
  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane;  // first sample of pair A
    const int sB = sA + 128;                        // first sample of pair B
    if (sA >= S) continue;
    const bool hasB = sB < S;
    SvColumn<T> col[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) col[j].init();
    double racur = __builtin_nan("");
    // the samples and the coefficient row of ping p + 1 are requested before ping p is processed
    T2 nA = {(T)0, (T)0}, nB = {(T)0, (T)0};
    epa::CoefRow nxtR = rowp0[pb < pe ? pb : 0];
    if (pb < pe) {
      nA = *reinterpret_cast<const T2*>(sv_c + (size_t)pb * S + sA);
      if (hasB) nB = *reinterpret_cast<const T2*>(sv_c + (size_t)pb * S + sB);
    }
    for (int p = pb; p < pe; ++p) {
      const epa::CoefRow r = nxtR;
      const T2 inA = nA, inB = nB;
      if (p + 1 < pe) {
        nxtR = rowp0[p + 1];
        const T* nx = sv_c + (size_t)(p + 1) * S;
        nA = *reinterpret_cast<const T2*>(nx + sA);
        if (hasB) nB = *reinterpret_cast<const T2*>(nx + sB);
      }
      if (!(r.ra == racur)) {  // uniform; once per column for a file with a constant sample_interval
        racur = r.ra;
        for (int j = 0; j < VEC; ++j) col[j].sra = (double)((j < 2 ? sA : sB) + (j & 1)) * r.ra;
      }
      bin_sv_sample<T, AS_STORED>(col[0], inA.x, r, bin, inv_bin, n_rbins, tab, lsum, lcnt);
      bin_sv_sample<T, AS_STORED>(col[1], inA.y, r, bin, inv_bin, n_rbins, tab, lsum, lcnt);
      if (hasB) {
        bin_sv_sample<T, AS_STORED>(col[2], inB.x, r, bin, inv_bin, n_rbins, tab, lsum, lcnt);
        bin_sv_sample<T, AS_STORED>(col[3], inB.y, r, bin, inv_bin, n_rbins, tab, lsum, lcnt);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) col[j].flush(lsum, lcnt);
  }
  __syncthreads();
  const size_t cell0 = ((size_t)c * a.n_tbins + tb) * n_rbins;
  T* out = mvbs_out + cell0;
  T* gsum = sum_out ? sum_out + cell0 : nullptr;
  uint32_t* gcnt = cnt_out ? cnt_out + cell0 : nullptr;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    const uint32_t n = lcnt[i];
    const T s = lsum[i];
    out[i] = n > 0u ? (T)10 * epa::M<T>::log10(s / (T)n) : (T)a.fill_value;
    if (gsum) gsum[i] = s;
    if (gcnt) gcnt[i] = n;
  }
}

template <typename T>
int launch_sv_rows(Args& a, const void* sv, const double* coef, const int32_t* bin_start, void* mvbs_out, void* sum_out,
                   uint32_t* cnt_out, int C, size_t lds_bytes, bool as_stored, hipStream_t st) {
  const dim3 grid((unsigned)a.n_tbins, (unsigned)C);
  a.tab_off = (unsigned)((lds_bytes + 15) & ~(size_t)15);
  lds_bytes = a.tab_off + epa::kMathTabBytes;
  a.xcd_map = epa::xcd_map_enabled() ? 1 : 0;
  static const bool fixed_off = [] {  // development knob: EPA_MVBS_FIXED=0 sends every time bin to the per-sample form
    const char* e = getenv("EPA_MVBS_FIXED");
    return e && e[0] == '0';
  }();
#define EPA_SR(KERN, AS)                                                                                 \
  do {                                                                                                   \
    auto kern = KERN<T, AS>;                                                                             \
    if (lds_bytes > 64 * 1024)                                                                           \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                             \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));    \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds_bytes, st, (const T*)sv,                       \
                       reinterpret_cast<const epa::CoefRow*>(coef), bin_start, (T*)mvbs_out, (T*)sum_out, \
                       cnt_out, a);                                                                      \
  } while (0)
  a.flagged_only = 0;
  if (!fixed_off) {
    if (as_stored) EPA_SR(mvbs_of_sv_fixed_kernel, true); else EPA_SR(mvbs_of_sv_fixed_kernel, false);
    if (int rc = epa::check_launch("mvbs_of_sv_fixed_kernel")) return rc;
    a.flagged_only = 1;
  }
  if (as_stored) EPA_SR(mvbs_of_sv_rows_kernel, true); else EPA_SR(mvbs_of_sv_rows_kernel, false);
#undef EPA_SR
  return epa::check_launch("mvbs_of_sv_rows_kernel");
}

}  // namespace epa_fused

// Called by epa_mvbs (block_reduce.hip): Sv in, range from the coefficient rows, default binning flags.
int epa_mvbs_rows_fast_path(const void* sv, const double* coef, int C, int P, int S, const int32_t* bin_start,
                            int n_tbins, double range_bin, int n_rbins, int as_stored, double fill_value,
                            void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype, size_t lds_bytes,
                            unsigned cnt_off, hipStream_t st) {
  epa_fused::Args a{};
  a.P = P; a.S = S; a.n_tbins = n_tbins; a.n_rbins = n_rbins;
  a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.fill_value = fill_value;
  a.skipna = 1;
  a.cnt_off = cnt_off;
  if (dtype == EPA_F64)
    return epa_fused::launch_sv_rows<double>(a, sv, coef, bin_start, mvbs_out, sum_out, cnt_out, C, lds_bytes,
                                             as_stored != 0, st);
  return epa_fused::launch_sv_rows<float>(a, sv, coef, bin_start, mvbs_out, sum_out, cnt_out, C, lds_bytes,
                                          as_stored != 0, st);
}

extern "C" int epa_selftest_lin_from_db(const double* u, double* out, size_t n, epa_stream_t stream) {
  EPA_CHECK_ARG(u && out, "epa_selftest_lin_from_db: NULL array argument");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  hipLaunchKernelGGL(epa_fused::selftest_lin_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)),
                     dim3(epa::kBlock), 0, (hipStream_t)stream, u, out, n);
  return epa::check_launch("selftest_lin_kernel");
}

extern "C" int epa_selftest_log10(const double* x, double* out, size_t n, epa_stream_t stream) {
  EPA_CHECK_ARG(x && out, "epa_selftest_log10: NULL array argument");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  hipLaunchKernelGGL(epa_fused::selftest_log_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)),
                     dim3(epa::kBlock), 0, (hipStream_t)stream, x, out, n);
  return epa::check_launch("selftest_log_kernel");
}

extern "C" int epa_selftest_log10_inline(const double* x, double* out, size_t n, epa_stream_t stream) {
  EPA_CHECK_ARG(x && out, "epa_selftest_log10_inline: NULL array argument");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  hipLaunchKernelGGL(epa_fused::selftest_log_inl_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)),
                     dim3(epa::kBlock), 0, (hipStream_t)stream, x, out, n);
  return epa::check_launch("selftest_log_inl_kernel");
}

// Called by epa_sv_mvbs_fused (block_reduce.hip) when the fast path applies.
int epa_fused_fast_path(const void* raw, int raw_is_i16, const int32_t* n_valid, const double* coef,
                        int C, int P, int S, double nspread,
                        unsigned cal_flags, const int32_t* bin_start, int n_tbins, double range_bin,
                        int n_rbins, unsigned bin_flags, double fill_value, void* sv_out,
                        void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                        size_t lds_bytes, unsigned cnt_off, unsigned long long* rmax_key,
                        unsigned long long* rstat, const double* dscale, const double* doffset, void* depth_out,
                        hipStream_t st) {
  epa_fused::Args a{};
  a.dscale = dscale; a.doffset = doffset; a.depth_out = dscale ? depth_out : nullptr;
  a.rmax_key = rmax_key;
  a.rstat = rmax_key ? rstat : nullptr;
  a.P = P; a.S = S; a.n_tbins = n_tbins; a.n_rbins = n_rbins;
  a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.nspread = nspread; a.fill_value = fill_value;
  a.guard = (cal_flags & EPA_FLAG_GUARD_POS) != 0;
  a.mask_range = (cal_flags & EPA_FLAG_MASK_RANGE) != 0;
  a.skipna = (bin_flags & EPA_BIN_SKIPNA) != 0;
  a.closed_right = (bin_flags & EPA_BIN_CLOSED_RIGHT) != 0;
  a.cnt_off = cnt_off;
#define EPA_GO(T, RawT)                                                                           \
  return epa_fused::launch<T, RawT>(a, reinterpret_cast<const RawT*>(raw), n_valid, coef, bin_start, \
                                    sv_out, mvbs_out, sum_out, cnt_out, C, lds_bytes, st)
  if (raw_is_i16) {
    if (dtype == EPA_F64) EPA_GO(double, int16_t);
    EPA_GO(float, int16_t);
  }
  if (dtype == EPA_F64) EPA_GO(double, float);
  EPA_GO(float, float);
#undef EPA_GO
}
