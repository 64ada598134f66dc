// Fast paths of the two-pass chain compute_Sv -> remove_background_noise -> compute_MVBS(Sv_corrected)
// for the hot configuration (EK power samples, R' <= 0 guard + masked range, sorted pings, skipna,
// left-closed bins, range grid in LDS): the structure of the headline kernel (fused_sv_mvbs.hip) --
// one workgroup per (channel, ping group), every lane on two sample PAIRS so that each load / store of a
// wavefront is one contiguous run, coefficient rows by scalar loads, logs cached per range column --
// applied to the two sources that the generic kernel of block_reduce.hip otherwise serves:
//
//   pass 1  sv_noise_fast_kernel      raw -> Sv (written once) + the noise estimate of
//                                     clean/api.py:397-422 per (channel, ping block)
//   pass 2  sv_denoise_mvbs_fast_kernel  raw -> Sv_noise / Sv_corrected (clean/api.py:425-430,485-487)
//                                     + MVBS of Sv_corrected (commongrid/utils.py:592-627) in the same sweep
//
// Per sample, fp64: pass 1 one exp10; pass 2 two exp10 + one log10 (table-driven, fast_math.h).  The
// transmission loss 20 log10(R) is separable like the spreading term: log10(R) = log10(k) + log10(s - d),
// d = -r0/k, cached per column.  Same arithmetic as the generic kernel (tests hold the two to <= 1e-9).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>

#include "fast_math.h"
#include "sample_math.h"

namespace epa_chain {

constexpr int VEC = 4;
constexpr int kChunk = epa::kBlock * VEC;

struct Args {
  int P, S;
  double nspread;
  // pass 1
  int ping_num, rsn, n_pblocks, n_rblocks;
  double noise_max;
  // a ping shard of a longer file: local ping p belongs to noise block (p + ping_phase) / ping_num (both passes); pass 1
  // also leaves the raw (sum, count) rows of its first / last block for the cross-shard merge (epa_noise_estimate's layout)
  int ping_phase;
  double* edge_sum;
  uint32_t* edge_cnt;
  // pass 2
  int n_tbins, n_rbins, noise_ping_num;
  double range_bin, inv_range_bin, fill_value, snr;
  unsigned cnt_off, tab_off;
  unsigned long long* rmax_key;
  unsigned long long* rstat;  // optional, with rmax_key: {min valid echo_range as a key, number of NaN echo_range values}
  int xcd_map;  // ping blocks / time-bin groups dealt to the XCDs in contiguous eighths (epa::xcd_contiguous)
  unsigned long long* mm_keys;  // pass 2, optional [4]: min/max of Sv_noise, min/max of Sv_corrected
  int flagged_only;  // pass 2, general kernel after the uniform-group kernel: only the groups that one left (kLeftToGeneral)
  int uni_bins;      // pass 2, uniform-group kernel: time bins per workgroup
};

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

__device__ __forceinline__ float vmin_num(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long b = __double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// log10 on the rare paths (column refresh, rounding residue).  Call-free by default: an out-of-line call inside
// the ping loop makes the compiler spill the ~90 live SGPRs around it (v_readlane / v_writelane per ping).
#ifdef EPA_CHAIN_CALLS
template <typename T>
__device__ __noinline__ T log10_slow(T x, const double2*) {
  return epa::M<T>::log10(x);
}
template <typename T>
__device__ __forceinline__ T log10_pos(T x, const double2* tab) {
  return epa::fast_log10(x, tab);
}
template <typename T>
__device__ __forceinline__ T log10_lin(T x, const double2* tab) {
  return epa::fast_log10(x, tab);
}
#else
// fp64: table-driven and branch-free (measured +7 % on pass 2, +4 % on pass 1).  fp32 keeps the call: ocml's
// log10f inlined at ten rare sites costs more registers than the spills it saves (measured -7 %).
__device__ __forceinline__ double log10_slow(double x, const double2* tab) {
  return epa::fast_log10_inl<false>(x, tab);
}
__device__ __noinline__ float log10_slow(float x, const double2*) { return ::log10f(x); }
__device__ __forceinline__ double log10_pos(double x, const double2* tab) {
  return epa::fast_log10_inl<true>(x, tab);
}
// per-sample log of the denoised linear value: no exact-at-1 select (a result of 1e-17 dB instead of 0)
__device__ __forceinline__ double log10_lin(double x, const double2* tab) {
  return epa::fast_log10_inl<true, false>(x, tab);
}
__device__ __forceinline__ float log10_lin(float x, const double2*) { return ::log10f(x); }
__device__ __forceinline__ float log10_pos(float x, const double2*) { return ::log10f(x); }
#endif
// The spreading log of a column stays ocml's log10 so that Sv is bit-identical to sv_power.hip / fused_sv_mvbs.hip.
template <typename T>
__device__ __noinline__ T log10_exact(T x) {
  return epa::M<T>::log10(x);
}

template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
  unsafeAtomicAdd(p, v);
}

// what both passes keep per range column across the pings of a group
template <typename T>
struct ColBase {
  double sra;  // fl(s * ra)
  T nL;        // n * log10(s - d)        spreading (TVG-shifted range)
  T lgs;       // log10(s - d_tl)         transmission loss (unshifted range)
  T c2;        // (s - d)^(n/10)          the spreading term in the linear domain; 0 where s - d <= 0
};

// echo_range, R', Sv of one sample -- as process_sample of fused_sv_mvbs.hip
// (r0v / A0: the ping's r0 and A0, which the callers keep in vector registers -- a VALU instruction takes one scalar
// operand, the compiler otherwise copies the second one per sample)
template <typename T>
__device__ __forceinline__ T calibrate(const ColBase<T>& c, float raw, const epa::CoefRow& r, double r0v, T g, T a2, T A0,
                                       T nspread, double& x, const double2* log_tab) {
  const T NaN = epa::M<T>::nan();
  x = fma(c.sra, r.rb, r0v);
  const double rtd = x - r.shift;
  const T rt = (T)rtd;
  const bool pos = rtd > 0.0;
  T s1 = fma(g, (T)raw, c.nL);
  if (pos & !(c.nL > -(T)__builtin_inf()))  // rounding residue of R - shift (rare)
    s1 = fma(g, (T)raw, nspread * (log10_exact<T>(rt) - log10_exact<T>((T)(r.ra * r.rb))));
  s1 = pos ? s1 : NaN;
  return s1 + fma(a2, rt, A0);
}

// clean/api.py:397-398: 20 log10(R if R >= 1 else 1) + 2 alpha R, R = echo_range (NaN where masked)
template <typename T>
__device__ __forceinline__ T transmission_loss(const ColBase<T>& c, T xr, T log10k, T na2) {
  return (T)20 * (xr >= (T)1 ? c.lgs + log10k : (T)0) + na2 * xr;
}

// log10(k), k = ra * rb, of the first kPingLogs pings of a workgroup's ping group: every wavefront needs it for
// every ping of every chunk (a uniform ~35-instruction log); computed once, one ping per lane, kept in LDS.
constexpr int kPingLogs = 64;
template <typename T>
__device__ __forceinline__ void fill_ping_logs(T* plog, const epa::CoefRow* __restrict__ rows, int n,
                                               const double2* log_tab) {
  for (int i = threadIdx.x; i < min(n, kPingLogs); i += epa::kBlock)
    plog[i] = log10_pos((T)(rows[i].ra * rows[i].rb), log_tab);
}

// Pass 2 in fp64 forms the LINEAR values as products instead of exponentials of sums (the kernel is bound by VALU
// issue, two exp10 + one log10 per sample; this leaves one exp10 + one log10):
//   10^(Sv/10)       = 10^(g raw/10) . (s - d)^(n/10) . 10^((A0 - a2 shift + a2 r0)/10) . E(s)
//   10^(Sv_noise/10) = 10^((noise + a2 r0)/10) . (R >= 1 ? R^2 : 1) . E(s)
//   E(s) = 10^(a2 k s / 10),  k = ra rb                      (absorption over the unshifted range k s)
// E is a geometric progression along the range: the lane evaluates it once per ping at its first column and
// reaches its other three columns (s + 1, s + 128, s + 129) with the per-ping ratios q1, q128, q129.  The per-ping
// constants come from LDS, one ping per lane (fill_ping_consts), like the log10(k) table.
struct PingConst {
  double a2k;   // a2 * k                       [dB per sample index]; a2 = the noise removal's 2 alpha
  double da2k;  // (a2 of the calibration - a2) * k: 0 unless the caller passed two different absorptions
  double q1, q128, q129;
  double csv;   // 10^((A0 - a2 shift + a2 r0)/10)
  double cn;    // 10^((noise of the ping's block + a2 r0)/10)
};
__device__ __forceinline__ PingConst ping_const(const epa::CoefRow& r, double na2, double nb, const double* exp2_tab) {
  PingConst c;
  const double k = r.ra * r.rb;
  c.a2k = na2 * k;
  c.da2k = (r.alpha2 - na2) * k;
  c.q1 = epa::lin_from_db(c.a2k, exp2_tab);
  c.q128 = epa::lin_from_db(128.0 * c.a2k, exp2_tab);
  c.q129 = c.q1 * c.q128;
  c.csv = epa::lin_from_db(r.A0 - r.alpha2 * r.shift + r.alpha2 * r.r0, exp2_tab);
  c.cn = epa::lin_from_db(nb + na2 * r.r0, exp2_tab);
  return c;
}
__device__ __forceinline__ void fill_ping_consts(PingConst* pc, const epa::CoefRow* __restrict__ rows,
                                                 const double* __restrict__ a2, const double* __restrict__ noise,
                                                 int p0, int n, int ping_num, const double* exp2_tab) {
  for (int i = threadIdx.x; i < min(n, kPingLogs); i += epa::kBlock)
    pc[i] = ping_const(rows[i], a2[i], noise[(p0 + i) / ping_num], exp2_tab);
}

// refresh of the cached column logs when the row constants they depend on change (uniform branch; once
// per column for a file with constant pulse length / sample interval / sound speed)
template <typename T>
struct RowCache {
  double d, ra, rb, dtl;
  T log10k;
  __device__ __forceinline__ RowCache() : d(__builtin_nan("")), ra(d), rb(d), dtl(d), log10k((T)0) {}
  // plog / pi: log10(k) of ping pi of the group, computed once per workgroup (fill_ping_logs), pi < kPingLogs
  template <typename C>
  __device__ __forceinline__ void update(const epa::CoefRow& r, C (&col)[VEC], int sA, int sB, T nspread,
                                         const double2* log_tab, const T* plog, int pi) {
    if (!((r.d == d) & (r.ra == ra))) {
      d = r.d;
      for (int j = 0; j < VEC; ++j) {
        const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
        col[j].nL = nspread * log10_exact<T>((T)(sj - r.d));
        col[j].sra = sj * r.ra;
        const T v = (T)(sj - r.d), v2 = v * v;
        col[j].c2 = v > (T)0 ? (nspread == (T)20 ? v2 : v2 * v2) : (T)0;
      }
    }
    const double k = r.ra * r.rb;
    if (pi < kPingLogs) {
      log10k = plog[pi];
    } else if (!((r.ra == ra) & (r.rb == rb))) {  // sound speed may drift from ping to ping: table-driven log
      log10k = log10_pos((T)k, log_tab);
    }
    rb = r.rb;
    ra = r.ra;
    const double dnew = r.r0 == 0.0 ? 0.0 : -r.r0 / k;  // EK rows: echo_range starts at 0
    if (!(dnew == dtl)) {
      dtl = dnew;
      for (int j = 0; j < VEC; ++j) {
        const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
        col[j].lgs = log10_slow((T)(sj - dnew), log_tab);
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------------
constexpr int kP1Blocks = 2;  // ping blocks per workgroup of pass 1

// Pass 1 keeps the two cached logarithms of a range column -- n log10(s - d) and log10(s - d_tl) -- in LDS, every lane
// its own four entries (written and read by the same lane: no barrier), instead of sixteen registers: with the sample
// temporaries of two samples at a time that is what fits four wavefronts per SIMD without scratch.
template <typename T>
struct NoiseCol {
  double sra;  // fl(s * ra)
  T acc_sum;
  uint32_t acc_cnt;
};

// The workgroup's {min, max} pairs (already reduced over each wavefront; an empty pair has min > max) -> global keys.
// The four wavefronts meet in LDS and ONE lane sends at most four atomics without a return value: the L2 serialises
// the atomics of a 128-byte line, and one per wavefront and key (6.4 M per 0.8 G samples on the four keys' line) is
// what a launch then waits for (round 6, measured on the fused kernel: fused_sv_mvbs.hip).  Every lane of the
// workgroup must call it (two barriers).
#ifndef EPA_WG_KEYS
#define EPA_WG_KEYS 1
#endif
template <typename U>
__device__ __forceinline__ void wg_minmax_keys(const U (&mm_)[4], unsigned long long* keys, int lane) {
  const double mm[4] = {(double)mm_[0], (double)mm_[1], (double)mm_[2], (double)mm_[3]};
#if !EPA_WG_KEYS  // (development knob: the round-5 form, one atomic per wavefront and key)
  if (lane == 0) {
    if (mm[0] <= mm[1]) {
      atomicMin(keys + 0, ordered_key(mm[0]));
      atomicMax(keys + 1, ordered_key(mm[1]));
    }
    if (mm[2] <= mm[3]) {
      atomicMin(keys + 2, ordered_key(mm[2]));
      atomicMax(keys + 3, ordered_key(mm[3]));
    }
  }
  return;
#endif
  __shared__ unsigned long long wk[4];
  if (threadIdx.x == 0) {
    wk[0] = wk[2] = ~0ull;
    wk[1] = wk[3] = 0ull;
  }
  __syncthreads();
  if (lane == 0) {
    if (mm[0] <= mm[1]) {
      atomicMin(&wk[0], ordered_key(mm[0]));
      atomicMax(&wk[1], ordered_key(mm[1]));
    }
    if (mm[2] <= mm[3]) {
      atomicMin(&wk[2], ordered_key(mm[2]));
      atomicMax(&wk[3], ordered_key(mm[3]));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (wk[0] != ~0ull) atomicMin(keys + 0, wk[0]);
    if (wk[1] != 0ull) atomicMax(keys + 1, wk[1]);
    if (wk[2] != ~0ull) atomicMin(keys + 2, wk[2]);
    if (wk[3] != 0ull) atomicMax(keys + 3, wk[3]);
  }
}

template <typename T, bool WRITE_SV, bool RMAX>
__global__ __launch_bounds__(epa::kBlock, sizeof(T) == 8 ? 4 : 1) void sv_noise_fast_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef,
    const double* __restrict__ alpha2, T* __restrict__ sv_out, double* __restrict__ noise_out, Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);
  __shared__ T red[8];

  // One workgroup takes kP1Blocks consecutive ping blocks: the per-column logs it caches (two ocml-grade logs per
  // column and chunk) are then paid once per 40 pings instead of once per 20 (interleaved A/B on one box, 2 G samples:
  // 5.45 -> 5.27 ms with 2 blocks, 5.30 with 4: the larger footprint of a workgroup starts to cost what the fixed work
  // saves).  lsum / lcnt hold one row of range-block sums per ping block.
  // (the workgroups of one XCD walk one contiguous eighth of the ping blocks: epa::xcd_contiguous)
  const int c = blockIdx.y, pbk0 = (a.xcd_map ? epa::xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x) * kP1Blocks;
  const int nb = min(kP1Blocks, a.n_pblocks - pbk0);
  const int S = a.S, Sb = a.n_rblocks;
  for (int i = threadIdx.x; i < kP1Blocks * Sb; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  __syncthreads();
  const T nspread = (T)a.nspread;
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const double* __restrict__ a2p = alpha2 + (size_t)c * a.P;
  const float* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  T* __restrict__ sv_c = WRITE_SV ? sv_out + (size_t)c * a.P * S : nullptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (a shard whose first ping is ping_phase pings into its first block: that block is shorter by as much)
  const int pb = max(0, pbk0 * a.ping_num - a.ping_phase), pe = min(a.P, (pbk0 + nb) * a.ping_num - a.ping_phase);
  double xmax = -__builtin_inf(), xmin = __builtin_inf();
  unsigned nnan = 0u;
  __shared__ T plog[kPingLogs];
  __shared__ T col_nL[kChunk], col_lgs[kChunk];
  // (the instance that also carries the echo_range statistics: fl(s * ra) of a column likewise -- eight registers, the
  // difference between 32 B of scratch per lane and none; a scratch reload in the ping loop waits for the stores in flight)
  constexpr bool SRA_LDS = RMAX && sizeof(T) == 8;
  __shared__ double col_sra[SRA_LDS ? kChunk : 1];
  fill_ping_logs<T>(plog, rowp0 + pb, pe - pb, mt.log_tab);
  __syncthreads();

  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane, sB = sA + 128;
    if (sA >= S) continue;
    const bool hasB = sB < S;
    const int eA = wave * 256 + 2 * lane;  // the lane's entries of col_nL / col_lgs: eA, eA + 1, eA + 128, eA + 129
    NoiseCol<T> col[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      col[j].sra = 0.0; col[j].acc_sum = (T)0; col[j].acc_cnt = 0u;
      col_nL[eA + (j < 2 ? 0 : 128) + (j & 1)] = epa::M<T>::nan();
      col_lgs[eA + (j < 2 ? 0 : 128) + (j & 1)] = (T)0;
    }
    int rbk[VEC];  // range block of each column
#pragma unroll
    for (int j = 0; j < VEC; ++j) rbk[j] = ((j < 2 ? sA : sB) + (j & 1)) / a.rsn;
    auto flush = [&](int g) {  // the columns' sums of ping block g -> its row of range-block sums
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (col[j].acc_cnt > 0u) {
          lds_add(lsum + g * Sb + rbk[j], col[j].acc_sum);
          atomicAdd(lcnt + g * Sb + rbk[j], col[j].acc_cnt);
        }
        col[j].acc_sum = (T)0;
        col[j].acc_cnt = 0u;
      }
    };
    RowCache<T> rc;
    float2 nA = make_float2(0.f, 0.f), nB = nA;
    epa::CoefRow nxtR = rowp0[pb];
    double nxtA2 = a2p[pb];  // (requested a ping ahead like the row: a scalar load consumed where it is issued stalls the ping)
    nA = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sA);
    if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sB);
    int left = pbk0 == 0 ? a.ping_num - a.ping_phase : a.ping_num, g = 0;  // pings left in the current ping block
    for (int p = pb; p < pe; ++p) {
      const epa::CoefRow r = nxtR;
      const double curA2 = nxtA2;
      const size_t row_off = (size_t)p * S;
      const float2 inA = nA, inB = nB;
      if (p + 1 < pe) {  // software prefetch of the next ping
        nxtR = rowp0[p + 1];
        nxtA2 = a2p[p + 1];
        nA = *reinterpret_cast<const float2*>(raw_c + row_off + S + sA);
        if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + row_off + S + sB);
      }
      // refresh of the cached column logs when the row constants they depend on change (as RowCache::update)
      if (!((r.d == rc.d) & (r.ra == rc.ra))) {
        rc.d = r.d;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
          col_nL[eA + (j < 2 ? 0 : 128) + (j & 1)] = nspread * log10_exact<T>((T)(sj - r.d));
          if (SRA_LDS) col_sra[eA + (j < 2 ? 0 : 128) + (j & 1)] = sj * r.ra;
          else col[j].sra = sj * r.ra;
        }
      }
      {
        const double k = r.ra * r.rb;
        if (p - pb < kPingLogs) {
          rc.log10k = plog[p - pb];
        } else if (!((r.ra == rc.ra) & (r.rb == rc.rb))) {  // sound speed may drift from ping to ping: table-driven log
          rc.log10k = log10_pos((T)k, mt.log_tab);
        }
        rc.rb = r.rb;
        rc.ra = r.ra;
        const double dnew = r.r0 == 0.0 ? 0.0 : -r.r0 / k;  // EK rows: echo_range starts at 0
        if (!(dnew == rc.dtl)) {
          rc.dtl = dnew;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
            col_lgs[eA + (j < 2 ? 0 : 128) + (j & 1)] = log10_slow((T)(sj - dnew), mt.log_tab);
          }
        }
      }
      const T g_ = (T)r.g, a2 = (T)r.alpha2, na2 = (T)curA2;
      T A0 = (T)r.A0;
      double r0v = r.r0;
      asm volatile("" : "+v"(A0), "+v"(r0v));  // one copy per ping into vector registers, not one per sample
      const float in[VEC] = {inA.x, inA.y, inB.x, inB.y};
      auto one = [&](int j) -> T {
        double x;
        const int e = eA + (j < 2 ? 0 : 128) + (j & 1);
        const ColBase<T> cb{SRA_LDS ? col_sra[e] : col[j].sra, col_nL[e], col_lgs[e], (T)0};
        const T svj = calibrate<T>(cb, in[j], r, r0v, g_, a2, A0, nspread, x, mt.log_tab);
        if (RMAX) {  // as stored (T); x + 0 * raw is the range or NaN, and v_max_f64 / v_min_f64 skip the NaN
          const double xq = fma((double)in[j], 0.0, (double)(T)x);
          xmax = vmax_num(xmax, xq);
          xmin = vmin_num(xmin, xq);
          nnan += (unsigned)__builtin_popcountll(__ballot(in[j] != in[j]));  // (per wavefront, in a scalar register)
        }
        // the block mean uses the UNMASKED range (the generic kernel does too: a masked sample has a NaN Sv)
        const T xr = (T)x;
        const T v = epa::lin_from_db_lean(svj - transmission_loss<T>(cb, xr, rc.log10k, na2), mt.exp2_tab);
        col[j].acc_sum += vmax_num(v, (T)0);  // v >= 0 or NaN: max(v, 0) adds nothing for a NaN
        col[j].acc_cnt += v == v ? 1u : 0u;
        return svj;
      };
      // pair A, its store, then pair B: two samples' temporaries alive at a time instead of four (147 -> <= 128 VGPRs)
      {
        const T s0 = one(0), s1 = one(1);
        if (WRITE_SV) epa::store_nt2(sv_c + row_off + sA, s0, s1);
      }
      if (hasB) {
        const T s2 = one(2), s3 = one(3);
        if (WRITE_SV) epa::store_nt2(sv_c + row_off + sB, s2, s3);
      }
      if (--left == 0) {  // (uniform) the ping block is complete
        flush(g);
        ++g;
        left = a.ping_num;
      }
    }
    if (g < nb) flush(g);  // the last, shorter ping block of the array (nothing left in the columns otherwise)
  }
  if (RMAX) {  // {max, min, NaN count} of the echo_range: per workgroup, one lane, no return value (see wg_minmax_keys)
    __shared__ unsigned long long wr[2];
    __shared__ unsigned wn;
    if (threadIdx.x == 0) {
      wr[0] = 0ull;
      wr[1] = ~0ull;
      wn = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_down(xmax, o, 64));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmin = fmin(xmin, __shfl_down(xmin, o, 64));  // (nnan is the wavefront's already)
    if (lane == 0) {
      if (xmax > -__builtin_inf()) atomicMax(&wr[0], ordered_key(xmax));
      if (xmin < __builtin_inf()) atomicMin(&wr[1], ordered_key(xmin));
      if (nnan > 0u) atomicAdd(&wn, nnan);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (wr[0] != 0ull) atomicMax(a.rmax_key, wr[0]);
      if (a.rstat) {  // the rest of {nanmin, nanmax, NaN count} of the echo_range
        if (wr[1] != ~0ull) atomicMin(a.rstat, wr[1]);
        if (wn > 0u) atomicAdd(a.rstat + 1, (unsigned long long)wn);
      }
    }
  }
  // min over the range blocks of 10 log10(block mean) (clean/api.py:402-411), optional clamp (:418-422)
  for (int g = 0; g < nb; ++g) {
    __syncthreads();  // the sums are complete / the previous block's `red` is consumed
    if (a.edge_sum && (pbk0 + g == 0 || pbk0 + g == a.n_pblocks - 1)) {  // (uniform) rows for the cross-shard merge
      const size_t e0 = ((size_t)(pbk0 + g == 0 ? 0 : 1) * gridDim.y + c) * Sb;  // (a single block: its first edge only)
      for (int i = threadIdx.x; i < Sb; i += epa::kBlock) {
        a.edge_sum[e0 + i] = (double)lsum[g * Sb + i];
        a.edge_cnt[e0 + i] = lcnt[g * Sb + i];
      }
    }
    T best = (T)__builtin_inf();
    int any = 0;
    for (int i = threadIdx.x; i < Sb; i += epa::kBlock) {
      const uint32_t n = lcnt[g * Sb + i];
      if (n > 0u) {
        const T db = (T)10 * epa::M<T>::log10(lsum[g * Sb + i] / (T)n);
        if (db == db) {
          best = fmin(best, db);
          any = 1;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      best = fmin(best, __shfl_down(best, o, 64));
      any |= __shfl_down(any, o, 64);
    }
    if (lane == 0) {
      red[wave] = best;
      red[4 + wave] = (T)any;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      T m = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
      const bool some = (red[4] != (T)0) | (red[5] != (T)0) | (red[6] != (T)0) | (red[7] != (T)0);
      double rr = some ? (double)m : __builtin_nan("");
      if (a.noise_max == a.noise_max) rr = (rr < a.noise_max) ? rr : a.noise_max;
      noise_out[(size_t)c * a.n_pblocks + pbk0 + g] = rr;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pass 2
// ------------------------------------------------------------------------------------------------
// written by the uniform-group kernel into the first MVBS cell of a group it leaves to the general kernel (a NaN
// payload no computation produces; the general kernel overwrites it)
constexpr unsigned long long kLeftToGeneral = 0x7ff8dead0c0ffee1ull;
constexpr unsigned kLeftToGeneral32 = 0x7fcdead1u;  // (float grids: a cell is four bytes)
__device__ __forceinline__ bool left_to_general(const double* cell) {
  return *reinterpret_cast<const unsigned long long*>(cell) == kLeftToGeneral;
}
__device__ __forceinline__ bool left_to_general(const float* cell) {
  return *reinterpret_cast<const unsigned*>(cell) == kLeftToGeneral32;
}
__device__ __forceinline__ void mark_left_to_general(double* cell) {
  *reinterpret_cast<unsigned long long*>(cell) = kLeftToGeneral;
}
__device__ __forceinline__ void mark_left_to_general(float* cell) { *reinterpret_cast<unsigned*>(cell) = kLeftToGeneral32; }

template <typename T>
struct BinCol : ColBase<T> {
  double blo, bhi;  // edges of the range bin the column currently sits in (empty: blo > bhi)
  T acc_sum;
  int acc_rb;
  uint32_t acc_cnt;
  __device__ __forceinline__ void flush(T* lsum, uint32_t* lcnt) {
    if (acc_rb >= 0 && acc_cnt > 0u) {
      lds_add(lsum + acc_rb, acc_sum);
      atomicAdd(lcnt + acc_rb, acc_cnt);
    }
  }
};

// at least 3 wavefronts per SIMD: the variant with every output and the min/max by-product wanted 174 VGPRs (two
// wavefronts); held to 168 it spills four registers and is 5 % faster
template <typename T, bool WRITE_NOISE, bool WRITE_CORR, bool MINMAX>
__global__ __launch_bounds__(epa::kBlock, 3) void sv_denoise_mvbs_fast_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef,
    const double* __restrict__ alpha2, const double* __restrict__ noise,
    const int32_t* __restrict__ bin_start, T* __restrict__ noise_out, T* __restrict__ corr_out,
    T* __restrict__ mvbs_out, T* __restrict__ sum_out, uint32_t* __restrict__ cnt_out, Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);

  const int c = blockIdx.y, tb = blockIdx.x;
  const int S = a.S, n_rbins = a.n_rbins;
  // blockIdx.x == n_tbins: pings that belong to no time bin still get their Sv_noise / Sv_corrected
  const bool extra = tb == a.n_tbins;
  if (extra && !(WRITE_NOISE || WRITE_CORR)) return;
  if (a.flagged_only && !extra && !left_to_general(mvbs_out + ((size_t)c * a.n_tbins + tb) * n_rbins))
    return;  // (uniform) the group was done by sv_denoise_mvbs_uniform_kernel
  const int nseg = extra ? 2 : 1;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  __syncthreads();

  const T nspread = (T)a.nspread, snr = (T)a.snr;
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const double* __restrict__ a2p = alpha2 + (size_t)c * a.P;
  const double* __restrict__ nzp = noise + (size_t)c * a.n_pblocks;
  const float* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  T* __restrict__ sn_c = WRITE_NOISE ? noise_out + (size_t)c * a.P * S : nullptr;
  T* __restrict__ sc_c = WRITE_CORR ? corr_out + (size_t)c * a.P * S : nullptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double mm[4] = {__builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf()};
  __shared__ T plog[kPingLogs];
  constexpr bool kProd = sizeof(T) == 8;  // linear values as products (see PingConst); fp32 keeps the exponentials
  __shared__ PingConst pcs[kProd ? kPingLogs : 1];

  for (int seg = 0; seg < nseg; ++seg) {
  const int pb = extra ? (seg == 0 ? 0 : bin_start[a.n_tbins]) : bin_start[tb];
  const int pe = extra ? (seg == 0 ? bin_start[0] : a.P) : bin_start[tb + 1];
  __syncthreads();  // all wavefronts are done with the previous segment's logs
  fill_ping_logs<T>(plog, rowp0 + pb, pe - pb, mt.log_tab);
  if (kProd) fill_ping_consts(pcs, rowp0 + pb, a2p + pb, nzp, pb + a.ping_phase, pe - pb, a.noise_ping_num, mt.exp2_tab);
  __syncthreads();
  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane, sB = sA + 128;
    if (sA >= S) continue;
    const bool hasB = sB < S;
    BinCol<T> col[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      col[j].sra = 0.0; col[j].nL = epa::M<T>::nan(); col[j].lgs = (T)0; col[j].c2 = (T)0;
      col[j].blo = 1.0; col[j].bhi = 0.0; col[j].acc_sum = (T)0; col[j].acc_rb = -1; col[j].acc_cnt = 0u;
    }
    RowCache<T> rc;
    float2 nA = make_float2(0.f, 0.f), nB = nA;
    // fp64 is short of SGPRs (about 90 spilled to VGPR lanes): there the next coefficient row is not prefetched
    constexpr bool kPrefetchRow = sizeof(T) == 4;
    epa::CoefRow nxtR = rowp0[pb < pe ? pb : 0];
    if (pb < pe) {
      nA = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sA);
      if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sB);
    }
    for (int p = pb; p < pe; ++p) {
      const epa::CoefRow r = kPrefetchRow ? nxtR : rowp0[p];
      const size_t row_off = (size_t)p * S;
      const float2 inA = nA, inB = nB;
      if (p + 1 < pe) {
        if (kPrefetchRow) nxtR = rowp0[p + 1];
        nA = *reinterpret_cast<const float2*>(raw_c + row_off + S + sA);
        if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + row_off + S + sB);
      }
      rc.update(r, col, sA, sB, nspread, mt.log_tab, plog, p - pb);
      const T g = (T)r.g, a2 = (T)r.alpha2, A0 = (T)r.A0, na2 = (T)a2p[p];
      const T nb = (T)nzp[(p + a.ping_phase) / a.noise_ping_num];
      const float in[VEC] = {inA.x, inA.y, inB.x, inB.y};
      T sn[VEC], sc[VEC];
      // product form (fp64): per-ping constants, E at the lane's first column, ratios to the other three
      PingConst pc{};
      T ecol[VEC] = {};
      if constexpr (kProd) {
        pc = (p - pb) < kPingLogs ? pcs[p - pb] : ping_const(r, (double)na2, (double)nb, mt.exp2_tab);
        const T e0 = (T)epa::lin_from_db_lean(pc.a2k * (double)sA, mt.exp2_tab);
        ecol[0] = e0; ecol[1] = e0 * (T)pc.q1; ecol[2] = e0 * (T)pc.q128; ecol[3] = e0 * (T)pc.q129;
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (j >= 2 && !hasB) break;
        BinCol<T>& cj = col[j];
        double x;
        const bool xok = in[j] == in[j];
        T lin;
        if constexpr (kProd) {
          x = cj.sra * r.rb + r.r0;
          const double rtd = x - r.shift;  // R' <= 0 -> NaN (calibrate_ek.py:107)
          T c2 = cj.c2;
          if ((rtd > 0.0) & !(c2 > (T)0)) {  // rounding residue of R - shift (rare): the range itself
            const T v = (T)(rtd / (r.ra * r.rb)), v2 = v * v;
            c2 = nspread == (T)20 ? v2 : v2 * v2;
          }
          c2 = rtd > 0.0 ? c2 : epa::M<T>::nan();
          const T xr = xok ? (T)x : epa::M<T>::nan();  // echo_range is NaN where the input is
          sn[j] = nb + transmission_loss<T>(cj, xr, rc.log10k, na2);
          // (R >= 1 ? R^2 : 1); a NaN sample needs no masking here: it makes lsv, hence lin, NaN by itself
          const T mx = fmax((T)x, (T)1), xx = mx * mx;
          T lsv = epa::lin_from_db_lean(g * (T)in[j], mt.exp2_tab) * (c2 * (T)pc.csv);
          if (pc.da2k != 0.0)  // (uniform, never through the Dataset API) the calibration used another absorption
            lsv *= (T)epa::lin_from_db_lean(pc.da2k * (double)((j < 2 ? sA : sB) + (j & 1)), mt.exp2_tab);
          lin = ecol[j] * (lsv - (T)pc.cn * xx);
        } else {
          const T sv = calibrate<T>(cj, in[j], r, r.r0, g, a2, A0, nspread, x, mt.log_tab);
          const T xr = xok ? (T)x : epa::M<T>::nan();  // echo_range is NaN where the input is
          sn[j] = nb + transmission_loss<T>(cj, xr, rc.log10k, na2);
          lin = epa::lin_from_db(sv, mt.exp2_tab) - epa::lin_from_db(sn[j], mt.exp2_tab);
        }
        const T corr = lin > (T)0 ? (T)10 * (kProd ? epa::fast_log10_lean(lin, mt.log_tab) : log10_lin(lin, mt.log_tab))
                                  : epa::M<T>::nan();
        const bool keep = corr - sn[j] > snr;
        sc[j] = keep ? corr : epa::M<T>::nan();
        if (MINMAX) {  // fmin / fmax ignore NaN operands
          mm[0] = fmin(mm[0], (double)sn[j]);
          mm[1] = fmax(mm[1], (double)sn[j]);
          mm[2] = fmin(mm[2], (double)sc[j]);
          mm[3] = fmax(mm[3], (double)sc[j]);
        }
        // range bin of the column (left-closed), as in the headline kernel
        const bool same = xok & (x >= cj.blo) & (x < cj.bhi);
        if (!same) {
          const int rb = xok ? epa::range_bin_index(x, bin, inv_bin, n_rbins, false) : -1;
          if (rb != cj.acc_rb) {
            if (!extra) cj.flush(lsum, lcnt);
            cj.acc_rb = rb;
            cj.acc_sum = (T)0;
            cj.acc_cnt = 0u;
          }
          cj.blo = rb >= 0 ? (double)rb * bin : 1.0;
          cj.bhi = rb >= 0 ? (double)(rb + 1) * bin : 0.0;
        }
        const bool take = (cj.acc_rb >= 0) & keep;  // keep implies a finite positive lin
        cj.acc_sum += take ? lin : (T)0;
        cj.acc_cnt += take ? 1u : 0u;
      }
      if (WRITE_NOISE) {
        epa::store_nt2(sn_c + row_off + sA, sn[0], sn[1]);
        if (hasB) epa::store_nt2(sn_c + row_off + sB, sn[2], sn[3]);
      }
      if (WRITE_CORR) {
        epa::store_nt2(sc_c + row_off + sA, sc[0], sc[1]);
        if (hasB) epa::store_nt2(sc_c + row_off + sB, sc[2], sc[3]);
      }
    }
    if (!extra) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) col[j].flush(lsum, lcnt);
    }
  }
  }
  if (MINMAX) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mm[0] = fmin(mm[0], __shfl_down(mm[0], o, 64));
      mm[1] = fmax(mm[1], __shfl_down(mm[1], o, 64));
      mm[2] = fmin(mm[2], __shfl_down(mm[2], o, 64));
      mm[3] = fmax(mm[3], __shfl_down(mm[3], o, 64));
    }
    wg_minmax_keys(mm, a.mm_keys, lane);
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

// ------------------------------------------------------------------------------------------------
// pass 2, ping groups whose pings share ONE range vector and ONE absorption (the sample interval, sound speed,
// pulse length and absorption of a file rarely change from ping to ping).  Then everything that depends on the range is
// a per-column constant of the group -- the spreading and absorption factors of the linear Sv, the linear noise shape,
// the transmission loss in dB, the range bin -- and a sample costs one exp10 (of g raw), one log10 and a few multiplies:
//   lin(Sv)       = 10^(g raw/10) . [ (s - d)^(n/10) E(s) ] . C_sv(ping)          E(s) = 10^(a2 k s / 10)
//   lin(Sv_noise) = C_n(ping) . [ max(R, 1)^2 E(s) ]
//   Sv_noise      = noise(ping) + [ 20 log10 max(R, 1) + a2 R ]
// (brackets: per column; C_sv, C_n as PingConst).  A workgroup that finds its group not uniform (or longer than
// kUniPings) marks the group's first MVBS cell and leaves it to sv_denoise_mvbs_fast_kernel, launched right after.
// ------------------------------------------------------------------------------------------------
constexpr int kUniPings = 256;
template <typename T>
struct PingLin {  // (the per-ping constants in the output's type: the float instance must not meet a double per sample)
  T g, csv, cn, nb;
};

// (round 6: templated on the output type -- the float chain's pass 2 ran the general kernel, 10.0 ms per 4 G samples)
template <typename T, bool WRITE_NOISE, bool WRITE_CORR, bool MINMAX>
__global__ __launch_bounds__(epa::kBlock, 4) void sv_denoise_mvbs_uniform_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef, const double* __restrict__ alpha2,
    const double* __restrict__ noise, const int32_t* __restrict__ bin_start, T* __restrict__ noise_out,
    T* __restrict__ corr_out, T* __restrict__ mvbs_out, T* __restrict__ sum_out,
    uint32_t* __restrict__ cnt_out, Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);
  __shared__ PingLin<T> pl[kUniPings];
  __shared__ int differs;

  // a workgroup takes a.uni_bins consecutive time bins (their pings are one run): the per-column constants are paid
  // once for all of them; lsum / lcnt hold one row of range bins per time bin
  const int c = blockIdx.y, tb0 = (a.xcd_map ? epa::xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x) * a.uni_bins;
  const int nbn = min(a.uni_bins, a.n_tbins - tb0);
  const int S = a.S, n_rbins = a.n_rbins;
  const int pb = bin_start[tb0], pe = bin_start[tb0 + nbn], np = pe - pb;
  const size_t cell0 = ((size_t)c * a.n_tbins + tb0) * n_rbins;
  for (int i = threadIdx.x; i < nbn * n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  if (threadIdx.x == 0) differs = np > kUniPings ? 1 : 0;
  __syncthreads();  // (also publishes the math tables)
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const double* __restrict__ a2p = alpha2 + (size_t)c * a.P;
  const double* __restrict__ nzp = noise + (size_t)c * a.n_pblocks;
  const epa::CoefRow r = rowp0[np > 0 ? pb : 0];
  const double na2 = a2p[np > 0 ? pb : 0];
  if ((int)threadIdx.x < min(np, kUniPings)) {
    const int p = pb + threadIdx.x;
    const epa::CoefRow ri = rowp0[p];
    const double a2i = a2p[p];
    const bool same = (ri.ra == r.ra) & (ri.rb == r.rb) & (ri.r0 == r.r0) & (ri.shift == r.shift) & (ri.d == r.d) &
                      (a2i == na2) & (ri.alpha2 == a2i);
    if (!same) differs = 1;
    const double nbi = nzp[(p + a.ping_phase) / a.noise_ping_num];
    pl[threadIdx.x] = PingLin<T>{(T)ri.g, (T)epa::lin_from_db(ri.A0 - ri.alpha2 * ri.shift + ri.alpha2 * ri.r0, mt.exp2_tab),
                                 (T)epa::lin_from_db(nbi + a2i * ri.r0, mt.exp2_tab), (T)nbi};
  }
  __syncthreads();
  if (differs) {
    if ((int)threadIdx.x < nbn) mark_left_to_general(mvbs_out + cell0 + (size_t)threadIdx.x * n_rbins);
    return;
  }

  const T nspread = (T)a.nspread, snr = (T)a.snr;
  const double k = r.ra * r.rb, a2k = na2 * k;
  const double dtl = r.r0 == 0.0 ? 0.0 : -r.r0 / k;  // EK rows: echo_range starts at 0
  const T log10k = log10_pos((T)k, mt.log_tab);
  const epa::LogCoef lk = epa::make_log_coef();
  const float* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  T* __restrict__ sn_c = WRITE_NOISE ? noise_out + (size_t)c * a.P * S : nullptr;
  T* __restrict__ sc_c = WRITE_CORR ? corr_out + (size_t)c * a.P * S : nullptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  T mm[4] = {(T)__builtin_inf(), -(T)__builtin_inf(), (T)__builtin_inf(), -(T)__builtin_inf()};

  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane, sB = sA + 128;
    if (sA >= S) continue;
    const bool hasB = sB < S;
    // ---- the group's per-column constants
    T c2E[VEC], xxE[VEC], sncol[VEC], acc_sum[VEC];
    int rbin[VEC];
    uint32_t acc_cnt[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
      const double x = (sj * r.ra) * r.rb + r.r0;  // echo_range (range.py:138 operation order)
      const double rtd = x - r.shift;              // R' <= 0 -> NaN (calibrate_ek.py:107)
      const T v = (T)(sj - r.d), v2 = v * v;
      T c2 = v > (T)0 ? (nspread == (T)20 ? v2 : v2 * v2) : (T)0;
      if ((rtd > 0.0) & !(c2 > (T)0)) {  // rounding residue of R - shift (rare): the range itself
        const T w = (T)(rtd / k), w2 = w * w;
        c2 = nspread == (T)20 ? w2 : w2 * w2;
      }
      c2 = rtd > 0.0 ? c2 : epa::M<T>::nan();
      const T E = (T)epa::lin_from_db_lean(a2k * sj, mt.exp2_tab);
      const T mx = fmax((T)x, (T)1);
      c2E[j] = c2 * E;
      xxE[j] = (mx * mx) * E;
      const T lgs = log10_slow((T)(sj - dtl), mt.log_tab);
      sncol[j] = (T)20 * (x >= 1.0 ? lgs + log10k : (T)0) + (T)na2 * (T)x;
      rbin[j] = epa::range_bin_index(x, a.range_bin, a.inv_range_bin, n_rbins, false);
      acc_sum[j] = (T)0;
      acc_cnt[j] = 0u;
    }
    auto flush = [&](int g) {  // the columns' sums of time bin g -> its row of range bins
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (rbin[j] >= 0 && acc_cnt[j] > 0u) {
          lds_add(lsum + g * n_rbins + rbin[j], acc_sum[j]);
          atomicAdd(lcnt + g * n_rbins + rbin[j], acc_cnt[j]);
        }
        acc_sum[j] = (T)0;
        acc_cnt[j] = 0u;
      }
    };
    float2 nA = make_float2(0.f, 0.f), nB = nA;
    if (np > 0) {
      nA = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sA);
      if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sB);
    }
    int g = 0, edge = bin_start[tb0 + 1];  // first ping of the next time bin
    for (int p = pb; p < pe; ++p) {
      while (p >= edge) {  // (uniform) ping p opens a later time bin; empty bins are stepped over
        flush(g);
        ++g;
        edge = bin_start[tb0 + g + 1];
      }
      const size_t row_off = (size_t)p * S;
      const float2 inA = nA, inB = nB;
      if (p + 1 < pe) {
        nA = *reinterpret_cast<const float2*>(raw_c + row_off + S + sA);
        if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + row_off + S + sB);
      }
      const PingLin<T> q = pl[p - pb];
      const float in[VEC] = {inA.x, inA.y, inB.x, inB.y};
      T sn[VEC], sc[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (j >= 2 && !hasB) break;
        const bool xok = in[j] == in[j];
        // a NaN sample needs no masking: it makes the exponential, hence lin, NaN by itself
        const T e = epa::lin_from_db_lean(q.g * (T)in[j], mt.exp2_tab);
        const T lin = fma(-q.cn, xxE[j], e * (c2E[j] * q.csv));
        const T corr = lin > (T)0 ? (T)10 * epa::fast_log10_lean(lin, mt.log_tab, lk) : epa::M<T>::nan();
        sn[j] = xok ? q.nb + sncol[j] : epa::M<T>::nan();  // echo_range is NaN where the input is
        const bool keep = corr - sn[j] > snr;
        sc[j] = keep ? corr : epa::M<T>::nan();
        if (MINMAX) {  // (IEEE minNum / maxNum: a NaN operand leaves the running value alone)
          mm[0] = vmin_num(mm[0], sn[j]);
          mm[1] = vmax_num(mm[1], sn[j]);
          mm[2] = vmin_num(mm[2], sc[j]);
          mm[3] = vmax_num(mm[3], sc[j]);
        }
        const bool take = (rbin[j] >= 0) & keep;  // keep implies a finite positive lin (and a valid input)
        acc_sum[j] += take ? lin : (T)0;
        acc_cnt[j] += take ? 1u : 0u;
      }
      if (WRITE_NOISE) {
        epa::store_nt2(sn_c + row_off + sA, sn[0], sn[1]);
        if (hasB) epa::store_nt2(sn_c + row_off + sB, sn[2], sn[3]);
      }
      if (WRITE_CORR) {
        epa::store_nt2(sc_c + row_off + sA, sc[0], sc[1]);
        if (hasB) epa::store_nt2(sc_c + row_off + sB, sc[2], sc[3]);
      }
    }
    flush(g);
  }
  if (MINMAX) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mm[0] = fmin(mm[0], __shfl_down(mm[0], o, 64));
      mm[1] = fmax(mm[1], __shfl_down(mm[1], o, 64));
      mm[2] = fmin(mm[2], __shfl_down(mm[2], o, 64));
      mm[3] = fmax(mm[3], __shfl_down(mm[3], o, 64));
    }
    wg_minmax_keys(mm, a.mm_keys, lane);
  }
  __syncthreads();
  T* out = mvbs_out + cell0;
  T* gsum = sum_out ? sum_out + cell0 : nullptr;
  uint32_t* gcnt = cnt_out ? cnt_out + cell0 : nullptr;
  for (int i = threadIdx.x; i < nbn * n_rbins; i += epa::kBlock) {
    const uint32_t n = lcnt[i];
    const T s = lsum[i];
    out[i] = n > 0u ? (T)10 * epa::M<T>::log10(s / (T)n) : (T)a.fill_value;
    if (gsum) gsum[i] = s;
    if (gcnt) gcnt[i] = n;
  }
}

// ------------------------------------------------------------------------------------------------
// pass 2, ping groups that the uniform kernel left because only the SOUND SPEED differs from ping to ping (an
// EK60 / EK80 records the sound speed of the moment with every ping: rb = c / 2 drifts, sample interval, pulse length
// and absorption stay).  With echo_range = k s, k = ra rb (EK rows: r0 = 0) and shift = d k:
//   lin(Sv)       = 10^(g raw/10) . (s - d)^(n/10) . C_sv(ping) . E_p(s)        E_p(s) = 10^(a2 k_p s / 10)
//   lin(Sv_noise) = C_n(ping) k_p^2 . s^2 . E_p(s)                              wherever k_p s >= 1
//   Sv_noise      = [noise(ping) + 20 log10 k_p] + 20 log10 s + (a2 k_p) s
// i.e. the per-COLUMN constants of the uniform kernel survive, joined by per-PING scalars (LDS, one ping per lane) and
// E_p along the lane's four columns (one exponential per ping and lane + the ratios q1, q128, q129).  The range bin of
// a column is fixed for the whole group when the slowest and the fastest ping of the group put it into the same bin
// (the range is monotone in rb); the few columns near a bin edge, below 1 m or at the R' <= 0 guard take the
// per-sample arithmetic of the general kernel (`plain` bit clear).  Anything else differing -> left to the general kernel.
// ------------------------------------------------------------------------------------------------
constexpr int kDriftPings = 128;
template <typename T>
struct PingDrift {  // what every sample of the ping needs ...
  T g, csv, cnk2, snp, a2k, q1, q128, q129;
};
template <typename T>
struct PingDriftRare {  // ... and what only the columns off the plain path read
  T cn, nb;
  double rb, shift;
};

template <typename T, bool WRITE_NOISE, bool WRITE_CORR, bool MINMAX>
__global__ __launch_bounds__(epa::kBlock, 4) void sv_denoise_mvbs_drift_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef, const double* __restrict__ alpha2,
    const double* __restrict__ noise, const int32_t* __restrict__ bin_start, T* __restrict__ noise_out,
    T* __restrict__ corr_out, T* __restrict__ mvbs_out, T* __restrict__ sum_out,
    uint32_t* __restrict__ cnt_out, Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);
  __shared__ PingDrift<T> pl[kDriftPings];
  __shared__ PingDriftRare<T> plr[kDriftPings];
  __shared__ int differs;
  __shared__ unsigned long long rb_lo_key, rb_hi_key;

  const int c = blockIdx.y, tb0 = (a.xcd_map ? epa::xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x) * a.uni_bins;
  const int nbn = min(a.uni_bins, a.n_tbins - tb0);
  const int S = a.S, n_rbins = a.n_rbins;
  const size_t cell0 = ((size_t)c * a.n_tbins + tb0) * n_rbins;
  // (uniform) only the groups the uniform kernel marked; a group of two bins is marked in both or in neither
  if (!left_to_general(mvbs_out + cell0)) return;
  const int pb = bin_start[tb0], pe = bin_start[tb0 + nbn], np = pe - pb;
  for (int i = threadIdx.x; i < nbn * n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  if (threadIdx.x == 0) {
    differs = np > kDriftPings ? 1 : 0;
    rb_lo_key = ~0ull;
    rb_hi_key = 0ull;
  }
  __syncthreads();  // (also publishes the math tables)
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const double* __restrict__ a2p = alpha2 + (size_t)c * a.P;
  const double* __restrict__ nzp = noise + (size_t)c * a.n_pblocks;
  const epa::CoefRow r = rowp0[np > 0 ? pb : 0];
  const double na2 = a2p[np > 0 ? pb : 0];
  if ((int)threadIdx.x < min(np, kDriftPings)) {
    const int p = pb + threadIdx.x;
    const epa::CoefRow ri = rowp0[p];
    const double a2i = a2p[p];
    const double ki = ri.ra * ri.rb;
    // shift = d k up to the roundings of either side (power_coef.hip builds both from the same numbers)
    const bool same = (ri.ra == r.ra) & (ri.r0 == 0.0) & (ri.d == r.d) & (a2i == na2) & (ri.alpha2 == a2i) &
                      (fabs(ri.shift - ri.d * ki) <= 8e-16 * fabs(ri.shift)) & (ri.rb > 0.0);
    if (!same) differs = 1;
    atomicMin(&rb_lo_key, ordered_key(ri.rb));
    atomicMax(&rb_hi_key, ordered_key(ri.rb));
    const double nbi = nzp[(p + a.ping_phase) / a.noise_ping_num];
    const double a2k = a2i * ki;
    const double cn = epa::lin_from_db(nbi, mt.exp2_tab);
    const double q1 = epa::lin_from_db(a2k, mt.exp2_tab), q128 = epa::lin_from_db(128.0 * a2k, mt.exp2_tab);
    pl[threadIdx.x] = PingDrift<T>{(T)ri.g, (T)epa::lin_from_db(ri.A0 - ri.alpha2 * ri.shift, mt.exp2_tab), (T)(cn * (ki * ki)),
                                   (T)(nbi + 20.0 * log10_pos(ki, mt.log_tab)), (T)a2k, (T)q1, (T)q128, (T)(q1 * q128)};
    plr[threadIdx.x] = PingDriftRare<T>{(T)cn, (T)nbi, ri.rb, ri.shift};
  }
  __syncthreads();
  if (differs) return;  // stays marked: the general kernel takes it
  auto unkey = [](unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
  };
  const double rb_lo = unkey(rb_lo_key), rb_hi = unkey(rb_hi_key);

  const T nspread = (T)a.nspread, snr = (T)a.snr;
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const epa::LogCoef lk = epa::make_log_coef();
  const float* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  T* __restrict__ sn_c = WRITE_NOISE ? noise_out + (size_t)c * a.P * S : nullptr;
  T* __restrict__ sc_c = WRITE_CORR ? corr_out + (size_t)c * a.P * S : nullptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // min / max of the two outputs (actual_range): one LDS slot per lane and quantity, updated by ds_min_f64 / ds_max_f64
  // (a NaN operand leaves the slot alone) -- four running doubles per lane are eight registers this kernel does not
  // have at 4 wavefronts / SIMD (held in registers they spilled into the ping loop: 15.0 -> 22.0 ms per 4 G samples)
  __shared__ T mm_slot[MINMAX ? 4 * epa::kBlock : 1];
  auto mm_min = [&](int k, T v) {
    __hip_atomic_fetch_min(&mm_slot[k * epa::kBlock + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto mm_max = [&](int k, T v) {
    __hip_atomic_fetch_max(&mm_slot[k * epa::kBlock + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (MINMAX) {
#pragma unroll
    for (int k = 0; k < 4; ++k) mm_slot[k * epa::kBlock + threadIdx.x] = (k & 1) ? -(T)__builtin_inf() : (T)__builtin_inf();
  }

  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int sA = chunk0 + wave * 256 + 2 * lane, sB = sA + 128;
    if (sA >= S) continue;
    const bool hasB = sB < S;
    // ---- the group's per-column constants
    T c2[VEC], sn20[VEC], acc_sum[VEC];
    int rbin[VEC];
    uint32_t acc_cnt[VEC];
    unsigned plain = 0;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const double sj = (double)((j < 2 ? sA : sB) + (j & 1));
      const double sra = sj * r.ra;
      const double x_lo = sra * rb_lo, x_hi = sra * rb_hi;  // the group's extreme ranges of this column (monotone in rb)
      const T v = (T)(sj - r.d), v2 = v * v;
      c2[j] = v > (T)0 ? (nspread == (T)20 ? v2 : v2 * v2) : (T)0;
      sn20[j] = (T)20 * log10_slow((T)sj, mt.log_tab);
      const int b_lo = epa::range_bin_index(x_lo, bin, inv_bin, n_rbins, false);
      const int b_hi = epa::range_bin_index(x_hi, bin, inv_bin, n_rbins, false);
      rbin[j] = b_lo;
      // plain: one range bin for every ping, echo_range >= 1 m (the transmission loss and the noise shape take their
      // R >= 1 form) and s - d >= 1 (R' > 0 whatever the rounding of R - shift)
      if (b_lo == b_hi && x_lo >= 1.0 && (sj - r.d) >= 1.0) plain |= 1u << j;
      acc_sum[j] = (T)0;
      acc_cnt[j] = 0u;
    }
    auto flush_col = [&](int g, int j) {
      if (rbin[j] >= 0 && acc_cnt[j] > 0u) {
        lds_add(lsum + g * n_rbins + rbin[j], acc_sum[j]);
        atomicAdd(lcnt + g * n_rbins + rbin[j], acc_cnt[j]);
      }
      acc_sum[j] = (T)0;
      acc_cnt[j] = 0u;
    };
    const bool allplain = __all((plain | (hasB ? 0u : 0xcu)) == 0xfu) != 0;  // no column of this wavefront is redone below
    float2 nA = make_float2(0.f, 0.f), nB = nA;
    if (np > 0) {
      nA = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sA);
      if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sB);
    }
    int g = 0, edge = bin_start[tb0 + 1];  // first ping of the next time bin
    // ---- every column as a plain one (what the off-plain columns get here is overwritten below, never accumulated)
    for (int p = pb; p < pe; ++p) {
      while (p >= edge) {  // (uniform) ping p opens a later time bin; empty bins are stepped over
#pragma unroll
        for (int j = 0; j < VEC; ++j) flush_col(g, j);
        ++g;
        edge = bin_start[tb0 + g + 1];
      }
      const size_t row_off = (size_t)p * S;
      const float2 inA = nA, inB = nB;
      if (p + 1 < pe) {
        nA = *reinterpret_cast<const float2*>(raw_c + row_off + S + sA);
        if (hasB) nB = *reinterpret_cast<const float2*>(raw_c + row_off + S + sB);
      }
      const PingDrift<T> q = pl[p - pb];
      const float in[VEC] = {inA.x, inA.y, inB.x, inB.y};
      T sn[VEC], sc[VEC];
      const T e0 = epa::lin_from_db_lean(q.a2k * (T)sA, mt.exp2_tab);
      const T ecol[VEC] = {e0, e0 * q.q1, e0 * q.q128, e0 * q.q129};
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (j >= 2 && !hasB) break;
        const bool xok = in[j] == in[j];
        const T sj = (T)((j < 2 ? sA : sB) + (j & 1));
        // a NaN sample needs no masking: it makes the exponential, hence lin, NaN by itself
        const T e = epa::lin_from_db_lean(q.g * (T)in[j], mt.exp2_tab);
        const T lin = ecol[j] * fma(-q.cnk2, sj * sj, (e * c2[j]) * q.csv);
        sn[j] = xok ? fma(q.a2k, sj, q.snp + sn20[j]) : epa::M<T>::nan();  // echo_range is NaN where the input is
        const T corr = lin > (T)0 ? (T)10 * epa::fast_log10_lean(lin, mt.log_tab, lk) : epa::M<T>::nan();
        const bool keep = corr - sn[j] > snr;
        sc[j] = keep ? corr : epa::M<T>::nan();
        const bool pj = ((plain >> j) & 1u) != 0u;
        if (MINMAX) {
          // (allplain: wavefront-uniform, the usual case; an off-plain column is redone -- and counted -- below)
          const T vn = allplain || pj ? sn[j] : epa::M<T>::nan(), vc = allplain || pj ? sc[j] : epa::M<T>::nan();
          mm_min(0, vn);
          mm_max(1, vn);
          mm_min(2, vc);
          mm_max(3, vc);
        }
        const bool take = pj & (rbin[j] >= 0) & keep;  // keep implies a finite positive lin (and a valid input)
        acc_sum[j] += take ? lin : (T)0;
        acc_cnt[j] += take ? 1u : 0u;
      }
      if (WRITE_NOISE) {
        epa::store_nt2(sn_c + row_off + sA, sn[0], sn[1]);
        if (hasB) epa::store_nt2(sn_c + row_off + sB, sn[2], sn[3]);
      }
      if (WRITE_CORR) {
        epa::store_nt2(sc_c + row_off + sA, sc[0], sc[1]);
        if (hasB) epa::store_nt2(sc_c + row_off + sB, sc[2], sc[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) flush_col(g, j);
    // ---- the columns off the plain path (a few per wavefront): one column at a time with the wavefront's lanes
    // spread over the PINGS of the group -- the general kernel's per-sample arithmetic, the range bin found per ping,
    // the sums added straight to the group's rows
    const unsigned have = hasB ? 0xfu : 0x3u;
    const unsigned off_plain = (~plain) & have;
    if (__ballot(off_plain != 0u) != 0ull) {
      // the lanes still here are 0 .. nact - 1 (sA grows with the lane): in the row's last, partial wavefront only
      // they exist to share the pings, so the stride is their number, not 64
      const int nact = (int)__popcll(__ballot(true));
#pragma unroll 1
      for (int j = 0; j < VEC; ++j) {
        unsigned long long todo = __ballot(((off_plain >> j) & 1u) != 0u);
        while (todo != 0ull) {  // (wave-uniform)
          const int src = __ffsll((long long)todo) - 1;
          todo &= todo - 1ull;
          const int sx = chunk0 + wave * 256 + 2 * src + (j < 2 ? 0 : 128) + (j & 1);
          const double sj = (double)sx;
          const T v = (T)(sj - r.d), v2 = v * v;
          const T c2j = v > (T)0 ? (nspread == (T)20 ? v2 : v2 * v2) : (T)0;
          const T sn20j = (T)20 * log10_slow((T)sj, mt.log_tab);
          for (int p = pb + lane; p < pe; p += nact) {
            int g2 = 0;
            for (int t = 1; t < nbn; ++t) g2 += p >= bin_start[tb0 + t] ? 1 : 0;
            const size_t o = (size_t)p * S + sx;
            const float inv = raw_c[o];
            const bool xok = inv == inv;
            const PingDrift<T> q = pl[p - pb];
            const PingDriftRare<T> qr = plr[p - pb];
            const double x = (sj * r.ra) * qr.rb;
            const double rtd = x - qr.shift;  // R' <= 0 -> NaN (calibrate_ek.py:107)
            T cc = c2j;
            if ((rtd > 0.0) & !(cc > (T)0)) {  // rounding residue of R - shift (rare): the range itself
              const T w = (T)(rtd / (r.ra * qr.rb)), w2 = w * w;
              cc = nspread == (T)20 ? w2 : w2 * w2;
            }
            cc = rtd > 0.0 ? cc : epa::M<T>::nan();
            const T e = epa::lin_from_db_lean(q.g * (T)inv, mt.exp2_tab);
            const T E = epa::lin_from_db_lean(q.a2k * (T)sj, mt.exp2_tab);
            const T mx = fmax((T)x, (T)1);
            const T lin = E * fma(-qr.cn, mx * mx, (e * cc) * q.csv);
            const T tl = x >= 1.0 ? (q.snp - qr.nb) + sn20j : (T)0;  // 20 log10(R >= 1 ? R : 1)
            const T snv = xok ? (qr.nb + tl) + q.a2k * (T)sj : epa::M<T>::nan();
            const T corr = lin > (T)0 ? (T)10 * epa::fast_log10_lean(lin, mt.log_tab, lk) : epa::M<T>::nan();
            const bool keep = corr - snv > snr;
            const T scv = keep ? corr : epa::M<T>::nan();
            if (WRITE_NOISE) sn_c[o] = snv;
            if (WRITE_CORR) sc_c[o] = scv;
            if (MINMAX) {
              mm_min(0, snv);
              mm_max(1, snv);
              mm_min(2, scv);
              mm_max(3, scv);
            }
            const int rb = xok ? epa::range_bin_index(x, bin, inv_bin, n_rbins, false) : -1;
            if ((rb >= 0) & keep) {
              lds_add(lsum + g2 * n_rbins + rb, lin);
              atomicAdd(lcnt + g2 * n_rbins + rb, 1u);
            }
          }
        }
      }
    }
  }
  if (MINMAX) {
    double mm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mm[k] = (double)mm_slot[k * epa::kBlock + threadIdx.x];  // (the lane's own slots: no barrier)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mm[0] = fmin(mm[0], __shfl_down(mm[0], o, 64));
      mm[1] = fmax(mm[1], __shfl_down(mm[1], o, 64));
      mm[2] = fmin(mm[2], __shfl_down(mm[2], o, 64));
      mm[3] = fmax(mm[3], __shfl_down(mm[3], o, 64));
    }
    wg_minmax_keys(mm, a.mm_keys, lane);
  }
  __syncthreads();
  T* out = mvbs_out + cell0;
  T* gsum = sum_out ? sum_out + cell0 : nullptr;
  uint32_t* gcnt = cnt_out ? cnt_out + cell0 : nullptr;
  for (int i = threadIdx.x; i < nbn * n_rbins; i += epa::kBlock) {
    const uint32_t n = lcnt[i];
    const T s = lsum[i];
    out[i] = n > 0u ? (T)10 * epa::M<T>::log10(s / (T)n) : (T)a.fill_value;  // (overwrites the mark of the group)
    if (gsum) gsum[i] = s;
    if (gcnt) gcnt[i] = n;
  }
}

template <typename K>
int set_lds(K kern, size_t lds) {
  if (lds > 64 * 1024)
    EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return EPA_OK;
}

template <typename T>
int launch_pass1(Args& a, const float* raw, const double* coef, const double* alpha2, void* sv_out,
                 double* noise_out, int C, hipStream_t st) {
  const size_t sum_bytes = ((size_t)kP1Blocks * a.n_rblocks * sizeof(T) + 15) & ~(size_t)15;
  a.cnt_off = (unsigned)sum_bytes;
  a.tab_off = (unsigned)((sum_bytes + (size_t)kP1Blocks * a.n_rblocks * 4 + 15) & ~(size_t)15);
  const size_t lds = a.tab_off + epa::kMathTabBytes;
  const dim3 grid((unsigned)((a.n_pblocks + kP1Blocks - 1) / kP1Blocks), (unsigned)C);
#define EPA_P1(W, R)                                                                                    \
  do {                                                                                                  \
    auto kern = sv_noise_fast_kernel<T, W, R>;                                                          \
    if (int rc = set_lds(kern, lds)) return rc;                                                         \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds, st, raw,                                     \
                       reinterpret_cast<const epa::CoefRow*>(coef), alpha2, (T*)sv_out, noise_out, a);  \
  } while (0)
  if (a.rmax_key) { if (sv_out) EPA_P1(true, true); else EPA_P1(false, true); }
  else { if (sv_out) EPA_P1(true, false); else EPA_P1(false, false); }
#undef EPA_P1
  return epa::check_launch("sv_noise_fast_kernel");
}

template <typename T>
int launch_pass2(Args& a, const float* raw, const double* coef, const double* alpha2, const double* noise,
                 const int32_t* bin_start, void* noise_out, void* corr_out, void* mvbs_out, void* sum_out,
                 uint32_t* cnt_out, int C, size_t lds_acc_bytes, hipStream_t st) {
  a.tab_off = (unsigned)((lds_acc_bytes + 15) & ~(size_t)15);
  const size_t lds = a.tab_off + epa::kMathTabBytes;
  const dim3 grid((unsigned)a.n_tbins + 1u, (unsigned)C);
  a.flagged_only = 0;
  static const bool uniform_off = [] {  // development knob: EPA_CHAIN_UNIFORM=0 leaves every group to the general kernel
    const char* e = getenv("EPA_CHAIN_UNIFORM");
    return e && e[0] == '0';
  }();
  if (a.n_tbins > 0 && a.n_rbins > 0 && !uniform_off) {
    // uniform ping groups first; the general kernel then takes the groups that one left, and the pings outside every bin
    // two time bins per workgroup (the per-column constants paid once for both) when both rows of accumulators fit
    Args au = a;
    au.uni_bins = (size_t)2 * a.n_rbins * (sizeof(T) + 4) + epa::kMathTabBytes <= 48 * 1024 ? 2 : 1;
    const size_t usum = ((size_t)au.uni_bins * a.n_rbins * sizeof(T) + 15) & ~(size_t)15;
    au.cnt_off = (unsigned)usum;
    au.tab_off = (unsigned)((usum + (size_t)au.uni_bins * a.n_rbins * 4 + 15) & ~(size_t)15);
    const size_t ulds = au.tab_off + epa::kMathTabBytes;
    const dim3 ugrid((unsigned)((a.n_tbins + au.uni_bins - 1) / au.uni_bins), (unsigned)C);
#define EPA_U2(N, K, M)                                                                                  \
  do {                                                                                                   \
    auto kern = sv_denoise_mvbs_uniform_kernel<T, N, K, M>;                                                 \
    if (int rc = set_lds(kern, ulds)) return rc;                                                         \
    hipLaunchKernelGGL(kern, ugrid, dim3(epa::kBlock), ulds, st, raw,                                    \
                       reinterpret_cast<const epa::CoefRow*>(coef), alpha2, noise, bin_start,            \
                       (T*)noise_out, (T*)corr_out, (T*)mvbs_out, (T*)sum_out, cnt_out, au);                \
  } while (0)
#define EPA_U2M(N, K)                                                                                    \
  do {                                                                                                   \
    if (a.mm_keys) EPA_U2(N, K, true); else EPA_U2(N, K, false);                                         \
  } while (0)
    if (noise_out) { if (corr_out) EPA_U2M(true, true); else EPA_U2M(true, false); }
    else { if (corr_out) EPA_U2M(false, true); else EPA_U2M(false, false); }
#undef EPA_U2M
#undef EPA_U2
    if (int rc = epa::check_launch("sv_denoise_mvbs_uniform_kernel")) return rc;
    static const bool drift_off = [] {  // development knob: EPA_CHAIN_DRIFT=0 skips the sound-speed-drift kernel
      const char* e = getenv("EPA_CHAIN_DRIFT");
      return e && e[0] == '0';
    }();
    if (!drift_off) {
#define EPA_D2(N, K, M)                                                                                  \
  do {                                                                                                   \
    auto kern = sv_denoise_mvbs_drift_kernel<T, N, K, M>;                                                 \
    if (int rc = set_lds(kern, ulds)) return rc;                                                         \
    hipLaunchKernelGGL(kern, ugrid, dim3(epa::kBlock), ulds, st, raw,                                    \
                       reinterpret_cast<const epa::CoefRow*>(coef), alpha2, noise, bin_start,            \
                       (T*)noise_out, (T*)corr_out, (T*)mvbs_out, (T*)sum_out, cnt_out, au);                \
  } while (0)
#define EPA_D2M(N, K)                                                                                    \
  do {                                                                                                   \
    if (a.mm_keys) EPA_D2(N, K, true); else EPA_D2(N, K, false);                                         \
  } while (0)
      if (noise_out) { if (corr_out) EPA_D2M(true, true); else EPA_D2M(true, false); }
      else { if (corr_out) EPA_D2M(false, true); else EPA_D2M(false, false); }
#undef EPA_D2M
#undef EPA_D2
      if (int rc = epa::check_launch("sv_denoise_mvbs_drift_kernel")) return rc;
    }
    a.flagged_only = 1;
  }
#define EPA_P2(N, K)                                                                                     \
  do {                                                                                                   \
    if (a.mm_keys) EPA_P2M(N, K, true); else EPA_P2M(N, K, false);                                       \
  } while (0)
#define EPA_P2M(N, K, M)                                                                                 \
  do {                                                                                                   \
    auto kern = sv_denoise_mvbs_fast_kernel<T, N, K, M>;                                                 \
    if (int rc = set_lds(kern, lds)) return rc;                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds, st, raw,                                      \
                       reinterpret_cast<const epa::CoefRow*>(coef), alpha2, noise, bin_start,            \
                       (T*)noise_out, (T*)corr_out, (T*)mvbs_out, (T*)sum_out, cnt_out, a);              \
  } while (0)
  if (noise_out) { if (corr_out) EPA_P2(true, true); else EPA_P2(true, false); }
  else { if (corr_out) EPA_P2(false, true); else EPA_P2(false, false); }
#undef EPA_P2
#undef EPA_P2M
  return epa::check_launch("sv_denoise_mvbs_fast_kernel");
}

}  // namespace epa_chain

// Called by epa_sv_noise_fused (block_reduce.hip) when the fast path applies.
int epa_chain_fast_pass1(const float* raw, const double* coef, const double* alpha2, int C, int P, int S,
                         double nspread, int ping_num, int rsn, int ping_phase, double noise_max, void* sv_out,
                         double* noise_out, double* edge_sum_out, uint32_t* edge_cnt_out,
                         unsigned long long* rmax_key, unsigned long long* rstat, int dtype, hipStream_t st) {
  epa_chain::Args a{};
  a.P = P; a.S = S; a.nspread = nspread;
  a.ping_num = ping_num; a.rsn = rsn; a.ping_phase = ping_phase;
  a.edge_sum = edge_sum_out; a.edge_cnt = edge_cnt_out;
  a.n_pblocks = (P + ping_phase + ping_num - 1) / ping_num; a.n_rblocks = (S + rsn - 1) / rsn;
  a.noise_max = noise_max; a.rmax_key = rmax_key; a.rstat = rmax_key ? rstat : nullptr;
  a.xcd_map = epa::xcd_map_enabled() ? 1 : 0;
  if (dtype == EPA_F64) return epa_chain::launch_pass1<double>(a, raw, coef, alpha2, sv_out, noise_out, C, st);
  return epa_chain::launch_pass1<float>(a, raw, coef, alpha2, sv_out, noise_out, C, st);
}

// Called by epa_sv_denoise_mvbs (block_reduce.hip) when the fast path applies.
int epa_chain_fast_pass2(const float* raw, const double* coef, const double* alpha2, const double* noise, int C,
                         int P, int S, double nspread, int ping_num, int ping_phase, double snr, const int32_t* bin_start,
                         int n_tbins, double range_bin, int n_rbins, double fill_value, void* noise_out,
                         void* corr_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                         size_t lds_acc_bytes, unsigned cnt_off, unsigned long long* mm_keys, hipStream_t st) {
  epa_chain::Args a{};
  a.P = P; a.S = S; a.nspread = nspread;
  a.noise_ping_num = ping_num; a.ping_phase = ping_phase; a.n_pblocks = (P + ping_phase + ping_num - 1) / ping_num;
  a.snr = snr;
  a.n_tbins = n_tbins; a.n_rbins = n_rbins; a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.fill_value = fill_value; a.cnt_off = cnt_off; a.mm_keys = mm_keys;
  a.xcd_map = epa::xcd_map_enabled() ? 1 : 0;
  if (dtype == EPA_F64)
    return epa_chain::launch_pass2<double>(a, raw, coef, alpha2, noise, bin_start, noise_out, corr_out, mvbs_out,
                                           sum_out, cnt_out, C, lds_acc_bytes, st);
  return epa_chain::launch_pass2<float>(a, raw, coef, alpha2, noise, bin_start, noise_out, corr_out, mvbs_out,
                                        sum_out, cnt_out, C, lds_acc_bytes, st);
}
