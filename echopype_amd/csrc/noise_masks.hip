// Ryan et al. (2015) noise masks and apply_mask (SURVEY 8f "next" row 2).
//
// Replaces, in /root/reference/echopype:
//   clean/utils.py:173-304  depth-bin down-sampling (linear nanmean) + forward-fill up-sampling
//   clean/utils.py:307-323  two-sided ping comparison of the impulse-noise mask
//   clean/utils.py:109-170  (2n+1) x (2m+1) pooled Sv (dask_image generic_filter, reflect boundary)
//   clean/utils.py:326-372  ping-median vs block-median attenuated-signal mask
//   clean/api.py:166        Sv - pooled > threshold
//   mask/api.py:402-432     logical AND of masks, where(mask, Sv, fill_value)
// Masks are uint8 [C*P*S] (1 = True) in the (channel, ping_time, range_sample) layout of Sv.
//
// Linear-domain sums are accumulated in double whatever the storage type; comparisons against the
// thresholds are done in the storage type, as numpy does for the reference's arrays.
#include "fast_math.h"

namespace {

using epa::kBlock;

// scipy.ndimage / dask_image mode="reflect":  d c b a | a b c d | d c b a   (period 2n)
__device__ __forceinline__ int reflect_index(int i, int n) {
  const int period = 2 * n;
  i %= period;
  if (i < 0) i += period;
  return i < n ? i : period - 1 - i;
}

// ------------------------------------------------------------------------------------------------
// workgroup reductions (256 threads = 4 wavefronts); every thread gets the result
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned block_sum(unsigned v, unsigned* sh4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

__device__ __forceinline__ unsigned long long block_min(unsigned long long v, unsigned long long* sh4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_down(v, o, 64);
    v = w < v ? w : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = sh4[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) r = sh4[i] < r ? sh4[i] : r;
  return r;
}

// lexicographic (value, index) minimum -- np.argmin's "first occurrence of the minimum"
__device__ __forceinline__ int block_argmin(double v, int idx, double* shv, int* shi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double w = __shfl_down(v, o, 64);
    const int j = __shfl_down(idx, o, 64);
    if (w < v || (w == v && j < idx)) { v = w; idx = j; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = v; shi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  double bv = shv[0];
  int bi = shi[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (shv[i] < bv || (shv[i] == bv && shi[i] < bi)) { bv = shv[i]; bi = shi[i]; }
  return bi;
}

// ------------------------------------------------------------------------------------------------
// depth-bin smoothing: up[c,p,s] = 10 log10( nanmean_{s' in bin(s)} 10^(Sv[c,p,s']/10) )
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double bin_edge(double r0, double delta, int j) {
  return __dadd_rn(r0, __dmul_rn((double)j, delta));  // np.arange: start + j*delta, two roundings
}

// FILL == false: flox membership in [e_j, e_j+1), -1 outside or NaN.
// FILL == true : np.digitize(d, left edges) - 1 clipped to [0, nb-1] (NaN sorts last -> nb-1).
template <bool FILL>
__device__ __forceinline__ int value_bin(double d, double r0, double delta, double inv, int nb) {
  if (!(d == d)) return FILL ? nb - 1 : -1;
  const double t = (d - r0) * inv;
  if (!(t > -2.0)) return FILL ? 0 : -1;
  if (t > (double)nb + 2.0) return FILL ? nb - 1 : -1;
  int j = (int)floor(t);
  if (d < bin_edge(r0, delta, j)) --j;
  else if (d >= bin_edge(r0, delta, j + 1)) ++j;
  if (FILL) return j < 0 ? 0 : (j >= nb ? nb - 1 : j);
  return (j >= 0 && j < nb) ? j : -1;
}

template <typename T, bool BY_VALUE>
__global__ __launch_bounds__(kBlock) void range_bin_smooth_kernel(
    const T* __restrict__ sv, const T* __restrict__ range, long long rows, int S, int nper, double r0,
    double delta, int nbins, int seg_bins, T* __restrict__ up) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  double* ssum = reinterpret_cast<double*>(smem + epa::kMathTabBytes);
  unsigned* scnt = reinterpret_cast<unsigned*>(ssum + seg_bins);
  const double inv = 1.0 / delta;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* svr = sv + (size_t)row * S;
    const T* rr = BY_VALUE ? range + (size_t)row * S : nullptr;
    // seg_bins == nbins unless the ping has more bins than the LDS accumulators hold: then the bins are taken
    // seg_bins at a time, the ping (a few KB, in L2) swept once per segment
    for (int b0 = 0; b0 < nbins; b0 += seg_bins) {
    const int nseg = min(seg_bins, nbins - b0);
    __syncthreads();
    for (int b = threadIdx.x; b < nseg; b += kBlock) {
      ssum[b] = 0.0;
      scnt[b] = 0u;
    }
    __syncthreads();
    // 4 consecutive samples per lane; a run of equal bins is merged before it touches LDS
    for (int base = 4 * threadIdx.x; base < S; base += 4 * kBlock) {
      int rb = -1;
      double rs = 0.0;
      unsigned rn = 0u;
      // the lane's four samples in 16-byte accesses where the row allows (one request instead of four)
      T vv[4], xx[4];
      if (base + 4 <= S) {
        typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
        const pair_t v01 = *reinterpret_cast<const pair_t*>(svr + base), v23 = *reinterpret_cast<const pair_t*>(svr + base + 2);
        vv[0] = v01.x; vv[1] = v01.y; vv[2] = v23.x; vv[3] = v23.y;
        if (BY_VALUE) {
          const pair_t x01 = *reinterpret_cast<const pair_t*>(rr + base), x23 = *reinterpret_cast<const pair_t*>(rr + base + 2);
          xx[0] = x01.x; xx[1] = x01.y; xx[2] = x23.x; xx[3] = x23.y;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          vv[j] = (base + j < S) ? svr[base + j] : epa::M<T>::nan();
          if (BY_VALUE) xx[j] = (base + j < S) ? rr[base + j] : epa::M<T>::nan();
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = base + j;
        if (s >= S) break;
        const T v = vv[j];
        const int b = (BY_VALUE ? value_bin<false>((double)xx[j], r0, delta, inv, nbins) : s / nper) - b0;
        if (!(v == v) || b < 0 || b >= nseg) continue;
        if (b != rb) {
          if (rn) {
            unsafeAtomicAdd(&ssum[rb], rs);
            atomicAdd(&scnt[rb], rn);
          }
          rb = b;
          rs = 0.0;
          rn = 0u;
        }
        rs += (double)epa::lin_from_db(v, mt.exp2_tab);
        ++rn;
      }
      if (rn) {
        unsafeAtomicAdd(&ssum[rb], rs);
        atomicAdd(&scnt[rb], rn);
      }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nseg; b += kBlock) {  // the dB value of a bin once, not once per sample
      const unsigned n = scnt[b];
      ssum[b] = n ? 10.0 * epa::fast_log10(ssum[b] / (double)n, mt.log_tab) : __builtin_nan("");
    }
    __syncthreads();
    T* ur = up + (size_t)row * S;
    for (int s = threadIdx.x; s < S; s += kBlock) {
      const int b = (BY_VALUE ? value_bin<true>((double)rr[s], r0, delta, inv, nbins) : s / nper) - b0;
      if (b >= 0 && b < nseg) ur[s] = (T)ssum[b];
    }
    }
  }
}

// mask = (up[p] - up[p+n] > thr) & (up[p] - up[p-n] > thr), NaN differences (and the missing side at
// the first / last n pings) count as +inf.
template <typename T>
__global__ __launch_bounds__(kBlock) void impulse_compare_kernel(const T* __restrict__ up, int P, int S,
                                                                 long long rows, int n, T thr,
                                                                 uint8_t* __restrict__ mask) {
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int p = (int)(row % P);
    const T* a = up + (size_t)row * S;
    const T* f = (long long)p + n < P ? a + (size_t)n * S : nullptr;
    const T* b = p - n >= 0 ? a - (size_t)n * S : nullptr;
    uint8_t* m = mask + (size_t)row * S;
    auto decide = [&](T x, T xf, T xb) -> uint8_t {
      T df = f ? x - xf : epa::M<T>::nan();
      T db = b ? x - xb : epa::M<T>::nan();
      if (!(df == df)) df = (T)__builtin_inf();
      if (!(db == db)) db = (T)__builtin_inf();
      return (df > thr && db > thr) ? 1 : 0;
    };
    // two samples per lane: 16-byte loads of the three rows (half the requests of one sample per lane)
    typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    const int S2 = S & ~1;
    for (int s = 2 * threadIdx.x; s < S2; s += 2 * kBlock) {
      const pair_t x = *reinterpret_cast<const pair_t*>(a + s);
      pair_t xf = x, xb = x;
      if (f) xf = *reinterpret_cast<const pair_t*>(f + s);
      if (b) xb = *reinterpret_cast<const pair_t*>(b + s);
      m[s] = decide(x.x, xf.x, xb.x);
      m[s + 1] = decide(x.y, xf.y, xb.y);
    }
    if (S2 < S && threadIdx.x == 0) m[S2] = decide(a[S2], f ? f[S2] : (T)0, b ? b[S2] : (T)0);
  }
}

// ------------------------------------------------------------------------------------------------
// pooled Sv, nanmean: separable box sums with reflect boundaries
//   pass 1 (range):  vsum/vcnt[c,p,s] = sum / count of the non-NaN linear Sv over s-m..s+m
//   pass 2 (ping) :  pooled = 10 log10( sum_{p-n..p+n} vsum / sum vcnt ),  mask = Sv - pooled > thr
// ------------------------------------------------------------------------------------------------
constexpr int kRangeTile = 2048;
constexpr int kGroup = 8;  // window sums = two ragged edges + whole groups of 8 (no subtraction anywhere)

template <typename T>
__global__ __launch_bounds__(kBlock) void box_range_kernel(const T* __restrict__ sv, long long rows,
                                                           int S, int s0, int m,
                                                           double* __restrict__ vsum,
                                                           int* __restrict__ vcnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  const int L = S - s0;
  const int t0 = blockIdx.y * kRangeTile;  // relative to s0
  const int nout = min(kRangeTile, L - t0);
  const int nin = nout + 2 * m;
  const int ngrp = nin / kGroup;
  double* lin = reinterpret_cast<double*>(smem + epa::kMathTabBytes);  // [nin], NaN kept
  double* gsum = lin + ((nin + 1) & ~1);                                 // [ngrp]
  int* gcnt = reinterpret_cast<int*>(gsum + ngrp);                       // [ngrp]
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
    const T* svr = sv + (size_t)row * S + s0;
    for (int i = threadIdx.x; i < nin; i += kBlock) {
      const T v = svr[reflect_index(t0 - m + i, L)];
      lin[i] = (v == v) ? (double)epa::lin_from_db(v, mt.exp2_tab) : __builtin_nan("");
    }
    __syncthreads();
    for (int g = threadIdx.x; g < ngrp; g += kBlock) {
      double sum = 0.0;
      int cnt = 0;
#pragma unroll
      for (int k = 0; k < kGroup; ++k) {
        const double x = lin[g * kGroup + k];
        if (x == x) {
          sum += x;
          ++cnt;
        }
      }
      gsum[g] = sum;
      gcnt[g] = cnt;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nout; o += kBlock) {
      const int a = o, b = o + 2 * m + 1;  // window [a, b)
      int a8 = (a + kGroup - 1) & ~(kGroup - 1), b8 = b & ~(kGroup - 1);
      if (a8 >= b8) a8 = b8 = b;  // no whole group inside: one plain run
      double sum = 0.0;
      int cnt = 0;
      for (int k = a; k < a8; ++k) {
        const double x = lin[k];
        if (x == x) {
          sum += x;
          ++cnt;
        }
      }
      for (int g = a8 / kGroup; g < b8 / kGroup; ++g) {
        sum += gsum[g];
        cnt += gcnt[g];
      }
      for (int k = b8; k < b; ++k) {
        const double x = lin[k];
        if (x == x) {
          sum += x;
          ++cnt;
        }
      }
      const size_t at = (size_t)row * S + s0 + t0 + o;
      vsum[at] = sum;
      vcnt[at] = cnt;
    }
  }
}

// The same range pass in O(1) additions per sample (van Herk / Gil-Werman without any subtraction): the tile is
// cut into blocks of w = 2m+1 samples; with pre[i] / suf[i] = running sums from the start / to the end of i's
// block, the window [o, o+w) is  suf[o] + pre[o+w-1]  (or pre alone when o starts a block).  The running sums
// of a block are built by kBlock / nblk lanes: each sums its run, the runs before / after it give its offsets,
// a second sweep writes pre and suf.  (PMC on the grouped kernel above: ~295 VALU instructions per sample,
// 97 % VALU-busy; this one is bound by its 20 B/sample of HBM traffic.)
#ifndef EPA_SCAN_TILE
#define EPA_SCAN_TILE 1024
#endif
constexpr int kScanTile = EPA_SCAN_TILE;
constexpr int kScanMinW = 8, kScanMaxW = 512;

template <typename T>
__global__ __launch_bounds__(kBlock) void box_range_scan_kernel(const T* __restrict__ sv, long long rows,
                                                                int S, int s0, int m,
                                                                double* __restrict__ vsum,
                                                                int* __restrict__ vcnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  const int L = S - s0, w = 2 * m + 1;
  const int t0 = blockIdx.y * kScanTile;  // relative to s0
  const int nout = min(kScanTile, L - t0);
  const int nin = nout + 2 * m;
  const int ninp = (kScanTile + 2 * m + 2) & ~1;  // array pitch, the same for every tile of the launch
  double* lin = reinterpret_cast<double*>(smem + epa::kMathTabBytes);  // [nin] NaN kept; becomes suf
  double* pre_s = lin + ninp;
  double* tot_s = pre_s + ninp;                                    // [nvirt <= 2 * kBlock]
  unsigned short* pre_c = reinterpret_cast<unsigned short*>(tot_s + 2 * kBlock);
  unsigned short* suf_c = pre_c + ninp;
  unsigned short* tot_c = suf_c + ninp;
  const int nblk = (nin + w - 1) / w;
  // at most 12 lanes share a block: more lanes shorten the runs but lengthen the offset sums (measured 12 best)
#ifndef EPA_SCAN_TPB
#define EPA_SCAN_TPB 12
#endif
  const int tpb = max(1, min(EPA_SCAN_TPB, kBlock / nblk));  // lanes per block
  const int r = (w + tpb - 1) / tpb;         // samples per lane
  const int nvirt = nblk * tpb;              // <= 2 * kBlock for w >= kScanMinW
  // the next row's samples are requested before this row is scanned (the scan phases are latency-bound)
  constexpr int kPf = (kScanTile + kScanMaxW + kBlock - 1) / kBlock;
  T pf[kPf];
  int src[kPf];
#pragma unroll
  for (int j = 0; j < kPf; ++j) {
    const int i = threadIdx.x + j * kBlock;
    src[j] = i < nin ? reflect_index(t0 - m + i, L) : -1;
    pf[j] = (src[j] >= 0 && blockIdx.x < rows) ? sv[(size_t)blockIdx.x * S + s0 + src[j]] : (T)0;
  }
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPf; ++j) {
      const int i = threadIdx.x + j * kBlock;
      const T v = pf[j];
      if (src[j] >= 0) lin[i] = (v == v) ? (double)epa::lin_from_db(v, mt.exp2_tab) : __builtin_nan("");
    }
    const long long nrow = row + gridDim.x;
    if (nrow < rows) {
      // every request of the lane in ONE basic block, by 32-bit offsets from the row's (uniform) address: behind a
      // per-element bound test the compiler computed each address into the register pair of the value it replaces and
      // waited for all earlier requests before doing so -- six round trips per row.  (A lane without an element reads
      // the row's first one and ignores it.)
      const char* svn = reinterpret_cast<const char*>(sv + (size_t)nrow * S + s0);
#pragma unroll
      for (int j = 0; j < kPf; ++j)
        pf[j] = *reinterpret_cast<const T*>(svn + (unsigned)max(src[j], 0) * (unsigned)sizeof(T));
    }
    __syncthreads();
    for (int v = threadIdx.x; v < nvirt; v += kBlock) {  // run totals
      const int k = v / tpb, t = v - k * tpb;
      const int lo = k * w + t * r, hi = min(min(lo + r, (k + 1) * w), nin);
      double sum = 0.0;
      int cnt = 0;
      for (int i = lo; i < hi; ++i) {
        const double x = lin[i];
        const bool ok = x == x;
        sum += ok ? x : 0.0;
        cnt += ok ? 1 : 0;
      }
      tot_s[v] = sum;
      tot_c[v] = (unsigned short)cnt;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < nvirt; v += kBlock) {  // running sums of the block, both directions
      const int k = v / tpb, t = v - k * tpb;
      const int lo = k * w + t * r, hi = min(min(lo + r, (k + 1) * w), nin);
      double fs = 0.0, bs = 0.0;
      int fc = 0, bc = 0;
      for (int u = 0; u < t; ++u) {
        fs += tot_s[k * tpb + u];
        fc += tot_c[k * tpb + u];
      }
      for (int u = tpb - 1; u > t; --u) {
        bs += tot_s[k * tpb + u];
        bc += tot_c[k * tpb + u];
      }
      for (int i = lo; i < hi; ++i) {
        const double x = lin[i];
        const bool ok = x == x;
        fs += ok ? x : 0.0;
        fc += ok ? 1 : 0;
        pre_s[i] = fs;
        pre_c[i] = (unsigned short)fc;
      }
      for (int i = hi - 1; i >= lo; --i) {
        const double x = lin[i];
        const bool ok = x == x;
        bs += ok ? x : 0.0;
        bc += ok ? 1 : 0;
        lin[i] = bs;  // suf
        suf_c[i] = (unsigned short)bc;
      }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < nout; o += kBlock) {
      const int k = o / w, e = o + w - 1;
      const bool whole = o == k * w;
      const double sum = whole ? pre_s[e] : lin[o] + pre_s[e];
      const int cnt = whole ? (int)pre_c[e] : (int)suf_c[o] + (int)pre_c[e];
      const size_t at = (size_t)row * S + s0 + t0 + o;
      vsum[at] = sum;
      vcnt[at] = cnt;
    }
  }
}

// The ping pass as a sliding window: one lane per range column walks a segment of kSlideSeg pings; the window
// sum is carried from ping to ping -- row p+n enters, row p-n-1 leaves -- in double-double arithmetic (Knuth
// two-sum, the rounding error of every addition is kept in `lo`), so the subtraction loses nothing: the result
// is the correctly rounded window sum to ~1e-30 relative, however large the values that have passed through.
// Every segment starts from a freshly summed window.  Reads 2.2 rows per output (the leaving row is 2n+1
// pings old: L2), ~60 VALU instructions per sample instead of ~290 (PMC on box_ping_kernel: 97 % VALU-busy,
// 1.7x the HBM bytes because of its ping halo).  +inf contributions are counted, not added (inf - inf).
#ifndef EPA_SLIDE_PAD
#define EPA_SLIDE_PAD 36864  // caps the kernel at 4 workgroups per CU: the rows about to leave the windows stay cached (-5 %)
#endif
#ifndef EPA_SLIDE_SEG
#define EPA_SLIDE_SEG 512
#endif
constexpr int kSlideSeg = EPA_SLIDE_SEG, kSlideUnroll = 4, kSlidePad = EPA_SLIDE_PAD;

struct DdSum {
  double hi = 0.0, lo = 0.0;
  long long cnt = 0;
  int ninf = 0;
  __device__ __forceinline__ void add(double x, int k, double sign) {
    cnt += sign > 0.0 ? k : -k;
    if (x == __builtin_inf()) {
      ninf += sign > 0.0 ? 1 : -1;
      return;
    }
    x *= sign;
    const double t = hi + x;
    const double bb = t - hi;
    lo += (hi - (t - bb)) + (x - bb);
    hi = t;
  }
  __device__ __forceinline__ double value() const { return ninf > 0 ? __builtin_inf() : hi + lo; }
};

template <typename T>
__global__ __launch_bounds__(kBlock) void box_ping_slide_kernel(const T* __restrict__ sv,
                                                                const double* __restrict__ vsum,
                                                                const int* __restrict__ vcnt, int P, int S,
                                                                int s0, int n, T thr, T* __restrict__ pooled,
                                                                uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  __syncthreads();
  const int s = blockIdx.y * kBlock + threadIdx.x;
  if (s >= S) return;
  const int p0 = blockIdx.x * kSlideSeg, p1 = min(P, p0 + kSlideSeg);
  const size_t cbase = (size_t)blockIdx.z * P * S;
  if (s < s0) {  // above the first pooled sample: NaN, never masked
    for (int p = p0; p < p1; ++p) {
      const size_t at = cbase + (size_t)p * S + s;
      if (pooled) pooled[at] = epa::M<T>::nan();
      if (mask) mask[at] = 0;
    }
    return;
  }
  const double* __restrict__ vs = vsum + cbase + s;
  const int* __restrict__ vc = vcnt + cbase + s;
  DdSum w;
  {  // the first window of the segment, summed afresh (independent loads, four rows at a time)
    int q = p0 - n;
    for (; q + 3 <= p0 + n; q += 4) {
      double a[4];
      int k[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t r = (size_t)reflect_index(q + j, P) * S;
        a[j] = vs[r];
        k[j] = vc[r];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) w.add(a[j], k[j], 1.0);
    }
    for (; q <= p0 + n; ++q) {
      const size_t r = (size_t)reflect_index(q, P) * S;
      w.add(vs[r], vc[r], 1.0);
    }
  }
  auto emit = [&](int p, T x) {
    T out = epa::M<T>::nan();
    if (w.cnt > 0) out = (T)(10.0 * epa::fast_log10(w.value() / (double)w.cnt, mt.log_tab));
    const size_t at = cbase + (size_t)p * S + s;
    if (pooled) pooled[at] = out;
    if (mask) mask[at] = (x - out > thr) ? 1 : 0;
  };
  const T* __restrict__ svc = sv + cbase + s;
  emit(p0, mask ? svc[(size_t)p0 * S] : (T)0);
  int p = p0 + 1;
  for (; p + kSlideUnroll <= p1; p += kSlideUnroll) {
    double ain[kSlideUnroll], aout[kSlideUnroll];
    int kin[kSlideUnroll], kout[kSlideUnroll];
    T x[kSlideUnroll];
#pragma unroll
    for (int j = 0; j < kSlideUnroll; ++j) {
      const size_t ri = (size_t)reflect_index(p + j + n, P) * S, ro = (size_t)reflect_index(p + j - n - 1, P) * S;
      ain[j] = vs[ri];
      kin[j] = vc[ri];
      aout[j] = vs[ro];
      kout[j] = vc[ro];
      x[j] = mask ? svc[(size_t)(p + j) * S] : (T)0;
    }
#pragma unroll
    for (int j = 0; j < kSlideUnroll; ++j) {
      w.add(ain[j], kin[j], 1.0);
      w.add(aout[j], kout[j], -1.0);
      emit(p + j, x[j]);
    }
  }
  for (; p < p1; ++p) {
    const size_t ri = (size_t)reflect_index(p + n, P) * S, ro = (size_t)reflect_index(p - n - 1, P) * S;
    w.add(vs[ri], vc[ri], 1.0);
    w.add(vs[ro], vc[ro], -1.0);
    emit(p, mask ? svc[(size_t)p * S] : (T)0);
  }
}

// ------------------------------------------------------------------------------------------------
// NaN-skipping median of the LINEAR values of a window, by radix selection on the dB values
// (10^(x/10) is monotone, so the order statistics are those of x; an even count averages the two
// middle values in the linear domain, as np.nanmedian(_log2lin(.)) does).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long sort_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
  return __longlong_as_double((long long)((k >> 63) ? (k ^ 0x8000000000000000ull) : ~k));
}

template <typename T>
struct Window {
  const T* base;  // channel base pointer
  int S;          // row stride
  int p_lo, np;   // pings p_lo .. p_lo+np-1
  int s_lo, ns;   // samples s_lo .. s_lo+ns-1
  int P, s0;      // reflect domain: pings [0,P), samples [s0,S)
  bool reflect;
  // calls f(value) for every element this thread owns
  template <typename F>
  __device__ __forceinline__ void for_each(F f) const {
    const int ne = np * ns;
    for (int i = threadIdx.x; i < ne; i += kBlock) {
      const int ip = i / ns, is = i - ip * ns;
      int p = p_lo + ip, s = s_lo + is;
      if (reflect) {
        p = reflect_index(p, P);
        s = s0 + reflect_index(s - s0, S - s0);
      }
      f((double)base[(size_t)p * S + s]);
    }
  }
};

// Window by VALUE of the range variable (pool_Sv, clean/utils.py:86-92): in ping q_lo + j the
// samples lo[j] .. hi[j]-1 (contiguous because the range variable increases along range_sample).
template <typename T>
struct RaggedWindow {
  const T* base;
  int S, q_lo, nq;
  const int* lo;  // LDS
  const int* hi;
  template <typename F>
  __device__ __forceinline__ void for_each(F f) const {
    for (int j = 0; j < nq; ++j) {
      const T* row = base + (size_t)(q_lo + j) * S;
      for (int s = lo[j] + (int)threadIdx.x; s < hi[j]; s += kBlock) f((double)row[s]);
    }
  }
};

constexpr int kCandCap = 2048;  // candidates kept in LDS once the selected radix bucket is this small

template <int CAP>
struct SelectScratchT {
  static constexpr int kCap = CAP;
  unsigned hist[256];
  unsigned u4[4];
  unsigned long long q4[4];
  unsigned digit, krem, bucket, ncand;
  unsigned long long cand[CAP];
};
using SelectScratch = SelectScratchT<kCandCap>;

// One 8-bit radix step of the selection: histogram of digit (key >> shift) & 255 over the keys that
// `each` enumerates and that match `prefix` on the bits above the digit; picks the bucket holding
// rank k.  Returns the total number of keys counted; updates prefix / k; *bucket = size of the bucket.
template <typename Each, typename SC>
__device__ __forceinline__ unsigned radix_step(Each each, SC* sc, int shift,
                                               unsigned long long& prefix, unsigned& k, unsigned& bucket,
                                               bool k_known, unsigned* total_out) {
  __syncthreads();
  sc->hist[threadIdx.x] = 0u;  // kBlock == 256
  __syncthreads();
  const unsigned long long hi_mask = shift == 56 ? 0ull : (~0ull << (shift + 8));
  unsigned* hist = sc->hist;
  const unsigned long long pre = prefix;
  each([&](unsigned long long key) {
    if ((key & hi_mask) == pre) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
  });
  __syncthreads();
  if (!k_known) {  // first step: the histogram total is the number of valid values; k = lower median rank
    const unsigned total = block_sum(sc->hist[threadIdx.x], sc->u4);
    *total_out = total;
    if (total == 0u) return 0u;
    k = (total - 1u) / 2u;
  }
  if (threadIdx.x < 64) {
    const unsigned l = threadIdx.x;
    const unsigned h0 = sc->hist[4 * l], h1 = sc->hist[4 * l + 1], h2 = sc->hist[4 * l + 2],
                   h3 = sc->hist[4 * l + 3];
    const unsigned tot = h0 + h1 + h2 + h3;
    unsigned incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o, 64);
      if ((int)l >= o) incl += t;
    }
    const unsigned excl = incl - tot;
    if (excl <= k && k < incl) {
      unsigned r = k - excl, d;
      if (r < h0) d = 0;
      else if ((r -= h0) < h1) d = 1;
      else if ((r -= h1) < h2) d = 2;
      else { r -= h2; d = 3; }
      sc->digit = 4 * l + d;
      sc->krem = r;
      sc->bucket = d == 0 ? h0 : (d == 1 ? h1 : (d == 2 ? h2 : h3));
    }
  }
  __syncthreads();
  prefix |= (unsigned long long)sc->digit << shift;
  k = sc->krem;
  bucket = sc->bucket;
  return 1u;
}

// Returns the median of 10^(x/10) over the non-NaN x of the window; n_valid = their count (the
// result is NaN when it is 0).  Must be called by all threads of the workgroup.
// The window is swept from memory only until the bucket that holds the median has at most kCandCap
// members (typically 2 sweeps for a 16 k-element block of dB values, 0 for a small window); those
// candidates are then gathered into LDS in one more sweep and the remaining digits are resolved there.
template <typename W, typename SC>
__device__ double window_median_lin(const W& w, SC* sc, const double* exp2_tab,
                                    unsigned& n_valid, int size_hint = 0x7fffffff) {
  constexpr int kCandCap = SC::kCap;  // (shadows the default capacity)
  auto each_global = [&](auto f) {
    w.for_each([&](double v) {
      if (v == v) f(sort_key(v));
    });
  };
  unsigned long long prefix = 0ull;
  unsigned k = 0u, bucket = 0xffffffffu, N = 0u;
  bool k_known = false;
  int shift = 64;  // bits [shift, 64) of the median's key are decided
  if (size_hint > kCandCap) {
    while (shift > 0 && bucket > (unsigned)kCandCap) {
      shift -= 8;
      if (!radix_step(each_global, sc, shift, prefix, k, bucket, k_known, &N)) {
        n_valid = 0u;
        return __builtin_nan("");
      }
      k_known = true;
    }
  }
  unsigned long long key1, key2;
  if (shift == 0) {
    // resolved entirely from memory (a bucket of > kCandCap equal values): one more sweep for the even case
    key1 = key2 = prefix;
    if ((N & 1u) == 0u) {
      unsigned le = 0;
      unsigned long long gt = ~0ull;
      each_global([&](unsigned long long key) {
        if (key <= prefix) ++le;
        else gt = key < gt ? key : gt;
      });
      const unsigned n_le = block_sum(le, sc->u4);
      const unsigned long long min_gt = block_min(gt, sc->q4);
      if (n_le < (N - 1u) / 2u + 2u) key2 = min_gt;
    }
  } else {
    // gather the candidates (keys matching the decided bits) into LDS; remember the smallest key above them
    const unsigned long long dmask = shift == 64 ? 0ull : (~0ull << shift);
    __syncthreads();
    if (threadIdx.x == 0) sc->ncand = 0u;
    __syncthreads();
    unsigned long long above = ~0ull;
    unsigned* ncand = &sc->ncand;
    unsigned long long* cand = sc->cand;
    const unsigned long long pre = prefix;
    each_global([&](unsigned long long key) {
      const unsigned long long hi = key & dmask;
      if (hi == pre) {
        const unsigned at = atomicAdd(ncand, 1u);
        if (at < (unsigned)kCandCap) cand[at] = key;
      } else if (hi > pre) {
        above = key < above ? key : above;
      }
    });
    const unsigned long long min_above = block_min(above, sc->q4);  // (barriers inside publish cand / ncand)
    const unsigned M = sc->ncand;
    if (!k_known) {  // small window gathered whole: M is the number of valid values
      N = M;
      if (N == 0u) {
        n_valid = 0u;
        return __builtin_nan("");
      }
      k = (N - 1u) / 2u;
    }
    const unsigned r0 = k;  // rank of the median inside the candidate set
    auto each_cand = [&](auto f) {
      for (unsigned i = threadIdx.x; i < M; i += kBlock) f(cand[i]);
    };
    unsigned dummy;
    while (shift > 0) {
      shift -= 8;
      radix_step(each_cand, sc, shift, prefix, k, bucket, true, &dummy);
    }
    key1 = key2 = prefix;
    if ((N & 1u) == 0u) {
      unsigned le = 0;
      unsigned long long gt = ~0ull;
      each_cand([&](unsigned long long key) {
        if (key <= prefix) ++le;
        else gt = key < gt ? key : gt;
      });
      const unsigned n_le = block_sum(le, sc->u4);
      const unsigned long long min_gt = block_min(gt, sc->q4);
      if (n_le < r0 + 2u) key2 = (min_gt != ~0ull) ? min_gt : min_above;
    }
  }
  n_valid = N;
  const double a = epa::lin_from_db(key_value(key1), exp2_tab);
  return key1 == key2 ? a : (a + epa::lin_from_db(key_value(key2), exp2_tab)) * 0.5;
}

// pooled Sv with func = nanmedian: one workgroup per output sample (the reference warns that this
// variant is "incredibly slow"; here it is exact and usable on subsets, O(window) per sample).
template <typename T>
__global__ __launch_bounds__(kBlock) void pool_median_kernel(const T* __restrict__ sv, int P, int S,
                                                             long long jobs, int s0, int n, int m,
                                                             T thr, T* __restrict__ pooled,
                                                             uint8_t* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  __shared__ SelectScratch sc;
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  for (long long job = blockIdx.x; job < jobs; job += gridDim.x) {
    const int s = (int)(job % S);
    const long long row = job / S;
    const int p = (int)(row % P);
    const long long c = row / P;
    T out = epa::M<T>::nan();
    if (s >= s0) {
      Window<T> w{sv + (size_t)c * P * S, S, p - n, 2 * n + 1, s - m, 2 * m + 1, P, s0, true};
      unsigned nv;
      const double med = window_median_lin(w, &sc, mt.exp2_tab, nv, (2 * n + 1) * (2 * m + 1));
      if (nv) out = (T)(10.0 * epa::fast_log10(med, mt.log_tab));
    }
    if (threadIdx.x == 0) {
      if (pooled) pooled[job] = out;
      if (mask) mask[job] = (sv[job] - out > thr) ? 1 : 0;
    }
  }
}

// The same pooled median with the window CARRIED from ping to ping (round 4).  A workgroup owns one range column s of
// one channel and walks a segment of pings (lane i owns window column i: the entering row p+n+1 is one coalesced
// read).  A two-level histogram of the (2n+1) x (2m+1) window -- 4096 bins + 64 coarse sums -- is kept up to date with
// 2 (2m+1) LDS atomics per step; the bin that holds the median rank comes from two 64-lane prefix scans (DPP; every
// searching wavefront does them redundantly: no broadcast, no barrier), the handful of values of that bin (and of the
// bin of the upper middle value when the count is even) are then ranked by counting.  The bins are 1/128 dB wide over
// +-16 dB around the median of the segment's first window (found with 1/16-dB bins over [-256, 0) dB first), values
// outside are clamped into the end bins: any monotone map selects exactly, a good one keeps the candidates few.  When
// the median has drifted more than 8 dB from the centre and the candidates are many, the map is re-centred and the
// histogram rebuilt.  Exact (the same two middle values, averaged in the linear domain); more than 64 values in the
// median's bin (a flat field) fall back to the radix selection over the window in memory.
// O(2m+1 + window/1000) per sample instead of O(window) per sample and sweep.
constexpr int kMedBins = 4096, kMedCoarse = 64, kMedSeg = 512, kMedCap = 512;

struct MedMap {
  float scale, off;  // bin = clamp(x * scale + off)
  __device__ __forceinline__ unsigned bin(double x) const {
    return (unsigned)fminf(fmaxf(fmaf((float)x, scale, off), 0.0f), (float)(kMedBins - 1));
  }
};

// inclusive prefix sum over the 64 lanes in 6 DPP additions (no LDS crossbar round trips): Hillis-Steele inside the
// rows of 16 lanes (a lane whose source falls outside its row adds 0), then the last lane of row 0 / 2 into rows 1 / 3,
// then lane 31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_add(unsigned v) {
  return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_scan_incl(unsigned v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

// lane (= bin of a 64-bin histogram held one count per lane) that holds 0-based rank k < total; rem = rank inside it
__device__ __forceinline__ unsigned wave_rank_lane(unsigned incl, unsigned v, unsigned k, unsigned& rem) {
  const unsigned long long bal = __ballot(incl > k);
  const int at = __ffsll((long long)bal) - 1;
  rem = k - (unsigned)__builtin_amdgcn_readlane((int)(incl - v), at);
  return (unsigned)at;
}

__device__ __forceinline__ unsigned long long wave_bcast64(unsigned long long v, int src) {  // src wavefront-uniform
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, src);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
  return ((unsigned long long)hi << 32) | lo;
}

// One workgroup per (channel, ping): layer limits from THIS ping's range row (np.argmin: first
// NaN, else first minimum), median of the ping's layer vs median of the 2n-ping block [p-n, p+n).
template <typename T>
__global__ __launch_bounds__(kBlock) void attenuated_mask_kernel(
    const T* __restrict__ sv, const T* __restrict__ range, int P, int S, long long rows, T upper,
    T lower, int n, T thr, uint8_t* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  __shared__ SelectScratch sc;
  __shared__ double shv[4];
  __shared__ int shi[4];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int p = (int)(row % P);
    const long long c = row / P;
    const T* rr = range + (size_t)row * S;
    double bu = __builtin_inf(), bl = __builtin_inf();
    int iu = 0x7fffffff, il = 0x7fffffff;
    for (int s = threadIdx.x; s < S; s += kBlock) {
      const T r = rr[s];
      T du = fabs(r - upper), dl = fabs(r - lower);
      const double vu = (du == du) ? (double)du : -1.0;  // NaN beats every |.| >= 0
      const double vl = (dl == dl) ? (double)dl : -1.0;
      if (vu < bu) { bu = vu; iu = s; }
      if (vl < bl) { bl = vl; il = s; }
    }
    const int up = block_argmin(bu, iu, shv, shi);
    const int lw = block_argmin(bl, il, shv, shi);
    bool flag = false;
    if (p - n >= 0 && (long long)p + n <= (long long)P - 1 && lw > up) {
      const T* cb = sv + (size_t)c * P * S;
      unsigned nv;
      Window<T> w1{cb, S, p, 1, up, lw - up, P, 0, false};
      const double m1 = window_median_lin(w1, &sc, mt.exp2_tab, nv, lw - up);
      if (nv) {
        Window<T> w2{cb, S, p - n, 2 * n, up, lw - up, P, 0, false};
        unsigned nv2;
        const double m2 = window_median_lin(w2, &sc, mt.exp2_tab, nv2, 2 * n * (lw - up));
        const T ping_db = (T)(10.0 * epa::fast_log10(m1, mt.log_tab));
        const T block_db = nv2 ? (T)(10.0 * epa::fast_log10(m2, mt.log_tab)) : epa::M<T>::nan();
        flag = (ping_db - block_db) < thr;
      }
    }
    uint8_t* m = mask + (size_t)row * S;
    const uint8_t f = flag ? 1 : 0;
    for (int s = threadIdx.x; s < S; s += kBlock) m[s] = f;
  }
}

// ------------------------------------------------------------------------------------------------
// Layer limits of every ping from its own range row (np.argmin: first NaN, else first minimum) and the median of the
// ping's own layer, stashed in the first sixteen bytes of the ping's row of the (not yet written) mask: one fully
// parallel pass over all pings instead of a dependent load + two selections in front of every step of the sequential
// walk of attenuated_walk_kernel.  One WAVEFRONT per ping, no workgroup barrier anywhere: the two argmins by a
// butterfly over the lanes; the median from 16-bit codes of the dB values (monotone, 0.004 dB) counted by their top
// byte, then by the low byte inside the one or two selected top bytes (LDS atomics of the wavefront's own 3 KB, the
// 256 counts scanned four per lane), then the one or two values of the selected code ranked exactly.  A layer of
// more than 2048 samples or more than 64 values inside one code leave kOwnUnknown: the walk takes that ping's
// medians from memory.
constexpr int kPrepCols = 32;  // layer samples per lane: layers up to 2048 samples
constexpr unsigned long long kOwnUnknown = 0x7ff8dead00000000ull;  // (a NaN no arithmetic produces)

__device__ __forceinline__ unsigned att_code_f(double v) {  // monotone 16-bit code over [-200, 50) dB, clamped; 0xffff: NaN
  const float t = fmaf((float)v, 65534.0f / 250.0f, 200.0f * 65534.0f / 250.0f);
  return (unsigned)fminf(fmaxf(t, 0.0f), 65534.0f);
}
// 256 counts held four per lane (h = bins 4l .. 4l+3, incl = inclusive scan of their sum tot): the bin of 0-based
// rank k < total and the rank inside it
__device__ __forceinline__ unsigned wave_locate4(const uint4& h, unsigned incl, unsigned tot, unsigned k, unsigned& r) {
  const int at = __ffsll((long long)__ballot(incl > k)) - 1;
  unsigned rem = k - (unsigned)__builtin_amdgcn_readlane((int)(incl - tot), at);
  const unsigned h0 = (unsigned)__builtin_amdgcn_readlane((int)h.x, at), h1 = (unsigned)__builtin_amdgcn_readlane((int)h.y, at),
                 h2 = (unsigned)__builtin_amdgcn_readlane((int)h.z, at);
  unsigned d = 0u;
  if (rem >= h0) { rem -= h0; d = 1u;
    if (rem >= h1) { rem -= h1; d = 2u;
      if (rem >= h2) { rem -= h2; d = 3u; } } }
  r = rem;
  return 4u * (unsigned)at + d;
}
// LDS traffic of ONE wavefront is processed in program order: this only keeps the compiler from reordering it
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// lexicographic (value, index) minimum over the 64 lanes -- np.argmin's "first occurrence of the minimum"
__device__ __forceinline__ int wave_argmin(double v, int idx) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double w = __shfl_xor(v, o, 64);
    const int j = __shfl_xor(idx, o, 64);
    if (w < v || (w == v && j < idx)) { v = w; idx = j; }
  }
  return idx;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void attenuated_prepare_kernel(const T* __restrict__ sv, const T* __restrict__ range,
                                                                    int P, int S, long long rows, T upper, T lower,
                                                                    int n, uint8_t* __restrict__ mask) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  __shared__ __attribute__((aligned(16))) unsigned hist[kBlock / 64][768];  // per wavefront: top[256], low[2][256]
  __shared__ unsigned long long cand[kBlock / 64][64];
  __shared__ unsigned ctag[kBlock / 64][64], ncand[kBlock / 64];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned* top = hist[wv];
  unsigned* low = top + 256;
  for (long long row = (long long)blockIdx.x * (kBlock / 64) + wv; row < rows; row += (long long)gridDim.x * (kBlock / 64)) {
    const int p = (int)(row % P);
    const T* rr = range + (size_t)row * S;
    double bu = __builtin_inf(), bl = __builtin_inf();
    int iu = 0x7fffffff, il = 0x7fffffff;
    for (int s0 = 0; s0 < S; s0 += 64 * 8) {  // eight requests in flight per lane
      T r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + lane + 64 * j;
        r[j] = s < S ? rr[s] : (T)0;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + lane + 64 * j;
        if (s < S) {
          T du = fabs(r[j] - upper), dl = fabs(r[j] - lower);
          const double vu = (du == du) ? (double)du : -1.0;  // NaN beats every |.| >= 0
          const double vl = (dl == dl) ? (double)dl : -1.0;
          if (vu < bu) { bu = vu; iu = s; }
          if (vl < bl) { bl = vl; il = s; }
        }
      }
    }
    const int up = wave_argmin(bu, iu), lw = wave_argmin(bl, il);
    const int L = lw - up;
    unsigned long long own = 0x7ff8000000000000ull;  // NaN: no valid sample, or no block around the ping
    if (p - n >= 0 && (long long)p + n <= (long long)P - 1 && L > 0) {
      own = kOwnUnknown;
      if (L <= kPrepCols * 64) {
        const T* layer = sv + (size_t)row * S + up;
        const int nj = (L + 63) >> 6;
#pragma unroll
        for (int i = 0; i < 12; ++i) top[lane + 64 * i] = 0u;  // top and both halves of low
        if (lane == 0) ncand[wv] = 0u;
        // only the codes stay in registers (the one or two values that decide are read again below): eight requests
        // in flight per lane
        unsigned short code[kPrepCols];
#pragma unroll
        for (int j0 = 0; j0 < kPrepCols; j0 += 8) {
          T val[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int col = lane + 64 * (j0 + j);
            val[j] = (j0 + j < nj && col < L) ? layer[col] : epa::M<T>::nan();
          }
#pragma unroll
          for (int j = 0; j < 8; ++j)
            code[j0 + j] = (unsigned short)((val[j] == val[j]) ? att_code_f((double)val[j]) : 0xffffu);
        }
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < kPrepCols; ++j)
          if (j < nj && code[j] != 0xffffu) atomicAdd(&top[code[j] >> 8], 1u);
        wave_lds_fence();
        const uint4 ht = reinterpret_cast<const uint4*>(top)[lane];
        const unsigned tt = ht.x + ht.y + ht.z + ht.w, ti = wave_scan_incl(tt);
        const unsigned N = (unsigned)__builtin_amdgcn_readlane((int)ti, 63);
        if (!N) {
          own = 0x7ff8000000000000ull;
        } else {
          const unsigned k1 = (N - 1u) >> 1, k2 = N >> 1;
          unsigned r1t, r2t;
          const unsigned t1 = wave_locate4(ht, ti, tt, k1, r1t), t2 = wave_locate4(ht, ti, tt, k2, r2t);
#pragma unroll
          for (int j = 0; j < kPrepCols; ++j) {
            if (j < nj && code[j] != 0xffffu) {
              const unsigned tb = code[j] >> 8;
              if (tb == t1) atomicAdd(&low[code[j] & 255u], 1u);
              else if (tb == t2) atomicAdd(&low[256u + (code[j] & 255u)], 1u);
            }
          }
          wave_lds_fence();
          const uint4 ha = reinterpret_cast<const uint4*>(low)[lane], hb = reinterpret_cast<const uint4*>(low + 256)[lane];
          const unsigned ta = ha.x + ha.y + ha.z + ha.w, ia = wave_scan_incl(ta);
          const unsigned tb = hb.x + hb.y + hb.z + hb.w, ib = wave_scan_incl(tb);
          unsigned rr1, rr2;
          const unsigned c1 = (t1 << 8) | wave_locate4(ha, ia, ta, r1t, rr1);
          const unsigned c2 = t2 == t1 ? ((t1 << 8) | wave_locate4(ha, ia, ta, r2t, rr2))
                                       : ((t2 << 8) | wave_locate4(hb, ib, tb, r2t, rr2));
          // the members of those codes: one per lane, ranked by counting (the others' keys by readlane)
#pragma unroll
          for (int j = 0; j < kPrepCols; ++j) {
            if (j < nj && (code[j] == c1 || code[j] == c2)) {
              const unsigned at = atomicAdd(&ncand[wv], 1u);
              if (at < 64u) {
                cand[wv][at] = sort_key((double)layer[lane + 64 * j]);
                ctag[wv][at] = code[j];
              }
            }
          }
          wave_lds_fence();
          const int M = __builtin_amdgcn_readfirstlane((int)ncand[wv]);
          if (M <= 64) {
            const unsigned long long ki = lane < M ? cand[wv][lane] : ~0ull;
            const unsigned tg = lane < M ? ctag[wv][lane] : 0xffffffffu;
            unsigned rank = 0u;
            for (int j = 0; j < M; ++j) {
              const unsigned long long kj = wave_bcast64(ki, j);
              const unsigned tj = (unsigned)__builtin_amdgcn_readlane((int)tg, j);
              rank += (tj == tg && (kj < ki || (kj == ki && j < lane))) ? 1u : 0u;
            }
            const int a1 = __ffsll((long long)__ballot(tg == c1 && rank == rr1)) - 1;
            const int a2 = __ffsll((long long)__ballot(tg == c2 && rank == rr2)) - 1;
            const unsigned long long o1 = wave_bcast64(ki, a1), o2 = wave_bcast64(ki, a2);
            const double la = epa::lin_from_db(key_value(o1), mt.exp2_tab);
            const double m1 = o1 == o2 ? la : (la + epa::lin_from_db(key_value(o2), mt.exp2_tab)) * 0.5;
            own = (unsigned long long)__double_as_longlong((double)(T)(10.0 * epa::fast_log10(m1, mt.log_tab)));
          }
        }
      }
    }
    if (lane == 0) {
      int* dst = reinterpret_cast<int*>(mask + (size_t)row * S);  // rows are S >= 16 bytes apart, S % 4 == 0 checked
      dst[0] = up;
      dst[1] = lw;
      dst[2] = (int)(unsigned)own;
      dst[3] = (int)(unsigned)(own >> 32);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pooled Sv by VALUE windows (pool_Sv, clean/utils.py:29-106): for every sample at depth d in ping p
// the aggregate of the linear Sv over pings p-n..p+n and depths [d - bin, d + bin], where feasible.
// The range variable must be non-decreasing along range_sample with NaN only as a tail
// (rows_check_kernel verifies this and yields the valid length of every row).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void rows_check_kernel(const T* __restrict__ x, long long rows, int S,
                                                            int* __restrict__ nvalid,
                                                            int* __restrict__ violations) {
  __shared__ int sfirst;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* xr = x + (size_t)row * S;
    __syncthreads();
    if (threadIdx.x == 0) sfirst = S;
    __syncthreads();
    int bad = 0;
    for (int s = threadIdx.x; s < S; s += kBlock) {
      const T v = xr[s];
      if (!(v == v)) atomicMin(&sfirst, s);
      else if (s + 1 < S && xr[s + 1] < v) bad = 1;
    }
    __syncthreads();
    const int first = sfirst;
    for (int s = first + threadIdx.x; s < S; s += kBlock)
      if (xr[s] == xr[s]) bad = 1;  // a number after the first NaN
    if (bad) atomicAdd(violations, 1);
    if (threadIdx.x == 0) nvalid[row] = first;
  }
}

// first index in [0, n) with row[idx] >= v (STRICT == false) or row[idx] > v (STRICT == true)
template <typename T, bool STRICT>
__device__ __forceinline__ int bound(const T* __restrict__ row, int n, T v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const T x = row[mid];
    if (STRICT ? (x > v) : (x >= v)) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}

template <typename T>
struct PoolValueArgs {
  const T* sv;
  const T* range;
  const int* nvalid;
  int P, S, n;
  T bin, exclude_above, rmin, rmax, thr;
  T* pooled;
  uint8_t* mask;
};

template <typename T>
__device__ __forceinline__ bool pool_feasible(const PoolValueArgs<T>& a, T d, int p) {
  // clean/utils.py:77-83 (ping_time_index_max = len(ping_time): p + n == P is accepted)
  return (d - a.bin >= a.rmin) && (d + a.bin <= a.rmax) && (d - a.bin >= a.exclude_above) &&
         (p - a.n >= 0) && ((long long)p + a.n <= (long long)a.P);
}

// (the part of pool_feasible that depends on the range value only)
template <typename T>
__device__ __forceinline__ bool pool_depth_feasible(const PoolValueArgs<T>& a, T d) {
  return (d - a.bin >= a.rmin) && (d + a.bin <= a.rmax) && (d - a.bin >= a.exclude_above);
}

// nanmean: one thread per output sample
template <typename T>
__global__ __launch_bounds__(kBlock) void pool_value_mean_kernel(PoolValueArgs<T> a, long long rows) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  const int s = blockIdx.y * kBlock + threadIdx.x;
  if (s >= a.S) return;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const int p = (int)(row % a.P);
    const long long c = row / a.P;
    const size_t at = (size_t)row * a.S + s;
    const T d = a.range[at];
    T out = epa::M<T>::nan();
    if (pool_feasible(a, d, p)) {
      const T lo_v = d - a.bin, hi_v = d + a.bin;
      double sum = 0.0;
      long long cnt = 0;
      const int q1 = min(p + a.n, a.P - 1);
      for (int q = p - a.n; q <= q1; ++q) {
        const size_t qrow = (size_t)(c * a.P + q);
        const T* rr = a.range + qrow * a.S;
        const T* vr = a.sv + qrow * a.S;
        const int nv = a.nvalid[qrow];
        const int lo = bound<T, false>(rr, nv, lo_v), hi = bound<T, true>(rr, nv, hi_v);
        for (int k = lo; k < hi; ++k) {
          const T v = vr[k];
          if (v == v) {
            sum += (double)epa::lin_from_db(v, mt.exp2_tab);
            ++cnt;
          }
        }
      }
      if (cnt > 0) out = (T)(10.0 * epa::fast_log10(sum / (double)cnt, mt.log_tab));
    }
    if (a.pooled) a.pooled[at] = out;
    if (a.mask) a.mask[at] = (a.sv[at] - out > a.thr) ? 1 : 0;
  }
}

// ---- nanmean through per-row running sums ------------------------------------------------------------------
// The brute-force kernel above visits every value of every window (and converts it to linear each time):
// (2n+1) x ~2*bin/step values per sample.  With the running sums of the linear Sv along each row,
//   W[k] = sum_{j<=k} lin(Sv[j]),  N[k] = number of non-NaN values among j<=k,
// a row contributes W[hi-1] - W[lo-1] for its index interval [lo, hi) -- O(1) per row once lo / hi are known, and
// they are known from the row above unless the range vectors differ (checked with two loads, re-searched if
// not).  W is kept in double-double (two-sum), so the difference of two running sums is the window sum to
// ~1e-30 of the ROW total: nothing is lost to cancellation.  Rows holding a +inf Sv are flagged and summed
// value by value (inf - inf).  Workspace: 20 B per sample + 1 B per row.
struct Dd {
  double hi, lo;
  __device__ __forceinline__ void add(double x) {  // x finite
    const double t = hi + x, bb = t - hi;
    lo += (hi - (t - bb)) + (x - bb);
    hi = t;
  }
  __device__ __forceinline__ void add(const Dd& b, double sign) {
    const double bh = sign * b.hi, bl = sign * b.lo;
    const double t = hi + bh, bb = t - hi;
    const double e = (hi - (t - bb)) + (bh - bb) + lo + bl;
    hi = t + e;
    lo = e - (hi - t);
  }
};

// One workgroup per row, one quarter of the row per wavefront, lanes interleaved (coalesced 512-B accesses):
// sweep 1 sums the quarter (wave totals -> offsets), sweep 2 scans it 64 samples at a time with a shuffle scan
// in double-double and carries the running value to the next 64.
__device__ __forceinline__ Dd dd_shfl_up(const Dd& v, int o) {
  return Dd{__shfl_up(v.hi, o, 64), __shfl_up(v.lo, o, 64)};
}

// The channels whose pings share one range vector do not take this route: their rows' interval sums come from
// row_interval_blocks_kernel below (no running sums, no subtraction) and this launch skips them (skip_same).
struct FuseArgs {
  const int* differ;
  const int* nvalid;
  const int* ilo;
  const int* ihi;
  double* rh;
  int* rn;
  int P, skip_same;
  // the run form (round 6): row_interval_blocks_kernel takes the rows of the long runs of the channels that differ (tables
  // ilo / ihi indexed by the row's slot); row_running_sum_kernel only the rows some staged ping still needs (need_w)
  const int* row_slot;
  const uint8_t* need_w;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void row_running_sum_kernel(const T* __restrict__ sv, long long rows, int S,
                                                                 double* __restrict__ wh, double* __restrict__ wl,
                                                                 int* __restrict__ wn, uint8_t* __restrict__ dirty,
                                                                 FuseArgs fa) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  __shared__ double th[4], tl[4];
  __shared__ int tc[4], any_inf;
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int Q = (((S + 3) / 4) + 127) & ~127;
  const int k0 = min(S, wave * Q), k1 = min(S, k0 + Q);
  // two consecutive samples per lane: 16-byte loads and stores (half the requests of one sample per lane)
  typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
  typedef double dpair_t __attribute__((ext_vector_type(2), aligned(8)));
  typedef int ipair_t __attribute__((ext_vector_type(2), aligned(4)));
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const long long chan = row / fa.P;
    if (fa.skip_same && fa.differ[chan] == 0) continue;  // uniform per workgroup (row_interval_blocks_kernel has it)
    if (fa.need_w && !fa.need_w[row]) continue;           // (uniform) no ping left to the staged kernels reads this row
    __syncthreads();
    if (threadIdx.x == 0) any_inf = 0;
    const T* svr = sv + (size_t)row * S;
    auto lin_of = [&](T v, double& x, int& c, bool& inf) {  // linear value (0 for NaN / +inf), count, inf flag
      x = 0.0;
      c = 0;
      if (v == v) {
        const double y = (double)epa::lin_from_db(v, mt.exp2_tab);
        if (y == __builtin_inf()) inf = true; else x = y;
        c = 1;
      }
    };
    Dd acc{0.0, 0.0};
    int cnt = 0;
    bool inf = false;
    for (int k = k0 + 2 * lane; k < k1; k += 128) {
      T v0, v1 = epa::M<T>::nan();
      if (k + 1 < k1) {
        const pair_t v = *reinterpret_cast<const pair_t*>(svr + k);
        v0 = v.x; v1 = v.y;
      } else {
        v0 = svr[k];
      }
      double x0, x1;
      int c0, c1;
      lin_of(v0, x0, c0, inf);
      lin_of(v1, x1, c1, inf);
      acc.add(x0);
      acc.add(x1);
      cnt += c0 + c1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      acc.add(Dd{__shfl_down(acc.hi, o, 64), __shfl_down(acc.lo, o, 64)}, 1.0);
      cnt += __shfl_down(cnt, o, 64);
    }
    __syncthreads();  // any_inf reset, previous row's totals consumed
    if (lane == 0) { th[wave] = acc.hi; tl[wave] = acc.lo; tc[wave] = cnt; }
    if (inf) any_inf = 1;
    __syncthreads();
    Dd carry{0.0, 0.0};
    int carry_n = 0;
    for (int w = 0; w < wave; ++w) {
      carry.add(Dd{th[w], tl[w]}, 1.0);
      carry_n += tc[w];
    }
    double* whr = wh + (size_t)row * S;
    double* wlr = wl + (size_t)row * S;
    int* wnr = wn + (size_t)row * S;
    for (int kb = k0; kb < k1; kb += 128) {
      const int k = kb + 2 * lane;
      T v0 = epa::M<T>::nan(), v1 = epa::M<T>::nan();
      if (k + 1 < k1) {
        const pair_t v = *reinterpret_cast<const pair_t*>(svr + k);
        v0 = v.x; v1 = v.y;
      } else if (k < k1) {
        v0 = svr[k];
      }
      double x0, x1;
      int c0, c1;
      bool dummy = false;
      lin_of(v0, x0, c0, dummy);
      lin_of(v1, x1, c1, dummy);
      Dd v{x0, 0.0};
      v.add(x1);          // the lane's pair total
      int c = c0 + c1;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {  // inclusive scan of the pair totals across the wavefront
        const Dd u = dd_shfl_up(v, o);
        const int uc = __shfl_up(c, o, 64);
        if (lane >= o) {
          v.add(u, 1.0);
          c += uc;
        }
      }
      // running value before this lane's pair = scan of the lane below (+ what came before this trip)
      Dd before = dd_shfl_up(v, 1);
      int before_n = __shfl_up(c, 1, 64);
      if (lane == 0) { before = Dd{0.0, 0.0}; before_n = 0; }
      before.add(carry, 1.0);
      before_n += carry_n;
      Dd r0 = before;
      r0.add(x0);
      Dd r1 = r0;
      r1.add(x1);
      if (k + 1 < k1) {
        *reinterpret_cast<dpair_t*>(whr + k) = dpair_t{r0.hi, r1.hi};
        *reinterpret_cast<dpair_t*>(wlr + k) = dpair_t{r0.lo, r1.lo};
        *reinterpret_cast<ipair_t*>(wnr + k) = ipair_t{before_n + c0, before_n + c0 + c1};
      } else if (k < k1) {
        whr[k] = r0.hi; wlr[k] = r0.lo; wnr[k] = before_n + c0;
      }
      Dd tot{__shfl(v.hi, 63, 64), __shfl(v.lo, 63, 64)};
      carry.add(tot, 1.0);
      carry_n += __shfl(c, 63, 64);
    }
    if (threadIdx.x == 0) dirty[row] = (uint8_t)any_inf;
  }
}

// The interval sums of a row WITHOUT a subtraction (round 4; channels whose pings share one range vector).  van Herk /
// Gil-Werman with blocks of 16 samples -- one DPP row: every sample gets the sum from its block's start up to it
// (pre) and from it to its block's end (suf), four DPP additions each, in plain doubles; an interval [lo, hi) that
// leaves lo's block is  suf[lo] + the whole blocks in between + pre[hi-1]  -- additions of non-negative numbers only,
// so a sample 10^14 times its neighbours disturbs nobody outside its own intervals and the double-double running sums
// of row_running_sum_kernel (two sweeps of the row, 6-step shuffle scans of pairs of doubles) are not needed.  An
// interval inside one block is summed value by value (<= 16, read again from memory).  Counts of valid / +inf
// samples the same way, packed.  One conversion per sample, the row read once.  LDS: 20 B per sample; the channel's
// intervals stay in registers from row to row.
template <int CTRL>
__device__ __forceinline__ double dpp_row_add(double v) {  // v + (v of the lane CTRL says, 0 where the row ends)
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)b >> 32), CTRL, 0xf, 0xf, false);
  return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_row_addu(unsigned v) {
  return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void row_interval_blocks_kernel(const T* __restrict__ sv, long long rows, int S,
                                                                     FuseArgs fa) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  const int Sp = (S + 63) & ~63;
  double* pre = reinterpret_cast<double*>(smem);  // [Sp]
  double* suf = pre + Sp;
  double* tot = suf + Sp;                         // [Sp / 16]
  unsigned short* cpre = reinterpret_cast<unsigned short*>(tot + Sp / 16);  // valid | inf << 8, inside the block
  unsigned short* csuf = cpre + Sp;
  unsigned short* ctot = csuf + Sp;               // [Sp / 16]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int trips = Sp >> 6;
  constexpr int kFly = 8;  // requests in flight per lane
  constexpr int kCols = 8; // columns per lane whose interval is kept in registers (rows up to 2048 samples)
  long long have = -1;
  int clo[kCols], chi[kCols];
  // (the run form walks CONSECUTIVE rows per workgroup: the interval table in registers changes with the run, not with
  //  every row)
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  for (long long it = 0; it < per; ++it) {
    const long long row = fa.row_slot ? (long long)blockIdx.x * per + it : (long long)blockIdx.x + it * gridDim.x;
    if (row >= rows) break;
    long long chan = row / fa.P;
    if (fa.row_slot) {  // uniform per workgroup
      if (fa.differ[chan] == 0 || fa.row_slot[row] < 0) continue;
      chan = fa.row_slot[row];
    } else if (fa.differ[chan] != 0) {
      continue;
    }
    if (chan != have) {
      have = chan;
#pragma unroll
      for (int j = 0; j < kCols; ++j) {
        const int s = threadIdx.x + j * kBlock;
        clo[j] = s < S ? fa.ilo[(size_t)chan * S + s] : -1;
        chi[j] = s < S ? fa.ihi[(size_t)chan * S + s] : -1;
      }
    }
    const T* svr = sv + (size_t)row * S;
    __syncthreads();  // the previous row's readers
    for (int t0 = wave; t0 < trips; t0 += 4 * kFly) {
      T v[kFly];
#pragma unroll
      for (int f = 0; f < kFly; ++f) {
        const int i = (t0 + 4 * f) * 64 + lane;
        v[f] = (t0 + 4 * f < trips && i < S) ? svr[i] : epa::M<T>::nan();
      }
#pragma unroll
      for (int f = 0; f < kFly; ++f) {
        if (t0 + 4 * f < trips) {
          const int i = (t0 + 4 * f) * 64 + lane;
          double x = 0.0;
          unsigned c = 0u;
          if (v[f] == v[f]) {
            const double y = (double)epa::lin_from_db(v[f], mt.exp2_tab);
            c = 1u;
            if (y == __builtin_inf()) c = 0x101u; else x = y;
          }
          double fs = x, bs = x;
          unsigned fc = c, bc = c;
          fs = dpp_row_add<0x111>(fs); fs = dpp_row_add<0x112>(fs); fs = dpp_row_add<0x114>(fs); fs = dpp_row_add<0x118>(fs);
          bs = dpp_row_add<0x101>(bs); bs = dpp_row_add<0x102>(bs); bs = dpp_row_add<0x104>(bs); bs = dpp_row_add<0x108>(bs);
          fc = dpp_row_addu<0x111>(fc); fc = dpp_row_addu<0x112>(fc); fc = dpp_row_addu<0x114>(fc); fc = dpp_row_addu<0x118>(fc);
          bc = dpp_row_addu<0x101>(bc); bc = dpp_row_addu<0x102>(bc); bc = dpp_row_addu<0x104>(bc); bc = dpp_row_addu<0x108>(bc);
          pre[i] = fs;
          suf[i] = bs;
          cpre[i] = (unsigned short)fc;
          csuf[i] = (unsigned short)bc;
          if ((lane & 15) == 15) {
            tot[i >> 4] = fs;
            ctot[i >> 4] = (unsigned short)fc;
          }
        }
      }
    }
    __syncthreads();
    const int nv = fa.nvalid[row];
    const size_t base = (size_t)row * S;
    auto column = [&](int s, int lo, int hi) {
      lo = min(lo, nv);
      hi = min(hi, nv);
      double sum = 0.0;
      unsigned valid = 0u, infs = 0u;
      if (lo >= 0 && hi > lo) {
        const int bl = lo >> 4, bh = (hi - 1) >> 4;
        if (bl == bh) {
          for (int k = lo; k < hi; ++k) {
            const T v = svr[k];
            if (v == v) {
              const double y = (double)epa::lin_from_db(v, mt.exp2_tab);
              if (y != __builtin_inf()) sum += y;
            }
          }
          const unsigned a = cpre[hi - 1], b = (lo & 15) ? cpre[lo - 1] : 0u;  // (integers: a difference is exact)
          valid = (a & 255u) - (b & 255u);
          infs = (a >> 8) - (b >> 8);
        } else {
          sum = suf[lo];
          unsigned c = csuf[lo];
          valid = c & 255u; infs = c >> 8;
          for (int b = bl + 1; b < bh; ++b) {
            sum += tot[b];
            c = ctot[b];
            valid += c & 255u; infs += c >> 8;
          }
          sum += pre[hi - 1];
          c = cpre[hi - 1];
          valid += c & 255u; infs += c >> 8;
        }
      }
      fa.rh[base + s] = infs ? __builtin_inf() : sum;
      fa.rn[base + s] = (int)valid;
    };
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
      const int s = threadIdx.x + j * kBlock;
      if (s < S) column(s, clo[j], chi[j]);
    }
    for (int s = threadIdx.x + kCols * kBlock; s < S; s += kBlock)
      column(s, fa.ilo[(size_t)chan * S + s], fa.ihi[(size_t)chan * S + s]);
  }
}

// first index with row[idx] >= v / > v, trying the answer of the previous row first and, when that is off, its
// neighbours (range rows that differ by a heave offset or a drifting sound speed move the answer by a sample or two)
template <typename T, bool STRICT>
__device__ __forceinline__ int bound_hint(const T* __restrict__ row, int n, T v, int hint) {
  if (hint >= 0 && hint <= n) {
    int h = hint;
#pragma unroll 1
    for (int tries = 0; tries < 4; ++tries) {
      const bool below = h == 0 || !(STRICT ? (row[h - 1] > v) : (row[h - 1] >= v));
      const bool above = h == n || (STRICT ? (row[h] > v) : (row[h] >= v));
      if (below && above) return h;
      h += below ? 1 : -1;  // (below: row[h] is still short of v; otherwise row[h - 1] already reaches it)
    }
  }
  return bound<T, STRICT>(row, n, v);
}

// ---- value windows over pings whose range rows differ: running sums W, neighbour rows staged in LDS -----------------
// One lane per sample with the bounds and W[hi-1], W[lo-1] read from global memory asks the L1 for ~14 scattered loads
// per (sample, neighbour ping): with 2n + 1 = 51 pings that is ~7 KB of cache traffic per sample, and the kernel ran
// at the L1's rate (2.4 Gsamp/s; 1.5 when a heave offset moved the bounds from ping to ping).  Here a workgroup takes
// kStageRows consecutive pings x 256 range columns.  For every neighbour ping q it needs the index span [kmin, kmax)
// of q's range row that the windows of ALL its samples can reach: the rows are non-decreasing, so two binary searches
// per neighbour, and the searches of all neighbours run side by side on the workgroup's lanes before the loop.  Per
// neighbour the span of the range row and of the running sums W is copied into LDS with coalesced loads (fetched into
// registers while the previous neighbour is being summed).  A lane's window bounds inside the span: range rows are
// affine in the sample index (echo_range, depth = echo_range x cos(tilt) + offset), so the position is computed from
// the span's two ends and confirmed on the four values around it (sentinels at both ends, no bounds checks, no
// branches) -- a wavefront with a lane that cannot confirm searches the span instead.  The window sum is
// (Wh[hi-1] - Wh[lo-1]) + (Wl[hi-1] - Wl[lo-1]).  Pings of the group with the same range value at a column -- all of
// them while the recorded sound speed holds -- share the interval and its sum: resolved once per (neighbour,
// wavefront), added to each ping that has q inside its ping window.  A span longer than kStageCap, a row holding a
// +inf Sv (summed value by value), or a ping window of more than kSpanMax neighbours reads global memory directly for
// that (ping, q) pair.  4 x 20 000 x 2000: 20 ms fp64 / 18 ms fp32 when the range vector changes every 2000 pings,
// 29 / 25 ms with a heave offset on every ping (57 / 109 ms before); three workgroups per CU (launch bound: 168
// registers; the fp64 instantiation spills 12 dwords outside the loop over neighbours).
constexpr int kStageRows = 8, kStageLoads = 3, kStageCap = kStageLoads * kBlock - 1, kSpanMax = 512;

// Runs (round 6): inside a channel whose pings do not all share one range vector, the pings of a RUN that does -- a
// recorded sound speed that holds for two thousand pings -- take the sliding route when their whole ping window lies
// inside the run (``elig``, per ping; see the run kernels below); a group of pings that are all eligible is none of the
// lean / staged kernels' business.
__device__ __forceinline__ bool group_all_eligible(const uint8_t* __restrict__ elig, long long row0, int nrows) {
  if (!elig) return false;
  bool all = true;
  for (int i = 0; i < nrows; ++i) all = all && elig[row0 + i] != 0;
  return all;
}

template <typename T>
__global__ __launch_bounds__(kBlock, 3) void pool_value_mean_staged_kernel(PoolValueArgs<T> a, int C,
                                                                        const double* __restrict__ wh,
                                                                        const double* __restrict__ wl,
                                                                        const int* __restrict__ wn,
                                                                        const uint8_t* __restrict__ dirty,
                                                                        const int* __restrict__ differ,
                                                                        const uint8_t* __restrict__ todo,
                                                                        const uint8_t* __restrict__ elig) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  // seg_r[2 + i] = range[kmin + i], two -inf before and two +inf behind: the three candidate positions around a guess
  // are tested without a bounds check
  __shared__ T seg_r[kStageCap + 4];
  __shared__ double seg_h[kStageCap + 1], seg_l[kStageCap + 1];  // entry i = W[kmin - 1 + i] (0 before the row)
  __shared__ int seg_n[kStageCap + 1];
  __shared__ double red_lo[4], red_hi[4];
  __shared__ int kspan[2 * kSpanMax];
  {  // (uniform) a workgroup with nothing to do -- the usual case behind the lean kernel -- leaves before the tables
    bool any = false;
    const int gpc = (a.P + kStageRows - 1) / kStageRows;
    for (long long grp = blockIdx.x; grp < (long long)C * gpc && !any; grp += gridDim.x) {
      const long long cc = grp / gpc;
      const int q0 = (int)(grp - cc * gpc) * kStageRows;
      any = differ[cc] != 0 && (!todo || todo[(long long)blockIdx.y * ((long long)C * gpc) + grp] != 0) &&
            !group_all_eligible(elig, cc * a.P + q0, min(kStageRows, a.P - q0));
    }
    if (!any) return;
  }
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  const int s = blockIdx.y * kBlock + threadIdx.x;
  const bool in_row = s < a.S;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups_per_channel = (a.P + kStageRows - 1) / kStageRows;
  const long long ngroups = (long long)C * groups_per_channel;
  const T inf = (T)__builtin_inf();
  for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const long long c = grp / groups_per_channel;
    if (!differ[c]) continue;  // every ping of the channel has the same range vector: value_slide_kernel
    // (after pool_value_mean_lean_kernel: only the groups it has left -- todo is indexed band-major as its work items)
    if (todo && !todo[(long long)blockIdx.y * ngroups + grp]) continue;
    const int p0 = (int)(grp - c * groups_per_channel) * kStageRows;
    const int nrows = min(kStageRows, a.P - p0);
    if (group_all_eligible(elig, c * a.P + p0, nrows)) continue;  // (uniform) the sliding route has these pings
    // ---- this lane's samples of the group's pings
    T d[kStageRows];
    unsigned feas = 0, dfeas = 0;  // dfeas: feasible but for the ping window (its interval may be SHARED with a later ping)
    double vmin = __builtin_inf(), vmax = -__builtin_inf();
    // (the eight requests of the lane first, from clamped positions: tested one by one they went out one by one)
    {
      const T* col = a.range + ((size_t)(c * a.P + p0)) * a.S + min(s, a.S - 1);
#pragma unroll
      for (int r = 0; r < kStageRows; ++r) d[r] = col[(size_t)min(r, nrows - 1) * a.S];
    }
#pragma unroll
    for (int r = 0; r < kStageRows; ++r) {
      if (!(r < nrows && in_row)) d[r] = epa::M<T>::nan();
      if (r < nrows && in_row) {
        if (pool_depth_feasible(a, d[r])) dfeas |= 1u << r;
        if (pool_feasible(a, d[r], p0 + r)) {
          feas |= 1u << r;
          vmin = fmin(vmin, (double)(d[r] - a.bin));
          vmax = fmax(vmax, (double)(d[r] + a.bin));
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = fmin(vmin, __shfl_down(vmin, o, 64));
      vmax = fmax(vmax, __shfl_down(vmax, o, 64));
    }
    __syncthreads();  // (the previous group is done with red / kspan / the staged span)
    if (lane == 0) {
      red_lo[wave] = vmin;
      red_hi[wave] = vmax;
    }
    __syncthreads();
    const T gmin = (T)fmin(fmin(red_lo[0], red_lo[1]), fmin(red_lo[2], red_lo[3]));
    const T gmax = (T)fmax(fmax(red_hi[0], red_hi[1]), fmax(red_hi[2], red_hi[3]));
    double sum[kStageRows];
    int cnt[kStageRows];
    unsigned has_inf = 0;
#pragma unroll
    for (int r = 0; r < kStageRows; ++r) {
      sum[r] = 0.0;
      cnt[r] = 0;
    }
    if (gmin <= gmax) {  // (uniform) some window of the group is feasible
      const int q_first = max(p0 - a.n, 0), q_last = min(p0 + nrows - 1 + a.n, a.P - 1);
      const int nq = q_last - q_first + 1;
      const bool can_stage = nq <= kSpanMax;  // (uniform)
      if (can_stage) {  // the spans of all neighbours: one binary search per lane, side by side
        for (int i = threadIdx.x; i < 2 * nq; i += kBlock) {
          const size_t row = (size_t)(c * a.P + q_first) + (i >> 1);
          const T* r2 = a.range + row * a.S;
          kspan[i] = (i & 1) ? bound<T, true>(r2, a.nvalid[row], gmax) : bound<T, false>(r2, a.nvalid[row], gmin);
        }
      }
      __syncthreads();
      // a neighbour's span travels HBM/L2 -> registers while the previous neighbour is being summed, registers -> LDS after
      T pre_r[kStageLoads];
      double pre_h[kStageLoads], pre_l[kStageLoads];
      int pre_n[kStageLoads];
      auto staged_at = [&](int qi) {
        return can_stage && kspan[2 * qi + 1] - kspan[2 * qi] <= kStageCap && dirty[(size_t)(c * a.P + q_first + qi)] == 0;
      };
      auto fetch = [&](int qi) {
        if (!staged_at(qi)) return;
        const size_t base = (size_t)(c * a.P + q_first + qi) * a.S;
        const int kmin = kspan[2 * qi], len = kspan[2 * qi + 1] - kmin;
        // (uniform row pointers + a lane offset: one address register per load instead of a 64-bit sum each)
        const T* rb = a.range + base + kmin;
        const size_t w0 = base + kmin - (kmin > 0 ? 1 : 0);
        const double *hb = wh + w0, *lb = wl + w0;
        const int* nb = wn + w0;
        const unsigned skip = kmin > 0 ? 0u : 1u;  // entry 0 = W[-1] = 0 when the span starts the row
#pragma unroll
        for (int u = 0; u < kStageLoads; ++u) {
          const unsigned i = threadIdx.x + u * kBlock;
          pre_r[u] = i < (unsigned)len ? rb[i] : inf;  // (+inf: the two sentinels behind the span)
          const bool in = i <= (unsigned)len && i >= skip;
          pre_h[u] = in ? hb[i - skip] : 0.0;
          pre_l[u] = in ? lb[i - skip] : 0.0;
          pre_n[u] = in ? nb[i - skip] : 0;
        }
      };
      auto store = [&](int qi) {
        if (!staged_at(qi)) return;
        const int len = kspan[2 * qi + 1] - kspan[2 * qi];
        if (threadIdx.x < 2) seg_r[threadIdx.x] = -inf;
#pragma unroll
        for (int u = 0; u < kStageLoads; ++u) {
          const int i = threadIdx.x + u * kBlock;
          if (i < len + 2) seg_r[2 + i] = pre_r[u];
          if (i <= len) {
            seg_h[i] = pre_h[u];
            seg_l[i] = pre_l[u];
            seg_n[i] = pre_n[u];
          }
        }
      };
      fetch(0);
      int hint_l = -1, hint_h = -1;  // (pairs that read global memory: the previous pair's answer first)
      for (int q = q_first; q <= q_last; ++q) {
        const int qi = q - q_first;
        const size_t qrow = (size_t)(c * a.P + q);
        const bool staged = staged_at(qi);  // (uniform)
        const int len = can_stage ? kspan[2 * qi + 1] - kspan[2 * qi] : 0;
        __syncthreads();  // the previous neighbour's span has been consumed
        store(qi);
        __syncthreads();
        if (q < q_last) fetch(qi + 1);  // (in flight while this neighbour is summed)
        // the group's pings that have q inside their ping window [p - n, min(p + n, P - 1)]
        const int r_lo = max(q - a.n - p0, 0), r_hi = min(q + a.n - p0, nrows - 1);
        if (staged) {
          // range rows are (nearly always) affine in the sample index: the position of a value is guessed from the
          // span's ends and confirmed on the values around the guess; anything else is searched
          const T first = seg_r[2], last = seg_r[len + 1];
          const T inv = (len > 1 && last > first) ? (T)(len - 1) / (last - first) : (T)0;
          const T reach = a.bin * inv;
          // pings of the group with the same range value at this column (all of them, while the sound speed holds)
          // share the interval in q's row and its sum: resolved once, added to each
          T same_d = epa::M<T>::nan();
          double same_w = 0.0;
          int same_c = 0;
#pragma unroll
          for (int r = 0; r < kStageRows; ++r) {
            if (r < r_lo || r > r_hi) continue;  // (uniform)
            // (depth-feasible, not only feasible: a ping too close to the file's start shares its interval with the next
            //  ping of the same range value, which may be feasible -- the search below must not leave its lanes out)
            const bool live = (dfeas >> r) & 1u;
            if (!__all(d[r] == same_d || !live)) {  // (uniform)
              same_d = d[r];
              const T lo_v = d[r] - a.bin, hi_v = d[r] + a.bin;
              const T x = (d[r] - first) * inv;
              // first index with range >= lo_v / > hi_v, as positions 0 .. len inside the span
              int gl = (int)ceil(x - reach), gh = (int)floor(x + reach) + 1;
              gl = min(max(gl, 0), len);
              gh = min(max(gh, 0), len);
              const T a0 = seg_r[gl], a1 = seg_r[gl + 1], a2 = seg_r[gl + 2], a3 = seg_r[gl + 3];
              const T b0 = seg_r[gh], b1 = seg_r[gh + 1], b2 = seg_r[gh + 2], b3 = seg_r[gh + 3];
              int l = (a1 < lo_v && lo_v <= a2) ? gl : (a0 < lo_v && lo_v <= a1) ? gl - 1 : (a2 < lo_v && lo_v <= a3) ? gl + 1 : -1;
              int h = (b1 <= hi_v && hi_v < b2) ? gh : (b0 <= hi_v && hi_v < b1) ? gh - 1 : (b2 <= hi_v && hi_v < b3) ? gh + 1 : -1;
              if (__any(live && (l < 0 || h < 0))) {  // (uniform, rare) a row that is not affine around here
                if (live && l < 0) l = bound<T, false>(seg_r + 2, len, lo_v);
                if (live && h < 0) h = bound<T, true>(seg_r + 2, len, hi_v);
              }
              l = max(l, 0);
              h = max(h, 0);
              // W[hi-1] - W[lo-1]: the high parts subtract exactly when they are close (a window far down a row whose
              // total is 1e14 times its own), the low parts carry what the running sum had rounded away
              const double w = (seg_h[h] - seg_h[l]) + (seg_l[h] - seg_l[l]);
              const int n_in = seg_n[h] - seg_n[l];
              same_w = h > l ? w : 0.0;
              same_c = h > l ? n_in : 0;
            }
            sum[r] += same_w;  // (a plain sum of the -- non-negative -- window sums)
            cnt[r] += same_c;
          }
        } else {
          // (rare: a span beyond the LDS copy, a row holding +inf, a very long ping window.  One pass of compact code
          // per ping, the ping's registers picked by selects, instead of eight unrolled copies)
          const T* rr = a.range + qrow * a.S;
          const int nv = a.nvalid[qrow];
          const size_t base = qrow * a.S;
          const bool dirty_q = dirty[qrow] != 0;
#pragma unroll 1
          for (int r = r_lo; r <= r_hi; ++r) {
            T dr = d[0];
#pragma unroll
            for (int k = 1; k < kStageRows; ++k) dr = r == k ? d[k] : dr;
            double add_w = 0.0;
            int add_c = 0;
            if ((feas >> r) & 1u) {
              hint_l = bound_hint<T, false>(rr, nv, dr - a.bin, hint_l);
              hint_h = bound_hint<T, true>(rr, nv, dr + a.bin, hint_h);
              if (hint_h > hint_l) {
                if (dirty_q) {  // a +inf Sv somewhere in this row: value by value
                  const T* vr = a.sv + base;
                  for (int k = hint_l; k < hint_h; ++k) {
                    const T v = vr[k];
                    if (v == v) {
                      const double x = (double)epa::lin_from_db(v, mt.exp2_tab);
                      if (x == __builtin_inf()) has_inf |= 1u << r; else add_w += x;
                      ++add_c;
                    }
                  }
                } else {
                  const size_t kh = base + hint_h - 1, kl = base + max(hint_l, 1) - 1;
                  const bool from0 = hint_l == 0;
                  add_w = (wh[kh] - (from0 ? 0.0 : wh[kl])) + (wl[kh] - (from0 ? 0.0 : wl[kl]));
                  add_c = wn[kh] - (from0 ? 0 : wn[kl]);
                }
              }
            }
#pragma unroll
            for (int k = 0; k < kStageRows; ++k) {
              sum[k] += r == k ? add_w : 0.0;
              cnt[k] += r == k ? add_c : 0;
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kStageRows; ++r) {
      if (r >= nrows || !in_row) continue;
      T out = epa::M<T>::nan();
      if (((feas >> r) & 1u) && cnt[r] > 0) {
        const double tot = ((has_inf >> r) & 1u) ? __builtin_inf() : sum[r];
        out = (T)(10.0 * epa::fast_log10(tot / (double)cnt[r], mt.log_tab));
      }
      const size_t at = ((size_t)(c * a.P + p0 + r)) * a.S + s;
      if (a.pooled) a.pooled[at] = out;
      if (a.mask) a.mask[at] = (a.sv[at] - out > a.thr) ? 1 : 0;
      __builtin_amdgcn_sched_barrier(0);  // (one row's logarithm at a time: eight side by side cost 40 registers)
    }
  }
}

// ---- the same, for the groups every neighbour of which can be staged (all of them, in a file without +inf Sv) -------
// The kernel above spends its time ISSUING instructions, not waiting for memory (rocprofv3 counters, 4 x 20 000 x 2000:
// VALU busy 62 % + scalar 32 % + LDS 7 % of the SIMDs' issue slots at three wavefronts each; ~415 vector and ~220 scalar
// instructions per wavefront and neighbour, 148 scalar registers spilled to vector lanes): the three ways a neighbour
// can be read, the hints of the global-memory route and the +inf bookkeeping all live in one loop.  This kernel keeps
// only the staged route -- a group with a neighbour it cannot stage (a span beyond the LDS copy, a row holding a +inf
// Sv, a ping window beyond kSpanMax) is flagged in ``todo`` and left to the kernel above -- and guesses window positions
// in float32 (they are confirmed on the staged values of type T either way).  Per neighbour and wavefront it issues
// ~150 vector + ~115 scalar instructions when the pings of a group share their range values (one interval per
// neighbour) and is still issue-bound: four wavefronts per SIMD, 29 % of each wavefront's cycles issuing.
// 4 x 100 000 x 2000, the range vector changing every 2000 pings / at every ping: 57 / 76 ms float64 and 45 / 63 ms
// float32 (91 / 137 and 80 / 118 ms with the kernel above alone); L2 misses 101 M instead of 392 M per 0.16 G samples
// with the rotated neighbour order (profiles/r05_pool_value_lean_ab.txt).
__device__ __forceinline__ bool same_bits(double x, double y) { return __double_as_longlong(x) == __double_as_longlong(y); }
__device__ __forceinline__ bool same_bits(float x, float y) { return __float_as_int(x) == __float_as_int(y); }

#ifndef EPA_LEAN_WGS  // (development knob) workgroups per CU the register budget is set for
#define EPA_LEAN_WGS 4
#endif
#ifndef EPA_LEAN_ROWS  // (development knob) pings per group: a multiple of kStageRows
#define EPA_LEAN_ROWS 8
#endif
constexpr int kLeanRows = EPA_LEAN_ROWS;
static_assert(kLeanRows % kStageRows == 0, "the general kernel takes over whole groups of its own");
template <typename T>
__global__ __launch_bounds__(kBlock, EPA_LEAN_WGS) void pool_value_mean_lean_kernel(PoolValueArgs<T> a, int C,
                                                                      const double* __restrict__ wh,
                                                                      const double* __restrict__ wl,
                                                                      const int* __restrict__ wn,
                                                                      const uint8_t* __restrict__ dirty,
                                                                      const int* __restrict__ differ, int nbands,
                                                                      int xcd_map, uint8_t* __restrict__ todo,
                                                                      const uint8_t* __restrict__ elig) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  typedef double dd_t __attribute__((ext_vector_type(2)));
  // seg_r[1 + i] = range[kmin + i], -inf before and +inf behind; seg_w[i] = {Wh, Wl}[kmin - 1 + i], seg_n[i] the count
  // (0 before the row)
  __shared__ T seg_r[kStageCap + 3];
  __shared__ __attribute__((aligned(16))) dd_t seg_w[kStageCap + 1];
  __shared__ int seg_n[kStageCap + 1];
  __shared__ double red_lo[4], red_hi[4];
  __shared__ int kspan[2 * kSpanMax];
  __shared__ float kinv[kSpanMax], kfirst[kSpanMax];  // per neighbour: samples per metre of its span, its first value - gmin
  epa::MathTabs mt{};
  bool have_tabs = false;  // (built by the first group that is this kernel's: a channel whose pings share one range vector
                           //  sends its workgroups through here with nothing to do)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups_per_channel = (a.P + kLeanRows - 1) / kLeanRows;
  const long long ngroups = (long long)C * groups_per_channel;
  const long long nwork = ngroups * nbands;
  const T inf = (T)__builtin_inf();
  for (long long w0 = blockIdx.x; w0 < nwork; w0 += gridDim.x) {
    // (the workgroups an XCD runs side by side take consecutive groups of one band of columns: neighbouring groups
    //  share all but kLeanRows of their neighbour rows, and every XCD has its own L2)
    const long long w = (xcd_map && nwork == (long long)gridDim.x) ? (long long)epa::xcd_contiguous((int)w0, (int)nwork) : w0;
    const int band = (int)(w / ngroups);
    const long long grp = w - (long long)band * ngroups;
    const long long c = grp / groups_per_channel;
    if (!differ[c]) continue;  // every ping of the channel has the same range vector: value_slide_kernel
    const int p0 = (int)(grp - c * groups_per_channel) * kLeanRows;
    const int nrows = min(kLeanRows, a.P - p0);
    if (group_all_eligible(elig, c * a.P + p0, nrows)) continue;  // (uniform) the sliding route has these pings
    if (!have_tabs) {
      mt = epa::build_math_tabs(tabs);
      have_tabs = true;  // (the barriers below come before the tables' first use)
    }
    const int s = band * kBlock + threadIdx.x;
    const bool in_row = s < a.S;
    T d[kLeanRows];
    unsigned feas = 0, dfeas = 0;  // dfeas: feasible but for the ping window (its interval may be SHARED with a later ping)
    double vmin = __builtin_inf(), vmax = -__builtin_inf();
    {
      const T* col = a.range + ((size_t)(c * a.P + p0)) * a.S + min(s, a.S - 1);
#pragma unroll
      for (int r = 0; r < kLeanRows; ++r) d[r] = col[(size_t)min(r, nrows - 1) * a.S];
    }
#pragma unroll
    for (int r = 0; r < kLeanRows; ++r) {
      if (!(r < nrows && in_row)) d[r] = epa::M<T>::nan();
      if (r < nrows && in_row) {
        if (pool_depth_feasible(a, d[r])) dfeas |= 1u << r;
        if (pool_feasible(a, d[r], p0 + r)) {
          feas |= 1u << r;
          vmin = fmin(vmin, (double)(d[r] - a.bin));
          vmax = fmax(vmax, (double)(d[r] + a.bin));
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = fmin(vmin, __shfl_down(vmin, o, 64));
      vmax = fmax(vmax, __shfl_down(vmax, o, 64));
    }
    __syncthreads();  // (the previous group is done with red / kspan / the staged span)
    if (lane == 0) {
      red_lo[wave] = vmin;
      red_hi[wave] = vmax;
    }
    __syncthreads();
    const T gmin = (T)fmin(fmin(red_lo[0], red_lo[1]), fmin(red_lo[2], red_lo[3]));
    const T gmax = (T)fmax(fmax(red_hi[0], red_hi[1]), fmax(red_hi[2], red_hi[3]));
    double sum[kLeanRows];
    int cnt[kLeanRows];
#pragma unroll
    for (int r = 0; r < kLeanRows; ++r) {
      sum[r] = 0.0;
      cnt[r] = 0;
    }
    bool bail = false;  // (uniform)
    if (gmin <= gmax) {  // (uniform) some window of the group is feasible
      const int q_first = max(p0 - a.n, 0), q_last = min(p0 + nrows - 1 + a.n, a.P - 1);
      const int nq = q_last - q_first + 1;
      int bad = nq > kSpanMax ? 1 : 0;
      if (!bad) {  // the spans of all neighbours: one binary search per lane, side by side
        for (int i = threadIdx.x; i < 2 * nq; i += kBlock) {
          const size_t row = (size_t)(c * a.P + q_first) + (i >> 1);
          const T* r2 = a.range + row * a.S;
          if ((i & 1) && dirty[row] != 0) bad = 1;
          kspan[i] = (i & 1) ? bound<T, true>(r2, a.nvalid[row], gmax) : bound<T, false>(r2, a.nvalid[row], gmin);
        }
        __syncthreads();
        // range rows are (nearly always) affine in the sample index: the position of a value inside a neighbour's span
        // is guessed from the span's two ends -- its slope and first value, worked out here for all neighbours at once
        for (int i = threadIdx.x; i < nq; i += kBlock) {
          const int kmin = kspan[2 * i], len = kspan[2 * i + 1] - kmin;
          bad |= len > kStageCap ? 1 : 0;
          const T* r2 = a.range + ((size_t)(c * a.P + q_first) + i) * a.S;
          const T first = len > 0 ? r2[kmin] : (T)0, last = len > 0 ? r2[kmin + len - 1] : (T)0;
          kinv[i] = (len > 1 && last > first) ? (float)(len - 1) / (float)(last - first) : 0.0f;
          kfirst[i] = (float)(first - gmin);
        }
      }
      bail = __syncthreads_or(bad) != 0;
      if (!bail) {
        // (two slots of 256 entries travel through registers ahead of time; the third -- spans beyond 511 samples, a
        //  window of more than +-125 samples -- is read when the span is stored: its registers cost the usual case
        //  a spill in every trip of the loop)
        constexpr int kAhead = 2;
        T pre_r[kAhead];
        double pre_h[kAhead], pre_l[kAhead];
        int pre_n[kAhead];
        // Requests without predicates (no branch round each): positions are clamped into the span, a slot of 256
        // entries that lies wholly beyond it is skipped by a uniform branch, and what a clamped request brought is
        // replaced when the registers go to LDS.  Uniform row pointers (stepped from neighbour to neighbour) + a
        // 32-bit lane offset per request.
        const size_t base0 = (size_t)(c * a.P + q_first) * a.S;
        const T* rb = a.range + base0;
        const double *hb = wh + base0, *lb = wl + base0;
        const int* nb = wn + base0;
        int kmin_n = 0, len_n = 0;  // (uniform) the span in the registers
        auto fetch = [&](int qi) {
          kmin_n = __builtin_amdgcn_readfirstlane(kspan[2 * qi]);
          len_n = __builtin_amdgcn_readfirstlane(kspan[2 * qi + 1]) - kmin_n;
#pragma unroll
          for (int u = 0; u < kAhead; ++u) {
            if (u * kBlock > len_n) continue;  // (uniform)
            const int i = threadIdx.x + u * kBlock;
            // (an empty span at the very end of a row: kmin = S -- the request stays inside the row)
            const unsigned kr = (unsigned)min(kmin_n + min(i, max(len_n - 1, 0)), a.S - 1);  // range[kmin + i], i < len
            const unsigned kw = (unsigned)max(kmin_n - 1 + min(i, len_n), 0);    // W[kmin - 1 + i], i <= len
            pre_r[u] = rb[kr];
            pre_h[u] = hb[kw];
            pre_l[u] = lb[kw];
            pre_n[u] = nb[kw];
          }
        };
        auto step_rows = [&](long long rows) {  // (uniform) the row pointers, `rows` rows on
          const long long o = rows * a.S;
          rb += o; hb += o; lb += o; nb += o;
        };
        auto store = [&](int kmin, int len, int qi_cur) {
          if (threadIdx.x == 0) seg_r[0] = -inf;
#pragma unroll
          for (int u = 0; u < kAhead; ++u) {
            if (u * kBlock > len) continue;  // (uniform)
            const int i = threadIdx.x + u * kBlock;
            if (i <= len) {
              seg_r[1 + i] = i == len ? inf : pre_r[u];  // (+inf: the sentinel behind the span)
              seg_w[i] = dd_t{pre_h[u], pre_l[u]};
              seg_n[i] = pre_n[u];
            }
          }
          if (len >= kAhead * kBlock) {  // (uniform, rare) the third slot, straight from this neighbour's row
            const int i = threadIdx.x + kAhead * kBlock;
            if (i <= len) {
              const size_t cur = base0 + (size_t)qi_cur * a.S;
              const unsigned kr = (unsigned)(kmin + min(i, len - 1)), kw = (unsigned)(kmin - 1 + i);
              seg_r[1 + i] = i == len ? inf : a.range[cur + kr];
              seg_w[i] = dd_t{wh[cur + kw], wl[cur + kw]};
              seg_n[i] = wn[cur + kw];
            }
          }
          if (kmin == 0 && threadIdx.x == 0) {  // (uniform, rare) the span starts the row: entry 0 = W[-1] = 0
            seg_w[0] = dd_t{0.0, 0.0};          // (the thread that wrote entry 0 above)
            seg_n[0] = 0;
          }
        };
        // Neighbours are taken in ROTATED order: at trip t the row of the window whose ping index is congruent to t
        // modulo the window length.  The groups an XCD runs side by side are consecutive and start together; taken
        // from the first row on, a row two groups share would be asked for 8 trips apart -- with 128 workgroups'
        // requests in between, 10 MB against the XCD's 4 MB of L2 (measured: 75 % of the requests missed, 226 GB over
        // the fabric per 0.8 G samples).  Rotated, the groups that share a row ask for it in the same trip.
        const int qi0 = (nq - q_first % nq) % nq;
        step_rows(qi0);
        fetch(qi0);
        const float binf = (float)a.bin;
        // Pings of the group with the same range value at a column -- all of them, while the recorded sound speed
        // holds -- share the interval in a neighbour's row and its sum.  Which pings differ from the one before them
        // in ANY lane of the wavefront is settled here, once: the loop over neighbours tests scalar bits.
        unsigned chg = 0;
#pragma unroll
        for (int r = 1; r < kLeanRows; ++r)
          chg |= __builtin_amdgcn_ballot_w64(!same_bits(d[r], d[r - 1])) != 0 ? 1u << r : 0u;
        int qi = qi0;
#pragma unroll 1
        for (int trip = 0; trip < nq; ++trip) {
          const int kmin = kmin_n, len = len_n;
          __syncthreads();  // the previous neighbour's span has been consumed
          store(kmin, len, qi);
          __syncthreads();
          const int qi_next = qi + 1 < nq ? qi + 1 : 0;
          if (trip + 1 < nq) {
            step_rows(qi_next ? 1 : 1 - nq);
            fetch(qi_next);  // (in flight while this neighbour is summed)
          }
          // the group's pings that have q inside their ping window [p - n, min(p + n, P - 1)], as a mask; the pings
          // whose interval has to be resolved: the first of them and those with a new range value
          const int q = q_first + qi;
          const int r_lo = max(q - a.n - p0, 0), r_hi = min(q + a.n - p0, nrows - 1);
          const unsigned rmask = r_hi >= r_lo ? (~0u >> (31 - r_hi)) & (~0u << r_lo) : 0u;
          const unsigned lmask = rmask & (chg | (1u << r_lo));
          const float firstf = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kfirst[qi])));
          const float invf = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, kinv[qi])));
          const float reachf = binf * invf;
          // the interval [l, h) of q's row inside [dv - bin, dv + bin] and its sum / count.  The position of a value is
          // guessed (in float32, from dvf = the value's offset from the group's lowest window edge) and confirmed on the two staged values it must lie between; anything else is searched.
          auto resolve = [&](T dv, float dvf, bool live, double& w_out, int& c_out) {
            const T lo_v = dv - a.bin, hi_v = dv + a.bin;
            const float x = (dvf - firstf) * invf;
            // first index with range >= lo_v / > hi_v, as positions 0 .. len inside the span
            int gl = (int)__builtin_ceilf(x - reachf), gh = (int)__builtin_floorf(x + reachf) + 1;
            gl = min(max(gl, 0), len);
            gh = min(max(gh, 0), len);
            const T a1 = seg_r[gl], a2 = seg_r[gl + 1];
            const T b1 = seg_r[gh], b2 = seg_r[gh + 1];
            // (& not &&: all four values are requested before the first comparison)
            const bool ok_l = (a1 < lo_v) & (lo_v <= a2), ok_h = (b1 <= hi_v) & (hi_v < b2);
            int l = gl, h = gh;
            if (__builtin_amdgcn_ballot_w64(live & !(ok_l & ok_h)) != 0) {  // (uniform, rare) off by one, a row not affine here
              if (live && !ok_l) l = bound<T, false>(seg_r + 1, len, lo_v);
              if (live && !ok_h) h = bound<T, true>(seg_r + 1, len, hi_v);
            }
            // W[hi-1] - W[lo-1]: the high parts subtract exactly when they are close (a window far down a row whose
            // total is 1e14 times its own), the low parts carry what the running sum had rounded away.  (h >= l in
            // every lane that counts: an empty interval subtracts an entry from itself.)
            const dd_t w_h = seg_w[h], w_l = seg_w[l];
            w_out = (w_h.x - w_l.x) + (w_h.y - w_l.y);
            c_out = seg_n[h] - seg_n[l];
          };
          double same_w = 0.0;
          int same_c = 0;
#pragma unroll
          for (int r = 0; r < kLeanRows; ++r) {
            if ((lmask >> r) & 1u)  // (uniform)
              resolve(d[r], (float)(d[r] - gmin), (dfeas >> r) & 1u, same_w, same_c);  // (dfeas: a later ping may share the interval)
            if ((rmask >> r) & 1u) {  // (uniform)
              sum[r] += same_w;  // (a plain sum of the -- non-negative -- window sums)
              cnt[r] += same_c;
            }
          }
          qi = qi_next;
        }
      }
    }
    if (bail) {  // (uniform) the general kernel takes this group: flags per group of ITS size, band-major
      const int gpc8 = (a.P + kStageRows - 1) / kStageRows;
      if (threadIdx.x < kLeanRows / kStageRows && p0 + (int)threadIdx.x * kStageRows < a.P)
        todo[((long long)band * C + c) * gpc8 + p0 / kStageRows + threadIdx.x] = 1;
      continue;
    }
#pragma unroll
    for (int r = 0; r < kLeanRows; ++r) {
      if (r >= nrows || !in_row) continue;
      T out = epa::M<T>::nan();
      if (((feas >> r) & 1u) && cnt[r] > 0) out = (T)(10.0 * epa::fast_log10(sum[r] / (double)cnt[r], mt.log_tab));
      const size_t at = ((size_t)(c * a.P + p0 + r)) * a.S + s;
      if (a.pooled) a.pooled[at] = out;
      if (a.mask) a.mask[at] = (a.sv[at] - out > a.thr) ? 1 : 0;
      __builtin_amdgcn_sched_barrier(0);  // (one row's logarithm at a time: eight side by side cost 40 registers)
    }
  }
}

// ---- channels whose pings all share one range vector (the usual echo_range) ---------------------------------
// Then the index interval [lo, hi) of a depth window depends on the sample only, a neighbour ping contributes
// R[q][s] = W_q[hi-1] - W_q[lo-1] whatever the ping it is needed for, and the window sum over pings is a sliding
// sum of R down the column (double-double as in box_ping_slide_kernel): ~140 B of HBM traffic per sample instead
// of ~500 cached loads.  Rows may be NaN-padded to different lengths (intervals are clipped to each row's length).
__global__ __launch_bounds__(kBlock) void ref_row_kernel(const int* __restrict__ nvalid, int P,
                                                         int* __restrict__ ref, int* __restrict__ differ) {
  __shared__ int best[kBlock], arg[kBlock];
  const int c = blockIdx.x;
  int b = -1, a = 0;
  for (int p = threadIdx.x; p < P; p += kBlock) {
    const int v = nvalid[(size_t)c * P + p];
    if (v > b) { b = v; a = p; }
  }
  best[threadIdx.x] = b; arg[threadIdx.x] = a;
  __syncthreads();
  for (int o = kBlock / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const int b2 = best[threadIdx.x + o], a2 = arg[threadIdx.x + o];
      if (b2 > best[threadIdx.x] || (b2 == best[threadIdx.x] && a2 < arg[threadIdx.x])) {
        best[threadIdx.x] = b2; arg[threadIdx.x] = a2;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { ref[c] = arg[0]; differ[c] = 0; }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void rows_same_kernel(const T* __restrict__ range,
                                                           const int* __restrict__ nvalid, long long rows, int P,
                                                           int S, const int* __restrict__ ref,
                                                           int* __restrict__ differ) {
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const long long c = row / P;
    // a channel already known to differ is not read again (and 400 000 atomics on one word would serialise)
    if (__atomic_load_n(&differ[c], __ATOMIC_RELAXED)) continue;
    const T* xr = range + (size_t)row * S;
    const T* rr = range + (size_t)(c * P + ref[c]) * S;
    const int nv = nvalid[row];
    int bad = 0;
    for (int k = threadIdx.x; k < nv; k += kBlock) bad |= xr[k] != rr[k];
    if (bad) __atomic_store_n(&differ[c], 1, __ATOMIC_RELAXED);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void value_intervals_kernel(PoolValueArgs<T> a, const int* __restrict__ ref,
                                                                 int* __restrict__ ilo, int* __restrict__ ihi) {
  const int s = blockIdx.x * kBlock + threadIdx.x, c = blockIdx.y;
  if (s >= a.S) return;
  const size_t rrow = (size_t)c * a.P + ref[c];
  const T* rr = a.range + rrow * a.S;
  const int nv = a.nvalid[rrow];
  int lo = -1, hi = -1;
  if (s < nv) {
    const T d = rr[s];
    if ((d - a.bin >= a.rmin) && (d + a.bin <= a.rmax) && (d - a.bin >= a.exclude_above)) {  // pool_feasible
      lo = bound<T, false>(rr, nv, d - a.bin);
      hi = bound<T, true>(rr, nv, d + a.bin);
    }
  }
  ilo[(size_t)c * a.S + s] = lo;
  ihi[(size_t)c * a.S + s] = hi;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void row_interval_sum_kernel(
    PoolValueArgs<T> a, long long rows, const double* __restrict__ wh, const double* __restrict__ wl,
    const int* __restrict__ wn, const uint8_t* __restrict__ dirty, const int* __restrict__ differ,
    const int* __restrict__ ilo, const int* __restrict__ ihi, double* __restrict__ rh,
    int* __restrict__ rn, const int* __restrict__ row_slot = nullptr) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  const int s = blockIdx.y * kBlock + threadIdx.x;
  if (s >= a.S) return;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    // row_slot (the run form): the interval table of the row's RUN inside a channel that differs; else the channel's
    long long c = row / a.P;
    if (row_slot) {
      if (!differ[c] || row_slot[row] < 0) continue;
      c = row_slot[row];
    } else if (differ[c]) {
      continue;
    }
    const int nv = a.nvalid[row];
    const int lo = min(ilo[(size_t)c * a.S + s], nv), hi = min(ihi[(size_t)c * a.S + s], nv);
    const size_t base = (size_t)row * a.S;
    Dd sum{0.0, 0.0};
    int cnt = 0;
    bool has_inf = false;
    if (lo >= 0 && hi > lo) {
      if (dirty[row]) {
        const T* vr = a.sv + base;
        for (int k = lo; k < hi; ++k) {
          const T v = vr[k];
          if (v == v) {
            const double x = (double)epa::lin_from_db(v, mt.exp2_tab);
            if (x == __builtin_inf()) has_inf = true; else sum.add(x);
            ++cnt;
          }
        }
      } else {
        sum.add(Dd{wh[base + hi - 1], wl[base + hi - 1]}, 1.0);
        cnt = wn[base + hi - 1];
        if (lo > 0) {
          sum.add(Dd{wh[base + lo - 1], wl[base + lo - 1]}, -1.0);
          cnt -= wn[base + lo - 1];
        }
      }
    }
    rh[base + s] = has_inf ? __builtin_inf() : sum.hi + sum.lo;
    rn[base + s] = cnt;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void value_slide_kernel(PoolValueArgs<T> a, const int* __restrict__ differ,
                                                             const int* __restrict__ ilo,
                                                             const double* __restrict__ rh,
                                                             const int* __restrict__ rn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  __syncthreads();
  const int s = blockIdx.y * kBlock + threadIdx.x, c = blockIdx.z;
  if (s >= a.S || differ[c]) return;
  const int P = a.P, S = a.S, n = a.n;
  const int p0 = blockIdx.x * kSlideSeg, p1 = min(P, p0 + kSlideSeg);
  const size_t cbase = (size_t)c * P * S;
  const bool depth_ok = ilo[(size_t)c * S + s] >= 0;
  const double* __restrict__ h = rh + cbase + s;
  const int* __restrict__ k = rn + cbase + s;
  DdSum w;
  if (depth_ok)
    for (int q = max(0, p0 - n); q <= min(P - 1, p0 + n); ++q) {
      const size_t r = (size_t)q * S;
      w.add(h[r], k[r], 1.0);
    }
  // four pings per trip: the rows entering and leaving at each of them (and the Sv compared with the result)
  // are requested together, then consumed in order
  constexpr int U = 4;
  for (int pb = p0; pb < p1; pb += U) {
    double hin[U], hout[U];
    int kin[U], kout[U], nvp[U];
    T x[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int p = min(pb + j, p1 - 1);
      const int in = p + n, out = p - n - 1;
      const bool gi = depth_ok && p > p0 && in <= P - 1, go = depth_ok && p > p0 && out >= 0;
      const size_t ri = (size_t)(gi ? in : 0) * S, ro = (size_t)(go ? out : 0) * S;
      hin[j] = gi ? h[ri] : 0.0;
      kin[j] = gi ? k[ri] : 0;
      hout[j] = go ? h[ro] : 0.0;
      kout[j] = go ? k[ro] : 0;
      nvp[j] = a.nvalid[(size_t)c * P + p];
      x[j] = a.mask ? a.sv[cbase + (size_t)p * S + s] : (T)0;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int p = pb + j;
      if (p >= p1) break;
      w.add(hin[j], kin[j], 1.0);
      w.add(hout[j], kout[j], -1.0);
      T res = epa::M<T>::nan();
      const bool ok = depth_ok && (p - n >= 0) && ((long long)p + n <= (long long)P) && s < nvp[j];
      if (ok && w.cnt > 0) res = (T)(10.0 * epa::fast_log10(w.value() / (double)w.cnt, mt.log_tab));
      const size_t at = cbase + (size_t)p * S + s;
      if (a.pooled) a.pooled[at] = res;
      if (a.mask) a.mask[at] = (x[j] - res > a.thr) ? 1 : 0;
    }
  }
}

// ---- runs of pings with one range vector inside a channel that has several (round 6) -----------------------------------
// An EK60 records the sound speed of the moment with every ping; an operator's setting holds for thousands of pings.
// The sliding route above then applies run by run: a ping whose whole window p - n .. p + n lies inside a run of pings
// that share a range vector pools exactly as in a channel with one vector.  Nothing here returns to the host:
//   run_flag      does ping p start a run?  (three samples of it against ping p - 1: the range is affine in the sample
//                 number -- a proposal; run_verify holds every row of a long run to the run's reference, value by value)
//   run_scan_*    start of the run of every ping (a max-scan of the flags: tiles of 1024 pings, then the tiles' carries)
//   run_stats     length of every run and its reference row (the longest; the first of those), by atomics at its start
//   run_verify    a row of a long run that differs from the reference on its own valid samples spoils the run
//   run_finish    per ping: eligible?  which interval table (slot) does its run use?  per slot: its reference row.
//                 Slots: run starts of long runs lie at least Lmin = 2n + 1 pings apart, so c * ceil(P / Lmin) +
//                 start / Lmin is a table index without collisions and without a compaction
// then value_intervals_kernel per slot, row_interval_sum_kernel with the row's slot, value_slide_runs_kernel; the lean
// and the staged kernel skip the groups whose pings are all eligible.
constexpr int kRunTile = 1024;

template <typename T>
__global__ __launch_bounds__(kBlock) void run_flag_kernel(const T* __restrict__ range, const int* __restrict__ nvalid,
                                                          const int* __restrict__ differ, long long rows, int P, int S,
                                                          uint8_t* __restrict__ flag) {
  for (long long row = (long long)blockIdx.x * kBlock + threadIdx.x; row < rows; row += (long long)gridDim.x * kBlock) {
    const long long c = row / P;
    const int p = (int)(row - c * P);
    uint8_t f = 1;
    if (differ[c] && p > 0) {
      const int m = min(nvalid[row], nvalid[row - 1]);
      const T* x = range + (size_t)row * S;
      const T* y = x - S;
      // (a row without a valid sample joins the run it lies in: nothing of it is ever pooled)
      f = (m > 0 && (x[0] != y[0] || x[m >> 1] != y[m >> 1] || x[m - 1] != y[m - 1])) ? 1 : 0;
    }
    flag[row] = f;
  }
}

// start[row] = the last ping <= p of the row's tile with a flag, or -1; tile_last[c * ntiles + t] likewise for the tile
__global__ __launch_bounds__(kBlock) void run_scan_tiles_kernel(const uint8_t* __restrict__ flag, int P, int ntiles,
                                                                int* __restrict__ start, int* __restrict__ tile_last) {
  __shared__ int wmax[kBlock / 64];
  const int c = blockIdx.y, t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = t * kRunTile + threadIdx.x * 4;  // four consecutive pings per lane
  int v[4];
  int run = -1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = p0 + j;
    if (p < P && flag[(size_t)c * P + p]) run = p;
    v[j] = run;
  }
  int incl = run;  // inclusive max-scan over the wavefront, then over the workgroup's four wavefronts
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl = max(incl, up);
  }
  if (lane == 63) wmax[wave] = incl;
  __syncthreads();
  int before = __shfl_up(incl, 1, 64);
  if (lane == 0) before = -1;
  for (int w = 0; w < wave; ++w) before = max(before, wmax[w]);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (p0 + j < P) start[(size_t)c * P + p0 + j] = max(v[j], before);
  if (threadIdx.x == kBlock - 1) tile_last[c * ntiles + t] = max(incl, before);
}

// tile_last -> the last flag BEFORE the tile (exclusive carry), in place; one lane per channel (a few thousand tiles)
__global__ void run_scan_carry_kernel(int* __restrict__ tile_last, int ntiles) {
  int* tl = tile_last + (size_t)blockIdx.x * ntiles;
  int carry = -1;
  for (int t = 0; t < ntiles; ++t) {
    const int v = tl[t];
    tl[t] = carry;
    carry = max(carry, v);
  }
}

// start[] completed with the carries; length of every run and the key of its reference row, at the run's first row
__global__ __launch_bounds__(kBlock) void run_stats_kernel(const int* __restrict__ nvalid, const int* __restrict__ differ,
                                                           const int* __restrict__ tile_carry, long long rows, int P,
                                                           int ntiles, int* __restrict__ start, int* __restrict__ run_len,
                                                           unsigned long long* __restrict__ run_key) {
  for (long long row = (long long)blockIdx.x * kBlock + threadIdx.x; row < rows; row += (long long)gridDim.x * kBlock) {
    const long long c = row / P;
    if (!differ[c]) continue;
    const int p = (int)(row - c * P);
    const int rs = max(start[row], tile_carry[c * ntiles + p / kRunTile]);  // (>= 0: ping 0 carries a flag)
    start[row] = rs;
    atomicAdd(&run_len[c * P + rs], 1);
    // the longest row; among those the first
    atomicMax(&run_key[c * P + rs], ((unsigned long long)(unsigned)nvalid[row] << 32) | (unsigned)(0x7fffffff - p));
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void run_verify_kernel(const T* __restrict__ range, const int* __restrict__ nvalid,
                                                            const int* __restrict__ differ, const int* __restrict__ start,
                                                            const int* __restrict__ run_len,
                                                            const unsigned long long* __restrict__ run_key, long long rows,
                                                            int P, int S, int lmin, uint8_t* __restrict__ run_bad) {
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const long long c = row / P;
    if (!differ[c]) continue;
    const long long r0 = c * P + start[row];
    if (run_len[r0] < lmin || __atomic_load_n(&run_bad[r0], __ATOMIC_RELAXED)) continue;  // (uniform)
    const int ref = 0x7fffffff - (int)(unsigned)(run_key[r0] & 0xffffffffull);
    if (c * P + ref == row) continue;
    const T* x = range + (size_t)row * S;
    const T* y = range + (size_t)(c * P + ref) * S;
    const int nv = nvalid[row];
    int bad = 0;
    for (int k = threadIdx.x; k < nv; k += kBlock) bad |= x[k] != y[k];
    if (bad) __atomic_store_n(&run_bad[r0], (uint8_t)1, __ATOMIC_RELAXED);
  }
}

__global__ __launch_bounds__(kBlock) void run_finish_kernel(const int* __restrict__ differ, const int* __restrict__ start,
                                                            const int* __restrict__ run_len,
                                                            const unsigned long long* __restrict__ run_key,
                                                            const uint8_t* __restrict__ run_bad, long long rows, int P, int n,
                                                            int lmin, int nslotc, uint8_t* __restrict__ elig,
                                                            int* __restrict__ row_slot, int* __restrict__ slot_ref) {
  for (long long row = (long long)blockIdx.x * kBlock + threadIdx.x; row < rows; row += (long long)gridDim.x * kBlock) {
    const long long c = row / P;
    uint8_t e = 0;
    int slot = -1;
    if (differ[c]) {
      const int p = (int)(row - c * P), rs = start[row];
      const long long r0 = c * P + rs;
      const int len = run_len[r0];
      if (len >= lmin && !run_bad[r0]) {
        slot = (int)(c * nslotc + rs / lmin);
        e = (p - n >= rs && p + n < rs + len) ? 1 : 0;
        if (p == rs) slot_ref[slot] = (int)(0x7fffffff - (int)(unsigned)(run_key[r0] & 0xffffffffull));
      }
    }
    elig[row] = e;
    row_slot[row] = slot;
  }
}

// need_w[row]: some ping within ``n`` (the caller's: side pings + a group of pings) of the row is not eligible -- the
// staged kernels, which read the running sums of their pings' neighbour rows, may ask for this one
__global__ __launch_bounds__(kBlock) void run_need_rows_kernel(const int* __restrict__ differ, const uint8_t* __restrict__ elig,
                                                               long long rows, int P, int n, uint8_t* __restrict__ need_w) {
  for (long long row = (long long)blockIdx.x * kBlock + threadIdx.x; row < rows; row += (long long)gridDim.x * kBlock) {
    const long long c = row / P;
    const int p = (int)(row - c * P);
    uint8_t need = 0;
    if (differ[c]) {
      const uint8_t* e = elig + c * P;
      for (int q = max(0, p - n); q <= min(P - 1, p + n) && !need; ++q) need = e[q] ? 0 : 1;
    }
    need_w[row] = need;
  }
}

// value_intervals_kernel per slot: ref row of the slot's run (a ping index of channel slot / nslotc), -1 = no run here
template <typename T>
__global__ __launch_bounds__(kBlock) void value_intervals_runs_kernel(PoolValueArgs<T> a, const int* __restrict__ slot_ref,
                                                                      int nslotc, int* __restrict__ ilo, int* __restrict__ ihi) {
  const int s = blockIdx.y * kBlock + threadIdx.x, slot = blockIdx.x;
  const int ref = slot_ref[slot];
  if (s >= a.S || ref < 0) return;
  const size_t rrow = (size_t)(slot / nslotc) * a.P + ref;
  const T* rr = a.range + rrow * a.S;
  const int nv = a.nvalid[rrow];
  int lo = -1, hi = -1;
  if (s < nv) {
    const T d = rr[s];
    if (pool_depth_feasible(a, d)) {
      lo = bound<T, false>(rr, nv, d - a.bin);
      hi = bound<T, true>(rr, nv, d + a.bin);
    }
  }
  ilo[(size_t)slot * a.S + s] = lo;
  ihi[(size_t)slot * a.S + s] = hi;
}

// value_slide_kernel for the eligible pings of the channels that differ: the window sum slides while consecutive pings
// are eligible in one run, and is taken afresh (2n + 1 rows) where eligibility resumes
template <typename T>
__global__ __launch_bounds__(kBlock) void value_slide_runs_kernel(PoolValueArgs<T> a, const int* __restrict__ differ,
                                                                  const uint8_t* __restrict__ elig,
                                                                  const int* __restrict__ row_slot,
                                                                  const int* __restrict__ ilo,
                                                                  const double* __restrict__ rh,
                                                                  const int* __restrict__ rn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int s = blockIdx.y * kBlock + threadIdx.x, c = blockIdx.z;
  if (!differ[c]) return;
  const int P = a.P, S = a.S, n = a.n;
  const int p0 = blockIdx.x * kSlideSeg, p1 = min(P, p0 + kSlideSeg);
  const uint8_t* __restrict__ el = elig + (size_t)c * P;
  {  // (uniform) a segment without an eligible ping leaves before the tables
    bool any = false;
    for (int p = p0; p < p1 && !any; ++p) any = el[p] != 0;
    if (!any) return;
  }
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  __syncthreads();
  if (s >= S) return;
  const size_t cbase = (size_t)c * P * S;
  const double* __restrict__ h = rh + cbase + s;
  const int* __restrict__ k = rn + cbase + s;
  const int* __restrict__ slots = row_slot + (size_t)c * P;
  DdSum w;
  int cur_slot = -1;   // (uniform) slot of the run the window sum belongs to; -1: no window sum
  bool depth_ok = false;
  constexpr int U = 4;
  for (int pb = p0; pb < p1; pb += U) {
    double hin[U], hout[U];
    int kin[U], kout[U], nvp[U], sl[U];
    bool cont[U];
    T x[U];
    int prev_slot = cur_slot;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int p = min(pb + j, p1 - 1);
      sl[j] = (pb + j < p1 && el[p]) ? slots[p] : -1;
      cont[j] = sl[j] >= 0 && sl[j] == prev_slot;  // slides on from the ping before (same run, both eligible)
      prev_slot = sl[j];
      const bool g = cont[j] && depth_ok;  // (depth_ok: of the current run; a fresh window re-reads it below)
      const size_t ri = (size_t)(g ? p + n : 0) * S, ro = (size_t)(g ? p - n - 1 : 0) * S;
      hin[j] = g ? h[ri] : 0.0;
      kin[j] = g ? k[ri] : 0;
      hout[j] = g ? h[ro] : 0.0;
      kout[j] = g ? k[ro] : 0;
      nvp[j] = a.nvalid[(size_t)c * P + p];
      x[j] = a.mask ? a.sv[cbase + (size_t)p * S + s] : (T)0;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int p = pb + j;
      if (p >= p1) break;
      if (sl[j] < 0) {  // (uniform) not this kernel's ping
        cur_slot = -1;
        continue;
      }
      if (!cont[j] || sl[j] != cur_slot) {  // (uniform) eligibility resumes, or another run: the window afresh
        cur_slot = sl[j];
        depth_ok = ilo[(size_t)cur_slot * S + s] >= 0;
        w = DdSum();
        if (depth_ok)
          for (int q = p - n; q <= p + n; ++q) w.add(h[(size_t)q * S], k[(size_t)q * S], 1.0);
      } else if (depth_ok) {
        // (the requests above were made with the depth_ok of the run as it stood at the top of the batch: a run that
        //  began inside this batch reads its rows here)
        const bool pre = hin[j] != 0.0 || kin[j] != 0 || hout[j] != 0.0 || kout[j] != 0;
        if (pre) {
          w.add(hin[j], kin[j], 1.0);
          w.add(hout[j], kout[j], -1.0);
        } else {
          w.add(h[(size_t)(p + n) * S], k[(size_t)(p + n) * S], 1.0);
          w.add(h[(size_t)(p - n - 1) * S], k[(size_t)(p - n - 1) * S], -1.0);
        }
      }
      T res = epa::M<T>::nan();
      const bool ok = depth_ok && s < nvp[j];  // (an eligible ping has its whole ping window inside the file)
      if (ok && w.cnt > 0) res = (T)(10.0 * epa::fast_log10(w.value() / (double)w.cnt, mt.log_tab));
      const size_t at = cbase + (size_t)p * S + s;
      if (a.pooled) a.pooled[at] = res;
      if (a.mask) a.mask[at] = (x[j] - res > a.thr) ? 1 : 0;
    }
  }
}

// nanmedian: one workgroup per output sample
constexpr int kMaxSidePings = 512;

template <typename T>
__global__ __launch_bounds__(kBlock) void pool_value_median_kernel(PoolValueArgs<T> a, long long jobs) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  __shared__ SelectScratch sc;
  __shared__ int wlo[2 * kMaxSidePings + 1], whi[2 * kMaxSidePings + 1];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  for (long long job = blockIdx.x; job < jobs; job += gridDim.x) {
    const long long row = job / a.S;
    const int p = (int)(row % a.P);
    const long long c = row / a.P;
    const T d = a.range[job];
    T out = epa::M<T>::nan();
    if (pool_feasible(a, d, p)) {
      const T lo_v = d - a.bin, hi_v = d + a.bin;
      const int q0 = p - a.n, nq = min(p + a.n, a.P - 1) - q0 + 1;
      __syncthreads();
      for (int j = threadIdx.x; j < nq; j += kBlock) {
        const size_t qrow = (size_t)(c * a.P + q0 + j);
        const T* rr = a.range + qrow * a.S;
        const int nv = a.nvalid[qrow];
        wlo[j] = bound<T, false>(rr, nv, lo_v);
        whi[j] = bound<T, true>(rr, nv, hi_v);
      }
      __syncthreads();
      RaggedWindow<T> w{a.sv + (size_t)c * a.P * a.S, a.S, q0, nq, wlo, whi};
      unsigned nv;
      const double med = window_median_lin(w, &sc, mt.exp2_tab, nv);
      if (nv) out = (T)(10.0 * epa::fast_log10(med, mt.log_tab));
    }
    if (threadIdx.x == 0) {
      if (a.pooled) a.pooled[job] = out;
      if (a.mask) a.mask[job] = (a.sv[job] - out > a.thr) ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// nanmedian pooling with the window carried from ping to ping (see the comment above MedMap)
// ------------------------------------------------------------------------------------------------
constexpr unsigned kMedNoBin = 0xffffu;  // bin of a NaN (and of the padding of the bin ring)
constexpr int kMedSel = 64;              // candidates the finishing wavefront ranks: one per lane
constexpr int kMedValueCap = 12288;      // window elements a value-window job may carry (its bins: 24 KB of LDS)

// Two 16-bit bins per word: non-zero iff one of them lies in [b1, b1 + span] -- packed subtract (wraps below b1),
// then saturating subtract from span + 1.  2 VALU instructions per 2 bins.
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned halves_in_span(unsigned x, unsigned b1b1, unsigned span1) {
  const ushort2_t d = __builtin_bit_cast(ushort2_t, x) - __builtin_bit_cast(ushort2_t, b1b1);
  const ushort2_t r = __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2_t, span1), d);
  return __builtin_bit_cast(unsigned, r);
}

template <typename T>
struct MedSlideArgs {
  const T* sv;
  int P, S, n, nseg;
  long long jobs;
  T thr;
  T* pooled;
  uint8_t* mask;
  int cap;  // window elements the bin ring holds
  // index windows (epa_pool_sv): first pooled sample, side samples
  int s0, m;
  // value windows (epa_pool_sv_value): per (channel, sample) intervals of the channels whose pings share one range
  // vector (differ[c] == 0); the others -- and windows wider than a workgroup or the ring -- take every window from memory
  PoolValueArgs<T> pv;
  const int* ilo;
  const int* ihi;
  const int* differ;
};

// the window of a value-window job whose channel has one range vector: samples [lo, hi) of every ping it has
template <typename T>
struct SameRowsWindow {
  const T* chan;
  const int* nvalid;  // of the channel
  int S, P, q_lo, nq, lo, hi;
  template <typename F>
  __device__ __forceinline__ void for_each(F f) const {
    for (int j = 0; j < nq; ++j) {
      const int q = q_lo + j;
      if (q < 0 || q >= P) continue;
      const int h = min(hi, nvalid[q]);
      const T* row = chan + (size_t)q * S;
      for (int k = lo + (int)threadIdx.x; k < h; k += kBlock) f((double)row[k]);
    }
  }
};

// What the workgroup keeps in LDS is the BIN of every window value (2 bytes), not the value: a leaving row is
// un-counted from its bins, the members of the median's bin are found by comparing bins (4 per lane and read), and
// only those few values are read again from memory (L2) -- by wavefront 3, one step later, while wavefronts 0-2
// already search the next window: it ranks the candidates (one per lane, the others' keys by readlane), converts
// and writes the result of the previous ping.  38 KB of LDS per workgroup: 4 workgroups per CU hide each other's
// barrier and LDS latencies.
// BY_VALUE == false: the (2n+1) x (2m+1) index window with reflected borders (pool_sv).
// BY_VALUE == true : pings p-n .. p+n that exist, samples [ilo, ihi) of the channel's range vector (pool_Sv);
//                    NaN where the reference declines to pool.
template <typename T, bool BY_VALUE>
__global__ __launch_bounds__(kBlock) void pool_median_slide_kernel(MedSlideArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ SelectScratchT<kMedCap> sc;  // the radix selection's scratch (flat fields, windows taken from memory)
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  unsigned* fine = reinterpret_cast<unsigned*>(smem + epa::kMathTabBytes);
  unsigned* coarse = fine + kMedBins;
  unsigned* ncand = coarse + kMedCoarse;                      // [2] (+ 2 words of padding), by parity of the ping
  unsigned* meta = ncand + 4;                                 // [2][4]: state, r1, dk, b1
  unsigned short* cidx = reinterpret_cast<unsigned short*>(meta + 8);  // [2][kMedSel] window elements of the candidates
  unsigned short* bins = cidx + 2 * kMedSel;                  // [8 * npack], 16-byte aligned
  const int tid = threadIdx.x, lane = tid & 63;
  const bool finisher = tid >= kBlock - 64;
  const int P = a.P, S = a.S, n = a.n, R = 2 * n + 1;
  for (long long job = blockIdx.x; job < a.jobs; job += gridDim.x) {
    const int s = (int)(job % S);
    const long long t = job / S;
    const int seg = (int)(t % a.nseg);
    const long long c = t / a.nseg;
    const size_t cbase = (size_t)c * P * S;
    const int p0 = seg * kMedSeg, p1 = min(P, p0 + kMedSeg);
    const T* __restrict__ chan = a.sv + cbase;
    const int* __restrict__ nvalid = BY_VALUE ? a.pv.nvalid + (size_t)c * P : nullptr;
    int w, c_lo = 0, c_hi = 0;  // window columns; value windows: the samples [c_lo, c_hi)
    bool trivial, carried = true;
    if (BY_VALUE) {
      carried = a.differ[c] == 0;
      c_lo = carried ? a.ilo[(size_t)c * S + s] : 0;
      c_hi = carried ? a.ihi[(size_t)c * S + s] : 0;
      w = c_hi - c_lo;
      trivial = carried && (c_lo < 0 || w <= 0);  // the depth is never pooled (or past the vector's end)
      carried = carried && !trivial && w <= kBlock && (long long)R * w <= a.cap;
    } else {
      w = 2 * a.m + 1;
      trivial = s < a.s0;  // above the first pooled sample
    }
    if (trivial) {  // NaN, never masked
      for (int p = p0 + tid; p < p1; p += kBlock) {
        const size_t at = cbase + (size_t)p * S + s;
        if (a.pooled) a.pooled[at] = epa::M<T>::nan();
        if (a.mask) a.mask[at] = 0;
      }
      continue;
    }
    if (BY_VALUE && !carried) {  // every window of the segment from memory (the round-2 kernel's step)
      int* wlo = reinterpret_cast<int*>(bins);
      int* whi = wlo + R;
      for (int p = p0; p < p1; ++p) {
        const size_t at = cbase + (size_t)p * S + s;
        const T d = a.pv.range[at];
        T out = epa::M<T>::nan();
        if (pool_feasible(a.pv, d, p)) {
          const T lo_v = d - a.pv.bin, hi_v = d + a.pv.bin;
          const int q0 = p - n, nq = min(p + n, P - 1) - q0 + 1;
          __syncthreads();
          for (int j = tid; j < nq; j += kBlock) {
            const T* rr = a.pv.range + cbase + (size_t)(q0 + j) * S;
            const int nv = nvalid[q0 + j];
            wlo[j] = bound<T, false>(rr, nv, lo_v);
            whi[j] = bound<T, true>(rr, nv, hi_v);
          }
          __syncthreads();
          RaggedWindow<T> rw{chan, S, q0, nq, wlo, whi};
          unsigned nv;
          const double med = window_median_lin(rw, &sc, mt.exp2_tab, nv);
          if (nv) out = (T)(10.0 * epa::fast_log10(med, mt.log_tab));
        }
        if (tid == 0) {
          if (a.pooled) a.pooled[at] = out;
          if (a.mask) a.mask[at] = (chan[(size_t)p * S + s] - out > a.thr) ? 1 : 0;
        }
      }
      continue;
    }
    const int W = R * w, npack = (W + 7) >> 3;  // bins are searched eight per 128-bit read
    const float inv_w = 1.0f / (float)w;
    // column i of the window / row of virtual ping q: where they live in memory (nullptr: not part of the window)
    auto column = [&](int i) -> int {
      return BY_VALUE ? c_lo + i : a.s0 + reflect_index(s - a.m + i - a.s0, S - a.s0);
    };
    auto value_at = [&](int q, int col) -> T {
      if (BY_VALUE) {
        if (q < 0 || q >= P || col >= nvalid[q]) return epa::M<T>::nan();
        return chan[(size_t)q * S + col];
      }
      return chan[(size_t)reflect_index(q, P) * S + col];
    };
    // does the reference pool at ping p?  (value windows: clean/utils.py:77-83; the sample must exist in ping p)
    auto pooled_at = [&](int p) -> bool {
      return !BY_VALUE || (p - n >= 0 && (long long)p + n <= (long long)P && s < nvalid[p]);
    };
    MedMap map{16.0f, 4096.0f};
    auto enter = [&](int e, T v) {  // value v becomes element e of the window
      unsigned b = kMedNoBin;
      if (v == v) {
        b = map.bin((double)v);
        atomicAdd(&fine[b], 1u);
        atomicAdd(&coarse[b >> 6], 1u);
      }
      bins[e] = (unsigned short)b;
    };
    // element e of the window of ping p (ring slot rs holds its oldest row p - n)
    auto element = [&](int e, int p, int rs) -> T {
      int slot = (int)(((float)e + 0.5f) * inv_w);
      if (slot * w > e) --slot;
      else if ((slot + 1) * w <= e) ++slot;
      int k = slot - rs;
      if (k < 0) k += R;
      return value_at(p - n + k, column(e - slot * w));
    };
    // bins (of rank k1 = lower middle, k1 + dk = upper middle) from the two-level histogram; false: no valid value
    auto middle_bins = [&](unsigned& b1, unsigned& b2, unsigned& r1, unsigned& dk) -> bool {
      const unsigned cv = coarse[lane];
      const unsigned ci = wave_scan_incl(cv);
      const unsigned N = (unsigned)__builtin_amdgcn_readlane((int)ci, 63);
      if (!N) return false;
      const unsigned k1 = (N - 1u) >> 1;
      dk = (N & 1u) ? 0u : 1u;
      unsigned rem1, rem2, r2;
      const unsigned cb1 = wave_rank_lane(ci, cv, k1, rem1);
      const unsigned cb2 = wave_rank_lane(ci, cv, k1 + dk, rem2);
      const unsigned fv = fine[cb1 * 64 + lane];
      const unsigned fi = wave_scan_incl(fv);
      b1 = cb1 * 64 + wave_rank_lane(fi, fv, rem1, r1);
      if (cb2 == cb1) {
        b2 = cb1 * 64 + wave_rank_lane(fi, fv, rem2, r2);
      } else {
        const unsigned gv = fine[cb2 * 64 + lane];
        const unsigned gi = wave_scan_incl(gv);
        b2 = cb2 * 64 + wave_rank_lane(gi, gv, rem2, r2);
      }
      return true;
    };
    // histogram and bins of the window of ping p from memory (called by all threads; nobody reads either meanwhile)
    auto rebuild = [&](int p, int rs) {
      for (int i = tid; i < kMedBins + kMedCoarse; i += kBlock) fine[i] = 0u;
      __syncthreads();
      for (int e = tid; e < W; e += kBlock) enter(e, element(e, p, rs));
      __syncthreads();
    };
    // 1/128-dB bins centred on the value of bin b of the current map
    auto recentre = [&](unsigned b) {
      const float centre = ((float)b + 0.5f - map.off) / map.scale;
      map.scale = 128.0f;
      map.off = (float)(kMedBins / 2) - centre * 128.0f;
    };
    // wavefront 3: the median of ping pp from its listed candidates (parity z, ring slot rs of its oldest row)
    auto finish = [&](int pp, int z, int rs, T xc) {
      const unsigned state = meta[4 * z], r1 = meta[4 * z + 1], dk = meta[4 * z + 2];
      if (state == 2u) return;  // (written by the radix selection already)
      T out = epa::M<T>::nan();
      if (state == 1u && pooled_at(pp)) {
        const int M = __builtin_amdgcn_readfirstlane((int)ncand[z]);  // <= kMedSel
        unsigned long long ki = ~0ull;
        if (lane < M) ki = sort_key((double)element((int)cidx[z * kMedSel + lane], pp, rs));
        unsigned rank = 0u;
        for (int j = 0; j < M; ++j) {
          const unsigned long long kj = wave_bcast64(ki, j);
          rank += (kj < ki || (kj == ki && j < lane)) ? 1u : 0u;
        }
        const int a1 = __ffsll((long long)__ballot(lane < M && rank == r1)) - 1;
        const int a2 = __ffsll((long long)__ballot(lane < M && rank == r1 + dk)) - 1;
        const unsigned long long ka = wave_bcast64(ki, a1), kb = wave_bcast64(ki, a2);
        const double la = epa::lin_from_db(key_value(ka), mt.exp2_tab);
        const double med = ka == kb ? la : (la + epa::lin_from_db(key_value(kb), mt.exp2_tab)) * 0.5;
        out = (T)(10.0 * epa::fast_log10(med, mt.log_tab));
      }
      if (lane == 0) {
        const size_t at = cbase + (size_t)pp * S + s;
        if (a.pooled) a.pooled[at] = out;
        if (a.mask) a.mask[at] = (xc - out > a.thr) ? 1 : 0;
      }
    };
    __syncthreads();  // the previous job's last readers
    if (tid < 4) ncand[tid] = 0u;
    if (tid < 8) bins[8 * (npack - 1) + tid] = (unsigned short)kMedNoBin;  // (the padding of the last pack)
    rebuild(p0, 0);
    {
      unsigned b1, b2, r1, dk;
      const bool any = middle_bins(b1, b2, r1, dk);
      __syncthreads();
      if (any) {
        recentre(b1);
        rebuild(p0, 0);
      }
    }
    const bool own = tid < w;
    const int col = own ? column(tid) : 0;
    int rs = 0;  // ring slot of the row that leaves next
    T xc = (T)0, xc_prev = (T)0;
    for (int p = p0; p < p1; ++p) {
      const bool more = p + 1 < p1;
      const int z = p & 1;
      T vin = epa::M<T>::nan();
      if (own && more) vin = value_at(p + n + 1, col);  // requested now, used after the search
      xc_prev = xc;
      if (tid == kBlock - 64 && a.mask) xc = chan[(size_t)p * S + s];
      unsigned* nc = &ncand[z];
      unsigned state, M;
      for (bool first = true;; first = false) {
        __syncthreads();  // B1: bins and histogram hold the window of ping p; *nc == 0
        if (!finisher) {
          unsigned b1 = 0u, b2 = 0u, r1 = 0u, dk = 0u;
          const bool any = middle_bins(b1, b2, r1, dk);
          if (tid == 0) {
            meta[4 * z] = any ? 1u : 0u;
            meta[4 * z + 1] = r1;
            meta[4 * z + 2] = dk;
            meta[4 * z + 3] = b1;
          }
          if (any) {  // the members of those bins (the bins between b1 and b2 are empty: "in [b1, b2]" selects them)
            const unsigned h1 = b1 * 0x00010001u, sp = (b2 - b1 + 1u) * 0x00010001u;
            const uint4* packs = reinterpret_cast<const uint4*>(bins);
            constexpr int kSearch = kBlock - 64, kFly = 4;  // 128-bit reads in flight per lane
            for (int j0 = tid; j0 < npack; j0 += kFly * kSearch) {
              uint4 pk[kFly];
              unsigned hit[kFly];
#pragma unroll
              for (int f = 0; f < kFly; ++f) {
                const int j = j0 + f * kSearch;
                pk[f] = packs[min(j, npack - 1)];
                if (j >= npack) pk[f].x = pk[f].y = pk[f].z = pk[f].w = 0xffffffffu;
              }
#pragma unroll
              for (int f = 0; f < kFly; ++f)
                hit[f] = halves_in_span(pk[f].x, h1, sp) | halves_in_span(pk[f].y, h1, sp) |
                         halves_in_span(pk[f].z, h1, sp) | halves_in_span(pk[f].w, h1, sp);
              if (hit[0] | hit[1] | hit[2] | hit[3]) {
#pragma unroll
                for (int f = 0; f < kFly; ++f) {
                  if (hit[f]) {
                    const unsigned wds[4] = {pk[f].x, pk[f].y, pk[f].z, pk[f].w};
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                      const unsigned q = (u & 1) ? (wds[u >> 1] >> 16) : (wds[u >> 1] & 0xffffu);
                      if (q - b1 <= b2 - b1) {
                        const unsigned slot = atomicAdd(nc, 1u);
                        if (slot < (unsigned)kMedSel)
                          cidx[z * kMedSel + slot] = (unsigned short)(8 * (j0 + f * kSearch) + u);
                      }
                    }
                  }
                }
              }
            }
          }
        } else if (first && p > p0) {
          finish(p - 1, z ^ 1, rs == 0 ? R - 1 : rs - 1, xc_prev);
          if (lane == 0) ncand[z ^ 1] = 0u;
        }
        __syncthreads();  // B2: candidates listed; nobody reads the histogram or the bins of ping p below
        state = meta[4 * z];
        M = state ? *nc : 0u;
        if (M <= (unsigned)kMedSel) break;
        const int drift = (int)meta[4 * z + 3] - kMedBins / 2;
        if (drift < kMedBins / 4 && drift > -kMedBins / 4) break;
        // many candidates and the median more than 8 dB from the centre of the map: re-centre, search again
        recentre(meta[4 * z + 3]);
        __syncthreads();  // (everybody has read *nc and meta)
        if (tid == 0) *nc = 0u;
        rebuild(p, rs);
      }
      if (M > (unsigned)kMedSel) {  // more than 64 values within 1/128 dB of the median: radix selection from memory
        unsigned nv;
        double med;
        if (BY_VALUE) {
          SameRowsWindow<T> gw{chan, nvalid, S, P, p - n, R, c_lo, c_hi};
          med = window_median_lin(gw, &sc, mt.exp2_tab, nv, W);
        } else {
          Window<T> gw{chan, S, p - n, R, s - a.m, w, P, a.s0, true};
          med = window_median_lin(gw, &sc, mt.exp2_tab, nv, W);
        }
        T out = epa::M<T>::nan();
        if (nv && pooled_at(p)) out = (T)(10.0 * epa::fast_log10(med, mt.log_tab));
        if (tid == kBlock - 64) {
          const size_t at = cbase + (size_t)p * S + s;
          if (a.pooled) a.pooled[at] = out;
          if (a.mask) a.mask[at] = (xc - out > a.thr) ? 1 : 0;
        }
        if (tid == 0) meta[4 * z] = 2u;
      }
      if (more && own) {  // row p-n leaves, row p+n+1 enters (same ring slot)
        const int e = rs * w + tid;
        const unsigned ob = bins[e];
        if (ob != kMedNoBin) {
          atomicSub(&fine[ob], 1u);
          atomicSub(&coarse[ob >> 6], 1u);
        }
        enter(e, vin);
      }
      rs = rs + 1 == R ? 0 : rs + 1;
    }
    __syncthreads();
    if (finisher) {  // the last ping of the segment
      finish(p1 - 1, (p1 - 1) & 1, rs == 0 ? R - 1 : rs - 1, xc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// attenuated-signal mask with the 2n-ping block carried from ping to ping on the same engine (round 4)
// ------------------------------------------------------------------------------------------------
// A workgroup walks a chunk of consecutive pings of one channel.  The block [p-n, p+n) x layer lives in LDS as BINS
// (2 bytes per value, 1/128 dB wide around the block's median, clamped ends) + the two-level histogram of
// pool_median_slide_kernel; ping p+n enters and ping p-n leaves at every step.  Wavefronts 0-2 locate the bins of the
// block's two middle values and list the positions of their members; wavefront 3 works one ping behind: it reads
// those few values again from memory (L2), ranks them exactly and compares the block median with the ping's own
// median (attenuated_prepare_kernel); everybody writes that ping's mask row after the next barrier.  Whenever the
// layer limits change, the block exceeds the ring or more than 64 values share the median's bin, the block is
// rebuilt / the ping's medians are taken from memory by all threads.
#ifndef EPA_ATT_SCALE
#define EPA_ATT_SCALE 128
#endif
#ifndef EPA_ATT_MARGIN
#define EPA_ATT_MARGIN 4
#endif
constexpr int kAttCols = 4;          // layer samples per lane: layers up to 1024 samples are carried
constexpr int kAttWalkChunkMax = 256;

template <typename T>
__global__ __launch_bounds__(kBlock, 2) void attenuated_walk_kernel(const T* __restrict__ sv, int P, int S, int nchunks,
                                                                    int chunk_len, int n, T thr, int ring_cap,
                                                                    uint8_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ SelectScratchT<kMedCap> sc;  // (medians taken from memory)
  __shared__ int lim[kAttWalkChunkMax][2];
  __shared__ double own_db[kAttWalkChunkMax];  // the pings' own medians in dB (NaN: no valid sample in the layer)
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  unsigned* fine = reinterpret_cast<unsigned*>(smem + epa::kMathTabBytes);
  unsigned* coarse = fine + kMedBins;
  unsigned* ncand = coarse + kMedCoarse;  // [2] (+ padding)
  unsigned* meta = ncand + 4;             // [2][4]: state, r1, dk, b1
  unsigned short* cidx = reinterpret_cast<unsigned short*>(meta + 8);  // [2][kMedSel]
  unsigned short* bins = cidx + 2 * kMedSel;                           // [ring_cap], 16-byte aligned
  const int tid = threadIdx.x, lane = tid & 63;
  const bool finisher = tid >= kBlock - 64;
  const int c = blockIdx.x / nchunks, chunk = blockIdx.x - c * nchunks;
  const int p0 = chunk * chunk_len, p1 = min(P, p0 + chunk_len);
  const T* __restrict__ cb = sv + (size_t)c * P * S;
  uint8_t* mb = mask + (size_t)c * P * S;
  const int R = 2 * n;  // pings of a block: [p - n, p + n)
  // the chunk's layer limits, read before any of its mask rows is overwritten
  for (int i = tid; i < p1 - p0; i += kBlock) {
    const int* src = reinterpret_cast<const int*>(mb + (size_t)(p0 + i) * S);
    lim[i][0] = src[0];
    lim[i][1] = src[1];
    own_db[i] = __hiloint2double(src[3], src[2]);
  }
  if (tid < 4) ncand[tid] = 0u;
  __syncthreads();

  MedMap map{16.0f, 4096.0f};
  bool valid = false, centred = false;
  // the block in LDS: ring column j of every slot is sample base + j (pitch Lp >= the layer + a margin on either
  // side: limits that move by a few samples from ping to ping -- a depth with heave in it -- shift the layer inside
  // the ring instead of rebuilding the block); layer [w_up, w_lw), first ping w_lo in ring slot rs.  Columns outside
  // the layer hold kMedNoBin in every slot.
  int base = 0, Lp = 8, w_up = -1, w_lw = -1, w_lo = 0, rs = 0, margin = EPA_ATT_MARGIN;
  // the ping wavefront 3 still has to finish
  bool pending = false;
  int f_p = 0, f_rs = 0, f_lo = 0;

  auto write_row = [&](int p, bool flag, int first, int step) {
    unsigned* m = reinterpret_cast<unsigned*>(mb + (size_t)p * S);  // S % 4 == 0, 4-byte aligned (checked by the launcher)
    const unsigned f4 = flag ? 0x01010101u : 0u;
    for (int i = first; i < S / 4; i += step) m[i] = f4;
  };
  auto enter = [&](int e, T v) {
    unsigned b = kMedNoBin;
    if (v == v) {
      b = map.bin((double)v);
      atomicAdd(&fine[b], 1u);
      atomicAdd(&coarse[b >> 6], 1u);
    }
    bins[e] = (unsigned short)b;
  };
  // element e of the block whose first ping lo sits in ring slot rs0
  auto element = [&](int e, int lo, int rs0) -> T {
    const float inv = 1.0f / (float)Lp;
    int slot = (int)(((float)e + 0.5f) * inv);
    if (slot * Lp > e) --slot;
    else if ((slot + 1) * Lp <= e) ++slot;
    int k = slot - rs0;
    if (k < 0) k += R;
    return cb[(size_t)(lo + k) * S + base + (e - slot * Lp)];
  };
  auto middle_bins = [&](unsigned& b1, unsigned& b2, unsigned& r1, unsigned& dk) -> bool {
    const unsigned cv = coarse[lane];
    const unsigned ci = wave_scan_incl(cv);
    const unsigned N = (unsigned)__builtin_amdgcn_readlane((int)ci, 63);
    if (!N) return false;
    const unsigned k1 = (N - 1u) >> 1;
    dk = (N & 1u) ? 0u : 1u;
    unsigned rem1, rem2, r2;
    const unsigned cb1 = wave_rank_lane(ci, cv, k1, rem1);
    const unsigned cb2 = wave_rank_lane(ci, cv, k1 + dk, rem2);
    const unsigned fv = fine[cb1 * 64 + lane];
    const unsigned fi = wave_scan_incl(fv);
    b1 = cb1 * 64 + wave_rank_lane(fi, fv, rem1, r1);
    if (cb2 == cb1) {
      b2 = cb1 * 64 + wave_rank_lane(fi, fv, rem2, r2);
    } else {
      const unsigned gv = fine[cb2 * 64 + lane];
      const unsigned gi = wave_scan_incl(gv);
      b2 = cb2 * 64 + wave_rank_lane(gi, gv, rem2, r2);
    }
    return true;
  };
  // histogram and bins of the block [lo, lo + R) x [up, lw) from memory, its first ping in slot 0 (all threads;
  // base / Lp set by the caller)
  auto rebuild = [&](int lo, int up, int lw) {
    for (int i = tid; i < kMedBins + kMedCoarse; i += kBlock) fine[i] = 0u;
    __syncthreads();
    // sixteen elements per lane and trip, all their requests in flight together (a dependent load per element would
    // cost the chunk as much as its whole walk)
    constexpr int kTrip = 16;
    const int total = R * Lp;
    const float inv = 1.0f / (float)Lp;
    for (int i0 = tid; i0 < total; i0 += kTrip * kBlock) {
      T v[kTrip];
      bool in[kTrip];
#pragma unroll
      for (int u = 0; u < kTrip; ++u) {
        const int i = i0 + u * kBlock;
        int slot = (int)(((float)i + 0.5f) * inv);
        if (slot * Lp > i) --slot;
        else if ((slot + 1) * Lp <= i) ++slot;
        const int smp = base + (i - slot * Lp);
        in[u] = i < total && smp >= up && smp < lw;
        v[u] = in[u] ? cb[(size_t)(lo + slot) * S + smp] : epa::M<T>::nan();
      }
#pragma unroll
      for (int u = 0; u < kTrip; ++u) {
        const int i = i0 + u * kBlock;
        if (in[u]) enter(i, v[u]);
        else if (i < total) bins[i] = (unsigned short)kMedNoBin;
      }
    }
    __syncthreads();
  };
  // the layer moves from [w_up, w_lw) to [up, lw) inside the ring: the samples that leave it are un-counted in every
  // slot, those that join it are read from memory (all threads, between two barriers)
  auto shift_layer = [&](int up, int lw) {
    auto each = [&](int a, int b, bool join) {  // samples [a, b) of every slot
      const int wd = b - a;
      if (wd <= 0) return;
      for (int i = tid; i < wd * R; i += kBlock) {
        const int slot = i / wd, smp = a + (i - slot * wd);
        const int e = slot * Lp + (smp - base);
        if (join) {
          int k = slot - rs;
          if (k < 0) k += R;
          enter(e, cb[(size_t)(w_lo + k) * S + smp]);
        } else {
          const unsigned ob = bins[e];
          if (ob != kMedNoBin) {
            atomicSub(&fine[ob], 1u);
            atomicSub(&coarse[ob >> 6], 1u);
          }
          bins[e] = (unsigned short)kMedNoBin;
        }
      }
    };
    each(w_up, min(up, w_lw), false);
    each(max(lw, w_up), w_lw, false);
    each(up, min(w_up, lw), true);
    each(max(w_lw, up), lw, true);
  };
  auto recentre = [&](unsigned b) {
    const float centre = ((float)b + 0.5f - map.off) / map.scale;
    map.scale = (float)EPA_ATT_SCALE;
    map.off = (float)(kMedBins / 2) - centre * (float)EPA_ATT_SCALE;
    centred = true;
  };
  // both medians of ping p from memory, by all threads
  auto sweep = [&](int p, int up, int L) -> bool {
    unsigned nv;
    Window<T> w1{cb, S, p, 1, up, L, P, 0, false};
    const double m1 = window_median_lin(w1, &sc, mt.exp2_tab, nv, L);
    if (!nv) return false;
    Window<T> w2{cb, S, p - n, R, up, L, P, 0, false};
    unsigned nv2;
    const double m2 = window_median_lin(w2, &sc, mt.exp2_tab, nv2, R * L);
    const T ping_db = (T)(10.0 * epa::fast_log10(m1, mt.log_tab));
    const T block_db = nv2 ? (T)(10.0 * epa::fast_log10(m2, mt.log_tab)) : epa::M<T>::nan();
    return (ping_db - block_db) < thr;
  };
  // wavefront 3: the two medians of the pending ping, the comparison, its mask row
  // wavefront 3: the block median of the pending ping from its listed candidates (read again from memory, ranked by
  // counting: one per lane, the others' keys by readlane), compared with the ping's own median (attenuated_prepare_kernel)
  auto finish = [&]() {
    const int z = f_p & 1;
    const unsigned state = meta[4 * z], r1 = meta[4 * z + 1], dk = meta[4 * z + 2];
    const int M = state == 1u ? __builtin_amdgcn_readfirstlane((int)ncand[z]) : 0;  // <= kMedSel
    const T ping_db = (T)own_db[f_p - p0];
    bool flag = false;
    if (M && ping_db == ping_db) {
      unsigned long long ki = ~0ull;
      if (lane < M) ki = sort_key((double)element((int)cidx[z * kMedSel + lane], f_lo, f_rs));
      unsigned rank = 0u;
      for (int j = 0; j < M; ++j) {
        const unsigned long long kj = wave_bcast64(ki, j);
        rank += (kj < ki || (kj == ki && j < lane)) ? 1u : 0u;
      }
      const int a1 = __ffsll((long long)__ballot(lane < M && rank == r1)) - 1;
      const int a2 = __ffsll((long long)__ballot(lane < M && rank == r1 + dk)) - 1;
      const unsigned long long ka = wave_bcast64(ki, a1), kb = wave_bcast64(ki, a2);
      const double la = epa::lin_from_db(key_value(ka), mt.exp2_tab);
      const double med = ka == kb ? la : (la + epa::lin_from_db(key_value(kb), mt.exp2_tab)) * 0.5;
      const T block_db = (T)(10.0 * epa::fast_log10(med, mt.log_tab));
      flag = (ping_db - block_db) < thr;
    }
    if (lane == 0) {
      ncand[z] = 0u;
      meta[4 * z + 3] = flag ? 1u : 0u;  // (b1 is no longer needed: the verdict takes its place)
    }
  };
  auto drain = [&]() {  // the pending ping finished before the block changes under it (uniform)
    if (pending) {
      __syncthreads();
      if (finisher) finish();
      __syncthreads();
      write_row(f_p, meta[4 * (f_p & 1) + 3] != 0u, tid, kBlock);
      pending = false;
    }
  };

  // the block moves on by one ping: the oldest ping leaves, ping w_lo + R (its layer samples in `pre`) takes its slot
  auto advance = [&](const T (&pre)[kAttCols]) {
    const int Lw = w_lw - w_up;
#pragma unroll
    for (int k = 0; k < kAttCols; ++k) {
      const int col = tid + k * kBlock;
      if (col < Lw) {
        const int e = rs * Lp + (w_up - base) + col;
        const unsigned ob = bins[e];
        if (ob != kMedNoBin) {
          atomicSub(&fine[ob], 1u);
          atomicSub(&coarse[ob >> 6], 1u);
        }
        enter(e, pre[k]);
      }
    }
    rs = rs + 1 == R ? 0 : rs + 1;
    ++w_lo;
  };

  for (int p = p0; p < p1; ++p) {
    const int up = lim[p - p0][0], lw = lim[p - p0][1], L = lw - up;
    if (!(p - n >= 0 && (long long)p + n <= (long long)P - 1 && lw > up)) {
      // no verdict for this ping (its range row gives no layer, or it has no 2n-ping block) -- the block in LDS stays
      // good for the pings behind it: it moves on over its own layer
      drain();
      write_row(p, false, tid, kBlock);
      if (valid && w_lo == p - n && p + 1 < p1 && (long long)p + n <= (long long)P - 1) {
        T pre[kAttCols];
#pragma unroll
        for (int k = 0; k < kAttCols; ++k) {
          const int col = tid + k * kBlock;
          pre[k] = col < w_lw - w_up ? cb[(size_t)(p + n) * S + w_up + col] : epa::M<T>::nan();
        }
        advance(pre);
      } else {
        valid = false;
      }
      continue;
    }
    if ((long long)R * ((L + 7) & ~7) > ring_cap || L > kAttCols * kBlock) {  // block beyond the ring
      drain();
      valid = false;
      __syncthreads();
      const bool flag = sweep(p, up, L);
      write_row(p, flag, tid, kBlock);
      continue;
    }
    if (valid && w_lo == p - n && up >= base && lw <= base + Lp) {
      if (up != w_up || lw != w_lw) {  // the layer moved inside the ring
        __syncthreads();
        shift_layer(up, lw);
        w_up = up; w_lw = lw;
      }
    } else {  // (re)build the block
      // a layer that walked out of the ring: a wider margin from now on (every margin sample is searched at every step)
      if (valid && w_lo == p - n) margin = min(64, 2 * margin);
      drain();
      // pitch: the layer + the margin on either side, as far as the ring and a lane's columns allow
      Lp = min(min((L + 2 * margin + 7) & ~7, (ring_cap / R) & ~7), (kAttCols + 1) * kBlock);
      base = max(0, up - (Lp - L) / 2);
      if (!centred) {  // the first block: 1/16-dB bins over [-256, 0) dB find its median, then the fine map
        rebuild(p - n, up, lw);
        unsigned b1, b2, r1, dk;
        const bool any = middle_bins(b1, b2, r1, dk);
        __syncthreads();
        if (any) recentre(b1);
      }
      rebuild(p - n, up, lw);
      valid = true;
      w_up = up; w_lw = lw; w_lo = p - n; rs = 0;
    }
    // will the block of the next ping be carried?  then its entering ping is requested now (over THIS ping's layer:
    // a layer that differs is shifted at the next step)
    bool slide = p + 1 < p1;  // (ping p + n exists: this ping has a block)
    if (slide) {
      const int nu = lim[p + 1 - p0][0], nl = lim[p + 1 - p0][1];
      slide = nl <= nu || (nu >= base && nl <= base + Lp && nl - nu <= kAttCols * kBlock);
    }
    T pre[kAttCols];
#pragma unroll
    for (int k = 0; k < kAttCols; ++k) {
      const int col = tid + k * kBlock;
      pre[k] = (slide && col < L) ? cb[(size_t)(p + n) * S + up + col] : epa::M<T>::nan();
    }
    const int z = p & 1;
    unsigned* nc = &ncand[z];
    const int total8 = (R * Lp) >> 3;  // (Lp is a multiple of 8)
    unsigned state, M;
    for (bool first = true;; first = false) {
      __syncthreads();  // B1: bins and histogram hold the block of ping p; *nc == 0
      if (!finisher) {
        unsigned b1 = 0u, b2 = 0u, r1 = 0u, dk = 0u;
        const bool any = middle_bins(b1, b2, r1, dk);
        if (tid == 0) {
          meta[4 * z] = any ? 1u : 0u;
          meta[4 * z + 1] = r1;
          meta[4 * z + 2] = dk;
          meta[4 * z + 3] = b1;
        }
        if (any) {  // the members of those bins, eight bins per 128-bit read
          const unsigned h1 = b1 * 0x00010001u, sp = (b2 - b1 + 1u) * 0x00010001u;
          const uint4* packs = reinterpret_cast<const uint4*>(bins);
          constexpr int kSearch = kBlock - 64, kFly = 4;  // 128-bit reads in flight per lane
          for (int j0 = tid; j0 < total8; j0 += kFly * kSearch) {
            uint4 pk[kFly];
            unsigned hit[kFly];
#pragma unroll
            for (int f = 0; f < kFly; ++f) {
              const int j = j0 + f * kSearch;
              pk[f] = packs[min(j, total8 - 1)];
              if (j >= total8) pk[f].x = pk[f].y = pk[f].z = pk[f].w = 0xffffffffu;
            }
#pragma unroll
            for (int f = 0; f < kFly; ++f)
              hit[f] = halves_in_span(pk[f].x, h1, sp) | halves_in_span(pk[f].y, h1, sp) |
                       halves_in_span(pk[f].z, h1, sp) | halves_in_span(pk[f].w, h1, sp);
            if (hit[0] | hit[1] | hit[2] | hit[3]) {
#pragma unroll
              for (int f = 0; f < kFly; ++f) {
                if (hit[f]) {
                  const unsigned wds[4] = {pk[f].x, pk[f].y, pk[f].z, pk[f].w};
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const unsigned q = (u & 1) ? (wds[u >> 1] >> 16) : (wds[u >> 1] & 0xffffu);
                    if (q - b1 <= b2 - b1) {
                      const unsigned slot = atomicAdd(nc, 1u);
                      if (slot < (unsigned)kMedSel)
                        cidx[z * kMedSel + slot] = (unsigned short)(8 * (j0 + f * kSearch) + u);
                    }
                  }
                }
              }
            }
          }
        }
      } else if (first && pending) {
        finish();
      }
      __syncthreads();  // B2: candidates listed; the pending ping is finished
      if (first && pending) write_row(f_p, meta[4 * (f_p & 1) + 3] != 0u, tid, kBlock);
      state = meta[4 * z];
      M = state ? *nc : 0u;
      if (M <= (unsigned)kMedSel) break;
      const int drift = (int)meta[4 * z + 3] - kMedBins / 2;
      if (drift < kMedBins / 4 && drift > -kMedBins / 4) break;
      // many candidates and the median more than 8 dB from the centre of the map: re-centre, search again
      recentre(meta[4 * z + 3]);
      __syncthreads();  // (everybody has read *nc and meta)
      if (tid == 0) *nc = 0u;
      pending = false;
      rebuild(p - n, up, lw);
      rs = 0;
    }
    pending = false;
    const bool own_unknown = (unsigned long long)__double_as_longlong(own_db[p - p0]) == kOwnUnknown;
    if (M > (unsigned)kMedSel || own_unknown) {  // more than 64 values in a median's bin: both medians from memory
      const bool flag = sweep(p, up, L);
      write_row(p, flag, tid, kBlock);
      __syncthreads();
      if (tid == 0) {
        meta[4 * z] = 2u;
        *nc = 0u;
      }
    } else {
      pending = true;
      f_p = p; f_rs = rs; f_lo = p - n;
    }
    if (slide) advance(pre);  // ping p-n leaves, ping p+n enters (same ring slot)
    else valid = false;
  }
  drain();
}

// out = mask ? src : fill   (fill: scalar, or an array like src when fill_arr != NULL)
template <typename T>
__global__ __launch_bounds__(kBlock) void apply_mask_kernel(const T* __restrict__ src,
                                                            const uint8_t* __restrict__ mask,
                                                            size_t n, size_t mask_period, T fill,
                                                            const T* __restrict__ fill_arr,
                                                            size_t fill_period, T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    out[i] = mask[i % mask_period] ? src[i] : (fill_arr ? fill_arr[i % fill_period] : fill);
}

// The same with up to four masks -- apply_mask(ds, [m1, m2, m3]) is where(m1 & m2 & m3, src, fill) (mask/api.py:402-432):
// one sweep, 8 + 3 + 8 B per fp64 sample instead of two mask_and passes (3 B each) before it and a min / max sweep (8 B)
// behind it for the variable's actual_range -- and the NaN-skipping {min, max} of what is written, one pair per workgroup
// in ``part`` (folded by minmax_pairs_kernel).  A mask whose period is the whole array is indexed without the division.
struct MaskSet {
  const uint8_t* m[4];
  size_t period[4];
  int n;
};
template <typename T>
__global__ __launch_bounds__(kBlock) void apply_masks_kernel(const T* __restrict__ src, MaskSet ms, size_t n, T fill,
                                                             const T* __restrict__ fill_arr, size_t fill_period,
                                                             T* __restrict__ out, double* __restrict__ part) {
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    bool keep = true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < ms.n) keep = keep & (ms.m[k][ms.period[k] == n ? i : i % ms.period[k]] != 0);
    const T v = keep ? src[i] : (fill_arr ? fill_arr[fill_period == n ? i : i % fill_period] : fill);
    out[i] = v;
    if (part) {  // fmin / fmax ignore a NaN operand
      lo = fmin(lo, (double)v);
      hi = fmax(hi, (double)v);
    }
  }
  if (part) {
    __shared__ double slo[kBlock / 64], shi[kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = fmin(lo, __shfl_down(lo, o, 64));
      hi = fmax(hi, __shfl_down(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      slo[threadIdx.x >> 6] = lo;
      shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      part[2 * (size_t)blockIdx.x] = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
      part[2 * (size_t)blockIdx.x + 1] = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
    }
  }
}

// {min, max} pairs of the workgroups -> out[0], out[1] (NaN when nothing was a number)
__global__ __launch_bounds__(kBlock) void minmax_pairs_kernel(const double* __restrict__ part, int npairs,
                                                              double* __restrict__ out) {
  __shared__ double slo[kBlock / 64], shi[kBlock / 64];
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int i = threadIdx.x; i < npairs; i += kBlock) {
    lo = fmin(lo, part[2 * i]);
    hi = fmax(hi, part[2 * i + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_down(lo, o, 64));
    hi = fmax(hi, __shfl_down(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
    hi = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
    out[0] = lo <= hi ? lo : __builtin_nan("");
    out[1] = lo <= hi ? hi : __builtin_nan("");
  }
}

__global__ __launch_bounds__(kBlock) void mask_and_kernel(const uint8_t* __restrict__ a,
                                                          const uint8_t* __restrict__ b, size_t n,
                                                          size_t b_period, uint8_t* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
    out[i] = (a[i] && b[i % b_period]) ? 1 : 0;
}

inline int row_grid(long long rows) { return (int)(rows < 65536 ? (rows > 0 ? rows : 1) : 65536); }

template <typename K>
int set_lds(K kern, size_t lds) {
  if (lds > 64 * 1024)
    EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return EPA_OK;
}

constexpr size_t kMaxLds = 156 * 1024;

// dynamic LDS of pool_median_slide_kernel for a ring of W window elements
inline size_t med_slide_lds(size_t W) {
  return epa::kMathTabBytes + (size_t)(kMedBins + kMedCoarse + 4 + 8) * 4 + 2 * kMedSel * 2 + ((W + 7) / 8) * 16;
}

}  // namespace

extern "C" int epa_range_bin_smooth(const void* sv, const void* range, int C, int P, int S, int nper,
                                    double r0, double bin, int nbins, void* up_out, int dtype,
                                    epa_stream_t stream) {
  EPA_CHECK_ARG(sv && up_out, "epa_range_bin_smooth: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_range_bin_smooth: sizes must be positive");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_range_bin_smooth: bad dtype %d", dtype);
  double delta = 1.0;
  if (range) {
    EPA_CHECK_ARG(bin > 0 && nbins > 0 && r0 == r0, "epa_range_bin_smooth: bad bin grid");
    delta = (r0 + bin) - r0;  // np.arange's step
  } else {
    EPA_CHECK_ARG(nper > 0, "epa_range_bin_smooth: samples per bin must be positive");
    nbins = (S + nper - 1) / nper;
  }
  // bins of a ping held in LDS at once; a finer grid is taken in segments (the ping is re-swept per segment)
  const int seg_cap = (int)((kMaxLds - epa::kMathTabBytes - 8) / 12) & ~1;
  const int seg_bins = nbins < seg_cap ? ((nbins + 1) & ~1) : seg_cap;
  const size_t lds = epa::kMathTabBytes + (size_t)seg_bins * 12 + 8;
  const long long rows = (long long)C * P;
  hipStream_t st = (hipStream_t)stream;
#define EPA_RBS(T, BV)                                                                            \
  do {                                                                                            \
    auto kern = range_bin_smooth_kernel<T, BV>;                                                   \
    if (int rc = set_lds(kern, lds)) return rc;                                                   \
    hipLaunchKernelGGL(kern, dim3(row_grid(rows)), dim3(kBlock), lds, st, (const T*)sv,           \
                       (const T*)range, rows, S, nper, r0, delta, nbins, seg_bins, (T*)up_out);   \
  } while (0)
  if (dtype == EPA_F64) { if (range) EPA_RBS(double, true); else EPA_RBS(double, false); }
  else { if (range) EPA_RBS(float, true); else EPA_RBS(float, false); }
#undef EPA_RBS
  return epa::check_launch("range_bin_smooth_kernel");
}

extern "C" int epa_impulse_mask(const void* up, int C, int P, int S, int num_side_pings,
                                double threshold, uint8_t* mask_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(up && mask_out, "epa_impulse_mask: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_impulse_mask: sizes must be positive");
  EPA_CHECK_ARG(num_side_pings >= 1, "epa_impulse_mask: num_side_pings must be >= 1");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_impulse_mask: bad dtype %d", dtype);
  const long long rows = (long long)C * P;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(impulse_compare_kernel<double>, dim3(row_grid(rows)), dim3(kBlock), 0, st,
                       (const double*)up, P, S, rows, num_side_pings, threshold, mask_out);
  else
    hipLaunchKernelGGL(impulse_compare_kernel<float>, dim3(row_grid(rows)), dim3(kBlock), 0, st,
                       (const float*)up, P, S, rows, num_side_pings, (float)threshold, mask_out);
  return epa::check_launch("impulse_compare_kernel");
}

extern "C" int epa_pool_sv(const void* sv, int C, int P, int S, int first_sample, int num_side_pings,
                           int num_side_samples, int func, double threshold, void* pooled_out,
                           uint8_t* mask_out, double* ws_sum, int32_t* ws_cnt, int dtype,
                           epa_stream_t stream) {
  EPA_CHECK_ARG(sv && (pooled_out || mask_out), "epa_pool_sv: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_pool_sv: sizes must be positive");
  EPA_CHECK_ARG(first_sample >= 0 && num_side_pings >= 0 && num_side_samples >= 0,
                "epa_pool_sv: negative window argument");
  EPA_CHECK_ARG(func == EPA_POOL_NANMEAN || func == EPA_POOL_NANMEDIAN, "epa_pool_sv: bad func %d", func);
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_pool_sv: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int s0 = first_sample < S ? first_sample : S;
  const int n = num_side_pings, m = num_side_samples;
  if (func == EPA_POOL_NANMEDIAN) {
    const size_t W = (size_t)(2 * n + 1) * (2 * m + 1);
    const size_t slide_lds = med_slide_lds(W);
    if (2 * m + 1 <= kBlock && W <= 65535 && slide_lds + sizeof(SelectScratchT<kMedCap>) + 512 <= kMaxLds) {
      // the window carried from ping to ping: one workgroup per (channel, ping segment, column)
      const int nseg = (P + kMedSeg - 1) / kMedSeg;
      const long long jobs = (long long)C * nseg * S;
      const int grid = (int)(jobs < (1 << 22) ? jobs : (1 << 22));
#define EPA_MS(T)                                                                                          \
  do {                                                                                                     \
    MedSlideArgs<T> a{};                                                                                   \
    a.sv = (const T*)sv; a.P = P; a.S = S; a.n = n; a.nseg = nseg; a.jobs = jobs; a.thr = (T)threshold;    \
    a.pooled = (T*)pooled_out; a.mask = mask_out; a.cap = (int)W; a.s0 = s0; a.m = m;                      \
    auto kern = pool_median_slide_kernel<T, false>;                                                        \
    if (int rc = set_lds(kern, slide_lds)) return rc;                                                      \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), slide_lds, st, a);                                  \
  } while (0)
      if (dtype == EPA_F64) EPA_MS(double); else EPA_MS(float);
#undef EPA_MS
      return epa::check_launch("pool_median_slide_kernel");
    }
    const long long jobs = (long long)C * P * S;
    const int grid = (int)(jobs < (1 << 20) ? jobs : (1 << 20));
    if (dtype == EPA_F64)
      hipLaunchKernelGGL(pool_median_kernel<double>, dim3(grid), dim3(kBlock), 0, st, (const double*)sv,
                         P, S, jobs, s0, n, m, threshold, (double*)pooled_out, mask_out);
    else
      hipLaunchKernelGGL(pool_median_kernel<float>, dim3(grid), dim3(kBlock), 0, st, (const float*)sv,
                         P, S, jobs, s0, n, m, (float)threshold, (float*)pooled_out, mask_out);
    return epa::check_launch("pool_median_kernel");
  }
  EPA_CHECK_ARG(ws_sum && ws_cnt, "epa_pool_sv: nanmean needs the f64 / int32 [C*P*S] workspaces");
  const size_t nin = (size_t)kRangeTile + 2 * m;
  const size_t lds1 = epa::kMathTabBytes + (nin + 2) * 8 + (nin / kGroup + 1) * 12;
  if (lds1 > kMaxLds) {
    epa::set_error("epa_pool_sv: window %d x %d exceeds the LDS budget", 2 * n + 1, 2 * m + 1);
    return EPA_EUNSUPPORTED;
  }
  const long long rows = (long long)C * P;
  const int w = 2 * m + 1;
  if (s0 < S && w >= kScanMinW && w <= kScanMaxW) {
    const size_t ninp = (size_t)((kScanTile + 2 * m + 2) & ~1);
    const size_t lds = epa::kMathTabBytes + (2 * ninp + 2 * kBlock) * 8 + (2 * ninp + 2 * kBlock) * 2;
    const dim3 g1(row_grid(rows) < 16384 ? row_grid(rows) : 16384, (S - s0 + kScanTile - 1) / kScanTile);
#define EPA_BR(T)                                                                                   \
  do {                                                                                              \
    auto kern = box_range_scan_kernel<T>;                                                           \
    if (int rc = set_lds(kern, lds)) return rc;                                                     \
    hipLaunchKernelGGL(kern, g1, dim3(kBlock), lds, st, (const T*)sv, rows, S, s0, m, ws_sum, ws_cnt); \
  } while (0)
    if (dtype == EPA_F64) EPA_BR(double); else EPA_BR(float);
#undef EPA_BR
    if (int rc = epa::check_launch("box_range_scan_kernel")) return rc;
  } else if (s0 < S) {
    const dim3 g1(row_grid(rows) < 16384 ? row_grid(rows) : 16384, (S - s0 + kRangeTile - 1) / kRangeTile);
#define EPA_BR(T)                                                                                   \
  do {                                                                                              \
    auto kern = box_range_kernel<T>;                                                                \
    if (int rc = set_lds(kern, lds1)) return rc;                                                    \
    hipLaunchKernelGGL(kern, g1, dim3(kBlock), lds1, st, (const T*)sv, rows, S, s0, m, ws_sum, ws_cnt); \
  } while (0)
    if (dtype == EPA_F64) EPA_BR(double); else EPA_BR(float);
#undef EPA_BR
    if (int rc = epa::check_launch("box_range_kernel")) return rc;
  }
  // ping pass: grid = (ping segments, column stripes, channels)
  const dim3 g2((P + kSlideSeg - 1) / kSlideSeg, (S + kBlock - 1) / kBlock, C);
  EPA_CHECK_ARG(g2.y <= 65535u && g2.z <= 65535u, "epa_pool_sv: more than 65535 channels or 16.7 M samples per ping");
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(box_ping_slide_kernel<double>, g2, dim3(kBlock), epa::kMathTabBytes + kSlidePad, st,
                       (const double*)sv, ws_sum, ws_cnt, P, S, s0, n, threshold, (double*)pooled_out, mask_out);
  else
    hipLaunchKernelGGL(box_ping_slide_kernel<float>, g2, dim3(kBlock), epa::kMathTabBytes + kSlidePad, st,
                       (const float*)sv, ws_sum, ws_cnt, P, S, s0, n, (float)threshold, (float*)pooled_out,
                       mask_out);
  return epa::check_launch("box_ping_slide_kernel");
}

extern "C" int epa_range_rows_check(const void* range, int C, int P, int S, int dtype, int32_t* nvalid_out,
                                    int32_t* violations_out, epa_stream_t stream) {
  EPA_CHECK_ARG(range && nvalid_out && violations_out, "epa_range_rows_check: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_range_rows_check: sizes must be positive");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_range_rows_check: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  EPA_CHECK_HIP(hipMemsetAsync(violations_out, 0, sizeof(int32_t), st));
  const long long rows = (long long)C * P;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(rows_check_kernel<double>, dim3(row_grid(rows)), dim3(kBlock), 0, st,
                       (const double*)range, rows, S, nvalid_out, violations_out);
  else
    hipLaunchKernelGGL(rows_check_kernel<float>, dim3(row_grid(rows)), dim3(kBlock), 0, st,
                       (const float*)range, rows, S, nvalid_out, violations_out);
  return epa::check_launch("rows_check_kernel");
}

namespace {
template <typename T>
int launch_pool_value(const void* sv, const void* range, const int32_t* nvalid, int C, int P, int S,
                      double bin, int n, double exclude_above, double rmin, double rmax, int func,
                      double thr, void* pooled, uint8_t* mask, void* ws, hipStream_t st) {
  PoolValueArgs<T> a{(const T*)sv, (const T*)range, nvalid, P, S, n, (T)bin, (T)exclude_above,
                     (T)rmin, (T)rmax, (T)thr, (T*)pooled, mask};
  if (func == EPA_POOL_NANMEAN) {
    const long long rows = (long long)C * P;
    const dim3 grid(row_grid(rows) < 32768 ? row_grid(rows) : 32768, (S + kBlock - 1) / kBlock);
    if (ws) {
      const size_t N = (size_t)rows * S;
      double* wh = static_cast<double*>(ws);
      double* wl = wh + N;
      double* rh = wl + N;
      double* rl = rh + N;
      int* wn = reinterpret_cast<int*>(rl + N);
      int* rn = wn + N;
      int* ilo = rn + N;
      int* ihi = ilo + (size_t)C * S;
      int* ref = ihi + (size_t)C * S;
      int* differ = ref + C;
      uint8_t* dirty = reinterpret_cast<uint8_t*>(differ + C);
      const dim3 rowg(row_grid(rows) < 16384 ? row_grid(rows) : 16384);
      // which channels have one range vector for all their pings?
      hipLaunchKernelGGL(ref_row_kernel, dim3(C), dim3(kBlock), 0, st, nvalid, P, ref, differ);
      hipLaunchKernelGGL(rows_same_kernel<T>, rowg, dim3(kBlock), 0, st, (const T*)range, nvalid, rows, P, S, ref, differ);
      hipLaunchKernelGGL(value_intervals_kernel<T>, dim3((S + kBlock - 1) / kBlock, C), dim3(kBlock), 0, st, a, ref, ilo, ihi);
      if (int rc = epa::check_launch("rows_same_kernel")) return rc;
      // channels whose pings differ in their range vectors: the RUNS of pings that do share one take the sliding route
      // (see run_flag_kernel); their bookkeeping lives in the workspace's spare quarter (rl is not used on this route),
      // behind the lean kernel's ``todo`` bytes.  EPA_POOL_RUNS=0: off (development knob)
      const uint8_t* elig = nullptr;
      const uint8_t* need_w = nullptr;
      int* row_slot = nullptr;
      int *ilo2 = nullptr, *ihi2 = nullptr;
      {
        static const bool runs_on = [] { const char* e = getenv("EPA_POOL_RUNS"); return !(e && e[0] == '0'); }();
        const int lmin = 2 * n + 1, nslotc = (P + lmin - 1) / lmin, ntiles = (P + kRunTile - 1) / kRunTile;
        const long long nslots = (long long)C * nslotc;
        const long long nwork0 = (long long)C * ((P + kStageRows - 1) / kStageRows) * (long long)grid.y;
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        size_t off = up16((size_t)nwork0);
        unsigned char* base = reinterpret_cast<unsigned char*>(rl);
        auto take = [&](size_t bytes) { unsigned char* q = base + off; off = up16(off + bytes); return q; };
        uint8_t* flag = take((size_t)rows);
        uint8_t* run_bad = take((size_t)rows);
        uint8_t* el = take((size_t)rows);
        uint8_t* nw = take((size_t)rows);
        int* start = reinterpret_cast<int*>(take((size_t)rows * 4));
        int* run_len = reinterpret_cast<int*>(take((size_t)rows * 4));
        int* rslot = reinterpret_cast<int*>(take((size_t)rows * 4));
        unsigned long long* run_key = reinterpret_cast<unsigned long long*>(take((size_t)rows * 8));
        int* tile_last = reinterpret_cast<int*>(take((size_t)C * ntiles * 4));
        int* slot_ref = reinterpret_cast<int*>(take((size_t)nslots * 4));
        int* tlo = reinterpret_cast<int*>(take((size_t)nslots * S * 4));
        int* thi = reinterpret_cast<int*>(take((size_t)nslots * S * 4));
        if (runs_on && n >= 1 && off <= N * sizeof(double) && nslots < (1ll << 31) && ntiles < (1 << 30)) {
          const dim3 flat((unsigned)std::min<long long>((rows + kBlock - 1) / kBlock, 65536));
          EPA_CHECK_HIP(hipMemsetAsync(run_bad, 0, (size_t)rows, st));
          EPA_CHECK_HIP(hipMemsetAsync(run_len, 0, (size_t)rows * 4, st));
          EPA_CHECK_HIP(hipMemsetAsync(run_key, 0, (size_t)rows * 8, st));
          EPA_CHECK_HIP(hipMemsetAsync(slot_ref, 0xff, (size_t)nslots * 4, st));
          hipLaunchKernelGGL(run_flag_kernel<T>, flat, dim3(kBlock), 0, st, (const T*)range, nvalid, differ, rows, P, S, flag);
          if (int rc = epa::check_launch("run_flag_kernel")) return rc;
          hipLaunchKernelGGL(run_scan_tiles_kernel, dim3((unsigned)ntiles, (unsigned)C), dim3(kBlock), 0, st, flag, P, ntiles,
                             start, tile_last);
          if (int rc = epa::check_launch("run_scan_tiles_kernel")) return rc;
          hipLaunchKernelGGL(run_scan_carry_kernel, dim3((unsigned)C), dim3(1), 0, st, tile_last, ntiles);
          if (int rc = epa::check_launch("run_scan_carry_kernel")) return rc;
          hipLaunchKernelGGL(run_stats_kernel, flat, dim3(kBlock), 0, st, nvalid, differ, tile_last, rows, P, ntiles, start,
                             run_len, run_key);
          if (int rc = epa::check_launch("run_stats_kernel")) return rc;
          hipLaunchKernelGGL(run_verify_kernel<T>, rowg, dim3(kBlock), 0, st, (const T*)range, nvalid, differ, start, run_len,
                             run_key, rows, P, S, lmin, run_bad);
          if (int rc = epa::check_launch("run_verify_kernel")) return rc;
          hipLaunchKernelGGL(run_finish_kernel, flat, dim3(kBlock), 0, st, differ, start, run_len, run_key, run_bad, rows, P,
                             n, lmin, nslotc, el, rslot, slot_ref);
          if (int rc = epa::check_launch("run_finish_kernel")) return rc;
          // (the staged kernels take whole groups of pings: an eligible ping of a mixed group is done there as well)
          hipLaunchKernelGGL(run_need_rows_kernel, flat, dim3(kBlock), 0, st, differ, el, rows, P,
                             n + std::max(kLeanRows, kStageRows) - 1, nw);
          if (int rc = epa::check_launch("run_need_rows_kernel")) return rc;
          hipLaunchKernelGGL(value_intervals_runs_kernel<T>, dim3((unsigned)nslots, (S + kBlock - 1) / kBlock), dim3(kBlock), 0,
                             st, a, slot_ref, nslotc, tlo, thi);
          if (int rc = epa::check_launch("value_intervals_runs_kernel")) return rc;
          elig = el; row_slot = rslot; ilo2 = tlo; ihi2 = thi; need_w = nw;
        }
      }
      // the channels with one vector, and the long runs of the others: running sums in LDS -> interval sums per row
      // (one kernel), then a sliding sum down every column; what is left: running sums to the workspace -- of the rows a
      // staged ping can still ask for --, then row by row
      const size_t Sp = ((size_t)S + 63) & ~(size_t)63;
      const size_t fuse_lds = Sp * 20 + (Sp / 16) * 10 + 16;
      const bool fuse = fuse_lds + epa::kMathTabBytes + 1024 <= kMaxLds;
      FuseArgs fa{differ, nvalid, ilo, ihi, rh, rn, P, fuse ? 1 : 0, nullptr, fuse ? need_w : nullptr};
      if (fuse) {
        auto kern = row_interval_blocks_kernel<T>;
        if (int rc = set_lds(kern, fuse_lds)) return rc;
        hipLaunchKernelGGL(kern, rowg, dim3(kBlock), fuse_lds, st, (const T*)sv, rows, S, fa);
        if (int rc = epa::check_launch("row_interval_blocks_kernel")) return rc;
        if (elig) {
          FuseArgs fr{differ, nvalid, ilo2, ihi2, rh, rn, P, 1, row_slot, nullptr};
          hipLaunchKernelGGL(kern, rowg, dim3(kBlock), fuse_lds, st, (const T*)sv, rows, S, fr);
          if (int rc = epa::check_launch("row_interval_blocks_kernel")) return rc;
        }
      }
      hipLaunchKernelGGL(row_running_sum_kernel<T>, rowg, dim3(kBlock), 0, st, (const T*)sv, rows, S, wh, wl, wn,
                         dirty, fa);
      if (int rc = epa::check_launch("row_running_sum_kernel")) return rc;
      if (!fuse) {
        hipLaunchKernelGGL(row_interval_sum_kernel<T>, grid, dim3(kBlock), 0, st, a, rows, wh, wl, wn, dirty, differ, ilo,
                           ihi, rh, rn);
        if (elig)
          hipLaunchKernelGGL(row_interval_sum_kernel<T>, grid, dim3(kBlock), 0, st, a, rows, wh, wl, wn, dirty, differ, ilo2,
                             ihi2, rh, rn, row_slot);
        if (int rc = epa::check_launch("row_interval_sum_kernel")) return rc;
      }
      const dim3 g2((P + kSlideSeg - 1) / kSlideSeg, (S + kBlock - 1) / kBlock, C);
      if (g2.y > 65535u || g2.z > 65535u) {
        epa::set_error("epa_pool_sv_value: more than 65535 channels or 16.7 M samples per ping");
        return EPA_EINVAL;
      }
      hipLaunchKernelGGL(value_slide_kernel<T>, g2, dim3(kBlock), epa::kMathTabBytes + kSlidePad, st, a, differ, ilo, rh,
                         rn);
      if (int rc = epa::check_launch("value_slide_kernel")) return rc;
      if (elig) {
        hipLaunchKernelGGL(value_slide_runs_kernel<T>, g2, dim3(kBlock), epa::kMathTabBytes + kSlidePad, st, a, differ, elig,
                           row_slot, ilo2, rh, rn);
        if (int rc = epa::check_launch("value_slide_runs_kernel")) return rc;
      }
      {  // ... and the pings the runs leave: neighbour rows staged in LDS
        const long long ngroups = (long long)C * ((P + kStageRows - 1) / kStageRows);
        const dim3 sgrid((unsigned)std::min<long long>(ngroups, 65535 * 4), grid.y);
        // the lean kernel first; the groups it flags in ``todo`` (kept in the workspace's spare quarter: rl is not used
        // on this route) go to the general one.  EPA_POOL_LEAN=0: the general kernel for everything (development knob)
        static const bool lean = [] { const char* e = getenv("EPA_POOL_LEAN"); return !(e && e[0] == '0'); }();
        uint8_t* todo = nullptr;
        const long long nwork = ngroups * (long long)grid.y;
        const long long nlean = (long long)C * ((P + kLeanRows - 1) / kLeanRows) * (long long)grid.y;
        if (lean && nwork <= (long long)(N * sizeof(double))) {
          todo = reinterpret_cast<uint8_t*>(rl);
          EPA_CHECK_HIP(hipMemsetAsync(todo, 0, (size_t)nwork, st));
          hipLaunchKernelGGL(pool_value_mean_lean_kernel<T>, dim3((unsigned)std::min<long long>(nlean, 1ll << 30)),
                             dim3(kBlock), 0, st, a, C, wh, wl, wn, dirty, differ, (int)grid.y,
                             epa::xcd_map_enabled() ? 1 : 0, todo, elig);
          if (int rc = epa::check_launch("pool_value_mean_lean_kernel")) return rc;
        }
        hipLaunchKernelGGL(pool_value_mean_staged_kernel<T>, sgrid, dim3(kBlock), 0, st, a, C, wh, wl, wn, dirty, differ,
                           todo, elig);
      }
      return epa::check_launch("pool_value_mean_staged_kernel");
    }
    hipLaunchKernelGGL(pool_value_mean_kernel<T>, grid, dim3(kBlock), 0, st, a, rows);
    return epa::check_launch("pool_value_mean_kernel");
  }
  if (ws && (long long)(2 * n + 1) * 2 * 4 <= (long long)kMedValueCap * 2) {
    // window carried from ping to ping in the channels whose pings share one range vector
    int* ilo = static_cast<int*>(ws);
    int* ihi = ilo + (size_t)C * S;
    int* ref = ihi + (size_t)C * S;
    int* differ = ref + C;
    const long long rows = (long long)C * P;
    const dim3 rowg(row_grid(rows) < 16384 ? row_grid(rows) : 16384);
    hipLaunchKernelGGL(ref_row_kernel, dim3(C), dim3(kBlock), 0, st, nvalid, P, ref, differ);
    hipLaunchKernelGGL(rows_same_kernel<T>, rowg, dim3(kBlock), 0, st, (const T*)range, nvalid, rows, P, S, ref, differ);
    hipLaunchKernelGGL(value_intervals_kernel<T>, dim3((S + kBlock - 1) / kBlock, C), dim3(kBlock), 0, st, a, ref, ilo, ihi);
    if (int rc = epa::check_launch("value_intervals_kernel<median>")) return rc;
    MedSlideArgs<T> ma{};
    ma.sv = (const T*)sv; ma.P = P; ma.S = S; ma.n = n; ma.nseg = (P + kMedSeg - 1) / kMedSeg;
    ma.jobs = (long long)C * ma.nseg * S; ma.thr = (T)thr; ma.pooled = (T*)pooled; ma.mask = mask;
    ma.cap = kMedValueCap; ma.pv = a; ma.ilo = ilo; ma.ihi = ihi; ma.differ = differ;
    const size_t lds = med_slide_lds(kMedValueCap);
    auto kern = pool_median_slide_kernel<T, true>;
    if (int rc = set_lds(kern, lds)) return rc;
    const int grid = (int)(ma.jobs < (1 << 22) ? ma.jobs : (1 << 22));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, ma);
    return epa::check_launch("pool_value_median_slide_kernel");
  }
  const long long jobs = (long long)C * P * S;
  const int grid = (int)(jobs < (1 << 20) ? jobs : (1 << 20));
  hipLaunchKernelGGL(pool_value_median_kernel<T>, dim3(grid), dim3(kBlock), 0, st, a, jobs);
  return epa::check_launch("pool_value_median_kernel");
}
}  // namespace

extern "C" int epa_pool_sv_value(const void* sv, const void* range, const int32_t* nvalid, int C, int P,
                                 int S, double depth_bin, int num_side_pings, double exclude_above,
                                 double range_min, double range_max, int func, double threshold,
                                 void* pooled_out, uint8_t* mask_out, void* ws, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(sv && range && nvalid && (pooled_out || mask_out), "epa_pool_sv_value: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_pool_sv_value: sizes must be positive");
  EPA_CHECK_ARG(num_side_pings >= 0, "epa_pool_sv_value: num_side_pings must be >= 0");
  EPA_CHECK_ARG(func == EPA_POOL_NANMEAN || func == EPA_POOL_NANMEDIAN, "epa_pool_sv_value: bad func %d", func);
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_pool_sv_value: bad dtype %d", dtype);
  if (func == EPA_POOL_NANMEDIAN && num_side_pings > kMaxSidePings) {
    epa::set_error("epa_pool_sv_value: nanmedian supports at most %d side pings", kMaxSidePings);
    return EPA_EUNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EPA_F64)
    return launch_pool_value<double>(sv, range, nvalid, C, P, S, depth_bin, num_side_pings, exclude_above,
                                     range_min, range_max, func, threshold, pooled_out, mask_out, ws, st);
  return launch_pool_value<float>(sv, range, nvalid, C, P, S, depth_bin, num_side_pings, exclude_above,
                                  range_min, range_max, func, threshold, pooled_out, mask_out, ws, st);
}

extern "C" int epa_attenuated_mask(const void* sv, const void* range, int C, int P, int S,
                                   double upper_limit, double lower_limit, int num_side_pings,
                                   double threshold, uint8_t* mask_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(sv && range && mask_out, "epa_attenuated_mask: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_attenuated_mask: sizes must be positive");
  EPA_CHECK_ARG(num_side_pings >= 0, "epa_attenuated_mask: num_side_pings must be >= 0");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_attenuated_mask: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (num_side_pings >= 1 && S >= 16 && S % 4 == 0 && (reinterpret_cast<uintptr_t>(mask_out) & 3u) == 0) {
    // block medians carried from ping to ping: layer limits and own median of every ping first (stashed in the mask
    // rows), then the walk
    const long long rows = (long long)C * P;
    // pings per workgroup: every chunk pays a rebuild of its 2n-ping block, and the workgroups run in rounds of two
    // per CU -- the length in [64, kAttWalkChunkMax] with the fewest sequential steps (rounds x (length + 2n)) is taken
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    const long long slots = 2ll * std::max(cus, 1);
    int chunk_len = std::min(P, kAttWalkChunkMax);
    long long best = -1;
    for (int len = std::min(P, kAttWalkChunkMax); len >= std::min(P, 64); --len) {
      const long long wgs = (long long)C * ((P + len - 1) / len);
      const long long steps = ((wgs + slots - 1) / slots) * (len + 2ll * num_side_pings);
      if (best < 0 || steps < best) {
        best = steps;
        chunk_len = len;
      }
    }
    const int nchunks = (P + chunk_len - 1) / chunk_len;
    const int ring_cap = 20480;  // u16 bins: 2 n x layer length up to this, longer blocks take both medians from memory
    const size_t wl = epa::kMathTabBytes + (size_t)(kMedBins + kMedCoarse + 4 + 8) * 4 + 2 * kMedSel * 2 + (size_t)ring_cap * 2;
    const dim3 grid((unsigned)((long long)C * nchunks));
#define EPA_AW(T)                                                                                                     \
  do {                                                                                                                \
    hipLaunchKernelGGL(attenuated_prepare_kernel<T>, dim3(row_grid(rows)), dim3(kBlock), 0, st, (const T*)sv,         \
                       (const T*)range, P, S, rows, (T)upper_limit, (T)lower_limit, num_side_pings, mask_out);        \
    if (int rc = epa::check_launch("attenuated_prepare_kernel")) return rc;                                           \
    hipLaunchKernelGGL(attenuated_walk_kernel<T>, grid, dim3(kBlock), wl, st, (const T*)sv, P, S, nchunks, chunk_len, \
                       num_side_pings, (T)threshold, ring_cap, mask_out);                                             \
  } while (0)
    if (dtype == EPA_F64) EPA_AW(double); else EPA_AW(float);
#undef EPA_AW
    return epa::check_launch("attenuated_walk_kernel");
  }
  const long long rows = (long long)C * P;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(attenuated_mask_kernel<double>, dim3(row_grid(rows)), dim3(kBlock), 0, st,
                       (const double*)sv, (const double*)range, P, S, rows, upper_limit, lower_limit,
                       num_side_pings, threshold, mask_out);
  else
    hipLaunchKernelGGL(attenuated_mask_kernel<float>, dim3(row_grid(rows)), dim3(kBlock), 0, st,
                       (const float*)sv, (const float*)range, P, S, rows, (float)upper_limit,
                       (float)lower_limit, num_side_pings, (float)threshold, mask_out);
  return epa::check_launch("attenuated_mask_kernel");
}

extern "C" int epa_apply_mask(const void* src, const uint8_t* mask, size_t n, size_t mask_period,
                              double fill_value, const void* fill_array, size_t fill_period, void* out,
                              int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(src && mask && out, "epa_apply_mask: NULL array argument");
  EPA_CHECK_ARG(mask_period > 0 && n % mask_period == 0, "epa_apply_mask: mask does not tile the source");
  EPA_CHECK_ARG(!fill_array || (fill_period > 0 && n % fill_period == 0),
                "epa_apply_mask: fill array does not tile the source");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_apply_mask: bad dtype %d", dtype);
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(apply_mask_kernel<double>, dim3(grid), dim3(kBlock), 0, st, (const double*)src,
                       mask, n, mask_period, fill_value, (const double*)fill_array,
                       fill_array ? fill_period : 1, (double*)out);
  else
    hipLaunchKernelGGL(apply_mask_kernel<float>, dim3(grid), dim3(kBlock), 0, st, (const float*)src, mask,
                       n, mask_period, (float)fill_value, (const float*)fill_array,
                       fill_array ? fill_period : 1, (float*)out);
  return epa::check_launch("apply_mask_kernel");
}

extern "C" int epa_apply_masks(const void* src, const uint8_t* const* masks, const size_t* mask_periods, int n_masks,
                               size_t n, double fill_value, const void* fill_array, size_t fill_period, void* out,
                               double* workspace, double* minmax_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(src && masks && mask_periods && out, "epa_apply_masks: NULL array argument");
  EPA_CHECK_ARG(n_masks >= 1 && n_masks <= 4, "epa_apply_masks: 1 to 4 masks (got %d)", n_masks);
  MaskSet ms{};
  ms.n = n_masks;
  for (int k = 0; k < n_masks; ++k) {
    EPA_CHECK_ARG(masks[k] != nullptr, "epa_apply_masks: mask %d is NULL", k);
    EPA_CHECK_ARG(mask_periods[k] > 0 && n % mask_periods[k] == 0, "epa_apply_masks: mask %d does not tile the source", k);
    ms.m[k] = masks[k];
    ms.period[k] = mask_periods[k];
  }
  EPA_CHECK_ARG(!fill_array || (fill_period > 0 && n % fill_period == 0),
                "epa_apply_masks: fill array does not tile the source");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_apply_masks: bad dtype %d", dtype);
  EPA_CHECK_ARG(!minmax_out || workspace, "epa_apply_masks: minmax_out needs the workspace");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < EPA_APPLY_MASKS_WS_DOUBLES / 2 ? blocks : EPA_APPLY_MASKS_WS_DOUBLES / 2);
  hipStream_t st = (hipStream_t)stream;
  double* part = minmax_out ? workspace : nullptr;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(apply_masks_kernel<double>, dim3(grid), dim3(kBlock), 0, st, (const double*)src, ms, n,
                       fill_value, (const double*)fill_array, fill_array ? fill_period : 1, (double*)out, part);
  else
    hipLaunchKernelGGL(apply_masks_kernel<float>, dim3(grid), dim3(kBlock), 0, st, (const float*)src, ms, n,
                       (float)fill_value, (const float*)fill_array, fill_array ? fill_period : 1, (float*)out, part);
  if (int rc = epa::check_launch("apply_masks_kernel")) return rc;
  if (minmax_out) {
    hipLaunchKernelGGL(minmax_pairs_kernel, dim3(1), dim3(kBlock), 0, st, part, grid, minmax_out);
    return epa::check_launch("minmax_pairs_kernel");
  }
  return EPA_OK;
}

extern "C" int epa_mask_and(const uint8_t* a, const uint8_t* b, size_t n, size_t b_period, uint8_t* out,
                            epa_stream_t stream) {
  EPA_CHECK_ARG(a && b && out, "epa_mask_and: NULL array argument");
  EPA_CHECK_ARG(b_period > 0 && n % b_period == 0, "epa_mask_and: second mask does not tile the first");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL(mask_and_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, b, n,
                     b_period, out);
  return epa::check_launch("mask_and_kernel");
}
