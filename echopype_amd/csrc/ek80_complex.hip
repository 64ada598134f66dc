// K3+K4: EK80 complex samples -- matched-filter pulse compression (BB) fused with the
// sector mean -> received power -> Sv/TS chain.  CW complex data take the same kernel with no
// replica (taps = 0).
//
// Reference arithmetic replaced (paths under /root/reference/echopype/calibrate):
// (replicas of 16 .. 1024 taps normally go through the LDS-FFT form of ek80_fft.hip; this direct form serves
// short and very long replicas, CW, and is the cross-check of the FFT form)
//   ek80_complex.py:285-369  compress_pulse: NaN -> 0, per (ping, sector) convolution with
//                            flipud(conj(tx)) cropped at m-1  ==  y[k] = sum_j x[k+j] * conj(tx[j]),
//                            NaN restored at input-NaN positions
//   ek80_complex.py:372-391  norm factor ||tx||^2 (here: wavefront __shfl reduction in-kernel)
//   calibrate_ek.py:483-490  prx = B*|mean_sector(y/||tx||^2)|^2 / 8 * (|z_er+z_et|/z_er)^2 / z_et
//   calibrate_ek.py:571-638  prx<=0 -> NaN; Sv/TS chain;  range.py:138-148,180-199 ranges
//
// Design.  One workgroup (256 lanes) produces a tile of 2048 consecutive range samples of one
// (channel, ping).  The matched filter is linear and the reference averages the sectors right
// after it, so the tile is staged ONCE as the sector SUM (NaN zero-filled) plus a per-sample
// validity bitmask: B x fewer MACs.  That is exact when all sectors share the NaN pattern at a
// sample (end-of-ping padding, multiplexed pings); a tile that contains a sample with a MIXED
// pattern falls back to one convolution per sector (block-uniform branch).  The replica (conj
// applied) sits in LDS; each lane owns 8 consecutive outputs and slides a 16-element register
// window over the staged tile (LDS rows padded 9/8 -> conflict-free ds_read_b64 / b128).
// Not a GEMM, no MFMA (BASELINE north_star); vector-FP32/FP64 FMA bound (8*m flops / sample).
#include "fast_math.h"

namespace {

#ifndef EPA_EK80_R
#define EPA_EK80_R 8
#endif
#ifndef EPA_EK80_MIN_WAVES
#define EPA_EK80_MIN_WAVES 1
#endif
constexpr int kR = EPA_EK80_R;             // outputs per lane (8 or 4)
constexpr int kRShift = kR == 8 ? 3 : 2;
constexpr int kTile = epa::kBlock * kR;    // 2048 outputs per workgroup
constexpr int kMaxBeams = 8;

template <typename A>
struct Cx {
  A re, im;
};

__device__ __forceinline__ int pad_idx(int a) { return a + (a >> kRShift); }  // one pad element per kR

// R' = R - shift decides the R' <= 0 guard: when the two are equal (tau = 2^k sample intervals) the
// difference must be the rounded difference of the rounded range, not the fused residue of its product
// (HIP compiles with -ffp-contract=fast: the backend fuses across statements, so the rounded range is
// pinned in a register with an empty asm before the subtraction)
__device__ __forceinline__ double sub_rn(double a, double b) {
  asm volatile("" : "+v"(a));
  return a - b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
  asm volatile("" : "+v"(a));
  return a - b;
}

struct CxArgs {
  const void* re;
  const void* im;
  const float* replica;        // interleaved (re, im) f32
  const int32_t* replica_off;  // [n_replicas + 1] in complex elements, or NULL (CW)
  const int32_t* replica_id;   // optional [C*P]: the replica of ping (c, p) (a file with several filter_time intervals:
                               // one replica per (channel, interval)); NULL: replica c for every ping of channel c
  const double* ccoef;
  int C, P, S, B;
  int tiles;
  double nspread;
  void* out;
  void* range_out;
  void* prx_out;
  unsigned rep_lds_off, mask_lds_off, tab_lds_off;  // byte offsets in dynamic LDS
  int stage_len;                       // kTile + max_taps - 1 (>= kTile)
  double* stats_part;                  // optional [3 * kCwStatSlots] {min, max, NaN count} of echo_range (CW kernel)
};
constexpr int kCwStatSlots = 1024;

// stage samples [k_begin, k_begin + len) of one ping: sector sum (or one sector when `only` >= 0)
// 16-byte vector loads of the NB sector values of one sample (NB * sizeof(InT) must be a multiple of
// 16 and the base 16-byte aligned: the launcher checks)
template <typename InT, int NB>
__device__ __forceinline__ void load_sectors(const InT* __restrict__ p, InT (&v)[NB]) {
  constexpr int kPer = 16 / sizeof(InT);
  typedef InT vec_t __attribute__((ext_vector_type(kPer)));
#pragma unroll
  for (int q = 0; q < NB / kPer; ++q) {
    const vec_t t = reinterpret_cast<const vec_t*>(p)[q];
#pragma unroll
    for (int e = 0; e < kPer; ++e) v[q * kPer + e] = t[e];
  }
}

template <typename InT, typename A, int NB>
__device__ __forceinline__ unsigned stage_tile(const InT* __restrict__ re, const InT* __restrict__ im,
                                               size_t ping_base, int S, int Brt, int k_begin, int len,
                                               int only, Cx<A>* xs, uint8_t* vmask) {
  unsigned mixed = 0;
  const int B = NB > 0 ? NB : Brt;
  const unsigned full = (1u << B) - 1u;
  for (int t = threadIdx.x; t < len; t += epa::kBlock) {
    const int s = k_begin + t;
    A sr = (A)0, si = (A)0;
    unsigned m = 0;
    if (s < S) {
      const InT* pr = re + ping_base + (size_t)s * B;
      const InT* pi = im + ping_base + (size_t)s * B;
      InT vrs[NB > 0 ? NB : 1], vis[NB > 0 ? NB : 1];
      if (NB > 0) {
        load_sectors<InT, (NB > 0 ? NB : 4)>(pr, reinterpret_cast<InT(&)[NB > 0 ? NB : 4]>(vrs));
        load_sectors<InT, (NB > 0 ? NB : 4)>(pi, reinterpret_cast<InT(&)[NB > 0 ? NB : 4]>(vis));
      }
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const InT vr = NB > 0 ? vrs[NB > 0 ? b : 0] : pr[b], vi = NB > 0 ? vis[NB > 0 ? b : 0] : pi[b];
        const bool ok = (vr == vr) && (vi == vi);
        if (ok) {
          m |= 1u << b;
          if (only < 0 || only == b) {
            sr += (A)vr;
            si += (A)vi;
          }
        }
      }
      // second mask byte, bit 0: beam-0 real part valid (echo_range mask, range.py:143-146)
      const InT r0v = NB > 0 ? vrs[0] : pr[0];
      if (r0v == r0v) m |= 0x100u;
    }
    xs[pad_idx(t)] = Cx<A>{sr, si};
    if (only < 0) {
      vmask[2 * t] = (uint8_t)(m & 0xffu);
      vmask[2 * t + 1] = (uint8_t)(m >> 8);
      if ((m & full) != 0u && (m & full) != full) mixed = 1u;
    }
  }
  return mixed;
}

// y[i] = sum_j x[k0+i+j] * conj(rep[j]), i = 0..7, for this lane's 8 outputs
template <typename A>
__device__ __forceinline__ void conv8(const Cx<A>* __restrict__ xs, const Cx<A>* __restrict__ rep,
                                      int taps, int k0, Cx<A> (&acc)[kR]) {
  Cx<A> w[2 * kR];
#pragma unroll
  for (int e = 0; e < kR; ++e) w[e] = xs[pad_idx(k0 + e)];
  for (int q = 0; q < taps; q += kR) {
#pragma unroll
    for (int e = 0; e < kR; ++e) w[kR + e] = xs[pad_idx(k0 + q + kR + e)];
#pragma unroll
    for (int jj = 0; jj < kR; ++jj) {
      // rep is zero-padded to a multiple of 8 taps in LDS
      const Cx<A> t = rep[q + jj];
#pragma unroll
      for (int i = 0; i < kR; ++i) {
        const Cx<A> x = w[i + jj];
        acc[i].re = fma(x.re, t.re, acc[i].re);
        acc[i].re = fma(x.im, t.im, acc[i].re);
        acc[i].im = fma(x.im, t.re, acc[i].im);
        acc[i].im = fma(-x.re, t.im, acc[i].im);
      }
    }
#pragma unroll
    for (int e = 0; e < kR; ++e) w[e] = w[kR + e];
  }
}

template <typename InT, typename T, typename A, int NB>
__global__ __launch_bounds__(epa::kBlock, EPA_EK80_MIN_WAVES) void sv_complex_kernel(CxArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Cx<A>* xs = reinterpret_cast<Cx<A>*>(smem);
  Cx<A>* rep = reinterpret_cast<Cx<A>*>(smem + a.rep_lds_off);
  uint8_t* vmask = smem + a.mask_lds_off;
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_lds_off);  // synchronised before use below
  __shared__ A red[8];

  const int c = blockIdx.y;
  const int p = blockIdx.x / a.tiles;
  const int tile = blockIdx.x - p * a.tiles;
  const int S = a.S, B = a.B;
  const int k_begin = tile * kTile;
  const InT* re = reinterpret_cast<const InT*>(a.re);
  const InT* im = reinterpret_cast<const InT*>(a.im);
  const size_t ping_base = ((size_t)c * a.P + p) * (size_t)S * B;

  // ---- replica -> LDS (conj is applied in the MAC), zero-padded to a multiple of 8 taps;
  //      ||tx||^2 by wavefront shuffle reduction
  int taps = 0;
  A norm2 = (A)1;
  if (a.replica) {
    // (a ping no interval covers has id -1 and a NaN coefficient row: whatever it is convolved with, its output is NaN)
    const int rid = a.replica_id ? max(a.replica_id[(size_t)c * a.P + p], 0) : c;
    const int r0 = a.replica_off[rid], r1 = a.replica_off[rid + 1];
    taps = r1 - r0;
    const int taps8 = (taps + kR - 1) / kR * kR;
    A part = (A)0;
    for (int j = threadIdx.x; j < taps8; j += epa::kBlock) {
      Cx<A> t{(A)0, (A)0};
      if (j < taps) {
        t.re = (A)a.replica[2 * (size_t)(r0 + j)];
        t.im = (A)a.replica[2 * (size_t)(r0 + j) + 1];
      }
      rep[j] = t;
      part += t.re * t.re + t.im * t.im;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    norm2 = red[0] + red[1] + red[2] + red[3];
  }
  const int taps8 = (taps + kR - 1) / kR * kR;
  // staged span: outputs [k_begin, k_begin+kTile) need inputs up to k_begin + kTile + taps8 - 1 (+8 lookahead)
  const int len = kTile + (taps8 > 0 ? taps8 + kR : 0);

  const unsigned mixed_l = stage_tile<InT, A, NB>(re, im, ping_base, S, B, k_begin, len, -1, xs, vmask);
  const int mixed = __syncthreads_or((int)mixed_l);

  const int k0 = threadIdx.x * kR;  // tile-local first output of this lane
  Cx<A> y[kR];
  unsigned nvalid[kR];
  const unsigned full = (1u << B) - 1u;
#pragma unroll
  for (int i = 0; i < kR; ++i) {
    y[i] = Cx<A>{(A)0, (A)0};
    nvalid[i] = __popc(vmask[2 * (k0 + i)] & full);
  }
  if (!mixed) {
    if (taps == 0) {
#pragma unroll
      for (int i = 0; i < kR; ++i) y[i] = xs[pad_idx(k0 + i)];
    } else {
      conv8<A>(xs, rep, taps8, k0, y);
    }
  } else {
    // per-sector convolutions; a sector contributes to output k only where it is valid at k
    for (int b = 0; b < B; ++b) {
      __syncthreads();
      stage_tile<InT, A, NB>(re, im, ping_base, S, B, k_begin, len, b, xs, vmask);
      __syncthreads();
      Cx<A> yb[kR];
#pragma unroll
      for (int i = 0; i < kR; ++i) yb[i] = Cx<A>{(A)0, (A)0};
      if (taps == 0) {
#pragma unroll
        for (int i = 0; i < kR; ++i) yb[i] = xs[pad_idx(k0 + i)];
      } else {
        conv8<A>(xs, rep, taps8, k0, yb);
      }
#pragma unroll
      for (int i = 0; i < kR; ++i) {
        if (vmask[2 * (k0 + i)] & (1u << b)) {
          y[i].re += yb[i].re;
          y[i].im += yb[i].im;
        }
      }
    }
  }

  // ---- sector mean -> prx -> Sv/TS (calibrate_ek.py:483-490, 571-638)
  const size_t row = (size_t)c * a.P + p;
  const double* cc = a.ccoef + row * EPA_NCCOEF;
  const double ra = cc[EPA_CC_RA], rb = cc[EPA_CC_RB];
  const T shift = (T)cc[EPA_CC_SHIFT], alpha2 = (T)cc[EPA_CC_ALPHA2], Aadd = (T)cc[EPA_CC_A];
  const T pscale = (T)(cc[EPA_CC_PSCALE]);
  const T nspread = (T)a.nspread;
  const T inv_norm = (T)1 / (T)norm2;
  const epa::LogCoef lk = epa::make_log_coef();
  T* out = reinterpret_cast<T*>(a.out);
  T* range_out = reinterpret_cast<T*>(a.range_out);
  T* prx_out = reinterpret_cast<T*>(a.prx_out);
#pragma unroll
  for (int i = 0; i < kR; ++i) {
    const int s = k_begin + k0 + i;
    if (s >= S) break;
    T mr, mi;
    if (nvalid[i] == 0u) {
      mr = mi = epa::M<T>::nan();
    } else {
      const T invn = inv_norm / (T)nvalid[i];
      mr = (T)y[i].re * invn;
      mi = (T)y[i].im * invn;
    }
    T prx = pscale * (mr * mr + mi * mi);
    if (!(prx > (T)0)) prx = epa::M<T>::nan();
    const double R = ((double)s * ra) * rb;  // range.py:138 operation order
    T rt = sub_rn((T)R, shift);  // never contracted with the range product into an fma
    if (!(rt > (T)0)) rt = epa::M<T>::nan();
    // the range the reference calibrates with is the MASKED echo_range (NaN where beam 0 is, range.py:143-148): a
    // sample whose beam 0 is missing is NaN even when its other sectors are valid (calibrate_ek.py:571-576)
    const bool range_ok = (vmask[2 * (k0 + i) + 1] & 1u) != 0u;
    if (!range_ok) rt = epa::M<T>::nan();
    // prx and rt are positive or NaN here: the lean log (zero / subnormal / inf / NaN through a rare branch)
    const T val = (T)10 * epa::fast_log10_lean(prx, mt.log_tab, lk) + nspread * epa::fast_log10_lean(rt, mt.log_tab, lk) + alpha2 * rt + Aadd;
    const size_t o = row * S + s;
    out[o] = val;
    if (range_out) range_out[o] = range_ok ? (T)R : epa::M<T>::nan();
    if (prx_out) prx_out[o] = prx;
  }
}

// ------------------------------------------------------------------------------------------------
// CW (no replica): nothing is convolved, a sample depends on its own sectors only -- a plain streaming kernel.  Lane j
// of a workgroup takes the samples j + 256 i of a 2048- (1024-) sample piece of one ping, so that every load of a wavefront is
// 1 KiB (float32 planes) contiguous per plane and every store 512 B (the tile kernel above gives a lane eight CONSECUTIVE outputs, which
// its sliding window needs and which makes each of its stores touch 64 separate 64-byte segments).  The NaN-skipping
// sector mean (calibrate_ek.py:483, xarray's skipna) needs no second pass here.
// ------------------------------------------------------------------------------------------------
// samples of a ping per workgroup: eight per lane for float32 planes, four for float64 planes (the lane holds all of
// its sectors in registers: 64 VGPRs either way)
template <typename InT>
constexpr int cw_piece() { return sizeof(InT) == 8 ? 1024 : 2048; }

template <typename InT, typename T, int NB, bool STATS>
__global__ __launch_bounds__(epa::kBlock) void sv_complex_cw_kernel(CxArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  const int c = blockIdx.y;
  const int p = blockIdx.x / a.tiles, piece = blockIdx.x - p * a.tiles;
  const int S = a.S, B = NB > 0 ? NB : a.B;
  const InT* re = reinterpret_cast<const InT*>(a.re);
  const InT* im = reinterpret_cast<const InT*>(a.im);
  const size_t row = (size_t)c * a.P + p;
  const size_t ping_base = row * (size_t)S * B;
  const double* cc = a.ccoef + row * EPA_NCCOEF;
  const double ra = cc[EPA_CC_RA], rb = cc[EPA_CC_RB];
  const T shift = (T)cc[EPA_CC_SHIFT], alpha2 = (T)cc[EPA_CC_ALPHA2], Aadd = (T)cc[EPA_CC_A];
  const T pscale = (T)(cc[EPA_CC_PSCALE]);
  const T nspread = (T)a.nspread;
  const epa::LogCoef lk = epa::make_log_coef();
  T* out = reinterpret_cast<T*>(a.out);
  T* range_out = reinterpret_cast<T*>(a.range_out);
  T* prx_out = reinterpret_cast<T*>(a.prx_out);
  constexpr int kCwPiece = cw_piece<InT>();
  constexpr int kPer = kCwPiece / epa::kBlock;
  double rmin = __builtin_inf(), rmax = -__builtin_inf();  // statistics of the echo_range (written or not)
  int rnan = 0;
  // all loads of the lane first (independent), then the arithmetic
  InT vr[kPer][NB > 0 ? NB : 1], vi[kPer][NB > 0 ? NB : 1];
  if (NB > 0) {
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int s = min(piece * kCwPiece + (int)threadIdx.x + epa::kBlock * i, S - 1);  // (clamped: no partial arrays)
      load_sectors<InT, (NB > 0 ? NB : 4)>(re + ping_base + (size_t)s * B, reinterpret_cast<InT(&)[NB > 0 ? NB : 4]>(vr[i]));
      load_sectors<InT, (NB > 0 ? NB : 4)>(im + ping_base + (size_t)s * B, reinterpret_cast<InT(&)[NB > 0 ? NB : 4]>(vi[i]));
    }
  }
#pragma unroll
  for (int i = 0; i < kPer; ++i) {
    const int s = piece * kCwPiece + (int)threadIdx.x + epa::kBlock * i;
    if (s >= S) break;
    T sr = (T)0, si = (T)0;
    unsigned nvalid = 0;
    bool range_ok;
    if (NB > 0) {
#pragma unroll
      for (int b = 0; b < (NB > 0 ? NB : 1); ++b) {
        const InT xr = vr[i][b], xi = vi[i][b];
        const bool ok = (xr == xr) && (xi == xi);
        sr += ok ? (T)xr : (T)0;
        si += ok ? (T)xi : (T)0;
        nvalid += ok ? 1u : 0u;
      }
      range_ok = vr[i][0] == vr[i][0];
    } else {
      const InT* pr = re + ping_base + (size_t)s * B;
      const InT* pi = im + ping_base + (size_t)s * B;
      for (int b = 0; b < B; ++b) {
        const InT xr = pr[b], xi = pi[b];
        const bool ok = (xr == xr) && (xi == xi);
        sr += ok ? (T)xr : (T)0;
        si += ok ? (T)xi : (T)0;
        nvalid += ok ? 1u : 0u;
      }
      range_ok = pr[0] == pr[0];
    }
    T mr, mi;
    if (nvalid == 0u) {
      mr = mi = epa::M<T>::nan();
    } else {
      const T invn = (T)1 / (T)nvalid;
      mr = sr * invn;
      mi = si * invn;
    }
    T prx = pscale * (mr * mr + mi * mi);
    if (!(prx > (T)0)) prx = epa::M<T>::nan();
    const double R = ((double)s * ra) * rb;  // range.py:138 operation order
    T rt = sub_rn((T)R, shift);
    if (!(rt > (T)0) || !range_ok) rt = epa::M<T>::nan();  // the masked echo_range calibrates (see the tile kernel)
    const T val = (T)10 * epa::fast_log10_lean(prx, mt.log_tab, lk) + nspread * epa::fast_log10_lean(rt, mt.log_tab, lk) +
                  alpha2 * rt + Aadd;
    const size_t o = row * S + s;
    out[o] = val;
    if (range_out) range_out[o] = range_ok ? (T)R : epa::M<T>::nan();
    if (prx_out) prx_out[o] = prx;
    if (STATS) {
      if (range_ok) {
        const double rr = (double)(T)R;
        rmin = fmin(rmin, rr);
        rmax = fmax(rmax, rr);
      } else {
        ++rnan;
      }
    }
  }
  if (STATS) {  // one merge per workgroup into one of kCwStatSlots slots (f64 atomics)
    __shared__ double sred[12];
    double cnt = (double)rnan;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      rmin = fmin(rmin, __shfl_down(rmin, o, 64));
      rmax = fmax(rmax, __shfl_down(rmax, o, 64));
      cnt += __shfl_down(cnt, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      sred[3 * wave] = rmin;
      sred[3 * wave + 1] = rmax;
      sred[3 * wave + 2] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* dst = a.stats_part + 3 * ((row * a.tiles + piece) & (size_t)(kCwStatSlots - 1));
      const double mn = fmin(fmin(sred[0], sred[3]), fmin(sred[6], sred[9]));
      const double mx = fmax(fmax(sred[1], sred[4]), fmax(sred[7], sred[10]));
      const double nn = (sred[2] + sred[5]) + (sred[8] + sred[11]);
      if (mn <= mx) {
        atomicMin(dst, mn);
        atomicMax(dst + 1, mx);
      }
      if (nn > 0.0) atomicAdd(dst + 2, nn);
    }
  }
}

__global__ __launch_bounds__(epa::kBlock) void cw_stats_init_kernel(double* __restrict__ sp) {
  for (int k = threadIdx.x; k < kCwStatSlots; k += epa::kBlock) {
    sp[3 * k] = __builtin_inf();
    sp[3 * k + 1] = -__builtin_inf();
    sp[3 * k + 2] = 0.0;
  }
}

template <typename InT, typename T>
int launch_cw(CxArgs& a, hipStream_t st) {
  a.tiles = (a.S + cw_piece<InT>() - 1) / cw_piece<InT>();
  const dim3 grid((unsigned)((long long)a.P * a.tiles), (unsigned)a.C);
  const bool b4 = a.B == 4 && (reinterpret_cast<uintptr_t>(a.re) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(a.im) & 15u) == 0;
  if (a.stats_part) {
    if (b4) hipLaunchKernelGGL((sv_complex_cw_kernel<InT, T, 4, true>), grid, dim3(epa::kBlock), 0, st, a);
    else hipLaunchKernelGGL((sv_complex_cw_kernel<InT, T, 0, true>), grid, dim3(epa::kBlock), 0, st, a);
  } else {
    if (b4) hipLaunchKernelGGL((sv_complex_cw_kernel<InT, T, 4, false>), grid, dim3(epa::kBlock), 0, st, a);
    else hipLaunchKernelGGL((sv_complex_cw_kernel<InT, T, 0, false>), grid, dim3(epa::kBlock), 0, st, a);
  }
  return epa::check_launch("sv_complex_cw_kernel");
}

template <typename InT, typename T, typename A>
int launch(CxArgs& a, int max_taps, hipStream_t st) {
  const int taps8 = (max_taps + kR - 1) / kR * kR;
  const int len = kTile + (taps8 > 0 ? taps8 + kR : 0);
  const size_t xs_bytes = ((size_t)(len + (len >> kRShift) + 1) * sizeof(Cx<A>) + 15) & ~(size_t)15;
  const size_t rep_bytes = ((size_t)(taps8 > 0 ? taps8 : kR) * sizeof(Cx<A>) + 15) & ~(size_t)15;
  const size_t mask_bytes = (size_t)2 * len;
  a.rep_lds_off = (unsigned)xs_bytes;
  a.mask_lds_off = (unsigned)(xs_bytes + rep_bytes);
  a.tab_lds_off = (unsigned)((xs_bytes + rep_bytes + mask_bytes + 15) & ~(size_t)15);
  const size_t lds = a.tab_lds_off + epa::kMathTabBytes;
  EPA_CHECK_ARG(lds <= 150 * 1024, "epa_sv_complex: replica of %d taps does not fit the LDS tile",
                max_taps);
  a.tiles = (a.S + kTile - 1) / kTile;
  const dim3 grid((unsigned)((long long)a.P * a.tiles), (unsigned)a.C);
  // four sectors (the usual split-beam transducer) with 16-byte aligned planes: vector loads
  const bool b4 = a.B == 4 && (reinterpret_cast<uintptr_t>(a.re) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(a.im) & 15u) == 0;
#define EPA_CX(NBV)                                                                              \
  do {                                                                                           \
    auto kern = sv_complex_kernel<InT, T, A, NBV>;                                               \
    if (lds > 64 * 1024)                                                                         \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds, st, a);                               \
  } while (0)
  if (b4) EPA_CX(4); else EPA_CX(0);
#undef EPA_CX
  return epa::check_launch("sv_complex_kernel");
}

}  // namespace

static int sv_complex_entry(const void* re, const void* im, int in_dtype, const float* replica,
                            const int32_t* replica_off, const int32_t* replica_id, int max_taps, const double* ccoef, int C,
                            int P, int S, int B, int cal_type, void* out, void* range_out,
                            void* prx_out, int out_dtype, double* stats_part, epa_stream_t stream) {
  EPA_CHECK_ARG(re && im && ccoef && out, "epa_sv_complex: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && B > 0, "epa_sv_complex: C=%d P=%d S=%d B=%d", C, P, S, B);
  EPA_CHECK_ARG(B <= kMaxBeams, "epa_sv_complex: at most %d sectors supported (got %d)", kMaxBeams, B);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_complex: bad cal_type");
  EPA_CHECK_ARG((replica == nullptr) == (replica_off == nullptr),
                "epa_sv_complex: replica and replica_off must both be given (BB) or both NULL (CW)");
  EPA_CHECK_ARG(!replica || max_taps > 0, "epa_sv_complex: max_taps must be positive for BB");
  CxArgs a{};
  a.re = re; a.im = im; a.replica = replica; a.replica_off = replica_off; a.replica_id = replica_id; a.ccoef = ccoef;
  a.C = C; a.P = P; a.S = S; a.B = B;
  a.nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  a.out = out; a.range_out = range_out; a.prx_out = prx_out;
  a.stats_part = stats_part;
  const int taps = replica ? max_taps : 0;
  hipStream_t st = (hipStream_t)stream;
  if (!replica) {  // CW: the streaming kernel
    if (in_dtype == EPA_F64 && out_dtype == EPA_F64) return launch_cw<double, double>(a, st);
    if (in_dtype == EPA_F64 && out_dtype == EPA_F32) return launch_cw<double, float>(a, st);
    if (in_dtype == EPA_F32 && out_dtype == EPA_F64) return launch_cw<float, double>(a, st);
    if (in_dtype == EPA_F32 && out_dtype == EPA_F32) return launch_cw<float, float>(a, st);
    epa::set_error("epa_sv_complex: bad dtype in=%d out=%d", in_dtype, out_dtype);
    return EPA_EINVAL;
  }
  if (in_dtype == EPA_F64 && out_dtype == EPA_F64) return launch<double, double, double>(a, taps, st);
  if (in_dtype == EPA_F64 && out_dtype == EPA_F32) return launch<double, float, float>(a, taps, st);
  if (in_dtype == EPA_F32 && out_dtype == EPA_F64) return launch<float, double, double>(a, taps, st);
  if (in_dtype == EPA_F32 && out_dtype == EPA_F32) return launch<float, float, float>(a, taps, st);
  epa::set_error("epa_sv_complex: bad dtype in=%d out=%d", in_dtype, out_dtype);
  return EPA_EINVAL;
}

extern "C" int epa_sv_complex(const void* re, const void* im, int in_dtype, const float* replica,
                              const int32_t* replica_off, int max_taps, const double* ccoef, int C,
                              int P, int S, int B, int cal_type, void* out, void* range_out,
                              void* prx_out, int out_dtype, epa_stream_t stream) {
  return sv_complex_entry(re, im, in_dtype, replica, replica_off, nullptr, max_taps, ccoef, C, P, S, B, cal_type, out,
                          range_out, prx_out, out_dtype, nullptr, stream);
}

extern "C" int epa_sv_complex_indexed(const void* re, const void* im, int in_dtype, const float* replica,
                                      const int32_t* replica_off, const int32_t* replica_id, int n_replicas,
                                      int max_taps, const double* ccoef, int C, int P, int S, int B, int cal_type,
                                      void* out, void* range_out, void* prx_out, int out_dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(replica && replica_off && replica_id && n_replicas > 0,
                "epa_sv_complex_indexed: replica, replica_off, replica_id and a positive n_replicas are needed");
  return sv_complex_entry(re, im, in_dtype, replica, replica_off, replica_id, max_taps, ccoef, C, P, S, B, cal_type, out,
                          range_out, prx_out, out_dtype, nullptr, stream);
}

extern "C" int epa_sv_complex_cw_stats(const void* re, const void* im, int in_dtype, const double* ccoef, int C, int P,
                                       int S, int B, int cal_type, void* out, void* range_out, void* prx_out,
                                       int out_dtype, double* workspace, double* range_stats_out,
                                       epa_stream_t stream) {
  EPA_CHECK_ARG(workspace && range_stats_out, "epa_sv_complex_cw_stats: NULL workspace / range_stats_out");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cw_stats_init_kernel, dim3(1), dim3(epa::kBlock), 0, st, workspace);
  if (int rc = epa::check_launch("cw_stats_init_kernel")) return rc;
  if (int rc = sv_complex_entry(re, im, in_dtype, nullptr, nullptr, nullptr, 0, ccoef, C, P, S, B, cal_type, out, range_out,
                                prx_out, out_dtype, workspace, stream))
    return rc;
  return epa_minmax_final(workspace, kCwStatSlots, range_stats_out, st);
}
