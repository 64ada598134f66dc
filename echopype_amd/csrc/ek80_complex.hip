// K3+K4: EK80 complex samples -- matched-filter pulse compression (BB) fused with the
// sector mean -> received power -> Sv/TS chain.  CW complex data take the same kernel with no
// replica (taps = 0).
//
// Reference arithmetic replaced (paths under /root/reference/echopype/calibrate):
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
  const int32_t* replica_off;  // [C+1] in complex elements, or NULL (CW)
  const double* ccoef;
  int C, P, S, B;
  int tiles;
  double nspread;
  void* out;
  void* range_out;
  void* prx_out;
  unsigned rep_lds_off, mask_lds_off, tab_lds_off;  // byte offsets in dynamic LDS
  int stage_len;                       // kTile + max_taps - 1 (>= kTile)
};

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
    const int r0 = a.replica_off[c], r1 = a.replica_off[c + 1];
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
    const T val = (T)10 * epa::fast_log10(prx, mt.log_tab) + nspread * epa::fast_log10(rt, mt.log_tab) + alpha2 * rt + Aadd;
    const size_t o = row * S + s;
    out[o] = val;
    if (range_out) range_out[o] = (vmask[2 * (k0 + i) + 1] & 1u) ? (T)R : epa::M<T>::nan();
    if (prx_out) prx_out[o] = prx;
  }
}

// ------------------------------------------------------------------------------------------------
// Long replicas: the same matched filter as a circular correlation of a 2048-sample tile through an
// LDS-resident FFT (what scipy.signal.convolve's method="auto" picks for these sizes in the
// reference, ek80_complex.py:334).  11x fewer flops than the 177-tap direct form: BB becomes
// HBM-bound like CW.  Always fp64, whatever the output type: FFT errors scale with the largest
// echo of the tile, not with the sample (a sea-floor echo next to 1e-12 water-column samples).
//   y[k] = sum_j x[k+j] conj(tx[j])  =  IFFT( FFT(x) * conj(FFT(tx)) )[k]   for k <= N - taps
// Stockham autosort, radix 8 x 8 x 8 x 4, in place (all reads of a pass complete before its writes),
// twiddles w^k from a 256-entry LDS table, w^(rk) by repeated complex multiplication.
// ------------------------------------------------------------------------------------------------
constexpr int kNfft = EPA_EK80_NFFT;
static_assert(kNfft == 2048 && epa::kBlock == 256 && kR == 8, "FFT path is written for N = 2048, 256 lanes");
typedef Cx<double> Cd;

__device__ __forceinline__ Cd cmul(Cd a, Cd b) {
  return Cd{fma(a.re, b.re, -a.im * b.im), fma(a.re, b.im, a.im * b.re)};
}
__device__ __forceinline__ Cd cadd(Cd a, Cd b) { return Cd{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cd csub(Cd a, Cd b) { return Cd{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cd mul_mi(Cd a) { return Cd{a.im, -a.re}; }  // * (-i)

__device__ __forceinline__ void dft4(Cd& u0, Cd& u1, Cd& u2, Cd& u3) {
  const Cd s02 = cadd(u0, u2), d02 = csub(u0, u2), s13 = cadd(u1, u3), d13 = mul_mi(csub(u1, u3));
  u0 = cadd(s02, s13);
  u2 = csub(s02, s13);
  u1 = cadd(d02, d13);
  u3 = csub(d02, d13);
}

__device__ __forceinline__ void dft8(Cd (&v)[8]) {
  constexpr double kH = 0.70710678118654752440;
  Cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  const Cd t1 = Cd{(o1.re + o1.im) * kH, (o1.im - o1.re) * kH};    // * e^{-i pi/4}
  const Cd t2 = mul_mi(o2);                                        // * e^{-i pi/2}
  const Cd t3 = Cd{(o3.im - o3.re) * kH, -(o3.re + o3.im) * kH};   // * e^{-3i pi/4}
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, t1); v[5] = csub(e1, t1);
  v[2] = cadd(e2, t2); v[6] = csub(e2, t2);
  v[3] = cadd(e3, t3); v[7] = csub(e3, t3);
}

// Forward DFT (e^{-2 pi i nk/N}) of the 2048 padded-LDS elements x[pad_idx(.)], natural order out.
__device__ void fft2048(Cd* x, const Cd* tw) {
  const int j = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const int Ns = pass == 0 ? 1 : (pass == 1 ? 8 : 64);
    const int k = j & (Ns - 1);
    Cd v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = x[pad_idx(j + r * (kNfft / 8))];
    if (pass > 0) {
      const Cd w1 = tw[k * (kNfft / (8 * Ns))];
      Cd w = w1;
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        v[r] = cmul(v[r], w);
        w = cmul(w, w1);
      }
    }
    dft8(v);
    __syncthreads();
    const int j0 = ((j - k) << 3) + k;
#pragma unroll
    for (int r = 0; r < 8; ++r) x[pad_idx(j0 + r * Ns)] = v[r];
    __syncthreads();
  }
  {  // radix 4, Ns = 512: two butterflies per lane
    Cd v[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = j + h * epa::kBlock;  // k == butterfly index < 512
#pragma unroll
      for (int r = 0; r < 4; ++r) v[h][r] = x[pad_idx(k + r * (kNfft / 4))];
      // 256-entry table: w^(j+256) = w^j * e^{-i pi/4}
      Cd w1 = tw[j];
      if (h == 1) w1 = Cd{(w1.re + w1.im) * 0.70710678118654752440, (w1.im - w1.re) * 0.70710678118654752440};
      const Cd w2 = cmul(w1, w1);
      v[h][1] = cmul(v[h][1], w1);
      v[h][2] = cmul(v[h][2], w2);
      v[h][3] = cmul(v[h][3], cmul(w2, w1));
      dft4(v[h][0], v[h][1], v[h][2], v[h][3]);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = j + h * epa::kBlock;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[pad_idx(k + r * (kNfft / 4))] = v[h][r];
    }
    __syncthreads();
  }
}

// workspace layout (doubles): [2*256 twiddles][per channel: ||tx||^2, first non-zero tap, one past the
// last non-zero tap, 0][2*C*N spectra conj(FFT(tx))/N]
__host__ __device__ inline size_t ws_chan_off() { return 2 * (kNfft / 8); }
__host__ __device__ inline size_t ws_spec_off(int C) { return ws_chan_off() + 4 * (size_t)C; }

__global__ __launch_bounds__(epa::kBlock) void replica_prepare_kernel(const float* __restrict__ replica,
                                                                      const int32_t* __restrict__ off,
                                                                      int C, double* __restrict__ ws) {
  __shared__ Cd xs[kNfft + kNfft / 8 + 1];
  __shared__ Cd tw[kNfft / 8];
  __shared__ double red[4];
  __shared__ int tap_lo, tap_hi;
  const int c = blockIdx.x;
  if (threadIdx.x == 0) {
    tap_lo = kNfft;
    tap_hi = 0;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < kNfft / 8; m += epa::kBlock) {
    double sn, cs;
    sincospi(-2.0 * (double)m / (double)kNfft, &sn, &cs);
    tw[m] = Cd{cs, sn};
    if (c == 0) reinterpret_cast<Cd*>(ws)[m] = tw[m];
  }
  const int r0 = off[c], taps = off[c + 1] - r0;
  double part = 0.0;
  for (int n = threadIdx.x; n < kNfft; n += epa::kBlock) {
    Cd t{0.0, 0.0};
    if (n < taps) {
      t.re = (double)replica[2 * (size_t)(r0 + n)];
      t.im = (double)replica[2 * (size_t)(r0 + n) + 1];
    }
    xs[pad_idx(n)] = t;
    part += t.re * t.re + t.im * t.im;
    if (t.re != 0.0 || t.im != 0.0) {  // tapered replicas start (and may end) with exact zeros
      atomicMin(&tap_lo, n);
      atomicMax(&tap_hi, n + 1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);  // ||tx||^2 (ek80_complex.py:372-391)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double* ch = ws + ws_chan_off() + 4 * (size_t)c;
    ch[0] = (red[0] + red[1]) + (red[2] + red[3]);
    ch[1] = (double)(tap_lo < tap_hi ? tap_lo : 0);
    ch[2] = (double)tap_hi;
    ch[3] = 0.0;
  }
  fft2048(xs, tw);
  Cd* spec = reinterpret_cast<Cd*>(ws + ws_spec_off(C)) + (size_t)c * kNfft;
  for (int f = threadIdx.x; f < kNfft; f += epa::kBlock) {
    const Cd X = xs[pad_idx(f)];
    spec[f] = Cd{X.re * (1.0 / kNfft), -X.im * (1.0 / kNfft)};
  }
}

// xs <- correlation of the staged tile with the channel's replica (valid for outputs 0 .. N - taps).
// The direct form returns an exact 0 where every staged sample under the non-zero taps [lo, hi) of the
// replica is 0 (NaN padding zero-filled, blanked samples, a sample met only by the zero first tap of
// a tapered chirp) and the chain turns prx == 0 into NaN; an FFT leaves rounding noise of the tile's
// strongest echo there.  To keep the NaN pattern, such outputs are set to 0: `nzp` (u16 [N]) holds the
// exclusive prefix count of non-zero staged samples.
__device__ __forceinline__ void fft_correlate(Cd* xs, const Cd* tw, const Cd* __restrict__ spec,
                                              unsigned short* nzp, unsigned* wave_tot, int lo, int hi,
                                              int n_out) {
  const int t0 = threadIdx.x * 8;
  unsigned loc[8], run = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const Cd v = xs[pad_idx(t0 + e)];
    loc[e] = run;
    run += (v.re != 0.0 || v.im != 0.0) ? 1u : 0u;
  }
  // the usual tile holds no zero sample at all: nothing to restore, one barrier
  const bool has_zero = !__syncthreads_and(run == 8u);
  if (has_zero) {
    unsigned incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o, 64);
      if ((int)(threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned base = incl - run;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_tot[w];
#pragma unroll
    for (int e = 0; e < 8; ++e) nzp[t0 + e] = (unsigned short)(base + loc[e]);
  }
  fft2048(xs, tw);  // begins with LDS reads of xs only; its first barrier also publishes nzp
  for (int f = threadIdx.x; f < kNfft; f += epa::kBlock) {
    const Cd z = cmul(xs[pad_idx(f)], spec[f]);
    xs[pad_idx(f)] = Cd{z.re, -z.im};  // conj: the inverse transform is conj(FFT(conj(.)))
  }
  __syncthreads();
  fft2048(xs, tw);  // result = conj(xs[.]); the 1/N sits in spec
  if (has_zero) {
    const unsigned total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    for (int t = threadIdx.x; t < n_out; t += epa::kBlock) {  // t + hi <= N for every valid output
      const unsigned a = nzp[t + lo];
      const unsigned b = (t + hi >= kNfft) ? total : nzp[t + hi];
      if (a == b) xs[pad_idx(t)] = Cd{0.0, 0.0};
    }
    __syncthreads();
  }
}

template <typename InT, typename T, int NB>
__global__ __launch_bounds__(epa::kBlock, 3) void sv_complex_fft_kernel(CxArgs a, const double* __restrict__ ws,
                                                                     int out_per_tile) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Cd* xs = reinterpret_cast<Cd*>(smem);
  Cd* tw = reinterpret_cast<Cd*>(smem + a.rep_lds_off);                                    // [256]
  unsigned short* nzp = reinterpret_cast<unsigned short*>(smem + a.rep_lds_off + (kNfft / 8) * sizeof(Cd));
  uint8_t* vmask = smem + a.mask_lds_off;
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_lds_off);
  if (threadIdx.x < kNfft / 8) tw[threadIdx.x] = reinterpret_cast<const Cd*>(ws)[threadIdx.x];
  __shared__ unsigned wave_tot[4];

  const int c = blockIdx.y;
  const int p = blockIdx.x / a.tiles;
  const int tile = blockIdx.x - p * a.tiles;
  const int S = a.S, B = a.B;
  const int k_begin = tile * out_per_tile;
  const double* chan = ws + ws_chan_off() + 4 * (size_t)c;
  const double norm2 = chan[0];
  const int tap_lo = (int)chan[1], tap_hi = (int)chan[2];
  const InT* re = reinterpret_cast<const InT*>(a.re);
  const InT* im = reinterpret_cast<const InT*>(a.im);
  const size_t ping_base = ((size_t)c * a.P + p) * (size_t)S * B;
  const Cd* spec = reinterpret_cast<const Cd*>(ws + ws_spec_off(a.C)) + (size_t)c * kNfft;

  const unsigned mixed_l = stage_tile<InT, double, NB>(re, im, ping_base, S, B, k_begin, kNfft, -1, xs, vmask);
  const int mixed = __syncthreads_or((int)mixed_l);

  Cd y[kR];  // this lane's outputs: tile-local samples threadIdx.x + 256 * i
  if (!mixed) {
    fft_correlate(xs, tw, spec, nzp, wave_tot, tap_lo, tap_hi, out_per_tile);
#pragma unroll
    for (int i = 0; i < kR; ++i) {
      const Cd v = xs[pad_idx(threadIdx.x + i * epa::kBlock)];
      y[i] = Cd{v.re, -v.im};
    }
  } else {
#pragma unroll
    for (int i = 0; i < kR; ++i) y[i] = Cd{0.0, 0.0};
    for (int b = 0; b < B; ++b) {  // one correlation per sector; a sector counts where it is valid
      __syncthreads();
      stage_tile<InT, double, NB>(re, im, ping_base, S, B, k_begin, kNfft, b, xs, vmask);
      __syncthreads();
      fft_correlate(xs, tw, spec, nzp, wave_tot, tap_lo, tap_hi, out_per_tile);
#pragma unroll
      for (int i = 0; i < kR; ++i) {
        const int t = threadIdx.x + i * epa::kBlock;
        if (vmask[2 * t] & (1u << b)) {
          const Cd v = xs[pad_idx(t)];
          y[i].re += v.re;
          y[i].im -= v.im;
        }
      }
    }
  }

  // ---- sector mean -> prx -> Sv/TS, as in sv_complex_kernel
  const size_t row = (size_t)c * a.P + p;
  const double* cc = a.ccoef + row * EPA_NCCOEF;
  const double ra = cc[EPA_CC_RA], rb = cc[EPA_CC_RB];
  const T shift = (T)cc[EPA_CC_SHIFT], alpha2 = (T)cc[EPA_CC_ALPHA2], Aadd = (T)cc[EPA_CC_A];
  const T pscale = (T)(cc[EPA_CC_PSCALE]);
  const T nspread = (T)a.nspread;
  const double inv_norm = 1.0 / norm2;
  const unsigned full = (1u << B) - 1u;
  T* out = reinterpret_cast<T*>(a.out);
  T* range_out = reinterpret_cast<T*>(a.range_out);
  T* prx_out = reinterpret_cast<T*>(a.prx_out);
#pragma unroll
  for (int i = 0; i < kR; ++i) {
    const int t = threadIdx.x + i * epa::kBlock;
    const int s = k_begin + t;
    if (t >= out_per_tile || s >= S) break;
    const unsigned nvalid = __popc(vmask[2 * t] & full);
    T mr, mi;
    if (nvalid == 0u) {
      mr = mi = epa::M<T>::nan();
    } else {
      const double invn = inv_norm / (double)nvalid;
      mr = (T)(y[i].re * invn);
      mi = (T)(y[i].im * invn);
    }
    T prx = pscale * (mr * mr + mi * mi);
    if (!(prx > (T)0)) prx = epa::M<T>::nan();
    const double R = ((double)s * ra) * rb;
    T rt = sub_rn((T)R, shift);  // never contracted with the range product into an fma
    if (!(rt > (T)0)) rt = epa::M<T>::nan();
    const T val = (T)10 * epa::fast_log10(prx, mt.log_tab) + nspread * epa::fast_log10(rt, mt.log_tab) + alpha2 * rt + Aadd;
    const size_t o = row * S + s;
    out[o] = val;
    if (range_out) range_out[o] = (vmask[2 * t + 1] & 1u) ? (T)R : epa::M<T>::nan();
    if (prx_out) prx_out[o] = prx;
  }
}

template <typename InT, typename T>
int launch_fft(CxArgs& a, int max_taps, double* ws, hipStream_t st) {
  hipLaunchKernelGGL(replica_prepare_kernel, dim3(a.C), dim3(epa::kBlock), 0, st, a.replica, a.replica_off,
                     a.C, ws);
  if (int rc = epa::check_launch("replica_prepare_kernel")) return rc;
  const int out_per_tile = kNfft - max_taps + 1;
  const size_t xs_bytes = ((size_t)(kNfft + kNfft / 8 + 1) * sizeof(Cd) + 15) & ~(size_t)15;
  // 256 twiddles + the u16 prefix counts of non-zero samples
  const size_t tw_bytes = (size_t)(kNfft / 8) * sizeof(Cd) + (size_t)kNfft * sizeof(unsigned short);
  const size_t mask_bytes = (size_t)2 * kNfft;
  a.rep_lds_off = (unsigned)xs_bytes;
  a.mask_lds_off = (unsigned)(xs_bytes + tw_bytes);
  a.tab_lds_off = (unsigned)((xs_bytes + tw_bytes + mask_bytes + 15) & ~(size_t)15);
  const size_t lds = a.tab_lds_off + epa::kMathTabBytes;
  a.tiles = (a.S + out_per_tile - 1) / out_per_tile;
  const dim3 grid((unsigned)((long long)a.P * a.tiles), (unsigned)a.C);
  const bool b4 = a.B == 4 && (reinterpret_cast<uintptr_t>(a.re) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(a.im) & 15u) == 0;
  if (b4)
    hipLaunchKernelGGL((sv_complex_fft_kernel<InT, T, 4>), grid, dim3(epa::kBlock), lds, st, a, ws, out_per_tile);
  else
    hipLaunchKernelGGL((sv_complex_fft_kernel<InT, T, 0>), grid, dim3(epa::kBlock), lds, st, a, ws, out_per_tile);
  return epa::check_launch("sv_complex_fft_kernel");
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

extern "C" int epa_sv_complex(const void* re, const void* im, int in_dtype, const float* replica,
                              const int32_t* replica_off, int max_taps, const double* ccoef, int C,
                              int P, int S, int B, int cal_type, void* out, void* range_out,
                              void* prx_out, int out_dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(re && im && ccoef && out, "epa_sv_complex: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && B > 0, "epa_sv_complex: C=%d P=%d S=%d B=%d", C, P, S, B);
  EPA_CHECK_ARG(B <= kMaxBeams, "epa_sv_complex: at most %d sectors supported (got %d)", kMaxBeams, B);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_complex: bad cal_type");
  EPA_CHECK_ARG((replica == nullptr) == (replica_off == nullptr),
                "epa_sv_complex: replica and replica_off must both be given (BB) or both NULL (CW)");
  EPA_CHECK_ARG(!replica || max_taps > 0, "epa_sv_complex: max_taps must be positive for BB");
  CxArgs a{};
  a.re = re; a.im = im; a.replica = replica; a.replica_off = replica_off; a.ccoef = ccoef;
  a.C = C; a.P = P; a.S = S; a.B = B;
  a.nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  a.out = out; a.range_out = range_out; a.prx_out = prx_out;
  const int taps = replica ? max_taps : 0;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == EPA_F64 && out_dtype == EPA_F64) return launch<double, double, double>(a, taps, st);
  if (in_dtype == EPA_F64 && out_dtype == EPA_F32) return launch<double, float, float>(a, taps, st);
  if (in_dtype == EPA_F32 && out_dtype == EPA_F64) return launch<float, double, double>(a, taps, st);
  if (in_dtype == EPA_F32 && out_dtype == EPA_F32) return launch<float, float, float>(a, taps, st);
  epa::set_error("epa_sv_complex: bad dtype in=%d out=%d", in_dtype, out_dtype);
  return EPA_EINVAL;
}

extern "C" int epa_sv_complex_fft(const void* re, const void* im, int in_dtype, const float* replica,
                                  const int32_t* replica_off, int max_taps, const double* ccoef, int C,
                                  int P, int S, int B, int cal_type, void* out, void* range_out,
                                  void* prx_out, int out_dtype, double* workspace, epa_stream_t stream) {
  EPA_CHECK_ARG(re && im && ccoef && out && replica && replica_off && workspace,
                "epa_sv_complex_fft: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && B > 0, "epa_sv_complex_fft: C=%d P=%d S=%d B=%d", C, P, S, B);
  EPA_CHECK_ARG(B <= kMaxBeams, "epa_sv_complex_fft: at most %d sectors supported (got %d)", kMaxBeams, B);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_complex_fft: bad cal_type");
  if (max_taps < 1 || max_taps > kNfft / 2) {
    epa::set_error("epa_sv_complex_fft: replicas of 1..%d taps only (got %d); use epa_sv_complex", kNfft / 2,
                   max_taps);
    return EPA_EUNSUPPORTED;
  }
  CxArgs a{};
  a.re = re; a.im = im; a.replica = replica; a.replica_off = replica_off; a.ccoef = ccoef;
  a.C = C; a.P = P; a.S = S; a.B = B;
  a.nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  a.out = out; a.range_out = range_out; a.prx_out = prx_out;
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == EPA_F64 && out_dtype == EPA_F64) return launch_fft<double, double>(a, max_taps, workspace, st);
  if (in_dtype == EPA_F64 && out_dtype == EPA_F32) return launch_fft<double, float>(a, max_taps, workspace, st);
  if (in_dtype == EPA_F32 && out_dtype == EPA_F64) return launch_fft<float, double>(a, max_taps, workspace, st);
  if (in_dtype == EPA_F32 && out_dtype == EPA_F32) return launch_fft<float, float>(a, max_taps, workspace, st);
  epa::set_error("epa_sv_complex_fft: bad dtype in=%d out=%d", in_dtype, out_dtype);
  return EPA_EINVAL;
}
