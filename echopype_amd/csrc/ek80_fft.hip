// K3+K4 for long replicas: EK80 broadband pulse compression as a circular correlation of a 2048-sample
// tile through an LDS-resident FFT, fused with the sector mean -> received power -> Sv/TS chain.
//
// Reference arithmetic replaced (paths under /root/reference/echopype/calibrate):
//   ek80_complex.py:285-369  compress_pulse (scipy.signal.convolve per (ping, sector); its method="auto"
//                            makes the same direct -> FFT switch for these sizes, :310-313)
//   ek80_complex.py:372-391  norm factor ||tx||^2
//   calibrate_ek.py:483-490  prx from the sector mean;  :571-638  Sv / TS chain;  range.py:138-148,180-199
//
//   y[k] = sum_j x[k+j] conj(tx[j])  =  IFFT( FFT(x) . conj(FFT(tx)) )[k]      for k <= N - taps
//
// Design (one workgroup of 256 lanes = one tile of N = 2048 samples of one (channel, ping)):
//   * lane j owns the samples j + 256 i (i = 0..7) from the global load to the Sv store: every wavefront
//     load is 1 KiB contiguous per plane, every store 512 B (f64) contiguous; the sector SUM (the matched
//     filter is linear and the reference averages the sectors right after it), the per-sample validity
//     bits and the final outputs never leave the lane's registers.
//   * forward transform = decimation in frequency, radix 4.8.8.8, IN PLACE: a lane reads and writes the
//     same LDS elements in a pass, so a pass needs ONE barrier and no second buffer.  Its output is in
//     digit-reversed order, which a convolution does not care about: the replica spectrum is stored in that
//     order by replica_prepare, and the inverse transform is the transposed algorithm (decimation in time,
//     radix 8.8.8.4) which takes digit-reversed input back to natural order.  The last forward pass, the
//     spectral product and the first inverse pass work on the same eight elements of a lane and are fused
//     in registers.  Per tile: 6 LDS round trips (the Stockham form this replaces needed 11 round trips and
//     ~20 barriers), first and last pass entirely in registers.
//   * after the first pass the transform is FOUR independent 512-point transforms, and a wavefront holds
//     exactly one of them (64 lanes x 8 elements): the three inner passes and their inverses exchange data
//     among the lanes of one wavefront only, which needs no workgroup barrier (LDS serves a wavefront's
//     requests in order).  Two barriers per tile are left -- after the first pass's stores, before the last
//     pass's loads.
//   * LDS elements are 8 bytes (float2; fp64 keeps separate re / im planes), one pad element per 8: the
//     stride-8 and stride-1 passes of a wavefront are bank-conflict free with the plain lane <-> butterfly
//     maps, the stride-64 pass and the lane's own samples collide two-fold on 3 of 32 slots.
//   * fft_dtype F32: complex64 butterflies (v_pk_* candidates), 21 KiB LDS per workgroup; F64: 39.6 KiB
//     (4 workgroups per CU: to stay under 40 960 B it keeps 68 twiddles instead of 256, see tw_any).
//     FFT errors scale with the strongest echo of the TILE, not with the sample:
//     F32 keeps 1e-3 relative on dB values up to ~90 dB of in-tile dynamic range and is the default for
//     float32 output only.
//   * the direct form returns an exact 0 where every staged sample under the replica's non-zero taps is 0
//     (zero-filled NaN padding, blanked samples) and the chain turns prx == 0 into NaN; an FFT leaves
//     rounding noise there.  To keep the NaN pattern such outputs are set to 0 from a bit mask of the
//     non-zero staged samples (wave ballots) and its prefix popcounts.
//   * a tile holding a sample whose sectors are only PARTLY NaN (never seen in files, allowed by the
//     reference) is redone one sector at a time (block-uniform branch).
#include <type_traits>

#include "fast_math.h"

namespace {

constexpr int kN = EPA_EK80_NFFT;
constexpr int kPlane = kN + kN / 8;  // padded element count
constexpr int kMaxBeams = 8;
static_assert(kN == 2048 && epa::kBlock == 256, "written for N = 2048, 256 lanes");

template <typename F>
struct C2 {
  F re, im;
};

// One pad element per 8: with the lane <-> butterfly maps below the stride-8 and stride-1 passes of a wavefront are
// free of bank conflicts (slot = element + element / 8 mod 32: 8 g + 9 r + o resp. 9 lane + r over the 32 lanes of a
// read group), the stride-64 pass and the lane's own samples collide two-fold on 3 of 32 slots.
__device__ __forceinline__ int pad(int a) { return a + (a >> 3); }

// ---- LDS element access: float2 elements / separate double planes (8-byte accesses either way)
template <typename F>
struct Xs;
template <>
struct Xs<float> {
  static constexpr size_t kBytes = (size_t)kPlane * 8;
  static __device__ __forceinline__ C2<float> ld(const unsigned char* xs, int a) {
    const float2 v = reinterpret_cast<const float2*>(xs)[pad(a)];
    return C2<float>{v.x, v.y};
  }
  static __device__ __forceinline__ void st(unsigned char* xs, int a, C2<float> v) {
    reinterpret_cast<float2*>(xs)[pad(a)] = make_float2(v.re, v.im);
  }
};
template <>
struct Xs<double> {
  static constexpr size_t kBytes = (size_t)kPlane * 16;
  static __device__ __forceinline__ C2<double> ld(const unsigned char* xs, int a) {
    const double* p = reinterpret_cast<const double*>(xs);
    return C2<double>{p[pad(a)], p[kPlane + pad(a)]};
  }
  static __device__ __forceinline__ void st(unsigned char* xs, int a, C2<double> v) {
    double* p = reinterpret_cast<double*>(xs);
    p[pad(a)] = v.re;
    p[kPlane + pad(a)] = v.im;
  }
};

template <typename F>
__device__ __forceinline__ C2<F> cmul(C2<F> a, C2<F> b) {
  return C2<F>{fma(a.re, b.re, -a.im * b.im), fma(a.re, b.im, a.im * b.re)};
}
template <typename F>
__device__ __forceinline__ C2<F> cmulc(C2<F> a, C2<F> b) {  // a * conj(b)
  return C2<F>{fma(a.re, b.re, a.im * b.im), fma(a.im, b.re, -a.re * b.im)};
}
template <typename F>
__device__ __forceinline__ C2<F> cadd(C2<F> a, C2<F> b) { return C2<F>{a.re + b.re, a.im + b.im}; }
template <typename F>
__device__ __forceinline__ C2<F> csub(C2<F> a, C2<F> b) { return C2<F>{a.re - b.re, a.im - b.im}; }
template <typename F>
__device__ __forceinline__ C2<F> mul_mi(C2<F> a) { return C2<F>{a.im, -a.re}; }  // * (-i)
template <typename F>
__device__ __forceinline__ C2<F> cswap(C2<F> a) { return C2<F>{a.im, a.re}; }

// forward 4-point DFT (e^{-2 pi i rq/4}), in place
template <typename F>
__device__ __forceinline__ void dft4(C2<F>& u0, C2<F>& u1, C2<F>& u2, C2<F>& u3) {
  const C2<F> s02 = cadd(u0, u2), d02 = csub(u0, u2), s13 = cadd(u1, u3), d13 = mul_mi(csub(u1, u3));
  u0 = cadd(s02, s13);
  u2 = csub(s02, s13);
  u1 = cadd(d02, d13);
  u3 = csub(d02, d13);
}
template <typename F>
__device__ __forceinline__ void dft8(C2<F> (&v)[8]) {
  const F kH = (F)0.70710678118654752440;
  C2<F> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  const C2<F> t1 = C2<F>{(o1.re + o1.im) * kH, (o1.im - o1.re) * kH};   // * e^{-i pi/4}
  const C2<F> t2 = mul_mi(o2);                                           // * e^{-i pi/2}
  const C2<F> t3 = C2<F>{(o3.im - o3.re) * kH, -(o3.re + o3.im) * kH};  // * e^{-3i pi/4}
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, t1); v[5] = csub(e1, t1);
  v[2] = cadd(e2, t2); v[6] = csub(e2, t2);
  v[3] = cadd(e3, t3); v[7] = csub(e3, t3);
}
// the conjugate transforms through swap(DFT(swap(.))): the swaps are register renames
template <typename F>
__device__ __forceinline__ void idft4(C2<F>& u0, C2<F>& u1, C2<F>& u2, C2<F>& u3) {
  u0 = cswap(u0); u1 = cswap(u1); u2 = cswap(u2); u3 = cswap(u3);
  dft4(u0, u1, u2, u3);
  u0 = cswap(u0); u1 = cswap(u1); u2 = cswap(u2); u3 = cswap(u3);
}
template <typename F>
__device__ __forceinline__ void idft8(C2<F> (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = cswap(v[r]);
  dft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = cswap(v[r]);
}

// v[q] *= w^q (CONJ: conj(w)^q), q = 1..7, powers by a depth-3 product tree
template <typename F, bool CONJ>
__device__ __forceinline__ void twiddle8(C2<F> (&v)[8], C2<F> w1) {
  if (CONJ) w1.im = -w1.im;
  const C2<F> w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
  v[1] = cmul(v[1], w1);
  v[2] = cmul(v[2], w2);
  v[3] = cmul(v[3], w3);
  v[4] = cmul(v[4], w4);
  v[5] = cmul(v[5], cmul(w4, w1));
  v[6] = cmul(v[6], cmul(w3, w3));
  v[7] = cmul(v[7], cmul(w4, w3));
}
template <typename F, bool CONJ>
__device__ __forceinline__ void twiddle4(C2<F>& u1, C2<F>& u2, C2<F>& u3, C2<F> w1) {
  if (CONJ) w1.im = -w1.im;
  const C2<F> w2 = cmul(w1, w1);
  u1 = cmul(u1, w1);
  u2 = cmul(u2, w2);
  u3 = cmul(u3, cmul(w2, w1));
}

// lane <-> butterfly maps (element index of r = 0 and the element stride); see the header comment
struct LaneMap {
  int a1, a2, a3;  // first element of the lane's butterfly in the stride-64, stride-8 and stride-1 passes
  int t1, t2;      // twiddle table index (w_2048^t) of those butterflies' offset
};
__device__ __forceinline__ LaneMap lane_map() {
  const int j = threadIdx.x;
  LaneMap m;
  // After the first pass the transform is four independent 512-point transforms, elements [512 w, 512 w + 512): a
  // wavefront holds exactly one of them (64 lanes x 8 elements), so its three inner passes -- and their inverses --
  // exchange data among its OWN lanes only: no workgroup barrier between them (LDS serves a wavefront's requests in
  // order), two barriers per tile instead of seven.
  const int w = j >> 6, l = j & 63;
  m.a1 = 512 * w + l;
  m.t1 = 4 * l;
  m.a2 = 512 * w + 64 * (l >> 3) + (l & 7);
  m.t2 = 32 * (l & 7);
  m.a3 = 512 * w + 8 * l;
  return m;
}

// The twiddle table w_2048^m, m < 256.  The double-precision tile leaves no room for all 256 entries next to its
// padded planes (4 workgroups per CU = 40 960 B each): SMALL keeps w^(4k), k < 64, and w^0..w^3 -- every index of
// the inner passes is a multiple of 4, the first pass pays one complex product.
template <typename F, bool SMALL>
__device__ __forceinline__ C2<F> tw_any(const C2<F>* tw, int m) {
  if (!SMALL) return tw[m];
  return cmul(tw[m >> 2], tw[64 + (m & 3)]);
}
template <typename F, bool SMALL>
__device__ __forceinline__ C2<F> tw_mul4(const C2<F>* tw, int t) {  // t % 4 == 0
  return SMALL ? tw[t >> 2] : tw[t];
}
template <typename F>
constexpr bool kSmallTw = sizeof(F) == 8;
template <typename F>
constexpr int kTwEntries = kSmallTw<F> ? 68 : 256;

// first forward pass (sub-size 2048, radix 4, two butterflies per lane), on the lane's registers:
// v[i] = sample j + 256 i.  tw = 256-entry table of w_2048^m.
// wa = w_2048^j comes from the caller (read from the full table in global memory next to the lane's samples): the
// LDS copy of the table is only published by correlate()'s first barrier, which lies behind this pass.
template <typename F>
__device__ __forceinline__ void fwd_pass0(C2<F> (&v)[8], C2<F> wa) {
  const F kH = (F)0.70710678118654752440;
  const C2<F> wb = C2<F>{(wa.re + wa.im) * kH, (wa.im - wa.re) * kH};  // w^(j+256) = w^j e^{-i pi/4}
  dft4(v[0], v[2], v[4], v[6]);
  twiddle4<F, false>(v[2], v[4], v[6], wa);
  dft4(v[1], v[3], v[5], v[7]);
  twiddle4<F, false>(v[3], v[5], v[7], wb);
}
template <typename F, bool SMALL>
__device__ __forceinline__ void inv_pass0(C2<F> (&v)[8], const C2<F>* tw) {
  const C2<F> wa = tw_any<F, SMALL>(tw, threadIdx.x);
  const F kH = (F)0.70710678118654752440;
  const C2<F> wb = C2<F>{(wa.re + wa.im) * kH, (wa.im - wa.re) * kH};
  twiddle4<F, true>(v[2], v[4], v[6], wa);
  idft4(v[0], v[2], v[4], v[6]);
  twiddle4<F, true>(v[3], v[5], v[7], wb);
  idft4(v[1], v[3], v[5], v[7]);
}

template <typename F, int STRIDE>
__device__ __forceinline__ void ld8(const unsigned char* xs, int a0, C2<F> (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = Xs<F>::ld(xs, a0 + STRIDE * r);
}
template <typename F, int STRIDE>
__device__ __forceinline__ void st8(unsigned char* xs, int a0, const C2<F> (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) Xs<F>::st(xs, a0 + STRIDE * r, v[r]);
}

// ---- experiment (25), compile-time EPA_FFT_XPOSE (bit 0: stride-64 <-> stride-8, bit 1: stride-8 <-> stride-1): the
// wave-local exchanges between two passes as register transposes across lanes instead of an LDS round trip.  With lane
// l = 8 a + b the stride-64 pass holds element l + 64 r in register r, the stride-8 pass element 64 a + b + 8 r, the
// stride-1 pass element 8 l + r: going from one to the next transposes the register index with the lane's bits 3..5
// (a) resp. 0..2 (b) -- three exchange stages each, a 32-bit word at a time: v_permlane32_swap / v_permlane16_swap
// (gfx950) for lane distances 32 and 16, DPP moves for 8, 4 (row_half_mirror then quad_perm), 2 and 1.
#ifndef EPA_FFT_XPOSE
#define EPA_FFT_XPOSE 0
#endif
namespace xp {
template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned x) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, false);
}
// exchange word a of the lanes with `hi` set against word b of their partner (lane ^ distance) without it
template <int DIST>
__device__ __forceinline__ void exch(unsigned& a, unsigned& b, bool hi) {
  if (DIST == 32) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);  // a of lanes 32..63 <-> b of lanes 0..31
    a = r[0];
    b = r[1];
  } else if (DIST == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);  // a of the odd rows <-> b of the even rows
    a = r[0];
    b = r[1];
  } else {
    const unsigned y = hi ? a : b;
    unsigned t;
    if (DIST == 8) t = dpp<0x128>(y);                 // row_ror:8
    else if (DIST == 4) t = dpp<0x1B>(dpp<0x141>(y));  // row_half_mirror (l ^ 7), then quad_perm [3,2,1,0] (l ^ 3)
    else if (DIST == 2) t = dpp<0x4E>(y);             // quad_perm [2,3,0,1]
    else t = dpp<0xB1>(y);                            // quad_perm [1,0,3,2]
    a = hi ? t : a;
    b = hi ? b : t;
  }
}
// transpose the register index of an 8-element lane set with three lane bits (LB = the lowest of them: 3 or 0)
template <int LB, int WORDS>
__device__ __forceinline__ void transpose(unsigned (&w)[8][WORDS]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int bit = 2; bit >= 0; --bit) {
    const bool hi = ((lane >> (LB + bit)) & 1) != 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (r & (1 << bit)) continue;
#pragma unroll
      for (int k = 0; k < WORDS; ++k) {
        if (LB + bit == 5) exch<32>(w[r][k], w[r | (1 << bit)][k], hi);
        else if (LB + bit == 4) exch<16>(w[r][k], w[r | (1 << bit)][k], hi);
        else if (LB + bit == 3) exch<8>(w[r][k], w[r | (1 << bit)][k], hi);
        else if (LB + bit == 2) exch<4>(w[r][k], w[r | (1 << bit)][k], hi);
        else if (LB + bit == 1) exch<2>(w[r][k], w[r | (1 << bit)][k], hi);
        else exch<1>(w[r][k], w[r | (1 << bit)][k], hi);
      }
    }
  }
}
template <int LB, typename V>
__device__ __forceinline__ void transpose_vals(V (&v)[8]) {
  constexpr int WORDS = sizeof(V) / 4;
  unsigned w[8][WORDS];
#pragma unroll
  for (int r = 0; r < 8; ++r) __builtin_memcpy(w[r], &v[r], sizeof(V));
  transpose<LB, WORDS>(w);
#pragma unroll
  for (int r = 0; r < 8; ++r) __builtin_memcpy(&v[r], w[r], sizeof(V));
}
}  // namespace xp

// Circular correlation of the tile held as v[i] = x[j + 256 i] with the channel's replica (spectrum `spec` in
// the digit-reversed order of the forward transform, conj and 1/N applied).  Result in v, same ownership.
// Barriers: the caller guarantees nobody still reads xs on entry; on exit xs holds nothing of value.
template <typename F>
__device__ __forceinline__ void correlate(C2<F> (&v)[8], unsigned char* xs, const C2<F>* tw,
                                          const C2<F>* __restrict__ spec, const LaneMap& lm, C2<F> w_lane) {
  const int j = threadIdx.x;
  constexpr bool SMALL = kSmallTw<F>;
  fwd_pass0<F>(v, w_lane);
#pragma unroll
  for (int i = 0; i < 8; ++i) Xs<F>::st(xs, j + 256 * i, v[i]);
  __syncthreads();
  ld8<F, 64>(xs, lm.a1, v);
  dft8(v);
  twiddle8<F, false>(v, tw_mul4<F, SMALL>(tw, lm.t1));
  if (EPA_FFT_XPOSE & 1) {
    xp::transpose_vals<3>(v);
  } else {
    st8<F, 64>(xs, lm.a1, v);
    __builtin_amdgcn_wave_barrier();  // (own wavefront's data: ordering for the compiler only)
    ld8<F, 8>(xs, lm.a2, v);
  }
  // the replica spectrum of the fused pass: 8 consecutive elements per lane, requested before the barrier
  C2<F> sp[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) sp[r] = spec[lm.a3 + r];
  dft8(v);
  twiddle8<F, false>(v, tw_mul4<F, SMALL>(tw, lm.t2));
  if (EPA_FFT_XPOSE & 2) {
    xp::transpose_vals<0>(v);
  } else {
    st8<F, 8>(xs, lm.a2, v);
    __builtin_amdgcn_wave_barrier();  // (own wavefront's data: ordering for the compiler only)
    ld8<F, 1>(xs, lm.a3, v);
  }
  dft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = cmul(v[r], sp[r]);
  idft8(v);
  if (EPA_FFT_XPOSE & 2) {
    xp::transpose_vals<0>(v);
  } else {
    st8<F, 1>(xs, lm.a3, v);
    __builtin_amdgcn_wave_barrier();  // (own wavefront's data: ordering for the compiler only)
    ld8<F, 8>(xs, lm.a2, v);
  }
  twiddle8<F, true>(v, tw_mul4<F, SMALL>(tw, lm.t2));
  idft8(v);
  if (EPA_FFT_XPOSE & 1) {
    xp::transpose_vals<3>(v);
  } else {
    st8<F, 8>(xs, lm.a2, v);
    __builtin_amdgcn_wave_barrier();  // (own wavefront's data: ordering for the compiler only)
    ld8<F, 64>(xs, lm.a1, v);
  }
  twiddle8<F, true>(v, tw_mul4<F, SMALL>(tw, lm.t1));
  idft8(v);
  st8<F, 64>(xs, lm.a1, v);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = Xs<F>::ld(xs, j + 256 * i);
  inv_pass0<F, SMALL>(v, tw);
}

// ---- complex64: the same transform on (re, im) register pairs with the packed-f32 instructions of gfx950.
// Left to the vectoriser, the scalar C2<float> code above becomes v_pk_* instructions glued together with moves and
// sign flips (189 v_mov / v_pk_mov and 35 v_xor in a 703-instruction transform): the operand modifiers of VOP3P --
// op_sel / op_sel_hi pick which half of a source feeds the low / high lane, neg_lo / neg_hi negate it -- do the
// swaps and negations of complex arithmetic for free, but the compiler does not use them for f32 pairs.  Written out:
// one instruction per complex add (also with a factor of -i or +i on the second operand), two per complex product,
// 26 per radix-8 butterfly.  Same operations in the same order as the scalar templates (products then fused
// multiply-adds, the 1/sqrt2 rotations fused into the following sum).
namespace pk {
typedef float f2 __attribute__((ext_vector_type(2)));
#define EPA_PK2(name, text)                                              \
  __device__ __forceinline__ f2 name(f2 a, f2 b) {                       \
    f2 r;                                                                \
    asm(text : "=v"(r) : "v"(a), "v"(b));                                \
    return r;                                                            \
  }
#define EPA_PK3(name, text)                                              \
  __device__ __forceinline__ f2 name(f2 a, f2 b, f2 c) {                 \
    f2 r;                                                                \
    asm(text : "=v"(r) : "v"(a), "v"(b), "v"(c));                        \
    return r;                                                            \
  }
EPA_PK2(add, "v_pk_add_f32 %0, %1, %2")
EPA_PK2(sub, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]")
// a + (-i) b = (a.re + b.im, a.im - b.re);  a + i b = (a.re - b.im, a.im + b.re)
EPA_PK2(add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
EPA_PK2(add_pi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
EPA_PK2(mul, "v_pk_mul_f32 %0, %1, %2")
// (a.im b.im, a.im b.re) and (a.im b.im, a.re b.im): the first halves of a b and of a conj(b)
EPA_PK2(mul_ii_ir, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]")
EPA_PK2(mul_ii_ri, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]")
// (a.re b.re - t.lo, a.re b.im + t.hi);  (a.re b.re + t.lo, a.im b.re - t.hi)
EPA_PK3(fma_rr_ri, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]")
EPA_PK3(fma_rr_ir, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]")
EPA_PK3(fma, "v_pk_fma_f32 %0, %1, %2, %3")
EPA_PK3(fnma, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]")  // c - a b
#undef EPA_PK2
#undef EPA_PK3
__device__ __forceinline__ f2 cmul(f2 a, f2 b) { return fma_rr_ri(a, b, mul_ii_ir(a, b)); }
__device__ __forceinline__ f2 conj(f2 a) { return f2{a.x, -a.y}; }

__device__ __forceinline__ void dft4(f2& u0, f2& u1, f2& u2, f2& u3) {
  const f2 s02 = add(u0, u2), d02 = sub(u0, u2), s13 = add(u1, u3), d13 = sub(u1, u3);
  u0 = add(s02, s13);
  u2 = sub(s02, s13);
  u1 = add_mi(d02, d13);
  u3 = add_pi(d02, d13);
}
__device__ __forceinline__ void idft4(f2& u0, f2& u1, f2& u2, f2& u3) {
  const f2 s02 = add(u0, u2), d02 = sub(u0, u2), s13 = add(u1, u3), d13 = sub(u1, u3);
  u0 = add(s02, s13);
  u2 = sub(s02, s13);
  u1 = add_pi(d02, d13);
  u3 = add_mi(d02, d13);
}
template <bool INV>
__device__ __forceinline__ void dft8(f2 (&v)[8], f2 kh) {
  f2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  if (INV) {
    idft4(e0, e1, e2, e3);
    idft4(o0, o1, o2, o3);
  } else {
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
  }
  // forward: o1 e^{-i pi/4} = kh (o1 - i o1), o3 e^{-3i pi/4} = -kh (o3 + i o3); inverse: the conjugate factors
  const f2 q1 = INV ? add_pi(o1, o1) : add_mi(o1, o1);
  const f2 q3 = INV ? add_mi(o3, o3) : add_pi(o3, o3);
  v[0] = add(e0, o0);
  v[4] = sub(e0, o0);
  v[1] = fma(q1, kh, e1);
  v[5] = fnma(q1, kh, e1);
  v[2] = INV ? add_pi(e2, o2) : add_mi(e2, o2);
  v[6] = INV ? add_mi(e2, o2) : add_pi(e2, o2);
  v[3] = fnma(q3, kh, e3);
  v[7] = fma(q3, kh, e3);
}
__device__ __forceinline__ void twiddle8(f2 (&v)[8], f2 w1) {
  const f2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
  v[1] = cmul(v[1], w1);
  v[2] = cmul(v[2], w2);
  v[3] = cmul(v[3], w3);
  v[4] = cmul(v[4], w4);
  v[5] = cmul(v[5], cmul(w4, w1));
  v[6] = cmul(v[6], cmul(w3, w3));
  v[7] = cmul(v[7], cmul(w4, w3));
}
__device__ __forceinline__ void twiddle4(f2& u1, f2& u2, f2& u3, f2 w1) {
  const f2 w2 = cmul(w1, w1);
  u1 = cmul(u1, w1);
  u2 = cmul(u2, w2);
  u3 = cmul(u3, cmul(w2, w1));
}
__device__ __forceinline__ f2 ld(const unsigned char* xs, int a) { return reinterpret_cast<const f2*>(xs)[pad(a)]; }
__device__ __forceinline__ void st(unsigned char* xs, int a, f2 v) { reinterpret_cast<f2*>(xs)[pad(a)] = v; }
template <int STRIDE>
__device__ __forceinline__ void ld8(const unsigned char* xs, int a0, f2 (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = ld(xs, a0 + STRIDE * r);
}
template <int STRIDE>
__device__ __forceinline__ void st8(unsigned char* xs, int a0, const f2 (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 8; ++r) st(xs, a0 + STRIDE * r, v[r]);
}
}  // namespace pk

template <>
__device__ __forceinline__ void correlate<float>(C2<float> (&vc)[8], unsigned char* xs, const C2<float>* tw,
                                                 const C2<float>* __restrict__ spec, const LaneMap& lm,
                                                 C2<float> w_lane) {
  using pk::f2;
  const int j = threadIdx.x;
  const f2* twp = reinterpret_cast<const f2*>(tw);
  const f2* specp = reinterpret_cast<const f2*>(spec);
  const f2 kh = f2{0.70710678118654752440f, 0.70710678118654752440f};
  f2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = f2{vc[i].re, vc[i].im};
  // first pass (sub-size 2048, radix 4, two butterflies per lane): w^(j + 256) = w^j e^{-i pi/4} = kh (w - i w)
  const f2 wa = f2{w_lane.re, w_lane.im};
  const f2 wb = pk::mul(pk::add_mi(wa, wa), kh);
  pk::dft4(v[0], v[2], v[4], v[6]);
  pk::twiddle4(v[2], v[4], v[6], wa);
  pk::dft4(v[1], v[3], v[5], v[7]);
  pk::twiddle4(v[3], v[5], v[7], wb);
#pragma unroll
  for (int i = 0; i < 8; ++i) pk::st(xs, j + 256 * i, v[i]);
  __syncthreads();
  pk::ld8<64>(xs, lm.a1, v);
  pk::dft8<false>(v, kh);
  pk::twiddle8(v, twp[lm.t1]);
  if (EPA_FFT_XPOSE & 1) {
    xp::transpose_vals<3>(v);
  } else {
    pk::st8<64>(xs, lm.a1, v);
    __builtin_amdgcn_wave_barrier();  // (own wavefront's data: ordering for the compiler only)
    pk::ld8<8>(xs, lm.a2, v);
  }
  f2 sp[8];  // the replica spectrum of the fused pass, requested early
#pragma unroll
  for (int r = 0; r < 8; ++r) sp[r] = specp[lm.a3 + r];
  pk::dft8<false>(v, kh);
  pk::twiddle8(v, twp[lm.t2]);
  if (EPA_FFT_XPOSE & 2) {
    xp::transpose_vals<0>(v);
  } else {
    pk::st8<8>(xs, lm.a2, v);
    __builtin_amdgcn_wave_barrier();
    pk::ld8<1>(xs, lm.a3, v);
  }
  pk::dft8<false>(v, kh);
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = pk::cmul(v[r], sp[r]);
  pk::dft8<true>(v, kh);
  if (EPA_FFT_XPOSE & 2) {
    xp::transpose_vals<0>(v);
  } else {
    pk::st8<1>(xs, lm.a3, v);
    __builtin_amdgcn_wave_barrier();
    pk::ld8<8>(xs, lm.a2, v);
  }
  pk::twiddle8(v, pk::conj(twp[lm.t2]));
  pk::dft8<true>(v, kh);
  if (EPA_FFT_XPOSE & 1) {
    xp::transpose_vals<3>(v);
  } else {
    pk::st8<8>(xs, lm.a2, v);
    __builtin_amdgcn_wave_barrier();
    pk::ld8<64>(xs, lm.a1, v);
  }
  pk::twiddle8(v, pk::conj(twp[lm.t1]));
  pk::dft8<true>(v, kh);
  pk::st8<64>(xs, lm.a1, v);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = pk::ld(xs, j + 256 * i);
  // last pass: conjugate twiddles, then the inverse radix-4 butterflies
  const f2 wl = twp[j];
  const f2 wlb = pk::mul(pk::add_mi(wl, wl), kh);
  pk::twiddle4(v[2], v[4], v[6], pk::conj(wl));
  pk::idft4(v[0], v[2], v[4], v[6]);
  pk::twiddle4(v[3], v[5], v[7], pk::conj(wlb));
  pk::idft4(v[1], v[3], v[5], v[7]);
#pragma unroll
  for (int i = 0; i < 8; ++i) vc[i] = C2<float>{v[i].x, v[i].y};
}

// ---- workspace layout (doubles):
//   [0, 512)            256 twiddles w_2048^m as double2
//   [512, 768)          the same as float2
//   [768, 768 + 4C)     per channel: ||tx||^2, first non-zero tap, one past the last non-zero tap, 0
//   then per channel    the replica spectrum conj(FFT(tx))/N in transform order: 2048 double2, then 2048 float2
__host__ __device__ inline size_t ws_tw64() { return 0; }
__host__ __device__ inline size_t ws_tw32() { return 512; }
__host__ __device__ inline size_t ws_chan() { return 768; }
__host__ __device__ inline size_t ws_spec64(int C, int c) { return 768 + 4 * (size_t)C + (size_t)c * 3 * kN; }
__host__ __device__ inline size_t ws_spec32(int C, int c) { return ws_spec64(C, c) + 2 * kN; }
//   then                kStatSlots x {min, max, NaN count} of the echo_range (merged by f64 atomics; optional)
constexpr int kStatSlots = 1024;
__host__ __device__ inline size_t ws_stats(int C) { return 768 + 4 * (size_t)C + (size_t)C * 3 * kN; }
//   then                [1 double] counter of deferred (partly-NaN) tiles, then one bit per tile
__host__ __device__ inline size_t ws_mixed(int C) { return ws_stats(C) + 3 * kStatSlots; }
//   then                per channel {ra, rb, shift, 2 alpha of its first ping} + S values of the time-varied gain
//                       n log10(R') + 2 alpha R' (tvg_table_kernel)
__host__ __device__ inline size_t ws_tvg(int C, int P, int S) {
  return ws_mixed(C) + 2 + ((size_t)C * (size_t)P * ((size_t)S / (kN / 2 + 1) + 1) + 63) / 64;
}
//   then                the 128 x {1/c, log10 c} table of fast_log10 (fast_math.h), copied to LDS by every workgroup
__host__ __device__ inline size_t ws_logtab(int C, int P, int S) { return ws_tvg(C, P, S) + (size_t)C * ((size_t)S + 4); }

__global__ __launch_bounds__(epa::kBlock) void replica_prepare_kernel(const float* __restrict__ replica,
                                                                      const int32_t* __restrict__ off, int C,
                                                                      double* __restrict__ ws, int init_stats,
                                                                      double* __restrict__ log_tab_out) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[Xs<double>::kBytes];
  __shared__ C2<double> tw[256];
  __shared__ double red[4];
  __shared__ int tap_lo, tap_hi;
  const int c = blockIdx.x, j = threadIdx.x;
  if (j == 0) {
    tap_lo = kN;
    tap_hi = 0;
  }
  if (c == 0 && j < epa::kLogTabN) {  // as build_math_tabs
    const double cc = 1.0 + ((double)j + 0.5) * (1.0 / epa::kLogTabN);
    const double inv = 1.0 / cc;
    double lg = -::log10(inv);
    if (cc > 1.4142135623730951) lg -= 0.30102999566398120;
    reinterpret_cast<double2*>(log_tab_out)[j] = make_double2(inv, lg);
  }
  if (init_stats && c == 0) {
    double* sp = ws + ws_stats(C);
    for (int k = j; k < kStatSlots; k += epa::kBlock) {
      sp[3 * k] = __builtin_inf();
      sp[3 * k + 1] = -__builtin_inf();
      sp[3 * k + 2] = 0.0;
    }
  }
  {
    double sn, cs;
    sincospi(-2.0 * (double)j / (double)kN, &sn, &cs);
    tw[j] = C2<double>{cs, sn};
    if (c == 0) {
      reinterpret_cast<C2<double>*>(ws + ws_tw64())[j] = tw[j];
      reinterpret_cast<C2<float>*>(ws + ws_tw32())[j] = C2<float>{(float)cs, (float)sn};
    }
  }
  __syncthreads();
  const int r0 = off[c], taps = off[c + 1] - r0;
  double part = 0.0;
  C2<double> v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int n = j + 256 * i;
    C2<double> t{0.0, 0.0};
    if (n < taps) {
      t.re = (double)replica[2 * (size_t)(r0 + n)];
      t.im = (double)replica[2 * (size_t)(r0 + n) + 1];
    }
    v[i] = t;
    part += t.re * t.re + t.im * t.im;
    if (t.re != 0.0 || t.im != 0.0) {  // tapered replicas start (and may end) with exact zeros
      atomicMin(&tap_lo, n);
      atomicMax(&tap_hi, n + 1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);  // ||tx||^2 (ek80_complex.py:372-391)
  if ((j & 63) == 0) red[j >> 6] = part;
  const LaneMap lm = lane_map();
  fwd_pass0<double>(v, tw[j]);
#pragma unroll
  for (int i = 0; i < 8; ++i) Xs<double>::st(xs, j + 256 * i, v[i]);
  __syncthreads();
  if (j == 0) {
    double* ch = ws + ws_chan() + 4 * (size_t)c;
    ch[0] = (red[0] + red[1]) + (red[2] + red[3]);
    ch[1] = (double)(tap_lo < tap_hi ? tap_lo : 0);
    ch[2] = (double)tap_hi;
    ch[3] = 1.0 / ch[0];  // (the epilogue of every tile used to divide)
  }
  ld8<double, 64>(xs, lm.a1, v);
  dft8(v);
  twiddle8<double, false>(v, tw[lm.t1]);
  st8<double, 64>(xs, lm.a1, v);
  __builtin_amdgcn_wave_barrier();
  ld8<double, 8>(xs, lm.a2, v);
  dft8(v);
  twiddle8<double, false>(v, tw[lm.t2]);
  st8<double, 8>(xs, lm.a2, v);
  __builtin_amdgcn_wave_barrier();
  ld8<double, 1>(xs, lm.a3, v);
  dft8(v);
  C2<double>* s64 = reinterpret_cast<C2<double>*>(ws + ws_spec64(C, c));
  C2<float>* s32 = reinterpret_cast<C2<float>*>(ws + ws_spec32(C, c));
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const C2<double> z{v[r].re * (1.0 / kN), -v[r].im * (1.0 / kN)};
    s64[lm.a3 + r] = z;
    s32[lm.a3 + r] = C2<float>{(float)z.re, (float)z.im};
  }
}

struct FftArgs {
  const void* re;
  const void* im;
  const double* ccoef;
  const double* ws;
  int C, P, S, B;
  int W;  // what the workspace layout was sized for: max(C, number of replicas)
  const int32_t* replica_id;  // optional [C*P]: the replica (spectrum, norm) of ping (c, p); NULL: replica c
  int tiles, out_per_tile;
  double nspread;
  void* out;
  void* range_out;
  void* prx_out;
  double* stats_part;   // optional [3 * kStatSlots]: {min, max, NaN count} of echo_range, merged by atomics
  unsigned* mixed_map;  // bit per tile (linear id (c * P + p) * tiles + tile): the tile holds a partly-NaN sample
  unsigned* mixed_cnt;  // number of bits set
  int map_words;
  const double* tvg;  // [C][4 + S], see ws_tvg
  const double* log_tab;  // see ws_logtab
};

// sector sum (or one sector when only >= 0) + validity bits of sample s (bits 0..B-1 sector valid, bit 8: beam-0
// real part valid = the echo_range mask of range.py:143-146)
template <typename InT, typename F, int NB>
__device__ __forceinline__ void load_sample(const InT* __restrict__ re, const InT* __restrict__ im, size_t ping_base,
                                            int S, int Brt, int s, int only, C2<F>& v, unsigned& m) {
  const int B = NB > 0 ? NB : Brt;
  F sr = (F)0, si = (F)0;
  m = 0;
  if (s < S) {
    const InT* pr = re + ping_base + (size_t)s * B;
    const InT* pi = im + ping_base + (size_t)s * B;
    if (NB > 0) {
      constexpr int kPer = 16 / sizeof(InT);
      typedef InT vec_t __attribute__((ext_vector_type(kPer)));
      InT vr[NB > 0 ? NB : 1], vi[NB > 0 ? NB : 1];
#pragma unroll
      for (int q = 0; q < (NB > 0 ? NB : kPer) / kPer; ++q) {
        const vec_t tr = reinterpret_cast<const vec_t*>(pr)[q], ti = reinterpret_cast<const vec_t*>(pi)[q];
#pragma unroll
        for (int e = 0; e < kPer; ++e) {
          vr[q * kPer + e] = tr[e];
          vi[q * kPer + e] = ti[e];
        }
      }
      // the usual sample holds 2 NB numbers: their sum is then a number too, and no per-sector test is needed
      F fr = (F)vr[0], fi = (F)vi[0];
#pragma unroll
      for (int b = 1; b < NB; ++b) {
        fr += (F)vr[b];
        fi += (F)vi[b];
      }
      if (__builtin_expect(only < 0 && fr == fr && fi == fi, 1)) {
        sr = fr;
        si = fi;
        m = ((1u << NB) - 1u) | 0x100u;
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const bool ok = (vr[b] == vr[b]) && (vi[b] == vi[b]);
          if (ok) {
            m |= 1u << b;
            if (only < 0 || only == b) {
              sr += (F)vr[b];
              si += (F)vi[b];
            }
          }
        }
        if (vr[0] == vr[0]) m |= 0x100u;
      }
      if (vr[0] == vr[0]) m |= 0x100u;
    } else {
      for (int b = 0; b < B; ++b) {
        const InT xr = pr[b], xi = pi[b];
        const bool ok = (xr == xr) && (xi == xi);
        if (ok) {
          m |= 1u << b;
          if (only < 0 || only == b) {
            sr += (F)xr;
            si += (F)xi;
          }
        }
      }
      if (pr[0] == pr[0]) m |= 0x100u;
    }
  }
  v = C2<F>{sr, si};
}

// The same in two steps, for the fixed sector counts of the fast form: every request of a lane goes out before the
// first value is looked at.  (load_sample's rare per-sector branch ends a basic block after every sample, and the
// requests of the next sample were only issued behind it: two 16-byte loads in flight per wavefront, eight round trips
// to memory per tile.)
template <typename InT, int NB>
struct RawSample {
  static constexpr int kPer = 16 / sizeof(InT);
  static constexpr int kVec = NB / kPer;
  typedef InT vec_t __attribute__((ext_vector_type(kPer)));
  vec_t r[kVec], i[kVec];
};
// (re_ping / im_ping: the ping's first sample, a uniform address; the lane's part is a 32-bit byte offset, so the
// requests take the scalar-base + 32-bit-offset form instead of a 64-bit address sum per request)
template <typename InT, int NB, bool INSIDE>
__device__ __forceinline__ void fetch_sample(const InT* __restrict__ re_ping, const InT* __restrict__ im_ping, int S,
                                             int s, RawSample<InT, NB>& raw) {
  typedef typename RawSample<InT, NB>::vec_t vec_t;
#pragma unroll
  for (int q = 0; q < RawSample<InT, NB>::kVec; ++q) raw.r[q] = raw.i[q] = (vec_t)(InT)0;
  if (INSIDE || s < S) {
    const unsigned off = (unsigned)s * (unsigned)(NB * sizeof(InT));
    const vec_t* pr = reinterpret_cast<const vec_t*>(reinterpret_cast<const char*>(re_ping) + off);
    const vec_t* pi = reinterpret_cast<const vec_t*>(reinterpret_cast<const char*>(im_ping) + off);
#pragma unroll
    for (int q = 0; q < RawSample<InT, NB>::kVec; ++q) {
      raw.r[q] = pr[q];
      raw.i[q] = pi[q];
    }
  }
}
// the plain sector sum (a NaN in any sector makes it NaN: the caller then takes the per-sector rules)
template <typename InT, typename F, int NB>
__device__ __forceinline__ C2<F> sum_plain(const RawSample<InT, NB>& raw) {
  constexpr int kPer = RawSample<InT, NB>::kPer;
  F fr = (F)raw.r[0][0], fi = (F)raw.i[0][0];
#pragma unroll
  for (int b = 1; b < NB; ++b) {
    fr += (F)raw.r[b / kPer][b % kPer];
    fi += (F)raw.i[b / kPer][b % kPer];
  }
  return C2<F>{fr, fi};
}
// v_min_f64 / v_max_f64 as the hardware has them (clang's fmin / fmax first canonicalise both operands)
__device__ __forceinline__ double vmax_num(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin_num(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ double sub_rn(double a, double b) {
  asm volatile("" : "+v"(a));
  return a - b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
  asm volatile("" : "+v"(a));
  return a - b;
}

#ifndef EPA_FFT_WAVES_F32
#define EPA_FFT_WAVES_F32 4
#endif
#ifndef EPA_FFT_WAVES_F64
#define EPA_FFT_WAVES_F64 4
#endif

template <typename F, typename T>
struct TileLds {
  unsigned char* xs;
  const C2<F>* tw;
  unsigned long long* nzw;  // [33]
  unsigned* wp;             // [34]
  unsigned* wflags;         // [4]
  const double2* log_tab;
  double* sred;             // [12]
};

// One tile.  MIXED = false: the fast form (sector sum; a tile that turns out to hold a partly-NaN sample is
// recorded in the bitmap and left to the MIXED = true pass, which convolves one sector at a time).
// Returns false when the tile was deferred.
// The time-varied gain n log10(R') + 2 alpha R' depends on the sample index and on four per-ping numbers that a file
// almost never changes from ping to ping (sample interval, sound speed, pulse length, absorption).  Tabulated once per
// channel for its first ping's numbers (double output only) -- with the roundings of the per-sample form -- it turns
// one of the two logarithms of a sample into an 8-byte read from L2; a ping with other numbers takes the per-sample form.
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void tvg_table_kernel(const double* __restrict__ ccoef, int P, int S,
                                                                double nspread, double* __restrict__ tvg) {
  const int c = blockIdx.y, s = blockIdx.x * epa::kBlock + threadIdx.x;
  const double* cc = ccoef + (size_t)c * P * EPA_NCCOEF;
  double* tab = tvg + (size_t)c * (S + 4);
  const double ra = cc[EPA_CC_RA], rb = cc[EPA_CC_RB], shift = cc[EPA_CC_SHIFT], alpha2 = cc[EPA_CC_ALPHA2];
  if (s == 0) {
    tab[0] = ra;
    tab[1] = rb;
    tab[2] = shift;
    tab[3] = alpha2;
  }
  if (s < S) {
    const double R = ((double)s * ra) * rb;
    T rt = sub_rn((T)R, (T)shift);
    if (!(rt > (T)0)) rt = epa::M<T>::nan();
    tab[4 + s] = (double)((T)nspread * epa::M<T>::log10(rt) + (T)alpha2 * rt);
  }
}

// twiddle and logarithm tables, L2 -> LDS (published by correlate()'s first barrier)
template <typename T, typename F>
__device__ __forceinline__ void stage_tables(const FftArgs& a, C2<F>* tw, double2* log_tab) {
  const int j = threadIdx.x;
  const C2<F>* wtab = reinterpret_cast<const C2<F>*>(a.ws + (sizeof(F) == 4 ? ws_tw32() : ws_tw64()));
  if (!kSmallTw<F>) tw[j] = wtab[j];
  else if (j < 68) tw[j] = j < 64 ? wtab[4 * j] : wtab[j - 64];
  if (sizeof(T) == 8 && j < epa::kLogTabN) log_tab[j] = reinterpret_cast<const double2*>(a.log_tab)[j];
}

template <typename InT, typename T, typename F, int NB, bool MIXED>
__device__ __forceinline__ void process_tile(const FftArgs& a, const TileLds<F, T>& L, const LaneMap& lm, int c, int p,
                                             int tile) {
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int S = a.S, B = NB > 0 ? NB : a.B;
  const int k_begin = tile * a.out_per_tile;
  // the ping's replica: its channel's, or -- a file with several filter_time intervals -- the one of its (channel,
  // interval) pair (a ping no interval covers has id -1 and a NaN coefficient row: its output is NaN whatever it is
  // correlated with)
  const int rid = a.replica_id ? max(a.replica_id[(size_t)c * a.P + p], 0) : c;
  const double* chan = a.ws + ws_chan() + 4 * (size_t)rid;
  const InT* re = reinterpret_cast<const InT*>(a.re);
  const InT* im = reinterpret_cast<const InT*>(a.im);
  const size_t ping_base = ((size_t)c * a.P + p) * (size_t)S * B;
  const C2<F>* spec = reinterpret_cast<const C2<F>*>(a.ws + (sizeof(F) == 4 ? ws_spec32(a.W, rid) : ws_spec64(a.W, rid)));
  const unsigned full = (1u << B) - 1u;

  // the ping's and the channel's numbers of the epilogue, read before anything is written: uniform addresses with no
  // store in front of them become scalar loads into SGPRs (behind the barriers they were vector loads from L2 with the
  // wavefront waiting on them at the start of its epilogue)
  const size_t row = (size_t)c * a.P + p;
  const double* cc = a.ccoef + row * EPA_NCCOEF;
  const double ra = cc[EPA_CC_RA], rb = cc[EPA_CC_RB];
  const double shift_d = cc[EPA_CC_SHIFT], alpha2_d = cc[EPA_CC_ALPHA2];
  const T shift = (T)shift_d, alpha2 = (T)alpha2_d, Aadd = (T)cc[EPA_CC_A];
  const T pscale = (T)(cc[EPA_CC_PSCALE]);
  const T nspread = (T)a.nspread;
  const double inv_norm = chan[3];  // 1 / ||replica||^2, divided once in replica_prepare_kernel
  const double inv_norm_b = (NB == 4 || NB == 2 || NB == 1 || NB == 8) ? inv_norm * (1.0 / (double)(NB > 0 ? NB : 1))
                                                                     : inv_norm / (double)B;  // (exact either way)
  const int zr_lo = (int)chan[1], zr_hi = (int)chan[2];
  const double* tkey = a.tvg + (size_t)c * (S + 4);
  const double* tvg_tab = tkey + 4;
  // (float output: the hardware logarithm is cheaper than the read -- measured; the table is not built then)
  const bool tabulated = sizeof(T) == 8 && ((tkey[0] == ra) & (tkey[1] == rb) & (tkey[2] == shift_d) & (tkey[3] == alpha2_d));
  const epa::LogCoef lk = epa::make_log_coef();
  // w_2048^j of the first pass, from the full table in the workspace (requested behind the samples)
  C2<F> w_lane;
  const C2<F>* w_table = reinterpret_cast<const C2<F>*>(a.ws + (sizeof(F) == 4 ? ws_tw32() : ws_tw64()));
  // ---- the lane's eight samples: sector sums + validity bits
  C2<F> v[8];
  unsigned m[MIXED ? 8 : 1];
  unsigned vbits = 0;  // bit i: sample i has valid sectors; bit 8 + i: its beam-0 real part is valid
  unsigned mixed_l = 0;
  // (wavefront-uniform) every sample of the wavefront inside the ping holds 2 B numbers: the epilogue without the NaN
  // rules of partly or wholly missing samples
  bool clean = !MIXED && NB > 0;
  // (a tile that lies inside the ping -- all but the last one or two -- needs no per-sample bound test: block-uniform)
  if (!MIXED && NB > 0) {
    constexpr int kBatch = sizeof(InT) == 4 ? 8 : 4;  // 64 registers of requested sectors per lane at a time
    constexpr int kNB = NB > 0 ? NB : 4;
    const bool inside = k_begin + kN <= S;
#pragma unroll
    for (int b0 = 0; b0 < 8; b0 += kBatch) {
      RawSample<InT, kNB> raw[kBatch];
      if (inside) {
#pragma unroll
        for (int i = 0; i < kBatch; ++i)
          fetch_sample<InT, kNB, true>(re + ping_base, im + ping_base, S, k_begin + j + 256 * (b0 + i), raw[i]);
      } else {
#pragma unroll
        for (int i = 0; i < kBatch; ++i)
          fetch_sample<InT, kNB, false>(re + ping_base, im + ping_base, S, k_begin + j + 256 * (b0 + i), raw[i]);
      }
      if (b0 == 0) w_lane = w_table[j];
      // One test per wavefront and batch instead of per-sample validity bits: the plain sums, and a ballot of "some
      // sum is NaN" (a scalar OR per sample).  Only a wavefront that holds a NaN sector looks at its samples one by
      // one (load_sample again: the lines are in L2).
      unsigned long long nan_lanes = 0ull;
#pragma unroll
      for (int i = 0; i < kBatch; ++i) {
        v[b0 + i] = sum_plain<InT, F, kNB>(raw[i]);
        nan_lanes |= __ballot(v[b0 + i].re != v[b0 + i].re || v[b0 + i].im != v[b0 + i].im);
      }
      if (__builtin_expect(nan_lanes == 0ull, 1)) {
        if (inside) {
          vbits |= (((1u << kBatch) - 1u) * 0x101u) << b0;
        } else {
#pragma unroll
          for (int i = 0; i < kBatch; ++i)
            vbits |= (k_begin + j + 256 * (b0 + i) < S ? 0x101u : 0u) << (b0 + i);
        }
      } else {
        clean = false;
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          unsigned mi;
          load_sample<InT, F, NB>(re, im, ping_base, S, B, k_begin + j + 256 * (b0 + i), -1, v[b0 + i], mi);
          vbits |= ((mi & full) != 0u ? 1u : 0u) << (b0 + i);
          vbits |= ((mi >> 8) & 1u) << (8 + b0 + i);
          mixed_l |= ((mi & full) != 0u && (mi & full) != full) ? 1u : 0u;
        }
      }
    }
  } else {
    w_lane = w_table[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned mi;
      load_sample<InT, F, NB>(re, im, ping_base, S, B, k_begin + j + 256 * i, -1, v[i], mi);
      if (MIXED) m[i] = mi;
      vbits |= ((mi & full) != 0u ? 1u : 0u) << i;
      vbits |= ((mi >> 8) & 1u) << (8 + i);
      mixed_l |= ((mi & full) != 0u && (mi & full) != full) ? 1u : 0u;
    }
  }

  // the tabulated time-varied gain of the lane's samples, requested inside correlate() (used when the ping has the
  // table's numbers -- the usual case; 8 bytes per sample from L2)
  const double* tvg_row = a.tvg + (size_t)c * (S + 4) + 4;
  constexpr bool kTvgRegs = !MIXED && sizeof(T) == 8;
  double tvg_early[kTvgRegs ? 8 : 1];
  C2<F> y[MIXED ? 8 : 1];
  for (int only = -1;;) {
    if (MIXED) {
      if (++only == B) break;
      if (only > 0) __syncthreads();  // the previous sector's pass is done with xs / nzw / wflags
      unsigned dummy;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        load_sample<InT, F, NB>(re, im, ping_base, S, B, k_begin + j + 256 * i, only, v[i], dummy);
        if (only == 0) y[i] = C2<F>{(F)0, (F)0};
      }
    }
    // bit mask of the non-zero staged samples (word 4 i + wave covers samples 256 i + 64 wave + lane)
    unsigned zero_l = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool nz = v[i].re != (F)0 || v[i].im != (F)0;
      const unsigned long long bal = __ballot(nz);
      if (lane == 0) L.nzw[4 * i + wave] = bal;
      zero_l |= nz ? 0u : 2u;
    }
    {
      const unsigned any = (__ballot(mixed_l != 0u) != 0ull ? 1u : 0u) | (__ballot(zero_l != 0u) != 0ull ? 2u : 0u);
      if (lane == 0) L.wflags[wave] = any;
      if (j == 0) L.nzw[32] = 0ull;
    }
    // correlate() publishes nzw / wflags (and, on the first tile, tw / log_tab) with its first barrier
    correlate<F>(v, L.xs, L.tw, spec, lm, w_lane);
    const unsigned flags = L.wflags[0] | L.wflags[1] | L.wflags[2] | L.wflags[3];
    if (!MIXED && (flags & 1u)) {  // block-uniform: leave the tile to the per-sector pass
      if (j == 0) {
        const size_t lin = ((size_t)c * a.P + p) * a.tiles + tile;
        atomicOr(a.mixed_map + (lin >> 5), 1u << (lin & 31));
        atomicAdd(a.mixed_cnt, 1u);
      }
      return;
    }
    if (flags & 2u) {  // some staged sample is 0: restore the exact zeros of the direct form
      __syncthreads();  // (block-uniform) everyone has read wflags; wp is rebuilt below
      if (j < 64) {
        const unsigned cnt = j < 32 ? (unsigned)__popcll(L.nzw[j]) : 0u;
        unsigned incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        if (j < 33) L.wp[j] = incl - cnt;  // exclusive prefix; wp[32] = total
      }
      __syncthreads();
      const int lo = zr_lo, hi = zr_hi;
      const int wave_u = __builtin_amdgcn_readfirstlane(wave);
      auto nz_before = [&](int n) {  // non-zero staged samples in [0, n)
        return L.wp[n >> 6] + (unsigned)__popcll(L.nzw[n >> 6] & ((1ull << (n & 63)) - 1ull));
      };
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = j + 256 * i;
        // (outputs past the ping's end are never stored; and when the samples COMMON to the 64 windows of the
        // wavefront's outputs hold a non-zero one -- a scalar test -- no window is all zeros)
        const int t0 = 256 * i + 64 * wave_u;
        const int ua = t0 + 63 + lo, ub = min(t0 + hi, kN);
        if (ua < ub && nz_before(ua) != nz_before(ub)) continue;
        if (t < a.out_per_tile && k_begin + t < S) {
          const int ta = t + lo, tb = min(t + hi, kN);
          if (nz_before(ta) == nz_before(tb)) v[i] = C2<F>{(F)0, (F)0};
        }
      }
    }
    if (!MIXED) break;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (m[MIXED ? i : 0] & (1u << only)) {
        y[MIXED ? i : 0].re += v[i].re;
        y[MIXED ? i : 0].im += v[i].im;
      }
    }
  }

  // ---- sector mean -> prx -> Sv/TS (calibrate_ek.py:483-490, 571-638); the ping's numbers were read up front
  // all eight table reads of the lane go out together (inside the per-sample branches below they were eight round
  // trips to L2, one after the other)
  if (kTvgRegs && tabulated) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int s = k_begin + j + 256 * i;
      tvg_early[i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(tvg_row) + (unsigned)(s < S ? s : S - 1) * 8u);
    }
  }
  T* out = reinterpret_cast<T*>(a.out);
  T* range_out = reinterpret_cast<T*>(a.range_out);
  T* prx_out = reinterpret_cast<T*>(a.prx_out);
  double rmin = __builtin_inf(), rmax = -__builtin_inf();
  unsigned rnan = 0;
  // the ping's rows of the outputs (uniform) + 32-bit offsets: scalar-base stores
  T* out_ping = out + row * S;
  T* range_ping = range_out ? range_out + row * S : nullptr;
  T* prx_ping = prx_out ? prx_out + row * S : nullptr;
  auto epilogue = [&](auto clean_tag) {
    constexpr bool CLEAN = decltype(clean_tag)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = j + 256 * i;
      const int s = k_begin + t;
      if (t < a.out_per_tile && s < S) {
        const unsigned nvalid = CLEAN ? (unsigned)B
                                      : (MIXED ? __popc(m[MIXED ? i : 0] & full) : (((vbits >> i) & 1u) ? (unsigned)B : 0u));
        const C2<F> yi = MIXED ? y[MIXED ? i : 0] : v[i];
        T mr, mi;
        if (!CLEAN && nvalid == 0u) {
          mr = mi = epa::M<T>::nan();
        } else {
          const double invn = MIXED ? inv_norm / (double)nvalid : inv_norm_b;
          mr = (T)((double)yi.re * invn);
          mi = (T)((double)yi.im * invn);
        }
        T prx = pscale * (mr * mr + mi * mi);
        if (!(prx > (T)0)) prx = epa::M<T>::nan();
        const double R = ((double)s * ra) * rb;  // range.py:138 operation order
        T tvg;
        if (tabulated) {  // (block-uniform)
          tvg = kTvgRegs ? (T)tvg_early[kTvgRegs ? i : 0] : (T)tvg_tab[s];
        } else {
          T rt = sub_rn((T)R, shift);  // never contracted with the range product into an fma
          if (!(rt > (T)0)) rt = epa::M<T>::nan();
          tvg = nspread * epa::fast_log10_lean(rt, L.log_tab, lk) + alpha2 * rt;
        }
        // prx (and rt) are positive or NaN here: the lean log (zero / subnormal / inf / NaN through a rare branch)
        // the range the reference calibrates with is the MASKED echo_range (NaN where beam 0 is, range.py:143-148): a
        // sample whose beam 0 is missing is NaN even when its other sectors are valid (calibrate_ek.py:571-576)
        const bool range_ok = CLEAN || ((vbits >> (8 + i)) & 1u) != 0u;
        const T val = range_ok ? ((T)10 * epa::fast_log10_lean(prx, L.log_tab, lk) + tvg) + Aadd : epa::M<T>::nan();
        const unsigned o = (unsigned)s;
        out_ping[o] = val;
        if (range_out || a.stats_part) {  // (statistics without the array: epa_range_complex writes it when asked for)
          if (range_out) range_ping[o] = range_ok ? (T)R : epa::M<T>::nan();
          const double rr = (double)(T)R;
          if (CLEAN) {  // (rr is a number: the hardware minimum / maximum without the canonicalising copy)
            rmin = vmin_num(rmin, rr);
            rmax = vmax_num(rmax, rr);
          } else if (range_ok) {
            rmin = fmin(rmin, rr);
            rmax = fmax(rmax, rr);
          } else {
            ++rnan;
          }
        }
        if (prx_out) prx_ping[o] = prx;
      }
    }
  };
  if (clean) epilogue(std::true_type{});
  else epilogue(std::false_type{});
  if (a.stats_part) {  // {nanmin, nanmax, NaN count} of the echo_range written by this workgroup
    double cnt = (double)rnan;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      rmin = fmin(rmin, __shfl_down(rmin, o, 64));
      rmax = fmax(rmax, __shfl_down(rmax, o, 64));
      cnt += __shfl_down(cnt, o, 64);
    }
    if (lane == 0) {
      L.sred[3 * wave] = rmin;
      L.sred[3 * wave + 1] = rmax;
      L.sred[3 * wave + 2] = cnt;
    }
    __syncthreads();
    if (j == 0) {
      const size_t lin = ((size_t)c * a.P + p) * a.tiles + tile;
      double* dst = a.stats_part + 3 * (lin & (kStatSlots - 1));
      const double mn = fmin(fmin(L.sred[0], L.sred[3]), fmin(L.sred[6], L.sred[9]));
      const double mx = fmax(fmax(L.sred[1], L.sred[4]), fmax(L.sred[7], L.sred[10]));
      const double nn = (L.sred[2] + L.sred[5]) + (L.sred[8] + L.sred[11]);
      if (mn <= mx) {  // the workgroup wrote at least one number
        atomicMin(dst, mn);
        atomicMax(dst + 1, mx);
      }
      if (nn > 0.0) atomicAdd(dst + 2, nn);
    }
  }
}

template <typename InT, typename T, typename F, int NB, bool MIXED>
__global__ __launch_bounds__(epa::kBlock, MIXED ? 1 : (sizeof(F) == 4 ? EPA_FFT_WAVES_F32 : EPA_FFT_WAVES_F64))
void sv_complex_fft_kernel(FftArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char xs[Xs<F>::kBytes];
  __shared__ C2<F> tw[kTwEntries<F>];
  __shared__ unsigned long long nzw[33];
  __shared__ unsigned wp[34];
  __shared__ unsigned wflags[4];
  __shared__ double2 log_tab[sizeof(T) == 8 ? epa::kLogTabN : 1];
  __shared__ double sred[12];
  if (MIXED && *a.mixed_cnt == 0u) return;  // the usual case: no tile was deferred

  stage_tables<T, F>(a, tw, log_tab);
  const LaneMap lm = lane_map();
  const TileLds<F, T> L{xs, tw, nzw, wp, wflags, log_tab, sred};
  if (!MIXED) {
    const int c = blockIdx.y;
    const int p = blockIdx.x / a.tiles;
    process_tile<InT, T, F, NB, false>(a, L, lm, c, p, blockIdx.x - p * a.tiles);
  } else {
    for (int w = blockIdx.x; w < a.map_words; w += gridDim.x) {
      unsigned bits = a.mixed_map[w];
      while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1u;
        const size_t lin = (size_t)w * 32 + bit;
        const int tile = (int)(lin % a.tiles);
        const size_t row = lin / a.tiles;
        __syncthreads();  // the previous tile is done with the LDS arrays
        process_tile<InT, T, F, NB, true>(a, L, lm, (int)(row / a.P), (int)(row % a.P), tile);
      }
    }
  }
}

template <typename InT, typename T, typename F>
int launch_fft(FftArgs& a, hipStream_t st) {
  const dim3 grid((unsigned)((long long)a.P * a.tiles), (unsigned)a.C);
  const bool b4 = a.B == 4 && (reinterpret_cast<uintptr_t>(a.re) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(a.im) & 15u) == 0;
  const int slow_grid = a.map_words < 2048 ? a.map_words : 2048;
  if (b4) {
    hipLaunchKernelGGL((sv_complex_fft_kernel<InT, T, F, 4, false>), grid, dim3(epa::kBlock), 0, st, a);
    hipLaunchKernelGGL((sv_complex_fft_kernel<InT, T, F, 4, true>), dim3(slow_grid), dim3(epa::kBlock), 0, st, a);
  } else {
    hipLaunchKernelGGL((sv_complex_fft_kernel<InT, T, F, 0, false>), grid, dim3(epa::kBlock), 0, st, a);
    hipLaunchKernelGGL((sv_complex_fft_kernel<InT, T, F, 0, true>), dim3(slow_grid), dim3(epa::kBlock), 0, st, a);
  }
  return epa::check_launch("sv_complex_fft_kernel");
}

// echo_range of complex samples alone: (s * ra) * rb per (channel, ping) coefficient row, NaN where the real part of
// sector 0 is (range.py:138-148) -- what the sample kernels write as range_out
template <typename InT, typename T>
__global__ __launch_bounds__(epa::kBlock) void range_complex_kernel(const InT* __restrict__ re,
                                                                    const double* __restrict__ ccoef, long long rows,
                                                                    int S, int B, T* __restrict__ range_out) {
  const int s = blockIdx.y * epa::kBlock + threadIdx.x;
  if (s >= S) return;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const double* cc = ccoef + (size_t)row * EPA_NCCOEF;
    const double ra = cc[EPA_CC_RA], rb = cc[EPA_CC_RB];
    const size_t o = (size_t)row * S + s;
    const InT v = re[o * B];
    range_out[o] = (v == v) ? (T)(((double)s * ra) * rb) : epa::M<T>::nan();
  }
}

}  // namespace

extern "C" int epa_range_complex(const void* re, int in_dtype, const double* ccoef, int C, int P, int S, int B,
                                 void* range_out, int out_dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(re && ccoef && range_out, "epa_range_complex: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && B > 0, "epa_range_complex: C=%d P=%d S=%d B=%d", C, P, S, B);
  EPA_CHECK_ARG((in_dtype == EPA_F32 || in_dtype == EPA_F64) && (out_dtype == EPA_F32 || out_dtype == EPA_F64),
                "epa_range_complex: bad dtype in=%d out=%d", in_dtype, out_dtype);
  const long long rows = (long long)C * P;
  const int chunks = (S + epa::kBlock - 1) / epa::kBlock;
  long long gx = 16384 / chunks;
  if (gx < 1) gx = 1;
  if (gx > rows) gx = rows;
  const dim3 grid((unsigned)gx, (unsigned)chunks);
  hipStream_t st = (hipStream_t)stream;
#define EPA_RC(IN, OUT)                                                                                          \
  hipLaunchKernelGGL((range_complex_kernel<IN, OUT>), grid, dim3(epa::kBlock), 0, st, (const IN*)re, ccoef, rows, S, B, \
                     (OUT*)range_out)
  if (in_dtype == EPA_F32 && out_dtype == EPA_F64) EPA_RC(float, double);
  else if (in_dtype == EPA_F32) EPA_RC(float, float);
  else if (out_dtype == EPA_F64) EPA_RC(double, double);
  else EPA_RC(double, float);
#undef EPA_RC
  return epa::check_launch("range_complex_kernel");
}

static int sv_complex_fft_entry(const void* re, const void* im, int in_dtype, const float* replica,
                                const int32_t* replica_off, const int32_t* replica_id, int n_replicas, int max_taps,
                                const double* ccoef, int C, int P, int S, int B, int cal_type, void* out,
                                void* range_out, void* prx_out, int out_dtype, int fft_dtype, double* workspace,
                                double* range_stats_out, epa_stream_t stream);

extern "C" int epa_sv_complex_fft(const void* re, const void* im, int in_dtype, const float* replica,
                                  const int32_t* replica_off, int max_taps, const double* ccoef, int C,
                                  int P, int S, int B, int cal_type, void* out, void* range_out,
                                  void* prx_out, int out_dtype, int fft_dtype, double* workspace,
                                  double* range_stats_out, epa_stream_t stream) {
  return sv_complex_fft_entry(re, im, in_dtype, replica, replica_off, nullptr, C, max_taps, ccoef, C, P, S, B, cal_type,
                              out, range_out, prx_out, out_dtype, fft_dtype, workspace, range_stats_out, stream);
}

extern "C" int epa_sv_complex_fft_indexed(const void* re, const void* im, int in_dtype, const float* replica,
                                          const int32_t* replica_off, const int32_t* replica_id, int n_replicas,
                                          int max_taps, const double* ccoef, int C, int P, int S, int B, int cal_type,
                                          void* out, void* range_out, void* prx_out, int out_dtype, int fft_dtype,
                                          double* workspace, double* range_stats_out, epa_stream_t stream) {
  EPA_CHECK_ARG(replica_id && n_replicas > 0, "epa_sv_complex_fft_indexed: replica_id and a positive n_replicas are needed");
  return sv_complex_fft_entry(re, im, in_dtype, replica, replica_off, replica_id, n_replicas, max_taps, ccoef, C, P, S, B,
                              cal_type, out, range_out, prx_out, out_dtype, fft_dtype, workspace, range_stats_out, stream);
}

static int sv_complex_fft_entry(const void* re, const void* im, int in_dtype, const float* replica,
                                const int32_t* replica_off, const int32_t* replica_id, int n_replicas, int max_taps,
                                const double* ccoef, int C, int P, int S, int B, int cal_type, void* out,
                                void* range_out, void* prx_out, int out_dtype, int fft_dtype, double* workspace,
                                double* range_stats_out, epa_stream_t stream) {
  EPA_CHECK_ARG(re && im && ccoef && out && replica && replica_off && workspace,
                "epa_sv_complex_fft: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && B > 0, "epa_sv_complex_fft: C=%d P=%d S=%d B=%d", C, P, S, B);
  EPA_CHECK_ARG(B <= kMaxBeams, "epa_sv_complex_fft: at most %d sectors supported (got %d)", kMaxBeams, B);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_complex_fft: bad cal_type");
  EPA_CHECK_ARG((in_dtype == EPA_F32 || in_dtype == EPA_F64) && (out_dtype == EPA_F32 || out_dtype == EPA_F64) &&
                    (fft_dtype == EPA_F32 || fft_dtype == EPA_F64),
                "epa_sv_complex_fft: bad dtype in=%d out=%d fft=%d", in_dtype, out_dtype, fft_dtype);
  if (max_taps < 1 || max_taps > kN / 2) {
    epa::set_error("epa_sv_complex_fft: replicas of 1..%d taps only (got %d); use epa_sv_complex", kN / 2, max_taps);
    return EPA_EUNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int W = n_replicas > C ? n_replicas : C;  // the workspace layout is that of W "channels" (header: WS_DOUBLES)
  hipLaunchKernelGGL(replica_prepare_kernel, dim3(n_replicas), dim3(epa::kBlock), 0, st, replica, replica_off, W, workspace,
                     range_stats_out ? 1 : 0, workspace + ws_logtab(W, P, S));
  if (int rc = epa::check_launch("replica_prepare_kernel")) return rc;
  FftArgs a{};
  a.re = re; a.im = im; a.ccoef = ccoef; a.ws = workspace;
  a.C = C; a.P = P; a.S = S; a.B = B;
  a.W = W; a.replica_id = replica_id;
  a.out_per_tile = kN - max_taps + 1;
  a.tiles = (S + a.out_per_tile - 1) / a.out_per_tile;
  a.nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  a.out = out; a.range_out = range_out; a.prx_out = prx_out;
  a.stats_part = range_stats_out ? workspace + ws_stats(W) : nullptr;
  const size_t ntiles = (size_t)C * P * a.tiles;
  EPA_CHECK_ARG(ntiles < ((size_t)1 << 36), "epa_sv_complex_fft: too many tiles");
  a.map_words = (int)((ntiles + 31) / 32);
  a.mixed_cnt = reinterpret_cast<unsigned*>(workspace + ws_mixed(W));
  a.mixed_map = a.mixed_cnt + 2;
  EPA_CHECK_HIP(hipMemsetAsync(a.mixed_cnt, 0, 8 + 4 * (size_t)a.map_words, st));
  double* tvg = workspace + ws_tvg(W, P, S);
  a.tvg = tvg;
  a.log_tab = workspace + ws_logtab(W, P, S);
  {
    const dim3 tg((unsigned)((S + epa::kBlock - 1) / epa::kBlock), (unsigned)C);
    if (out_dtype == EPA_F64) {
      hipLaunchKernelGGL(tvg_table_kernel<double>, tg, dim3(epa::kBlock), 0, st, ccoef, P, S, a.nspread, tvg);
      if (int rc2 = epa::check_launch("tvg_table_kernel")) return rc2;
    }
  }
  int rc;
#define EPA_FFT_CASE(IN, OUT, FF) rc = launch_fft<IN, OUT, FF>(a, st)
  const int key = (in_dtype == EPA_F64 ? 4 : 0) | (out_dtype == EPA_F64 ? 2 : 0) | (fft_dtype == EPA_F64 ? 1 : 0);
  switch (key) {
    case 0: EPA_FFT_CASE(float, float, float); break;
    case 1: EPA_FFT_CASE(float, float, double); break;
    case 2: EPA_FFT_CASE(float, double, float); break;
    case 3: EPA_FFT_CASE(float, double, double); break;
    case 4: EPA_FFT_CASE(double, float, float); break;
    case 5: EPA_FFT_CASE(double, float, double); break;
    case 6: EPA_FFT_CASE(double, double, float); break;
    default: EPA_FFT_CASE(double, double, double); break;
  }
#undef EPA_FFT_CASE
  if (rc) return rc;
  if (range_stats_out) return epa_minmax_final(a.stats_part, kStatSlots, range_stats_out, st);
  return EPA_OK;
}
