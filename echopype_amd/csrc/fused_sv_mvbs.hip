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
};

// ---- 10^(u/10) for the linear-domain average -----------------------------------------------------
// f64: 2^(u*K), K = log2(10)/10, argument reduced to m/256 + r/256 (|r| <= 1/2): 2^(m>>8) from the
// exponent, 2^((m&255)/256) from a 256-entry LDS table (built per workgroup with the full-precision
// exp2), 2^(r/256) from a degree-5 Taylor polynomial (|z| <= 1.4e-3 -> truncation < 1e-18 relative).
// ~16 fp64 instructions instead of ~35 for ocml exp10, same 1-ulp class accuracy (checked by
// epa_selftest_exp10 in the GPU tests).
__device__ __forceinline__ double lin_from_db(double u, const double* __restrict__ tab) {
  constexpr double K256_HI = 85.04135922911648;       // 256*log2(10)/10, rounded to double
  constexpr double K256_LO = -4.272771985668806e-15;  // 256*log2(10)/10 - K256_HI
  constexpr double Z = 0.0027076061740622863;         // ln(2)/256
  const double t = u * K256_HI;
  const double m = __builtin_rint(t);
  double r = fma(u, K256_HI, -m);
  r = fma(u, K256_LO, r);
  const double z = r * Z;
  double p = fma(z, 1.0 / 120.0, 1.0 / 24.0);
  p = fma(p, z, 1.0 / 6.0);
  p = fma(p, z, 0.5);
  p = fma(p, z, 1.0);
  p = fma(p, z, 1.0);
  // |u| beyond ~ +-3000 dB saturates like exp10 (inf / 0); NaN propagates through t
  const double mc = fmin(fmax(m, -300000.0), 300000.0);
  const int mi = (int)mc;
  const double v = ldexp(p * tab[mi & 255], mi >> 8);
  // non-finite arguments: NaN -> NaN, +inf -> +inf, -inf -> 0 (as exp10)
  return (fabs(t) < __builtin_inf()) ? v : (t < 0.0 ? 0.0 : t);
}
__device__ __forceinline__ float lin_from_db(float u, const double*) {
  return epa::M<float>::exp10(u * 0.1f);
}

template <typename T>
__device__ __noinline__ T log10_slow(T x) {
  return epa::M<T>::log10(x);
}

template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
  unsafeAtomicAdd(p, v);
}

template <typename T, bool WRITE_SV>
__global__ __launch_bounds__(epa::kBlock) void fused_sv_mvbs_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef,
    const int32_t* __restrict__ bin_start, T* __restrict__ sv_out, T* __restrict__ mvbs_out,
    T* __restrict__ sum_out, uint32_t* __restrict__ cnt_out, Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  double* tab = reinterpret_cast<double*>(smem + a.tab_off);
  if (sizeof(T) == 8) tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 256.0));  // kBlock == 256

  const int c = blockIdx.y, tb = blockIdx.x;
  const int S = a.S, n_rbins = a.n_rbins;
  const int pb = bin_start[tb], pe = bin_start[tb + 1];
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  __syncthreads();

  const T nspread = (T)a.nspread;
  const bool guard = a.guard, mask_range = a.mask_range, skipna = a.skipna, cr = a.closed_right;
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const T NaN = epa::M<T>::nan();
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const float* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  T* __restrict__ sv_c = WRITE_SV ? sv_out + (size_t)c * a.P * S : nullptr;

  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int s0 = chunk0 + threadIdx.x * VEC;
    if (s0 >= S) continue;
    double sdbl[VEC], blo[VEC], bhi[VEC];
    T nL[VEC], acc_sum[VEC];
    int acc_rb[VEC];
    uint32_t acc_cnt[VEC];
    double dcur = __builtin_nan("");
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      sdbl[j] = (double)(s0 + j);
      blo[j] = 1.0;
      bhi[j] = 0.0;
      nL[j] = NaN;
      acc_sum[j] = (T)0;
      acc_rb[j] = -1;
      acc_cnt[j] = 0u;
    }
    for (int p = pb; p < pe; ++p) {
      const epa::CoefRow r = rowp0[p];  // wave-uniform address -> scalar loads
      const size_t off = (size_t)p * S + s0;
      const float4 in4 = *reinterpret_cast<const float4*>(raw_c + off);
      const float in[VEC] = {in4.x, in4.y, in4.z, in4.w};
      if (!(r.d == dcur)) {  // uniform; once per column for a file with constant tau / interval
        dcur = r.d;
        for (int j = 0; j < VEC; ++j) nL[j] = nspread * log10_slow<T>((T)(sdbl[j] - r.d));
      }
      const T g = (T)r.g, a2 = (T)r.alpha2, A0 = (T)r.A0;
      T sv[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const double x = (sdbl[j] * r.ra) * r.rb + r.r0;  // echo_range, range.py:138 order
        const double rtd = x - r.shift;
        const T rt = (T)rtd;
        T spread = nL[j];
        if (guard) {
          const bool pos = rtd > 0.0;
          if (pos && !(spread > -(T)__builtin_inf()))  // rounding residue of R - shift (rare)
            spread = nspread * (log10_slow<T>(rt) - log10_slow<T>((T)(r.ra * r.rb)));
          spread = pos ? spread : NaN;
        }
        sv[j] = fma(g, (T)in[j], spread) + fma(a2, rt, A0);
        const T v = lin_from_db(sv[j], tab);
        // range-bin membership: still inside the bin of the previous ping?  (NaN raw -> NaN range
        // when masked: never inside)
        const bool xok = !mask_range || (in[j] == in[j]);
        const bool same = xok && (cr ? (x > blo[j] && x <= bhi[j]) : (x >= blo[j] && x < bhi[j]));
        if (!same) {
          const int rb = xok ? epa::range_bin_index(x, bin, inv_bin, n_rbins, cr) : -1;
          if (rb != acc_rb[j]) {
            if (acc_rb[j] >= 0 && acc_cnt[j] > 0u) {
              lds_add(lsum + acc_rb[j], acc_sum[j]);
              atomicAdd(lcnt + acc_rb[j], acc_cnt[j]);
            }
            acc_rb[j] = rb;
            acc_sum[j] = (T)0;
            acc_cnt[j] = 0u;
          }
          blo[j] = rb >= 0 ? (double)rb * bin : 1.0;
          bhi[j] = rb >= 0 ? (double)(rb + 1) * bin : 0.0;
        }
        const bool take = (acc_rb[j] >= 0) && (!skipna || v == v);
        acc_sum[j] += take ? v : (T)0;
        acc_cnt[j] += take ? 1u : 0u;
      }
      if (WRITE_SV) epa::store_vec<T, VEC>(sv_c + off, sv);
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (acc_rb[j] >= 0 && acc_cnt[j] > 0u) {
        lds_add(lsum + acc_rb[j], acc_sum[j]);
        atomicAdd(lcnt + acc_rb[j], acc_cnt[j]);
      }
    }
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
int launch(Args& a, const float* raw, const double* coef, const int32_t* bin_start, void* sv_out,
           void* mvbs_out, void* sum_out, uint32_t* cnt_out, int C, size_t lds_bytes, hipStream_t st) {
  const dim3 grid((unsigned)a.n_tbins, (unsigned)C);
  a.tab_off = (unsigned)((lds_bytes + 15) & ~(size_t)15);
  lds_bytes = a.tab_off + 256 * sizeof(double);
#define EPA_FL(W)                                                                              \
  do {                                                                                         \
    auto kern = fused_sv_mvbs_kernel<T, W>;                                                    \
    if (lds_bytes > 64 * 1024)                                                                 \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                        (int)lds_bytes));                                      \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds_bytes, st, raw,                      \
                       reinterpret_cast<const epa::CoefRow*>(coef), bin_start, (T*)sv_out,     \
                       (T*)mvbs_out, (T*)sum_out, cnt_out, a);                                 \
  } while (0)
  if (sv_out) EPA_FL(true); else EPA_FL(false);
#undef EPA_FL
  return epa::check_launch("fused_sv_mvbs_kernel");
}

__global__ __launch_bounds__(epa::kBlock) void selftest_lin_kernel(const double* __restrict__ u,
                                                                   double* __restrict__ out,
                                                                   size_t n) {
  __shared__ double tab[256];
  tab[threadIdx.x] = exp2((double)threadIdx.x * (1.0 / 256.0));
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = lin_from_db(u[i], tab);
}

}  // namespace epa_fused

extern "C" int epa_selftest_lin_from_db(const double* u, double* out, size_t n, epa_stream_t stream) {
  EPA_CHECK_ARG(u && out, "epa_selftest_lin_from_db: NULL array argument");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  hipLaunchKernelGGL(epa_fused::selftest_lin_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)),
                     dim3(epa::kBlock), 0, (hipStream_t)stream, u, out, n);
  return epa::check_launch("selftest_lin_kernel");
}

// Called by epa_sv_mvbs_fused (block_reduce.hip) when the fast path applies.
int epa_fused_fast_path(const float* raw, const double* coef, int C, int P, int S, double nspread,
                        unsigned cal_flags, const int32_t* bin_start, int n_tbins, double range_bin,
                        int n_rbins, unsigned bin_flags, double fill_value, void* sv_out,
                        void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                        size_t lds_bytes, unsigned cnt_off, hipStream_t st) {
  epa_fused::Args a{};
  a.P = P; a.S = S; a.n_tbins = n_tbins; a.n_rbins = n_rbins;
  a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.nspread = nspread; a.fill_value = fill_value;
  a.guard = (cal_flags & EPA_FLAG_GUARD_POS) != 0;
  a.mask_range = (cal_flags & EPA_FLAG_MASK_RANGE) != 0;
  a.skipna = (bin_flags & EPA_BIN_SKIPNA) != 0;
  a.closed_right = (bin_flags & EPA_BIN_CLOSED_RIGHT) != 0;
  a.cnt_off = cnt_off;
  if (dtype == EPA_F64)
    return epa_fused::launch<double>(a, raw, coef, bin_start, sv_out, mvbs_out, sum_out, cnt_out, C,
                                     lds_bytes, st);
  return epa_fused::launch<float>(a, raw, coef, bin_start, sv_out, mvbs_out, sum_out, cnt_out, C,
                                  lds_bytes, st);
}
