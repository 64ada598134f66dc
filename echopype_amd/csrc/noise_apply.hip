// K7: background-noise removal, one elementwise pass.
//
// Replaces /root/reference/echopype/clean/api.py:425-430 (forward-fill of the per-ping-block noise
// to every ping + transmission loss) and :485-487 (linear subtraction, SNR threshold):
//   TL        = 20*log10(R if R >= 1 else 1 [NaN -> 1]) + 2*alpha*R
//   Sv_noise  = noise[c, (p + ping_phase) // ping_num] + TL     (ping_phase: the shard's offset into its first block)
//   L         = 10^(Sv/10) - 10^(Sv_noise/10);  Sv_corr = 10*log10(L) if L > 0 else NaN
//   Sv_corr   = NaN unless Sv_corr - Sv_noise > SNR_threshold
// HBM-bound: reads Sv (+ echo_range unless affine), writes Sv_noise and Sv_corrected.
#include "fast_math.h"
#include "sample_math.h"

namespace {

// SEGLEN samples per 16-B access: f64 -> LaneMap<double> pairs; f32 -> 4 consecutive.  VEC == false:
// scalar fallback for odd sizes / unaligned buffers.
template <typename T, bool VEC>
__global__ __launch_bounds__(epa::kBlock) void noise_apply_kernel(
    const T* __restrict__ sv, const T* __restrict__ range, const epa::CoefRow* __restrict__ coef,
    const float* __restrict__ mask_raw, const double* __restrict__ alpha2, const double* __restrict__ noise, int P, int S,
    long long rows, int ping_num, int ping_phase, int n_pblocks, T snr, T* __restrict__ sv_noise,
    T* __restrict__ sv_corr, unsigned long long* __restrict__ mm_keys) {
  using LM = epa::LaneMap<T>;
  constexpr int NSEG = VEC ? LM::NSEG : 1, LEN = VEC ? LM::LEN : 1;
  __shared__ __attribute__((aligned(16))) unsigned char tabs[epa::kMathTabBytes];
  const epa::MathTabs mt = epa::build_math_tabs(tabs);
  __syncthreads();
  int s0[NSEG];
#pragma unroll
  for (int g = 0; g < NSEG; ++g)
    s0[g] = VEC ? LM::first(blockIdx.y * 1024, g) : (int)(blockIdx.y * epa::kBlock + threadIdx.x);
  // optional by-product: NaN-skipping min / max of both outputs (actual_range, clean/utils.py:392-395)
  double mm[4] = {__builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf()};
  // lanes beyond the row stay alive for the wavefront reduction below (no early return)
  for (long long row = blockIdx.x; row < rows && s0[0] < S; row += gridDim.x) {
    const int c = (int)(row / P), p = (int)(row - (long long)c * P);
    const T nb = (T)noise[(size_t)c * n_pblocks + (p + ping_phase) / ping_num];
    const T a2 = (T)alpha2[row];
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
      if (s0[g] >= S) continue;
      const size_t off = (size_t)row * S + s0[g];
      T v[LEN], x[LEN], on[LEN], oc[LEN];
      epa::load_vec<T, LEN>(sv + off, v);
      if (range) {
        epa::load_vec<T, LEN>(range + off, x);
      } else {
        const epa::CoefRow cr = coef[row];
#pragma unroll
        for (int j = 0; j < LEN; ++j) x[j] = (T)epa::row_range(cr, s0[g] + j);
        if (mask_raw) {  // the echo_range array is NaN where the raw sample is (range.py:143-148): so is Sv_noise
          epa::RawVec<LEN> in;
          in.load(mask_raw + off);
#pragma unroll
          for (int j = 0; j < LEN; ++j)
            if (!(in.v[j] == in.v[j])) x[j] = epa::M<T>::nan();
        }
      }
#pragma unroll
      for (int j = 0; j < LEN; ++j) {
        const T tl = (T)20 * epa::fast_log10(x[j] >= (T)1 ? x[j] : (T)1, mt.log_tab) + a2 * x[j];
        const T sn = nb + tl;
        const T lin = epa::lin_from_db(v[j], mt.exp2_tab) - epa::lin_from_db(sn, mt.exp2_tab);
        T corr = lin > (T)0 ? (T)10 * epa::fast_log10(lin, mt.log_tab) : epa::M<T>::nan();
        if (!(corr - sn > snr)) corr = epa::M<T>::nan();
        on[j] = sn;
        oc[j] = corr;
        if (mm_keys) {  // fmin / fmax ignore NaN operands
          mm[0] = fmin(mm[0], (double)sn);
          mm[1] = fmax(mm[1], (double)sn);
          mm[2] = fmin(mm[2], (double)corr);
          mm[3] = fmax(mm[3], (double)corr);
        }
      }
      if (sv_noise) epa::store_vec<T, LEN>(sv_noise + off, on);
      if (sv_corr) epa::store_vec<T, LEN>(sv_corr + off, oc);
    }
  }
  if (mm_keys) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mm[0] = fmin(mm[0], __shfl_down(mm[0], o, 64));
      mm[1] = fmax(mm[1], __shfl_down(mm[1], o, 64));
      mm[2] = fmin(mm[2], __shfl_down(mm[2], o, 64));
      mm[3] = fmax(mm[3], __shfl_down(mm[3], o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      auto key = [](double v) {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
      };
      if (mm[0] <= mm[1]) {
        atomicMin(mm_keys + 0, key(mm[0]));
        atomicMax(mm_keys + 1, key(mm[1]));
      }
      if (mm[2] <= mm[3]) {
        atomicMin(mm_keys + 2, key(mm[2]));
        atomicMax(mm_keys + 3, key(mm[3]));
      }
    }
  }
}

__global__ void init_minmax_kernel(unsigned long long* p) { p[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull; }
__global__ void decode_minmax_kernel(double* p) {
  const int i = threadIdx.x;
  const unsigned long long k = reinterpret_cast<unsigned long long*>(p)[i];
  const bool none = (i & 1) ? k == 0ull : k == ~0ull;
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  p[i] = none ? __builtin_nan("") : __longlong_as_double(b);
}

inline bool al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
int launch(const void* sv, const void* range, const double* coef, const float* mask_raw, const double* alpha2,
           const double* noise, int C, int P, int S, int ping_num, int ping_phase, double snr, void* sv_noise,
           void* sv_corr, double* minmax_out, hipStream_t st) {
  const long long rows = (long long)C * P;
  const int need = sizeof(T) == 8 ? 2 : 4;
  const bool vec = S % need == 0 && al16(sv) && al16(range) && al16(sv_noise) && al16(sv_corr) && al16(mask_raw);
  const int chunk = vec ? 1024 : epa::kBlock;
  const int chunks_per_row = (S + chunk - 1) / chunk;
  long long gx = 8192 / chunks_per_row;
  if (gx < 1) gx = 1;
  if (gx > rows) gx = rows;
  const dim3 grid((unsigned)gx, (unsigned)chunks_per_row);
  const int n_pblocks = (P + ping_phase + ping_num - 1) / ping_num;
  const epa::CoefRow* cf = reinterpret_cast<const epa::CoefRow*>(coef);
  unsigned long long* mm = reinterpret_cast<unsigned long long*>(minmax_out);
  if (mm) hipLaunchKernelGGL(init_minmax_kernel, dim3(1), dim3(4), 0, st, mm);
  if (vec)
    hipLaunchKernelGGL((noise_apply_kernel<T, true>), grid, dim3(epa::kBlock), 0, st, (const T*)sv,
                       (const T*)range, cf, mask_raw, alpha2, noise, P, S, rows, ping_num, ping_phase, n_pblocks, (T)snr,
                       (T*)sv_noise, (T*)sv_corr, mm);
  else
    hipLaunchKernelGGL((noise_apply_kernel<T, false>), grid, dim3(epa::kBlock), 0, st, (const T*)sv,
                       (const T*)range, cf, mask_raw, alpha2, noise, P, S, rows, ping_num, ping_phase, n_pblocks, (T)snr,
                       (T*)sv_noise, (T*)sv_corr, mm);
  if (int rc = epa::check_launch("noise_apply_kernel")) return rc;
  if (mm) {
    hipLaunchKernelGGL(decode_minmax_kernel, dim3(1), dim3(4), 0, st, minmax_out);
    return epa::check_launch("decode_minmax_kernel");
  }
  return EPA_OK;
}

}  // namespace

extern "C" int epa_noise_apply(const void* sv, const void* range, const double* coef,
                               const double* alpha2, const double* noise, int C, int P, int S,
                               int ping_num, int ping_phase, double snr_threshold, void* sv_noise_out,
                               void* sv_corrected_out, double* minmax_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(sv && alpha2 && noise, "epa_noise_apply: NULL array argument");
  EPA_CHECK_ARG(range || coef, "epa_noise_apply: either range or coef must be given");
  EPA_CHECK_ARG(sv_noise_out || sv_corrected_out, "epa_noise_apply: no output requested");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0, "epa_noise_apply: sizes must be positive");
  EPA_CHECK_ARG(ping_phase >= 0 && ping_phase < ping_num, "epa_noise_apply: ping_phase %d not in [0, %d)", ping_phase,
                ping_num);
  if (dtype == EPA_F64)
    return launch<double>(sv, range, coef, nullptr, alpha2, noise, C, P, S, ping_num, ping_phase, snr_threshold,
                          sv_noise_out, sv_corrected_out, minmax_out, (hipStream_t)stream);
  if (dtype == EPA_F32)
    return launch<float>(sv, range, coef, nullptr, alpha2, noise, C, P, S, ping_num, ping_phase, snr_threshold,
                         sv_noise_out, sv_corrected_out, minmax_out, (hipStream_t)stream);
  epa::set_error("epa_noise_apply: bad dtype %d", dtype);
  return EPA_EINVAL;
}

extern "C" int epa_noise_apply_rows(const void* sv, const double* coef, const float* mask_raw, const double* alpha2,
                                    const double* noise, int C, int P, int S, int ping_num, int ping_phase,
                                    double snr_threshold, void* sv_noise_out, void* sv_corrected_out,
                                    double* minmax_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(sv && coef && alpha2 && noise, "epa_noise_apply_rows: NULL array argument");
  EPA_CHECK_ARG(sv_noise_out || sv_corrected_out, "epa_noise_apply_rows: no output requested");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0, "epa_noise_apply_rows: sizes must be positive");
  EPA_CHECK_ARG(ping_phase >= 0 && ping_phase < ping_num, "epa_noise_apply_rows: ping_phase %d not in [0, %d)",
                ping_phase, ping_num);
  if (dtype == EPA_F64)
    return launch<double>(sv, nullptr, coef, mask_raw, alpha2, noise, C, P, S, ping_num, ping_phase, snr_threshold,
                          sv_noise_out, sv_corrected_out, minmax_out, (hipStream_t)stream);
  if (dtype == EPA_F32)
    return launch<float>(sv, nullptr, coef, mask_raw, alpha2, noise, C, P, S, ping_num, ping_phase, snr_threshold,
                         sv_noise_out, sv_corrected_out, minmax_out, (hipStream_t)stream);
  epa::set_error("epa_noise_apply_rows: bad dtype %d", dtype);
  return EPA_EINVAL;
}
