// K1: fused power-sample calibration -- one HBM-coalesced pass replaces the ~15 whole-array
// xarray temporaries of
//   /root/reference/echopype/calibrate/range.py:98-201
//   /root/reference/echopype/calibrate/calibrate_ek.py:104-110,154-184      (EK60 / EK80 power)
//   /root/reference/echopype/calibrate/range.py:69-95 + calibrate_azfp.py:64-97  (AZFP)
//
// Layout: raw f32 (C,P,S) row-major; one coefficient row (64 B) per (c,p).  A workgroup of 256
// threads owns one 1024-sample range chunk (blockIdx.y) and strides over the rows; lanes map to
// samples so that every wavefront access is a contiguous run of 16 B per lane (LaneMap in
// sample_math.h).  The row constants are wave-uniform -> scalar loads.  HBM-bound: 4 B in + 8 B (f64) out per sample (+8 B with
// echo_range); MFMA is irrelevant here (no contraction).
#include "sample_math.h"

namespace {

// Vector path: S even (f64) / S % 4 == 0 (f32) and 16-byte aligned buffers.
template <typename T, bool RANGE, bool STATS = false>
__global__ __launch_bounds__(epa::kBlock) void sv_power_kernel(const float* __restrict__ raw,
                                                               const epa::CoefRow* __restrict__ coef,
                                                               long long rows, int S, T nspread,
                                                               unsigned flags, T* __restrict__ out,
                                                               T* __restrict__ range_out,
                                                               double* __restrict__ part) {
  // blockIdx.y = range chunk (fixed for the life of the block, so that the lane's range columns
  // and their cached log10(s - d) never change); blockIdx.x strides over the (channel, ping) rows.
  using LM = epa::LaneMap<T>;
  constexpr int NSEG = LM::NSEG, LEN = LM::LEN;
  const bool guard = flags & EPA_FLAG_GUARD_POS;
  const bool mask_range = flags & EPA_FLAG_MASK_RANGE;
  int s0[NSEG];
  bool act[NSEG];
#pragma unroll
  for (int g = 0; g < NSEG; ++g) {
    s0[g] = LM::first(blockIdx.y * 1024, g);
    act[g] = s0[g] < S;
  }
  if (!STATS && !act[0]) return;
  // STATS: {min, max, NaN count} of the echo_range (written when RANGE, else left to epa_range_power for whoever asks
  // for the array), one partial per workgroup (no lane leaves early: the wavefront reduction at the end needs them all)
  double lo = __builtin_inf(), hi = -__builtin_inf(), nn = 0.0;
  epa::ColumnLog<T, LEN> col[NSEG];
  for (long long row = blockIdx.x; row < rows && act[0]; row += gridDim.x) {
    const epa::RowK<T> rk(coef[row]);
#pragma unroll
    for (int g = 0; g < NSEG; ++g) {
      if (!act[g]) continue;
      col[g].update(rk.d, s0[g], nspread);
      const size_t off = (size_t)row * S + s0[g];
      epa::RawVec<LEN> in;
      in.load(raw + off);
      T o[LEN], rg[LEN];
#pragma unroll
      for (int j = 0; j < LEN; ++j) {
        const double r = rk.range(s0[g] + j);
        o[j] = epa::cal_power_sample<T>(in.v[j], s0[g] + j, rk, nspread, col[g].nL[j], guard, r);
        if (RANGE || STATS) rg[j] = (mask_range && !(in.v[j] == in.v[j])) ? epa::M<T>::nan() : (T)r;
        if (STATS) {
          const double x = (double)rg[j];
          lo = fmin(lo, x);  // fmin / fmax ignore a NaN operand
          hi = fmax(hi, x);
          nn += x == x ? 0.0 : 1.0;
        }
      }
      epa::store_vec<T, LEN>(out + off, o);
      if (RANGE) epa::store_vec<T, LEN>(range_out + off, rg);
    }
  }
  if (STATS) {
    __shared__ double slo[4], shi[4], snn[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = fmin(lo, __shfl_down(lo, o, 64));
      hi = fmax(hi, __shfl_down(hi, o, 64));
      nn += __shfl_down(nn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; snn[threadIdx.x >> 6] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* pp = part + 3 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
      pp[0] = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
      pp[1] = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
      pp[2] = (snn[0] + snn[1]) + (snn[2] + snn[3]);
    }
  }
}

// Scalar path for odd sizes / unaligned buffers: one sample per lane.
template <typename T, bool RANGE>
__global__ __launch_bounds__(epa::kBlock) void sv_power_scalar_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef, long long rows, int S,
    T nspread, unsigned flags, T* __restrict__ out, T* __restrict__ range_out) {
  const bool guard = flags & EPA_FLAG_GUARD_POS;
  const bool mask_range = flags & EPA_FLAG_MASK_RANGE;
  const int s = blockIdx.y * epa::kBlock + threadIdx.x;
  if (s >= S) return;
  epa::ColumnLog<T, 1> col;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const epa::RowK<T> rk(coef[row]);
    col.update(rk.d, s, nspread);
    const size_t off = (size_t)row * S + s;
    const float v = raw[off];
    const double r = rk.range(s);
    out[off] = epa::cal_power_sample<T>(v, s, rk, nspread, col.nL[0], guard, r);
    if (RANGE) range_out[off] = (mask_range && !(v == v)) ? epa::M<T>::nan() : (T)r;
  }
}

template <typename T>
int launch(const float* raw, const double* coef, int C, int P, int S, int cal_type, unsigned flags,
           void* out, void* range_out, double* part, double* stats_out, hipStream_t st) {
  const long long rows = (long long)C * P;
  const T nspread = cal_type == EPA_CAL_SV ? (T)20 : (T)40;
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  const int need = sizeof(T) == 8 ? 2 : 4;  // samples per 16-B output access
  const bool vec = (S % need == 0) && al16(raw) && al16(out) && al16(range_out);
  const int chunk = vec ? 1024 : epa::kBlock;
  const int chunks_per_row = (S + chunk - 1) / chunk;
  long long gx = 8192 / chunks_per_row;
  if (gx < 1) gx = 1;
  if (gx > rows) gx = rows;
  const dim3 grid((unsigned)gx, (unsigned)chunks_per_row);
  const epa::CoefRow* cf = reinterpret_cast<const epa::CoefRow*>(coef);
#define EPA_LAUNCH(K, R)                                                                        \
  hipLaunchKernelGGL((K<T, R>), grid, dim3(epa::kBlock), 0, st, raw, cf, rows, S, nspread, flags, \
                     (T*)out, (T*)range_out)
  if (!vec && stats_out && !range_out) {
    epa::set_error("epa_sv_power_stats: range statistics without the echo_range array need S %% %d == 0 and 16-byte "
                   "aligned buffers (S=%d)", need, S);
    return EPA_EUNSUPPORTED;
  }
  if (vec && stats_out) {  // echo_range statistics as a by-product: one partial per workgroup, then one small kernel
    if (range_out)
      hipLaunchKernelGGL((sv_power_kernel<T, true, true>), grid, dim3(epa::kBlock), 0, st, raw, cf, rows, S, nspread,
                         flags, (T*)out, (T*)range_out, part);
    else
      hipLaunchKernelGGL((sv_power_kernel<T, false, true>), grid, dim3(epa::kBlock), 0, st, raw, cf, rows, S, nspread,
                         flags, (T*)out, (T*)range_out, part);
    if (int rc = epa::check_launch("sv_power_kernel")) return rc;
    return epa_minmax_final(part, (int)(grid.x * grid.y), stats_out, st);
  }
  if (vec) {
    if (range_out)
      hipLaunchKernelGGL((sv_power_kernel<T, true>), grid, dim3(epa::kBlock), 0, st, raw, cf, rows, S, nspread, flags,
                         (T*)out, (T*)range_out, (double*)nullptr);
    else
      hipLaunchKernelGGL((sv_power_kernel<T, false>), grid, dim3(epa::kBlock), 0, st, raw, cf, rows, S, nspread, flags,
                         (T*)out, (T*)range_out, (double*)nullptr);
  } else {
    if (range_out) EPA_LAUNCH(sv_power_scalar_kernel, true); else EPA_LAUNCH(sv_power_scalar_kernel, false);
  }
#undef EPA_LAUNCH
  if (int rc = epa::check_launch("sv_power_kernel")) return rc;
  if (stats_out)  // odd sizes / unaligned buffers: a separate sweep of the echo_range just written
    return epa_nanminmax(range_out, (size_t)rows * S, sizeof(T) == 8 ? EPA_F64 : EPA_F32, part, stats_out,
                         (epa_stream_t)st);
  return EPA_OK;
}

// echo_range alone, for a caller that asked epa_sv_power_stats not to write it and needs the array after all
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void range_power_kernel(const float* __restrict__ raw,
                                                                  const epa::CoefRow* __restrict__ coef,
                                                                  long long rows, int S, unsigned flags,
                                                                  T* __restrict__ range_out) {
  const bool mask_range = flags & EPA_FLAG_MASK_RANGE;
  const int s = blockIdx.y * epa::kBlock + threadIdx.x;
  if (s >= S) return;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const epa::CoefRow cr = coef[row];
    const size_t off = (size_t)row * S + s;
    const bool nan_in = mask_range && !(raw[off] == raw[off]);
    range_out[off] = nan_in ? epa::M<T>::nan() : (T)epa::row_range(cr, s);
  }
}

}  // namespace

extern "C" int epa_range_power(const float* raw, const double* coef, int C, int P, int S, unsigned flags,
                               void* range_out, int out_dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(coef && range_out, "epa_range_power: NULL array argument");
  EPA_CHECK_ARG(raw || !(flags & EPA_FLAG_MASK_RANGE), "epa_range_power: EPA_FLAG_MASK_RANGE needs the raw samples");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_range_power: C=%d P=%d S=%d must be positive", C, P, S);
  EPA_CHECK_ARG(out_dtype == EPA_F64 || out_dtype == EPA_F32, "epa_range_power: bad out_dtype %d", out_dtype);
  const long long rows = (long long)C * P;
  const int chunks = (S + epa::kBlock - 1) / epa::kBlock;
  long long gx = 16384 / chunks;
  if (gx < 1) gx = 1;
  if (gx > rows) gx = rows;
  const dim3 grid((unsigned)gx, (unsigned)chunks);
  const epa::CoefRow* cf = reinterpret_cast<const epa::CoefRow*>(coef);
  if (out_dtype == EPA_F64)
    hipLaunchKernelGGL(range_power_kernel<double>, grid, dim3(epa::kBlock), 0, (hipStream_t)stream, raw, cf, rows, S,
                       flags, (double*)range_out);
  else
    hipLaunchKernelGGL(range_power_kernel<float>, grid, dim3(epa::kBlock), 0, (hipStream_t)stream, raw, cf, rows, S,
                       flags, (float*)range_out);
  return epa::check_launch("range_power_kernel");
}

extern "C" int epa_sv_power(const float* raw, const double* coef, int C, int P, int S, int cal_type,
                            unsigned flags, void* out, void* range_out, int out_dtype,
                            epa_stream_t stream) {
  EPA_CHECK_ARG(raw && coef && out, "epa_sv_power: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_sv_power: C=%d P=%d S=%d must be positive", C, P, S);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_power: bad cal_type %d",
                cal_type);
  if (out_dtype == EPA_F64)
    return launch<double>(raw, coef, C, P, S, cal_type, flags, out, range_out, nullptr, nullptr, (hipStream_t)stream);
  if (out_dtype == EPA_F32)
    return launch<float>(raw, coef, C, P, S, cal_type, flags, out, range_out, nullptr, nullptr, (hipStream_t)stream);
  epa::set_error("epa_sv_power: bad out_dtype %d", out_dtype);
  return EPA_EINVAL;
}

extern "C" int epa_sv_power_stats(const float* raw, const double* coef, int C, int P, int S, int cal_type,
                                  unsigned flags, void* out, void* range_out, int out_dtype, double* workspace,
                                  double* range_stats_out, epa_stream_t stream) {
  EPA_CHECK_ARG(raw && coef && out && workspace && range_stats_out, "epa_sv_power_stats: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_sv_power_stats: C=%d P=%d S=%d must be positive", C, P, S);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_power_stats: bad cal_type %d", cal_type);
  if (out_dtype == EPA_F64)
    return launch<double>(raw, coef, C, P, S, cal_type, flags, out, range_out, workspace, range_stats_out,
                          (hipStream_t)stream);
  if (out_dtype == EPA_F32)
    return launch<float>(raw, coef, C, P, S, cal_type, flags, out, range_out, workspace, range_stats_out,
                         (hipStream_t)stream);
  epa::set_error("epa_sv_power_stats: bad out_dtype %d", out_dtype);
  return EPA_EINVAL;
}
