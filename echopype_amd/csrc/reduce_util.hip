// NaN-skipping min/max of a device array: the reductions the reference forces with
//   ds_Sv[range_var].max(skipna=True)            commongrid/api.py:108-110 (range bin edges)
//   round(float(da.min())), round(float(da.max()))  clean/utils.py:392-395, commongrid/api.py:252-255
// Two launches: per-workgroup partials (wave __shfl reduction, then LDS), then one workgroup.
#include "epa_internal.h"

namespace {

template <typename T>
__device__ __forceinline__ void wave_minmax(double& lo, double& hi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_down(lo, o, 64));
    hi = fmax(hi, __shfl_down(hi, o, 64));
  }
}

template <typename T>
__global__ __launch_bounds__(epa::kBlock) void minmax_partial_kernel(const T* __restrict__ x, size_t n,
                                                                     double* __restrict__ part) {
  __shared__ double slo[4], shi[4];
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const double v = (double)x[i];
    if (v == v) {
      lo = fmin(lo, v);
      hi = fmax(hi, v);
    }
  }
  wave_minmax<T>(lo, hi);
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
    part[2 * blockIdx.x + 1] = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
  }
}

__global__ __launch_bounds__(epa::kBlock) void minmax_final_kernel(const double* __restrict__ part,
                                                                   int nparts,
                                                                   double* __restrict__ out) {
  __shared__ double slo[4], shi[4];
  double lo = __builtin_inf(), hi = -__builtin_inf();
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    lo = fmin(lo, part[2 * i]);
    hi = fmax(hi, part[2 * i + 1]);
  }
  wave_minmax<double>(lo, hi);
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
    hi = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
    // no finite-or-inf value seen at all -> NaN (numpy nanmin/nanmax of an all-NaN array)
    const bool none = lo > hi;
    out[0] = none ? __builtin_nan("") : lo;
    out[1] = none ? __builtin_nan("") : hi;
  }
}

// depth = offset[c,p] + scale[c,p] * echo_range   (consolidate/api.py:226, add_depth)
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void affine_rows_kernel(const T* __restrict__ x,
                                                                  const double* __restrict__ scale,
                                                                  const double* __restrict__ offset,
                                                                  long long rows, int S,
                                                                  T* __restrict__ out) {
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T a = (T)scale[row], b = (T)offset[row];
    const T* xr = x + (size_t)row * S;
    T* orow = out + (size_t)row * S;
    for (int s = threadIdx.x; s < S; s += blockDim.x) orow[s] = b + a * xr[s];
  }
}

}  // namespace

extern "C" int epa_affine_rows(const void* x, const double* scale, const double* offset, int C, int P,
                               int S, void* out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(x && scale && offset && out, "epa_affine_rows: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_affine_rows: sizes must be positive");
  const long long rows = (long long)C * P;
  const int grid = (int)(rows < 16384 ? rows : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(affine_rows_kernel<double>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const double*)x, scale, offset, rows, S, (double*)out);
  else if (dtype == EPA_F32)
    hipLaunchKernelGGL(affine_rows_kernel<float>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const float*)x, scale, offset, rows, S, (float*)out);
  else {
    epa::set_error("epa_affine_rows: bad dtype %d", dtype);
    return EPA_EINVAL;
  }
  return epa::check_launch("affine_rows_kernel");
}

extern "C" int epa_nanminmax(const void* x, size_t n, int dtype, double* workspace, double* out,
                             epa_stream_t stream) {
  EPA_CHECK_ARG(x && workspace && out, "epa_nanminmax: NULL array argument");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_nanminmax: bad dtype %d", dtype);
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  const int grid = (int)(blocks < 1024 ? (blocks ? blocks : 1) : 1024);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(minmax_partial_kernel<double>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const double*)x, n, workspace);
  else
    hipLaunchKernelGGL(minmax_partial_kernel<float>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const float*)x, n, workspace);
  if (int rc = epa::check_launch("minmax_partial_kernel")) return rc;
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(epa::kBlock), 0, st, workspace, grid, out);
  return epa::check_launch("minmax_final_kernel");
}
