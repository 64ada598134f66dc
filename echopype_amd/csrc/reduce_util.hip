// NaN-skipping min/max of a device array: the reductions the reference forces with
//   ds_Sv[range_var].max(skipna=True)            commongrid/api.py:108-110 (range bin edges)
//   round(float(da.min())), round(float(da.max()))  clean/utils.py:392-395, commongrid/api.py:252-255
// Two launches: per-workgroup partials (wave __shfl reduction, then LDS), then one workgroup.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "sample_math.h"

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
  __shared__ double slo[4], shi[4], snan[4];
  double lo = __builtin_inf(), hi = -__builtin_inf(), nn = 0.0;
  auto take = [&](double v) {
    lo = fmin(lo, v);  // fmin / fmax ignore a NaN operand
    hi = fmax(hi, v);
    nn += v == v ? 0.0 : 1.0;
  };
  // four elements per lane and trip in 16-byte loads (an aligned base: torch allocations are), then the tail
  constexpr int kPer = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(kPer)));
  const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
  const size_t nvec = aligned ? n / (2 * kPer) : 0;
  const vec_t* xv = reinterpret_cast<const vec_t*>(x);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const vec_t a = xv[2 * i], b = xv[2 * i + 1];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      take((double)a[j]);
      take((double)b[j]);
    }
  }
  for (size_t i = nvec * 2 * kPer + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    take((double)x[i]);
  wave_minmax<T>(lo, hi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o, 64);
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
    snan[threadIdx.x >> 6] = nn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[3 * blockIdx.x] = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
    part[3 * blockIdx.x + 1] = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
    part[3 * blockIdx.x + 2] = (snan[0] + snan[1]) + (snan[2] + snan[3]);
  }
}

__global__ __launch_bounds__(epa::kBlock) void minmax_final_kernel(const double* __restrict__ part,
                                                                   int nparts,
                                                                   double* __restrict__ out, int stride = 3) {
  __shared__ double slo[4], shi[4], snan[4];
  double lo = __builtin_inf(), hi = -__builtin_inf(), nn = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    lo = fmin(lo, part[(size_t)stride * i]);
    hi = fmax(hi, part[(size_t)stride * i + 1]);
    nn += part[(size_t)stride * i + 2];
  }
  wave_minmax<double>(lo, hi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o, 64);
  if ((threadIdx.x & 63) == 0) {
    slo[threadIdx.x >> 6] = lo;
    shi[threadIdx.x >> 6] = hi;
    snan[threadIdx.x >> 6] = nn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
    hi = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
    // no finite-or-inf value seen at all -> NaN (numpy nanmin/nanmax of an all-NaN array)
    const bool none = lo > hi;
    out[0] = none ? __builtin_nan("") : lo;
    out[1] = none ? __builtin_nan("") : hi;
    out[2] = (snan[0] + snan[1]) + (snan[2] + snan[3]);  // number of NaN elements
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
    for (int s = threadIdx.x; s < S; s += blockDim.x) orow[s] = epa::depth_of(a, b, xr[s]);
  }
}


// mean range step per channel: np.nanmean(np.diff(range, axis=2), axis=(1, 2))  (clean/utils.py:131).
// Deterministic two-stage sum: one (sum, count) per (c, p) row, then one workgroup per channel.
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void step_rows_kernel(const T* __restrict__ x, long long rows,
                                                                int S, double* __restrict__ part) {
  __shared__ double ssum[4], scnt[4];
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T* xr = x + (size_t)row * S;
    double sum = 0.0, cnt = 0.0;
    for (int s = threadIdx.x; s + 1 < S; s += blockDim.x) {
      const T d = xr[s + 1] - xr[s];  // in the storage type, as np.diff
      if (d == d) {
        sum += (double)d;
        cnt += 1.0;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sum += __shfl_down(sum, o, 64);
      cnt += __shfl_down(cnt, o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      ssum[threadIdx.x >> 6] = sum;
      scnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      part[2 * row] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
      part[2 * row + 1] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
    }
  }
}

__global__ __launch_bounds__(epa::kBlock) void step_final_kernel(const double* __restrict__ part, int P,
                                                                 double* __restrict__ out) {
  __shared__ double ssum[4], scnt[4];
  const double* pc = part + (size_t)blockIdx.x * P * 2;
  double sum = 0.0, cnt = 0.0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    sum += pc[2 * p];
    cnt += pc[2 * p + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sum += __shfl_down(sum, o, 64);
    cnt += __shfl_down(cnt, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    ssum[threadIdx.x >> 6] = sum;
    scnt[threadIdx.x >> 6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
    const double n = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
    out[blockIdx.x] = n > 0.0 ? s / n : __builtin_nan("");
  }
}

// flat index of the first element that is NOT <= limit (NaN counts), n if there is none:
// np.argmin(x <= limit) over the flattened array (clean/utils.py:143), except that an array with no
// such element reports n where np.argmin reports 0.
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void first_not_le_kernel(const T* __restrict__ x, size_t n,
                                                                   T limit,
                                                                   unsigned long long* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    if (i >= *reinterpret_cast<volatile unsigned long long*>(out)) return;  // already beaten
    if (!(x[i] <= limit)) {
      atomicMin(out, (unsigned long long)i);
      return;
    }
  }
}

__global__ void set_u64_kernel(unsigned long long* p, unsigned long long v) { *p = v; }

// depth = offset[c,p] + scale[c,p] * echo_range with echo_range read from the array or -- x == NULL -- evaluated from
// the power-sample coefficient rows (NaN where the raw sample is, when mask_raw is given), and {min, max, NaN count}
// of the depth written as a by-product: one partial per workgroup in `part` (deterministic, no atomics)
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void depth_rows_kernel(const T* __restrict__ x,
                                                                 const epa::CoefRow* __restrict__ coef,
                                                                 const float* __restrict__ mask_raw,
                                                                 const double* __restrict__ scale,
                                                                 const double* __restrict__ offset, long long rows,
                                                                 int S, T* __restrict__ out,
                                                                 double* __restrict__ part) {
  __shared__ double slo[4], shi[4], snan[4];
  double lo = __builtin_inf(), hi = -__builtin_inf(), nn = 0.0;
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const T a = (T)scale[row], b = (T)offset[row];
    const size_t base = (size_t)row * S;
    T* orow = out + base;
    epa::CoefRow cr{};
    if (!x) cr = coef[row];
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      T r;
      if (x) {
        r = x[base + s];
      } else {
        r = (T)epa::row_range(cr, s);
        if (mask_raw && !(mask_raw[base + s] == mask_raw[base + s])) r = epa::M<T>::nan();
      }
      const T d = epa::depth_of(a, b, r);
      orow[s] = d;
      if (part) {
        const double dd = (double)d;
        lo = fmin(lo, dd);  // fmin / fmax ignore a NaN operand
        hi = fmax(hi, dd);
        nn += dd == dd ? 0.0 : 1.0;
      }
    }
  }
  if (part) {
    wave_minmax<double>(lo, hi);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nn += __shfl_down(nn, o, 64);
    if ((threadIdx.x & 63) == 0) {
      slo[threadIdx.x >> 6] = lo;
      shi[threadIdx.x >> 6] = hi;
      snan[threadIdx.x >> 6] = nn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double* pp = part + 3 * (size_t)blockIdx.x;
      pp[0] = fmin(fmin(slo[0], slo[1]), fmin(slo[2], slo[3]));
      pp[1] = fmax(fmax(shi[0], shi[1]), fmax(shi[2], shi[3]));
      pp[2] = (snan[0] + snan[1]) + (snan[2] + snan[3]);
    }
  }
}
}  // namespace

// ---- one-piece workgroups for the affine row passes (round 6) ---------------------------------------------------------
// echo_range (epa_range_power) and depth = offset + scale * echo_range (epa_depth_rows without the statistics) written
// from the coefficient rows: a workgroup = ONE 1024-sample piece of one row, 16-byte accesses, then it ends -- the
// loop-free form that streams at 6.2 TB/s where workgroups striding over the rows reach 5.0-5.4 (DESIGN 4.2, and K1 in
// sv_power.hip).  NaN where the raw sample is, when the mask is given.  DEPTH: 0 the range itself, 1 the depth.
namespace {
constexpr int kPieceSlots = 256, kPieceSlotStride = 16;  // {min, max, NaN count} slots of the statistics, 128 bytes apart
__global__ void piece_slots_init_kernel(double* slots) {
  double* p = slots + (size_t)threadIdx.x * kPieceSlotStride;
  p[0] = __builtin_inf();
  p[1] = -__builtin_inf();
  p[2] = 0.0;
}

// SRC: 0 the range evaluated from the coefficient rows (NaN where the raw sample is, with MASK), 1 read from the array x.
// STATS: {min, max, NaN count} of what is written -- the workgroup's lanes meet in LDS (ds_min / ds_max on 32 slots, a
// NaN operand leaves a slot alone), one lane sends three atomics WITHOUT a return value to one of kPieceSlots lines
// (nobody waits for them: fused_sv_mvbs.hip, round 6), epa_minmax_final folds the slots.
template <typename T, int DEPTH, bool MASK, int SRC, bool STATS>
__global__ __launch_bounds__(epa::kBlock) void rows_piece_kernel(const float* __restrict__ raw, const T* __restrict__ x,
                                                                 const epa::CoefRow* __restrict__ coef,
                                                                 const double* __restrict__ scale,
                                                                 const double* __restrict__ offset, int S,
                                                                 int chunks_per_row, long long pieces, T* __restrict__ out,
                                                                 int xcd_map, double* __restrict__ slots) {
  using LM = epa::LaneMap<T>;
  __shared__ T wlo[STATS ? 32 : 1], whi[STATS ? 32 : 1];
  __shared__ unsigned wnn;
  if (STATS) {
    if (threadIdx.x < 32) {
      wlo[threadIdx.x] = (T)__builtin_inf();
      whi[threadIdx.x] = -(T)__builtin_inf();
    }
    if (threadIdx.x == 0) wnn = 0u;
    __syncthreads();
  }
  const long long piece = xcd_map ? epa::xcd_contiguous((int)blockIdx.x, (int)pieces) : (long long)blockIdx.x;
  const long long row = piece / chunks_per_row;
  const int chunk0 = (int)(piece - row * chunks_per_row) * 1024;
  epa::CoefRow cr{};
  if (SRC == 0) cr = coef[row];
  T a = (T)1, b = (T)0;
  if (DEPTH) {
    a = (T)scale[row];
    b = (T)offset[row];
  }
  T lo = (T)__builtin_inf(), hi = -(T)__builtin_inf();
  unsigned nn = 0u;  // (the wavefront's: a scalar)
#pragma unroll
  for (int g = 0; g < LM::NSEG; ++g) {
    const int s0 = LM::first(chunk0, g);
    if (s0 >= S) continue;
    const size_t off = (size_t)row * S + s0;
    T o[LM::LEN], xin[LM::LEN];
    epa::RawVec<LM::LEN> in;
    if (SRC == 0 && MASK) in.load(raw + off);
    if (SRC == 1) epa::load_vec<T, LM::LEN>(x + off, xin);
#pragma unroll
    for (int j = 0; j < LM::LEN; ++j) {
      T r = SRC == 1 ? xin[j] : (T)epa::row_range(cr, s0 + j);
      if (SRC == 0 && MASK && !(in.v[j] == in.v[j])) r = epa::M<T>::nan();
      o[j] = DEPTH ? epa::depth_of(a, b, r) : r;
      if (STATS) {
        lo = fmin(lo, o[j]);  // fmin / fmax ignore a NaN operand
        hi = fmax(hi, o[j]);
        nn += (unsigned)__builtin_popcountll(__ballot(!(o[j] == o[j])));
      }
    }
    epa::store_vec<T, LM::LEN>(out + off, o);
  }
  if (STATS) {
    __hip_atomic_fetch_min(&wlo[threadIdx.x & 31], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_max(&whi[threadIdx.x & 31], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((threadIdx.x & 63) == 0 && nn > 0u) atomicAdd(&wnn, nn);
    __syncthreads();
    if (threadIdx.x < 64) {
      double l2 = (double)wlo[threadIdx.x & 31], h2 = (double)whi[threadIdx.x & 31];
      wave_minmax<double>(l2, h2);
      if (threadIdx.x == 0) {
        double* sl = slots + (size_t)(blockIdx.x & (kPieceSlots - 1)) * kPieceSlotStride;
        if (l2 <= h2) {
          unsafeAtomicMin(sl + 0, l2);
          unsafeAtomicMax(sl + 1, h2);
        }
        if (wnn > 0u) unsafeAtomicAdd(sl + 2, (double)wnn);
      }
    }
  }
}
}  // namespace

// (sv_power.hip's epa_range_power and epa_depth_rows below) -> EPA_OK when the pieces served the call, -1 when the
// shape does not take the vector form (S not a multiple of the 16-byte access, unaligned buffers, too many pieces).
// x: the range as an array instead of the coefficient rows; workspace + stats_out: {min, max, NaN count} of the output.
int epa_rows_piece_launch(const float* mask_raw, const void* x, const double* coef, const double* scale,
                          const double* offset, long long rows, int S, void* out, int dtype, double* workspace,
                          double* stats_out, hipStream_t st) {
  const int need = dtype == EPA_F64 ? 2 : 4;
  auto al16 = [](const void* q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
  static const bool off = [] { const char* e = getenv("EPA_ROW_PIECES"); return e && e[0] == '0'; }();  // development knob
  const int chunks = (S + 1023) / 1024;
  const long long pieces = rows * chunks;
  if (off || S % need != 0 || !al16(mask_raw) || !al16(x) || !al16(out) || pieces >= (1ll << 31)) return -1;
  const epa::CoefRow* cf = reinterpret_cast<const epa::CoefRow*>(coef);
  const int xm = epa::xcd_map_enabled() ? 1 : 0;
  const bool stats = stats_out != nullptr;
  if (stats) {
    hipLaunchKernelGGL(piece_slots_init_kernel, dim3(1), dim3(kPieceSlots), 0, st, workspace);
    if (int rc = epa::check_launch("piece_slots_init_kernel")) return rc;
  }
#define EPA_RP(T, D, M, SRC, ST)                                                                                   \
  hipLaunchKernelGGL((rows_piece_kernel<T, D, M, SRC, ST>), dim3((unsigned)pieces), dim3(epa::kBlock), 0, st,      \
                     mask_raw, (const T*)x, cf, scale, offset, S, chunks, pieces, (T*)out, xm, workspace)
#define EPA_RP4(T, D)                                                                                              \
  do {                                                                                                             \
    if (x) { if (stats) EPA_RP(T, D, false, 1, true); else EPA_RP(T, D, false, 1, false); }                        \
    else if (mask_raw) { if (stats) EPA_RP(T, D, true, 0, true); else EPA_RP(T, D, true, 0, false); }              \
    else { if (stats) EPA_RP(T, D, false, 0, true); else EPA_RP(T, D, false, 0, false); }                          \
  } while (0)
  if (dtype == EPA_F64) { if (scale) EPA_RP4(double, 1); else EPA_RP4(double, 0); }
  else { if (scale) EPA_RP4(float, 1); else EPA_RP4(float, 0); }
#undef EPA_RP4
#undef EPA_RP
  if (int rc = epa::check_launch("rows_piece_kernel")) return rc;
  if (stats) {
    hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(epa::kBlock), 0, st, workspace, kPieceSlots, stats_out,
                       kPieceSlotStride);
    return epa::check_launch("minmax_final_kernel");
  }
  return EPA_OK;
}

extern "C" int epa_depth_rows(const void* range, const double* coef, const float* mask_raw, const double* scale,
                              const double* offset, int C, int P, int S, void* out, int dtype, double* workspace,
                              double* stats_out, epa_stream_t stream) {
  EPA_CHECK_ARG((range || coef) && scale && offset && out, "epa_depth_rows: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_depth_rows: sizes must be positive");
  EPA_CHECK_ARG(!stats_out || workspace, "epa_depth_rows: the statistics need the workspace");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_depth_rows: bad dtype %d", dtype);
  const long long rows = (long long)C * P;
  const int grid = (int)(rows < 16384 ? rows : 16384);
  hipStream_t st = (hipStream_t)stream;
  const epa::CoefRow* cf = reinterpret_cast<const epa::CoefRow*>(coef);
  double* part = stats_out ? workspace : nullptr;
  {  // one-piece workgroups where the shape takes 16-byte accesses
    const int rc = epa_rows_piece_launch(range ? nullptr : mask_raw, range, coef, scale, offset, rows, S, out, dtype, workspace,
                                         stats_out, st);
    if (rc >= 0) return rc;
  }
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(depth_rows_kernel<double>, dim3(grid), dim3(epa::kBlock), 0, st, (const double*)range, cf,
                       mask_raw, scale, offset, rows, S, (double*)out, part);
  else
    hipLaunchKernelGGL(depth_rows_kernel<float>, dim3(grid), dim3(epa::kBlock), 0, st, (const float*)range, cf,
                       mask_raw, scale, offset, rows, S, (float*)out, part);
  if (int rc = epa::check_launch("depth_rows_kernel")) return rc;
  if (stats_out) return epa_minmax_final(workspace, grid, stats_out, st);
  return EPA_OK;
}

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

// {min, max, NaN count} partials of another kernel (3 doubles per workgroup) -> out[3]; used by sv_power.hip
int epa_minmax_final(const double* part, int nparts, double* out, hipStream_t st) {
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(epa::kBlock), 0, st, part, nparts, out);
  return epa::check_launch("minmax_final_kernel");
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

extern "C" int epa_range_step_mean(const void* range, int C, int P, int S, int dtype, double* workspace,
                                   double* out, epa_stream_t stream) {
  EPA_CHECK_ARG(range && workspace && out, "epa_range_step_mean: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_range_step_mean: sizes must be positive");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_range_step_mean: bad dtype %d", dtype);
  const long long rows = (long long)C * P;
  const int grid = (int)(rows < 16384 ? rows : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(step_rows_kernel<double>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const double*)range, rows, S, workspace);
  else
    hipLaunchKernelGGL(step_rows_kernel<float>, dim3(grid), dim3(epa::kBlock), 0, st, (const float*)range,
                       rows, S, workspace);
  if (int rc = epa::check_launch("step_rows_kernel")) return rc;
  hipLaunchKernelGGL(step_final_kernel, dim3(C), dim3(epa::kBlock), 0, st, workspace, P, out);
  return epa::check_launch("step_final_kernel");
}

extern "C" int epa_first_not_le(const void* x, size_t n, double limit, int dtype, uint64_t* out,
                                epa_stream_t stream) {
  EPA_CHECK_ARG(x && out, "epa_first_not_le: NULL array argument");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_first_not_le: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, st, (unsigned long long*)out,
                     (unsigned long long)n);
  if (n == 0) return epa::check_launch("set_u64_kernel");
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(first_not_le_kernel<double>, dim3(grid), dim3(epa::kBlock), 0, st, (const double*)x,
                       n, limit, (unsigned long long*)out);
  else
    hipLaunchKernelGGL(first_not_le_kernel<float>, dim3(grid), dim3(epa::kBlock), 0, st, (const float*)x, n,
                       (float)limit, (unsigned long long*)out);
  return epa::check_launch("first_not_le_kernel");
}
