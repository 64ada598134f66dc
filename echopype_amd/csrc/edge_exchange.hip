// Cross-shard edge exchange: the only data that crosses a ping-shard edge are the raw linear
// (sum, count) rows of MVBS time bins / background-noise ping blocks that a shard or tile edge cuts
// (commongrid/utils.py:614-627 sums a bin over ALL its pings; clean/api.py:402-411 takes the block mean before the
// minimum).  One step of the exchange is
//     edge_pack_kernel   rows of the local partial-sum arrays -> slots of ONE communication buffer (slots, 2, C, R) f64
//     all-reduce(SUM)    torch.distributed (RCCL over xGMI; the buffer stays in HBM)
//     edge_gather_kernel per local shared edge: the sum over the slots of every holder of that bin, in slot order
//                        (the same order on every rank: all holders read bit-identical totals)
// Both kernels are pure data movement over a few hundred KB: launch-latency bound, no library GEMM on the path.
#include "epa_internal.h"

namespace {

constexpr int kMaxRows = 16;  // rows per launch (kernel-argument struct: 16 x 32 B)

struct PackRow {
  const void* sum;      // element (c, r) at sum[c * stride + r]
  const uint32_t* cnt;  // same indexing
  long long stride;     // elements between channels
  int slot;
  int pad;
};
struct PackArgs {
  PackRow row[kMaxRows];
  int C, R, sum_f32;
  double* buf;
};

__global__ __launch_bounds__(epa::kBlock) void edge_pack_kernel(PackArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int CR = a.C * a.R;
  if (e >= CR) return;
  const PackRow& row = a.row[blockIdx.y];
  const int c = e / a.R, j = e - c * a.R;
  const size_t off = (size_t)c * (size_t)row.stride + j;
  const double s = a.sum_f32 ? (double)static_cast<const float*>(row.sum)[off] : static_cast<const double*>(row.sum)[off];
  double* dst = a.buf + (size_t)row.slot * 2 * CR;
  dst[e] = s;
  dst[CR + e] = (double)row.cnt[off];
}

template <typename T>
__global__ __launch_bounds__(epa::kBlock) void edge_gather_kernel(const double* __restrict__ buf,
                                                                  const int32_t* __restrict__ goff,
                                                                  const int32_t* __restrict__ gslots, int CR,
                                                                  double* __restrict__ tot, T* __restrict__ sum_typed,
                                                                  uint32_t* __restrict__ cnt_u32) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;  // element of the (2, C, R) slot
  if (e >= 2 * CR) return;
  const int edge = blockIdx.y;
  double acc = 0.0;
  for (int k = goff[edge]; k < goff[edge + 1]; ++k) acc += buf[(size_t)gslots[k] * 2 * CR + e];
  if (tot) tot[(size_t)edge * 2 * CR + e] = acc;
  if (e < CR) {
    if (sum_typed) sum_typed[(size_t)edge * CR + e] = (T)acc;
  } else if (cnt_u32) {
    cnt_u32[(size_t)edge * CR + (e - CR)] = (uint32_t)(acc + 0.5);  // counts are exact integers in f64 (< 2^53)
  }
}

// owners of a cut MVBS time bin: totals -> 10 log10(sum / count) straight into the bin's row of the MVBS array
struct FinRow {
  void* dst;         // element (c, r) at dst[c * stride + r]
  long long stride;
  int edge;
  int pad;
};
struct FinArgs {
  FinRow row[kMaxRows];
  int C, R;
  double fill;
  const double* buf;
  const int32_t *goff, *gslots;
};

template <typename T>
__global__ __launch_bounds__(epa::kBlock) void edge_finalize_mvbs_kernel(FinArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int CR = a.C * a.R;
  if (e >= CR) return;
  const FinRow& row = a.row[blockIdx.y];
  double s = 0.0, n = 0.0;
  for (int k = a.goff[row.edge]; k < a.goff[row.edge + 1]; ++k) {
    const double* slot = a.buf + (size_t)a.gslots[k] * 2 * CR;
    s += slot[e];
    n += slot[CR + e];
  }
  const uint32_t cnt = (uint32_t)(n + 0.5);
  const int c = e / a.R, j = e - c * a.R;
  // the arithmetic of mvbs_finalize_kernel (block_reduce.hip) on the merged totals rounded to the MVBS dtype
  static_cast<T*>(row.dst)[(size_t)c * (size_t)row.stride + j] =
      cnt > 0u ? (T)10 * epa::M<T>::log10((T)s / (T)cnt) : (T)a.fill;
}

}  // namespace

extern "C" int epa_edge_pack(const void* const* sum_rows, const uint32_t* const* cnt_rows, const long long* chan_stride,
                             const int* slots, int n_rows, int sum_dtype, int C, int R, int n_slots, int zero_first,
                             double* buf, epa_stream_t stream) {
  EPA_CHECK_ARG(buf != nullptr, "epa_edge_pack: NULL buffer");
  EPA_CHECK_ARG(C > 0 && R > 0 && n_slots > 0 && n_rows >= 0, "epa_edge_pack: C=%d R=%d n_slots=%d n_rows=%d", C, R,
                n_slots, n_rows);
  EPA_CHECK_ARG(sum_dtype == EPA_F32 || sum_dtype == EPA_F64, "epa_edge_pack: bad dtype");
  EPA_CHECK_ARG(n_rows == 0 || (sum_rows && cnt_rows && chan_stride && slots), "epa_edge_pack: NULL row table");
  for (int i = 0; i < n_rows; ++i) {
    EPA_CHECK_ARG(sum_rows[i] && cnt_rows[i], "epa_edge_pack: row %d is NULL", i);
    EPA_CHECK_ARG(slots[i] >= 0 && slots[i] < n_slots, "epa_edge_pack: slot %d of row %d outside [0, %d)", slots[i], i,
                  n_slots);
    EPA_CHECK_ARG(chan_stride[i] >= R, "epa_edge_pack: channel stride of row %d shorter than a row", i);
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t CR = (size_t)C * R;
  if (zero_first) EPA_CHECK_HIP(hipMemsetAsync(buf, 0, (size_t)n_slots * 2 * CR * sizeof(double), st));
  for (int r0 = 0; r0 < n_rows; r0 += kMaxRows) {
    PackArgs a{};
    const int n = n_rows - r0 < kMaxRows ? n_rows - r0 : kMaxRows;
    for (int i = 0; i < n; ++i) a.row[i] = PackRow{sum_rows[r0 + i], cnt_rows[r0 + i], chan_stride[r0 + i], slots[r0 + i], 0};
    a.C = C; a.R = R; a.sum_f32 = sum_dtype == EPA_F32; a.buf = buf;
    hipLaunchKernelGGL(edge_pack_kernel, dim3((unsigned)((CR + epa::kBlock - 1) / epa::kBlock), (unsigned)n), dim3(epa::kBlock),
                       0, st, a);
    const int rc = epa::check_launch("edge_pack_kernel");
    if (rc != EPA_OK) return rc;
  }
  return EPA_OK;
}

extern "C" int epa_edge_gather(const double* buf, int n_slots, const int32_t* group_off, const int32_t* group_slots,
                               int n_edges, int C, int R, double* totals_out, void* sum_out, uint32_t* cnt_out,
                               int sum_dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(buf && group_off && group_slots, "epa_edge_gather: NULL array argument");
  EPA_CHECK_ARG(C > 0 && R > 0 && n_slots > 0 && n_edges >= 0, "epa_edge_gather: C=%d R=%d n_slots=%d n_edges=%d", C, R,
                n_slots, n_edges);
  EPA_CHECK_ARG(sum_dtype == EPA_F32 || sum_dtype == EPA_F64, "epa_edge_gather: bad dtype");
  EPA_CHECK_ARG(totals_out || sum_out || cnt_out, "epa_edge_gather: no output requested");
  if (n_edges == 0) return EPA_OK;
  const int CR = C * R;
  const dim3 grid((unsigned)((2 * (size_t)CR + epa::kBlock - 1) / epa::kBlock), (unsigned)n_edges);
  hipStream_t st = (hipStream_t)stream;
  if (sum_dtype == EPA_F32)
    hipLaunchKernelGGL(edge_gather_kernel<float>, grid, dim3(epa::kBlock), 0, st, buf, group_off, group_slots, CR, totals_out,
                       static_cast<float*>(sum_out), cnt_out);
  else
    hipLaunchKernelGGL(edge_gather_kernel<double>, grid, dim3(epa::kBlock), 0, st, buf, group_off, group_slots, CR,
                       totals_out, static_cast<double*>(sum_out), cnt_out);
  return epa::check_launch("edge_gather_kernel");
}

extern "C" int epa_edge_finalize_mvbs(const double* buf, int n_slots, const int32_t* group_off, const int32_t* group_slots,
                                      const int* edges, void* const* dst_rows, const long long* chan_stride, int n_rows,
                                      int C, int R, double fill_value, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(buf && group_off && group_slots, "epa_edge_finalize_mvbs: NULL array argument");
  EPA_CHECK_ARG(C > 0 && R > 0 && n_slots > 0 && n_rows >= 0, "epa_edge_finalize_mvbs: C=%d R=%d n_slots=%d n_rows=%d", C,
                R, n_slots, n_rows);
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_edge_finalize_mvbs: bad dtype");
  EPA_CHECK_ARG(n_rows == 0 || (edges && dst_rows && chan_stride), "epa_edge_finalize_mvbs: NULL row table");
  for (int i = 0; i < n_rows; ++i) {
    EPA_CHECK_ARG(dst_rows[i] != nullptr && edges[i] >= 0, "epa_edge_finalize_mvbs: bad row %d", i);
    EPA_CHECK_ARG(chan_stride[i] >= R, "epa_edge_finalize_mvbs: channel stride of row %d shorter than a row", i);
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t CR = (size_t)C * R;
  for (int r0 = 0; r0 < n_rows; r0 += kMaxRows) {
    FinArgs a{};
    const int n = n_rows - r0 < kMaxRows ? n_rows - r0 : kMaxRows;
    for (int i = 0; i < n; ++i) a.row[i] = FinRow{dst_rows[r0 + i], chan_stride[r0 + i], edges[r0 + i], 0};
    a.C = C; a.R = R; a.fill = fill_value; a.buf = buf; a.goff = group_off; a.gslots = group_slots;
    const dim3 grid((unsigned)((CR + epa::kBlock - 1) / epa::kBlock), (unsigned)n);
    if (dtype == EPA_F32) hipLaunchKernelGGL(edge_finalize_mvbs_kernel<float>, grid, dim3(epa::kBlock), 0, st, a);
    else hipLaunchKernelGGL(edge_finalize_mvbs_kernel<double>, grid, dim3(epa::kBlock), 0, st, a);
    const int rc = epa::check_launch("edge_finalize_mvbs_kernel");
    if (rc != EPA_OK) return rc;
  }
  return EPA_OK;
}

// The other cross-shard number that lives in HBM: nanmax(echo_range) of a shard, left by the fused kernel as NaN when
// the shard holds no valid range.  all-reduce(MAX) is not defined on NaN: the operand is prepared here (NaN -> -inf),
// the all-reduce runs in place on the RCCL group, ordered on the device behind the kernel -- no host wait.
namespace {
__global__ void nan_to_neg_inf_kernel(double* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !(p[i] == p[i])) p[i] = -__builtin_inf();
}
}  // namespace

extern "C" int epa_edge_prepare_max(double* values, int n, epa_stream_t stream) {
  EPA_CHECK_ARG(values != nullptr && n > 0, "epa_edge_prepare_max: empty argument");
  hipLaunchKernelGGL(nan_to_neg_inf_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, values, n);
  return epa::check_launch("nan_to_neg_inf_kernel");
}
