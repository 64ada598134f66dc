// compute_NASC array pass (SURVEY 8f "next" row 3).
//
// Replaces /root/reference/echopype/commongrid/utils.py:97-205 (compute_raw_NASC):
//   sv_mean[c,d,r] = (nan)mean of 10^(Sv/10) over the pings of distance bin d and samples whose depth
//                    lies in range bin r                      (_groupby_x_along_channels, :150-160)
//   h_num[c,d,r]   = nansum of diff(depth, range_sample) labelled by its LOWER sample (:191-200)
//   h_mean         = h_num / (pings in distance bin d)                          (:177-201)
//   NASC           = sv_mean * h_mean * 4 * pi * 1852^2                         (:204)
// One pass over Sv and depth (16 B/sample in f64).  Stage 1: a workgroup owns a slice of the pings
// of one (channel, distance bin), accumulates (sv sum, sv count, NaN count, height sum) per range
// bin in LDS (runs of equal bins are merged in registers first) and adds its partials to a global
// workspace; stage 2 turns the workspace into NASC.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "fast_math.h"

namespace {

using epa::kBlock;

struct Cell {  // workspace layout per (c, d, r)
  double ssum, hsum;
  unsigned cnt, nnan;
};
static_assert(sizeof(Cell) == 24, "workspace cell");

// USE_LDS = false: a depth grid too fine for the LDS accumulators -- the lane's runs go straight to the global
// workspace cells with atomics (same sums, more atomic traffic), as epa_mvbs does for its large grids.
template <typename T, bool USE_LDS>
__global__ __launch_bounds__(kBlock) void nasc_accumulate_kernel(
    const T* __restrict__ sv, const T* __restrict__ depth, int P, int S,
    const int32_t* __restrict__ bin_start, int n_dbins, int nparts, double range_bin, double inv_bin,
    int n_rbins, int closed_right, Cell* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const epa::MathTabs mt = epa::build_math_tabs(smem);
  double* lss = reinterpret_cast<double*>(smem + epa::kMathTabBytes);
  double* lhs = lss + n_rbins;
  unsigned* lcn = reinterpret_cast<unsigned*>(lhs + n_rbins);
  unsigned* lnn = lcn + n_rbins;
  const int part = blockIdx.x, d = blockIdx.y, c = blockIdx.z;
  const int b0 = bin_start[d], b1 = bin_start[d + 1];
  const int per = (b1 - b0 + nparts - 1) / nparts;
  const int p0 = b0 + part * per, p1 = min(b1, p0 + per);
  Cell* cell = ws + ((size_t)c * n_dbins + d) * n_rbins;
  auto flush = [&](int rb, double rs, unsigned rn, unsigned rnan, double rh) {
    if (USE_LDS) {
      if (rn) { unsafeAtomicAdd(&lss[rb], rs); atomicAdd(&lcn[rb], rn); }
      if (rnan) atomicAdd(&lnn[rb], rnan);
      if (rh != 0.0) unsafeAtomicAdd(&lhs[rb], rh);
    } else {
      if (rn) { unsafeAtomicAdd(&cell[rb].ssum, rs); atomicAdd(&cell[rb].cnt, rn); }
      if (rnan) atomicAdd(&cell[rb].nnan, rnan);
      if (rh != 0.0) unsafeAtomicAdd(&cell[rb].hsum, rh);
    }
  };
  if (p0 >= p1) return;
  if (USE_LDS) {
    for (int i = threadIdx.x; i < n_rbins; i += kBlock) {
      lss[i] = 0.0;
      lhs[i] = 0.0;
      lcn[i] = 0u;
      lnn[i] = 0u;
    }
  }
  __syncthreads();  // (also publishes the exp table)
  for (int p = p0; p < p1; ++p) {
    const size_t row = ((size_t)c * P + p) * S;
    const T* svr = sv + row;
    const T* dr = depth + row;
    for (int base = 4 * threadIdx.x; base < S; base += 4 * kBlock) {
      int rb = -1;
      double rs = 0.0, rh = 0.0;
      unsigned rn = 0u, rnan = 0u;
      bool any = false;
      // the lane's four samples and the depth after them in 16-byte accesses where the row allows
      T dv[5], vv[4];
      if (base + 4 <= S) {
        typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
        const pair_t d01 = *reinterpret_cast<const pair_t*>(dr + base), d23 = *reinterpret_cast<const pair_t*>(dr + base + 2);
        const pair_t v01 = *reinterpret_cast<const pair_t*>(svr + base), v23 = *reinterpret_cast<const pair_t*>(svr + base + 2);
        dv[0] = d01.x; dv[1] = d01.y; dv[2] = d23.x; dv[3] = d23.y;
        vv[0] = v01.x; vv[1] = v01.y; vv[2] = v23.x; vv[3] = v23.y;
        dv[4] = (base + 4 < S) ? dr[base + 4] : epa::M<T>::nan();
      } else {
#pragma unroll
        for (int j = 0; j < 5; ++j) dv[j] = (base + j < S) ? dr[base + j] : epa::M<T>::nan();
#pragma unroll
        for (int j = 0; j < 4; ++j) vv[j] = (base + j < S) ? svr[base + j] : epa::M<T>::nan();
      }
      T dcur = dv[0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int s = base + j;
        if (s >= S) break;
        const T dnext = dv[j + 1];
        const int b = epa::range_bin_index((double)dcur, range_bin, inv_bin, n_rbins, closed_right != 0);
        if (b >= 0) {
          if (b != rb) {
            if (any) flush(rb, rs, rn, rnan, rh);
            rb = b; rs = 0.0; rh = 0.0; rn = 0u; rnan = 0u; any = true;
          }
          const T v = vv[j];
          if (v == v) {
            rs += (double)epa::lin_from_db(v, mt.exp2_tab);
            ++rn;
          } else {
            ++rnan;
          }
          const T h = dnext - dcur;  // in the storage type, as xarray's diff
          if (h == h) rh += (double)h;
        }
        dcur = dnext;
      }
      if (any) flush(rb, rs, rn, rnan, rh);
    }
  }
  if (!USE_LDS) return;
  __syncthreads();
  for (int i = threadIdx.x; i < n_rbins; i += kBlock) {
    if (lcn[i]) {
      unsafeAtomicAdd(&cell[i].ssum, lss[i]);
      atomicAdd(&cell[i].cnt, lcn[i]);
    }
    if (lnn[i]) atomicAdd(&cell[i].nnan, lnn[i]);
    if (lhs[i] != 0.0) unsafeAtomicAdd(&cell[i].hsum, lhs[i]);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void nasc_finalize_kernel(const Cell* __restrict__ ws,
                                                               const int32_t* __restrict__ bin_start,
                                                               int n_dbins, int n_rbins, size_t n,
                                                               int skipna, T* __restrict__ nasc,
                                                               T* __restrict__ sv_mean,
                                                               T* __restrict__ h_mean) {
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
    const Cell w = ws[i];
    const int d = (int)((i / n_rbins) % n_dbins);
    const double npings = (double)(bin_start[d + 1] - bin_start[d]);
    double m = __builtin_nan("");
    if (w.cnt + w.nnan > 0u && (skipna || w.nnan == 0u) && w.cnt > 0u) m = w.ssum / (double)w.cnt;
    const double h = w.hsum / npings;  // 0 / 0 -> NaN for a distance bin without pings
    if (sv_mean) sv_mean[i] = (T)m;
    if (h_mean) h_mean[i] = (T)h;
    nasc[i] = (T)((((m * h) * 4.0) * 3.141592653589793) * 3429904.0);  // reference evaluation order
  }
}

// Along-track step of every ping: geodesic distance on WGS-84 to the NEXT ping (commongrid/utils.py:208-231 takes it from
// geopy.distance.distance, ping by ping on the host: half a second per 100 000 pings there, 30 ms with the vectorised
// NumPy Vincenty of the drop-in -- the whole of compute_NASC's time on a resident dataset whose kernel takes 3 ms).
// Vincenty's inverse formula, one lane per pair, iterated until |d lambda| < 1e-14 (at most 200 times); NaN positions
// give NaN (the host drops such pairs as the reference does), coincident points 0.
__global__ __launch_bounds__(kBlock) void geodesic_step_kernel(const double* __restrict__ lat, const double* __restrict__ lon,
                                                               int P, double* __restrict__ step_m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  if (i == P - 1) {
    step_m[i] = __builtin_nan("");
    return;
  }
  constexpr double a = 6378137.0, f = 1.0 / 298.257223563, kRad = 0.017453292519943295;
  constexpr double b = (1.0 - f) * a;
  const double la1 = lat[i], lo1 = lon[i], la2 = lat[i + 1], lo2 = lon[i + 1];
  const double U1 = atan((1.0 - f) * tan(la1 * kRad)), U2 = atan((1.0 - f) * tan(la2 * kRad));
  const double L = (lo2 - lo1) * kRad;
  const double sU1 = sin(U1), cU1 = cos(U1), sU2 = sin(U2), cU2 = cos(U2);
  double lam = L, sin_sig = 0.0, cos_sig = 1.0, sig = 0.0, cos2_al = 1.0, cos_2sm = 0.0;
  for (int it = 0; it < 200; ++it) {
    const double sl = sin(lam), cl = cos(lam);
    sin_sig = hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl);
    cos_sig = sU1 * sU2 + cU1 * cU2 * cl;
    sig = atan2(sin_sig, cos_sig);
    const double sin_al = sin_sig == 0.0 ? 0.0 : cU1 * cU2 * sl / sin_sig;
    cos2_al = 1.0 - sin_al * sin_al;
    cos_2sm = cos2_al == 0.0 ? 0.0 : cos_sig - 2.0 * sU1 * sU2 / cos2_al;
    const double Cc = f / 16.0 * cos2_al * (4.0 + f * (4.0 - 3.0 * cos2_al));
    const double lam_new = L + (1.0 - Cc) * f * sin_al *
                                   (sig + Cc * sin_sig * (cos_2sm + Cc * cos_sig * (-1.0 + 2.0 * cos_2sm * cos_2sm)));
    const bool done = !(fabs(lam_new - lam) >= 1e-14);  // (NaN: stop)
    lam = lam_new;
    if (done) break;
  }
  const double u2 = cos2_al * (a * a - b * b) / (b * b);
  const double A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)));
  const double B = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)));
  const double dsig = B * sin_sig * (cos_2sm + B / 4.0 * (cos_sig * (-1.0 + 2.0 * cos_2sm * cos_2sm) -
                                                           B / 6.0 * cos_2sm * (-3.0 + 4.0 * sin_sig * sin_sig) *
                                                               (-3.0 + 4.0 * cos_2sm * cos_2sm)));
  const double sm = b * A * (sig - dsig);
  step_m[i] = sin_sig == 0.0 ? 0.0 : sm;
}

}  // namespace

extern "C" int epa_geodesic_steps(const double* lat, const double* lon, int P, double* step_m_out, epa_stream_t stream) {
  EPA_CHECK_ARG(lat && lon && step_m_out, "epa_geodesic_steps: NULL array argument");
  EPA_CHECK_ARG(P > 0, "epa_geodesic_steps: P must be positive");
  hipLaunchKernelGGL(geodesic_step_kernel, dim3((unsigned)((P + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, lat, lon, P, step_m_out);
  return epa::check_launch("geodesic_step_kernel");
}

extern "C" int epa_nasc(const void* sv, const void* depth, int C, int P, int S, const int32_t* bin_start,
                        int n_dbins, double range_bin, int n_rbins, unsigned bin_flags, void* workspace,
                        void* nasc_out, void* sv_mean_out, void* h_mean_out, int dtype,
                        epa_stream_t stream) {
  EPA_CHECK_ARG(sv && depth && bin_start && workspace && nasc_out, "epa_nasc: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && n_dbins > 0 && n_rbins > 0, "epa_nasc: sizes must be positive");
  EPA_CHECK_ARG(range_bin > 0, "epa_nasc: range_bin must be positive");
  EPA_CHECK_ARG(dtype == EPA_F32 || dtype == EPA_F64, "epa_nasc: bad dtype %d", dtype);
  EPA_CHECK_ARG(C <= 65535 && n_dbins <= 65535, "epa_nasc: more than 65535 channels / distance bins");
  const bool use_lds = epa::kMathTabBytes + (size_t)n_rbins * 24 <= 156 * 1024;
  const size_t lds = epa::kMathTabBytes + (use_lds ? (size_t)n_rbins * 24 : 0);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)C * n_dbins * n_rbins;
  EPA_CHECK_HIP(hipMemsetAsync(workspace, 0, n * sizeof(Cell), st));
  // enough workgroups for 256 CUs whatever the number of distance bins
  const long long cells = (long long)C * n_dbins;
  long long nparts = (4096 + cells - 1) / cells;
  const long long avg_pings = ((long long)P + n_dbins - 1) / n_dbins;
  if (nparts > avg_pings) nparts = avg_pings;
  if (nparts < 1) nparts = 1;
  const dim3 grid((unsigned)nparts, n_dbins, C);
  const int cr = (bin_flags & EPA_BIN_CLOSED_RIGHT) ? 1 : 0, skipna = (bin_flags & EPA_BIN_SKIPNA) ? 1 : 0;
#define EPA_NASC(T)                                                                                   \
  do {                                                                                                \
    auto kern = use_lds ? nasc_accumulate_kernel<T, true> : nasc_accumulate_kernel<T, false>;         \
    if (lds > 64 * 1024)                                                                              \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, st, (const T*)sv, (const T*)depth, P, S,         \
                       bin_start, n_dbins, (int)nparts, range_bin, 1.0 / range_bin, n_rbins, cr,      \
                       (Cell*)workspace);                                                             \
    if (int rc = epa::check_launch("nasc_accumulate_kernel")) return rc;                              \
    const size_t blocks = (n + kBlock - 1) / kBlock;                                                  \
    hipLaunchKernelGGL(nasc_finalize_kernel<T>, dim3((unsigned)(blocks < 4096 ? blocks : 4096)),      \
                       dim3(kBlock), 0, st, (const Cell*)workspace, bin_start, n_dbins, n_rbins, n,   \
                       skipna, (T*)nasc_out, (T*)sv_mean_out, (T*)h_mean_out);                        \
  } while (0)
  if (dtype == EPA_F64) EPA_NASC(double); else EPA_NASC(float);
#undef EPA_NASC
  return epa::check_launch("nasc_finalize_kernel");
}
