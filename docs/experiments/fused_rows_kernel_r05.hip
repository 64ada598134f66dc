// EXPERIMENT, NOT COMPILED INTO THE LIBRARY (round 5) -- the row-major form of fused_sv_mvbs_kernel.
//
// Why it was written: on two boxes of the pool the headline traffic mix (4 B read + 8 B written per sample) streamed at
// 6.0-6.3 TB/s when a workgroup takes its 20 rows WHOLE, one after the other (one contiguous run of 20 x S samples,
// XCD-contiguous order of the bins), against 5.1-5.4 TB/s for every walk that strides through the rows a 1024-sample
// piece at a time (scripts/probes/hbm_walk{2,3,3b}_probe.hip).  The side-by-side probe (hbm_walk5_probe.hip, three
// walks on ONE box, run on two more boxes) then showed that figure to be a property of those boxes: elsewhere the
// row-major walk is 3 % ahead of the shipped one (5.12-5.22 against 4.95-5.07 TB/s), and only the loop-free walk
// (one 1024-sample piece per workgroup, 6.2 TB/s everywhere) is really faster -- and that one cannot own a time bin
// (its bins through global atomics: 4.3-5.5 TB/s).  All of it in profiles/r05_walk_probes.txt.
// Why it is not shipped: a lane of the row-major form meets 16 columns per ping and cannot carry lane-private sums
// for them, so every pair of samples goes to the LDS accumulators with atomics and the bin index is computed per sample.
// Measured (profiles/r05_fused_rows_experiment.txt, 4 x 250 000 x 4096, fp64, same box, interleaved):
//     this kernel   12.0-12.4 ms with the Sv store, 10.2 ms without      (fp32: 14.2 ms -- the LDS float add)
//     shipped        9.2- 9.4 ms                      5.2 ms
// On 4 x 500 000 x 2000 (two chunks, cached logs) the two are within 3 % although this one does 7.0 ms of arithmetic +
// LDS traffic against 5.35.  It passed the GPU suite (257 tests up to the first one that asserts the kernel's NAME).
// (This file is the FIRST form, four chunks unrolled flat -- ~200 registers; the measured one took a ping in two groups
//  of two chunks with the logarithm evaluated per sample: git show 5427237:echopype_amd/csrc/fused_sv_mvbs.hip.)
// Drop-in: paste above `launch()` in csrc/fused_sv_mvbs.hip and dispatch from epa_fused_fast_path for S <= 4096.

// ---- the ROW-MAJOR form of the same kernel (round 5) -----------------------------------------------------------------
// fused_sv_mvbs_kernel above walks its time bin chunk by chunk (1024 columns x 20 pings, then the next 1024 columns):
// a lane carries the state of four columns down the pings.  Measured on the traffic mix alone (scripts/probes/
// hbm_walk2_probe.hip, hbm_walk3_probe.hip; profiles/r05_walk_probes.txt): EVERY walk in which a workgroup strides through
// its rows a piece at a time streams at 5.1-5.4 TB/s, the same 20 rows taken WHOLE, one after the other -- a workgroup
// then reads and writes one contiguous run of 20 x S samples -- at 6.0-6.3 TB/s with the XCD-contiguous order of the time
// bins.  A lane then meets 4 * NCH columns per ping, too many to carry lane-private sums and bin edges for; so this form
// carries nothing per column but the cached n log10(s - d): the range bin of every sample is computed, and the linear
// value goes to the bin's LDS accumulators with one LDS atomic per pair of samples that share a bin (the probe's
// "rows + lds" lines: the atomics cost 2 %).  Same arithmetic for Sv (operation for operation: the same bits), same
// exponential for the bins as the lean path above; the sums of a bin are added in another order (an ulp).
// Serves S <= 4096 (NCH = ceil(S / 1024) chunks per ping); longer rows and int16 samples keep the kernel above.
template <typename T>
__device__ __forceinline__ void rows_bins(T v0, T v1, bool ok0, bool ok1, double x0, double x1, double bin, double inv_bin,
                                          int n_rbins, T* lsum, uint32_t* lcnt) {
  // (ok = the value counts: a number, from a sample that is one; a sample outside the grid has bin -1)
  const int b0 = ok0 ? epa::range_bin_index(x0, bin, inv_bin, n_rbins, false) : -1;
  const int b1 = ok1 ? epa::range_bin_index(x1, bin, inv_bin, n_rbins, false) : -1;
  if (b0 == b1) {
    if (b0 >= 0) {
      lds_add(lsum + b0, v0 + v1);
      atomicAdd(lcnt + b0, 2u);
    }
  } else {
    if (b0 >= 0) {
      lds_add(lsum + b0, v0);
      atomicAdd(lcnt + b0, 1u);
    }
    if (b1 >= 0) {
      lds_add(lsum + b1, v1);
      atomicAdd(lcnt + b1, 1u);
    }
  }
}

template <typename T, bool STATS, bool WRITE_SV>
__device__ __forceinline__ void rows_pair(double sra0, double sra1, float2 in, bool clean, bool bins, const epa::CoefRow& r,
                                          double r0v, T g, T a2, T A0v, T nL0, T nL1, T nspread, double bin, double inv_bin,
                                          int n_rbins, const double* tab, T* lsum, uint32_t* lcnt, T* __restrict__ sv_dst,
                                          double& xmax, double& xmin, unsigned& nnan, double& xfirst, double& xlast) {
  const double x0 = fma(sra0, r.rb, r0v), x1 = fma(sra1, r.rb, r0v);  // echo_range = (s*ra)*rb [+0]
  const double rtd0 = x0 - r.shift, rtd1 = x1 - r.shift;
  const T rt0 = (T)rtd0, rt1 = (T)rtd1;
  T s10 = fma(g, (T)in.x, nL0), s11 = fma(g, (T)in.y, nL1);
  if (!clean) {  // (scalar) as process_pair
    const T NaN = epa::M<T>::nan();
    const bool pos0 = rtd0 > 0.0, pos1 = rtd1 > 0.0;
    if (pos0 & !(nL0 > -(T)__builtin_inf()))
      s10 = fma(g, (T)in.x, nspread * (log10_slow<T>(rt0) - log10_slow<T>((T)(r.ra * r.rb))));
    if (pos1 & !(nL1 > -(T)__builtin_inf()))
      s11 = fma(g, (T)in.y, nspread * (log10_slow<T>(rt1) - log10_slow<T>((T)(r.ra * r.rb))));
    s10 = pos0 ? s10 : NaN;
    s11 = pos1 ? s11 : NaN;
    if (STATS) {
      const double xq0 = fma((double)in.x, 0.0, x0), xq1 = fma((double)in.y, 0.0, x1);
      xmax = vmax_num(vmax_num(xmax, xq0), xq1);
      xmin = vmin_num(vmin_num(xmin, xq0), xq1);
      nnan += (unsigned)__builtin_popcountll(__ballot(in.x != in.x)) + (unsigned)__builtin_popcountll(__ballot(in.y != in.y));
    }
  }
  const T sv0 = s10 + fma(a2, rt0, A0v), sv1 = s11 + fma(a2, rt1, A0v);
  if (WRITE_SV) epa::store_nt2(sv_dst, sv0, sv1);
  if (bins) {  // (scalar: not for the pings outside every time bin)
    T v0 = lin_bins(sv0, tab), v1 = lin_bins(sv1, tab);
    bool ok0 = true, ok1 = true;
    if (!clean) {
      if (__builtin_expect(__builtin_isinf(sv0), 0)) v0 = sv0 < (T)0 ? (T)0 : sv0;
      if (__builtin_expect(__builtin_isinf(sv1), 0)) v1 = sv1 < (T)0 ? (T)0 : sv1;
      ok0 = (v0 == v0) & (in.x == in.x);
      ok1 = (v1 == v1) & (in.y == in.y);
      v0 = vmax_num(v0, (T)0);
      v1 = vmax_num(v1, (T)0);
    }
    rows_bins<T>(v0, v1, ok0, ok1, x0, x1, bin, inv_bin, n_rbins, lsum, lcnt);
  }
  xfirst = x0;
  xlast = x1;
}

#ifndef EPA_ROWS_MIN_WAVES
#define EPA_ROWS_MIN_WAVES 4
#endif
// A ping is taken in NG GROUPS of HCH chunks (NG * HCH = NCH rounded up): the HCH chunks of a group are unrolled -- their
// samples were requested a group ahead -- and the groups of a ping are a rolled loop (NCH = 4 unrolled flat wants ~200
// registers: address arithmetic and temporaries of four chunks in flight).  Up to two chunks (S <= 2048) there is one
// group per ping and the columns' n log10(s - d) are cached in registers (refreshed when the rows' d / ra change, once
// per workgroup for a file with constant tau / sample_interval); with two groups the column a register would belong to
// depends on the loop counter, so the logarithm is evaluated per sample instead (table-driven, ~22 instructions: the
// kernel waits for memory, not for the vector unit).
template <typename T, int NCH, bool WRITE_SV, bool RMAX>
__global__ __launch_bounds__(epa::kBlock, EPA_ROWS_MIN_WAVES) void fused_sv_mvbs_rows_kernel(
    const float* __restrict__ raw, const epa::CoefRow* __restrict__ coef, const int32_t* __restrict__ bin_start,
    T* __restrict__ sv_out, T* __restrict__ mvbs_out, T* __restrict__ sum_out, uint32_t* __restrict__ cnt_out, Args a) {
  constexpr int HCH = NCH > 2 ? 2 : NCH, NG = (NCH + HCH - 1) / HCH;
  constexpr bool FLYLOG = NG > 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);
  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);  // synchronised below
  const double* tab = mt.exp2_tab;

  const int c = blockIdx.y, tb = a.xcd_map ? epa::xcd_contiguous(blockIdx.x, a.n_tbins) : (int)blockIdx.x;
  const int S = a.S, n_rbins = a.n_rbins;
  const bool extra = tb == a.n_tbins;  // the pings that belong to NO time bin still get their Sv (two segments)
  if (extra && !WRITE_SV) return;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    lsum[i] = (T)0;
    lcnt[i] = 0u;
  }
  __syncthreads();

  const T nspread = (T)a.nspread;
  const double bin = a.range_bin, inv_bin = a.inv_range_bin;
  const epa::CoefRow* __restrict__ rowp0 = coef + (size_t)c * a.P;
  const float* __restrict__ raw_c = raw + (size_t)c * a.P * S;
  T* __restrict__ sv_c = WRITE_SV ? sv_out + (size_t)c * a.P * S : nullptr;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s00 = wave * 256 + 2 * lane;  // first sample of pair A in chunk 0; chunk k: + 1024 k, pair B: + 128
  const double s00d = (double)s00;        // (fl(s * ra) of a column is recomputed per sample: an add and a multiplication)
  double xmax = -__builtin_inf(), xmin = __builtin_inf();
  unsigned nnan = 0u;
  T nL[FLYLOG ? 1 : HCH][4];
  unsigned plain = 0u;  // (scalar) bit k: s - d >= 1 for every column of the wavefront in chunk k (its logs are finite)
  long long dcur = 0x7ff8dead00000001ll, racur = 0x7ff8dead00000002ll;

  const int nseg = extra ? 2 : 1;
  for (int seg = 0; seg < nseg; ++seg) {
    const int pb = extra ? (seg == 0 ? 0 : bin_start[a.n_tbins]) : bin_start[tb];
    const int pe = extra ? (seg == 0 ? bin_start[0] : a.P) : bin_start[tb + 1];
    if (pb >= pe) continue;
    // the samples of the NEXT group (the next chunks of the ping, or the first ones of the next ping) are requested
    // before the current group is processed
    float2 nxt[HCH][2];
    epa::CoefRow nxtR = rowp0[pb];
#pragma unroll
    for (int u = 0; u < HCH; ++u) {
      const int sA = s00 + 1024 * u, sB = sA + 128;
      nxt[u][0] = nxt[u][1] = make_float2(0.f, 0.f);
      if (sA < S) nxt[u][0] = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sA);
      if (sB < S) nxt[u][1] = *reinterpret_cast<const float2*>(raw_c + (size_t)pb * S + sB);
    }
    for (int p = pb; p < pe; ++p) {
      const epa::CoefRow r = nxtR;
      const size_t row_off = (size_t)p * S;
      const bool more = p + 1 < pe;
      if (more) nxtR = rowp0[p + 1];
      if (!((__double_as_longlong(r.d) == dcur) & (__double_as_longlong(r.ra) == racur))) {  // (scalar)
        dcur = __double_as_longlong(r.d);
        racur = __double_as_longlong(r.ra);
        plain = 0u;
#pragma unroll
        for (int k = 0; k < NG * HCH; ++k) {
          bool fin = true;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int sx = s00 + 1024 * k + (j < 2 ? 0 : 128) + (j & 1);
            const double sd = (double)sx - r.d;
            if (FLYLOG) {
              fin = fin & ((sx >= S) | ((sd >= 1.0) & (sd < 1e300)));
            } else {
              const T nl = nspread * log10_slow<T>((T)sd);
              nL[FLYLOG ? 0 : k][j] = nl;
              fin = fin & ((sx >= S) | (fabs(nl) < (T)__builtin_inf()));
            }
          }
          if (__ballot(!fin) == 0ull) plain |= 1u << k;
        }
      }
      const T g = (T)r.g, a2 = (T)r.alpha2;
      T A0 = (T)r.A0;
      double r0v = r.r0;
      asm volatile("" : "+v"(A0), "+v"(r0v));  // one copy per ping into vector registers, not one per sample
      const bool kpos = (__double2hiint(r.ra) > 0) & (__double2hiint(r.rb) > 0);  // (scalar) ra, rb > 0
#pragma unroll 1
      for (int h = 0; h < NG; ++h) {
        float2 cur[HCH][2];
#pragma unroll
        for (int u = 0; u < HCH; ++u) {
          cur[u][0] = nxt[u][0];
          cur[u][1] = nxt[u][1];
        }
        {  // request the next group
          const bool same_ping = h + 1 < NG;
          if (same_ping | more) {
            const float* nrow = raw_c + row_off + (same_ping ? 0 : S);
            const int k0 = same_ping ? (h + 1) * HCH : 0;
#pragma unroll
            for (int u = 0; u < HCH; ++u) {
              const int sA = s00 + 1024 * (k0 + u), sB = sA + 128;
              if (sA < S) nxt[u][0] = *reinterpret_cast<const float2*>(nrow + sA);
              if (sB < S) nxt[u][1] = *reinterpret_cast<const float2*>(nrow + sB);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < HCH; ++u) {
          const int k = h * HCH + u;
          const int sA = s00 + 1024 * k, sB = sA + 128;
          if (sA < S) {
            const float2 inA = cur[u][0], inB = cur[u][1];
            const double sAd = s00d + (double)(1024 * k);
            const double sa0 = sAd * r.ra, sa1 = (sAd + 1.0) * r.ra, sb0 = (sAd + 128.0) * r.ra, sb1 = (sAd + 129.0) * r.ra;
            const bool pl = ((plain >> k) & 1u) != 0u;
            // the chunk is CLEAN for this wavefront (see fused_sv_mvbs_kernel): lean forms, no per-sample NaN handling
            bool clean = false;
#if EPA_FUSED_LEAN
            {
              const double xa = fma(sa0, r.rb, r0v);
              const bool bad = not_finite(inA.x) | not_finite(inA.y) | not_finite(inB.x) | not_finite(inB.y) |
                               !(xa - r.shift > 0.0);
              clean = pl & kpos & (__ballot(bad) == 0ull);
            }
#endif
            T nl0, nl1, nl2, nl3;
            if (FLYLOG) {
              if (pl) {  // (scalar) every s - d of the wavefront's columns is a positive normal number
                nl0 = nspread * epa::fast_log10_lean((T)(sAd - r.d), mt.log_tab);
                nl1 = nspread * epa::fast_log10_lean((T)((sAd + 1.0) - r.d), mt.log_tab);
                nl2 = nspread * epa::fast_log10_lean((T)((sAd + 128.0) - r.d), mt.log_tab);
                nl3 = nspread * epa::fast_log10_lean((T)((sAd + 129.0) - r.d), mt.log_tab);
              } else {
                nl0 = nspread * log10_slow<T>((T)(sAd - r.d));
                nl1 = nspread * log10_slow<T>((T)((sAd + 1.0) - r.d));
                nl2 = nspread * log10_slow<T>((T)((sAd + 128.0) - r.d));
                nl3 = nspread * log10_slow<T>((T)((sAd + 129.0) - r.d));
              }
            } else {
              nl0 = nL[FLYLOG ? 0 : u][0]; nl1 = nL[FLYLOG ? 0 : u][1]; nl2 = nL[FLYLOG ? 0 : u][2]; nl3 = nL[FLYLOG ? 0 : u][3];
            }
            double xf, xl, xdummy;
            rows_pair<T, RMAX, WRITE_SV>(sa0, sa1, inA, clean, !extra, r, r0v, g, a2, A0, nl0, nl1, nspread, bin, inv_bin,
                                         n_rbins, tab, lsum, lcnt, WRITE_SV ? sv_c + row_off + sA : nullptr, xmax, xmin,
                                         nnan, xf, xl);
            if (sB < S)
              rows_pair<T, RMAX, WRITE_SV>(sb0, sb1, inB, clean, !extra, r, r0v, g, a2, A0, nl2, nl3, nspread, bin, inv_bin,
                                           n_rbins, tab, lsum, lcnt, WRITE_SV ? sv_c + row_off + sB : nullptr, xmax, xmin,
                                           nnan, xdummy, xl);
            if (RMAX && clean) {  // no NaN among the wavefront's samples: the lane's smallest / largest range of the chunk
              xmin = vmin_num(xmin, xf);
              xmax = vmax_num(xmax, xl);
            }
          }
        }
      }
    }
  }
  if (RMAX) {  // max valid echo_range seen by this workgroup -> one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_down(xmax, o, 64));
    if (lane == 0 && xmax > -__builtin_inf()) atomicMax(a.rmax_key, ordered_key(xmax));
    if (a.rstat) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) xmin = fmin(xmin, __shfl_down(xmin, o, 64));  // (nnan is the wavefront's already)
      if (lane == 0 && xmin < __builtin_inf()) atomicMin(a.rstat, ordered_key(xmin));
      if (lane == 0 && nnan > 0u) atomicAdd(a.rstat + 1, (unsigned long long)nnan);
    }
  }
  if (extra) return;
  __syncthreads();
  const size_t cell0 = ((size_t)c * a.n_tbins + tb) * n_rbins;
  T* out = mvbs_out + cell0;
  T* gsum = sum_out ? sum_out + cell0 : nullptr;
  uint32_t* gcnt = cnt_out ? cnt_out + cell0 : nullptr;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    const uint32_t n = lcnt[i];
    const T sm = lsum[i];
    out[i] = n > 0u ? (T)10 * epa::M<T>::log10(sm / (T)n) : (T)a.fill_value;
    if (gsum) gsum[i] = sm;
    if (gcnt) gcnt[i] = n;
  }
}

template <typename T>
int launch_rows(Args& a, const float* raw, const double* coef, const int32_t* bin_start, void* sv_out, void* mvbs_out,
                void* sum_out, uint32_t* cnt_out, int C, size_t lds_bytes, hipStream_t st) {
  const dim3 grid((unsigned)a.n_tbins + 1u, (unsigned)C);  // +1: pings outside every time bin
  a.tab_off = (unsigned)((lds_bytes + 15) & ~(size_t)15);
  lds_bytes = a.tab_off + epa::kMathTabBytes;
  a.xcd_map = epa::xcd_map_enabled() ? 1 : 0;
  const int nch = (a.S + 1023) / 1024;
#define EPA_FR(N, W, R)                                                                        \
  do {                                                                                         \
    auto kern = fused_sv_mvbs_rows_kernel<T, N, W, R>;                                         \
    if (lds_bytes > 64 * 1024)                                                                 \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,            \
                                        (int)lds_bytes));                                      \
    hipLaunchKernelGGL(kern, grid, dim3(epa::kBlock), lds_bytes, st, raw,                      \
                       reinterpret_cast<const epa::CoefRow*>(coef), bin_start, (T*)sv_out,     \
                       (T*)mvbs_out, (T*)sum_out, cnt_out, a);                                 \
  } while (0)
#define EPA_FRN(N)                                                                             \
  do {                                                                                         \
    if (a.rmax_key) {                                                                          \
      if (sv_out) EPA_FR(N, true, true); else EPA_FR(N, false, true);                          \
    } else {                                                                                   \
      if (sv_out) EPA_FR(N, true, false); else EPA_FR(N, false, false);                        \
    }                                                                                          \
  } while (0)
  if (nch == 1) EPA_FRN(1); else if (nch == 2) EPA_FRN(2); else if (nch == 3) EPA_FRN(3); else EPA_FRN(4);
#undef EPA_FRN
#undef EPA_FR
  return epa::check_launch("fused_sv_mvbs_rows_kernel");
}

