// K5 / K6 (and the fused K1+K5): linear-domain (ping-bin x range-bin) block reductions.
//
// One workgroup owns one (channel, ping-bin) pair -- 20 pings x S samples at the BASELINE
// configs -- walks its pings with every lane on a fixed 4-sample column (16-B coalesced loads),
// keeps a private (bin, sum, count) accumulator per column across the pings of the bin, flushes
// to per-range-bin partial sums in LDS (ds_add_f64 / ds_add_u32) only when the range bin under
// the column changes, and finishes the bin itself: mean -> 10*log10 -> MVBS row (or, for the
// noise estimate, min over range blocks).  Nothing is re-read from HBM; with the fused source the
// raw power is read once (4 B/sample) and Sv written once (8 B/sample in f64).
//
// Reference arithmetic replaced (paths under /root/reference/echopype):
//   commongrid/utils.py:592,614-627 + :92   flox group-by nanmean/mean in the linear domain
//   commongrid/api.py:108-128               range / ping bin edges (uniform, left or right closed)
//   commongrid/api.py:217-238               index binning (coarsen ... "pad" mean, echo_range min)
//   clean/api.py:397-422                    noise estimate (block mean of calibrated power, min)
//   calibrate_ek.py / range.py              (fused source) as in sv_power.hip
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>

#include "fast_math.h"
#include "sample_math.h"

namespace {

enum { SRC_RAW = 0, SRC_SV = 1, SRC_SV_DENOISE = 2, SRC_RAW_DENOISE = 3 };
enum { OP_MVBS = 0, OP_NOISE = 1 };
enum { BIN_PHYS = 0, BIN_INDEX = 1 };

struct ReduceArgs {
  double* range_max_out;      // fused fast path only
  double* range_stats_out;    // fused fast path only: {nanmin, nanmax, NaN count} of the echo_range, with range_max_out
  int range_stats_filled;     // set by run_mvbs when the kernel it chose leaves them
  const int16_t* raw_i16;     // fused fast path only: instrument int16 power samples ...
  const int32_t* n_valid;     // ... with the recorded length of every ping
  const double* dscale;       // fused fast path only: bin on depth = doffset[c,p] + dscale[c,p] * echo_range ...
  const double* doffset;
  void* depth_out;            // ... and optionally write that array
  const float* raw;
  const void* sv;
  const void* range;
  const epa::CoefRow* coef;
  const double* alpha2;
  int C, P, S;
  double nspread;
  unsigned cal_flags;
  const int32_t* bin_start;
  const int32_t* ping_perm;
  int n_tbins, ping_num, nparts;
  int ping_phase;  // index blocks: local ping p belongs to block (p + ping_phase) / ping_num (shards of a longer file)
  int bin_mode;
  double range_bin, inv_range_bin;
  int n_rbins, range_sample_num;
  unsigned bin_flags;
  double fill_value, noise_max;
  void* sv_out;
  void* range_out;
  void* out;
  void* sum_out;
  uint32_t* cnt_out;
  // SRC_SV_DENOISE: background-noise removal applied on the fly (K7 fused into the reduction)
  const double* noise;   // [C * n_pblocks] per ping-block noise (epa_noise_estimate layout)
  int noise_ping_num, n_pblocks, noise_phase;
  // OP_NOISE, optional: raw linear (sum, count) per range block of the FIRST and LAST ping block, [2][C][n_rbins]
  // (slot 1 is written only when the last block is not the first) -- what a ping-sharded run merges across ranks
  double* edge_sum_out;
  uint32_t* edge_cnt_out;
  double snr;
  void* sv_noise_out;    // optional Sv_noise output (Sv_corrected goes to sv_out)
  unsigned long long* mm_keys;  // optional [4]: ordered keys of min/max(Sv_noise), min/max(Sv_corrected)
  int use_lds;
  unsigned cnt_off;  // byte offset of the count array in dynamic LDS
  unsigned tab_off;  // byte offset of the exp/log tables (fast_math.h) in dynamic LDS
};

template <typename T>
__device__ __forceinline__ void atomic_add(T* p, T v) {
  unsafeAtomicAdd(p, v);
}

// NaN-skipping min over the workgroup; returns NaN when no lane contributed.
template <typename T>
__device__ T block_nanmin(T v, bool valid, T* scratch /* >= 8 elements of LDS */) {
  const T inf = (T)__builtin_inf();
  T m = valid ? v : inf;
  int any = valid ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmin(m, __shfl_down(m, o, 64));
    any |= __shfl_down(any, o, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
    scratch[wave] = m;
    scratch[4 + wave] = (T)any;
  }
  __syncthreads();
  T r = inf;
  int a = 0;
  for (int w = 0; w < epa::kBlock / 64; ++w) {
    r = fmin(r, scratch[w]);
    a |= (scratch[4 + w] != (T)0);
  }
  return a ? r : epa::M<T>::nan();
}

template <typename T, int SRC, int OP, int VEC>
__global__ __launch_bounds__(epa::kBlock, 3) void block_reduce_kernel(ReduceArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kChunk = epa::kBlock * VEC;
  T* lsum = reinterpret_cast<T*>(smem);
  uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + a.cnt_off);

  const int c = blockIdx.y;
  const int tb = blockIdx.x / a.nparts;
  const int part = blockIdx.x - tb * a.nparts;
  // tb == n_tbins (fused source only): pings that belong to no time bin still get their Sv/echo_range
  const bool extra = tb == a.n_tbins;
  if (extra && (part != 0 || !(a.sv_out || a.range_out || a.sv_noise_out))) return;
  const int nseg = extra ? 2 : 1;
  const int S = a.S, n_rbins = a.n_rbins;
  const bool use_lds = a.use_lds != 0;
  const size_t cell0 = ((size_t)c * a.n_tbins + tb) * n_rbins;
  T* gsum = a.sum_out ? reinterpret_cast<T*>(a.sum_out) + cell0 : nullptr;
  uint32_t* gcnt = a.cnt_out ? a.cnt_out + cell0 : nullptr;

  const epa::MathTabs mt = epa::build_math_tabs(smem + a.tab_off);
  if (use_lds) {
    for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
      lsum[i] = (T)0;
      lcnt[i] = 0u;
    }
  }
  __syncthreads();
  T* asum = use_lds ? lsum : gsum;
  uint32_t* acnt = use_lds ? lcnt : gcnt;

  const bool skipna = a.bin_flags & EPA_BIN_SKIPNA;
  const bool closed_right = a.bin_flags & EPA_BIN_CLOSED_RIGHT;
  const bool range_as_stored = a.bin_flags & EPA_BIN_RANGE_AS_STORED;
  const bool guard = a.cal_flags & EPA_FLAG_GUARD_POS;
  const bool mask_range = a.cal_flags & EPA_FLAG_MASK_RANGE;
  const bool phys = a.bin_mode == BIN_PHYS;
  const T nspread = (T)a.nspread;
  const T* svp = reinterpret_cast<const T*>(a.sv);
  const T* rgp = reinterpret_cast<const T*>(a.range);
  T* sv_out = reinterpret_cast<T*>(a.sv_out);
  T* range_out = reinterpret_cast<T*>(a.range_out);
  T* sv_noise_out = reinterpret_cast<T*>(a.sv_noise_out);

  double xmax = -__builtin_inf();  // max valid echo_range seen by this lane (raw sources, optional)
  double mm[4] = {__builtin_inf(), -__builtin_inf(), __builtin_inf(), -__builtin_inf()};  // denoise by-product
  for (int seg = 0; seg < nseg; ++seg) {
  int pb, pe;
  if (extra) {
    pb = seg == 0 ? 0 : a.bin_start[a.n_tbins];
    pe = seg == 0 ? a.bin_start[0] : a.P;
  } else if (a.bin_start) {
    pb = a.bin_start[tb];
    pe = a.bin_start[tb + 1];
  } else {
    pb = max(0, tb * a.ping_num - a.ping_phase);
    pe = min(a.P, (tb + 1) * a.ping_num - a.ping_phase);
  }
  if (a.nparts > 1 && !extra) {
    const int per = (pe - pb + a.nparts - 1) / a.nparts;
    const int b2 = pb + part * per;
    pe = min(pe, b2 + per);
    pb = b2;
  }
  for (int chunk0 = 0; chunk0 < S; chunk0 += kChunk) {
    const int s0 = chunk0 + threadIdx.x * VEC;
    if (s0 < S) {
      // per-column state kept across the pings of the bin
      int acc_rb[VEC];
      T acc_sum[VEC];
      uint32_t acc_cnt[VEC];
      double blo[VEC], bhi[VEC];  // edges of the range bin the column currently sits in
      epa::ColumnLog<T, VEC> col;
      // transmission-loss log of the UNSHIFTED range, separable like the spreading term:
      // log10(R) = log10(k) + log10(s - d_tl), d_tl = -r0/k (0 for EK); cached per column
      epa::ColumnLog<T, VEC> col_tl;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        acc_rb[j] = phys ? -1 : (s0 + j) / a.range_sample_num;
        acc_sum[j] = (T)0;
        acc_cnt[j] = 0u;
        blo[j] = 1.0;
        bhi[j] = 0.0;  // empty interval: first sample always takes the slow path
      }
      for (int pi = pb; pi < pe; ++pi) {
        // wave-uniform by construction; readfirstlane lets the compiler keep the row index in an
        // SGPR and fetch the 64-B coefficient row with scalar loads
        const int p = __builtin_amdgcn_readfirstlane(a.ping_perm ? a.ping_perm[pi] : pi);
        const size_t row = (size_t)c * a.P + p;
        const size_t off = row * S + s0;
        T sv[VEC];
        T vlin[VEC];  // SRC_SV_DENOISE: linear value of the corrected Sv (NaN where removed)
        T logR[VEC];  // raw sources: log10(echo_range) from the cached column logs
        double x[VEC];
        bool xok[VEC];
        constexpr bool kFromRaw = SRC == SRC_RAW || SRC == SRC_RAW_DENOISE;
        constexpr bool kDenoise = SRC == SRC_SV_DENOISE || SRC == SRC_RAW_DENOISE;
        if (kFromRaw) {
          epa::RawVec<VEC> in;
          in.load(a.raw + off);
          const epa::RowK<T> rk(a.coef[row]);
          col.update(rk.d, s0, nspread);
          if (OP == OP_NOISE || kDenoise) {
            const double k = rk.ra * rk.rb;
            col_tl.update(-rk.r0 / k, s0, (T)1);
            const T log10k = epa::fast_log10((T)k, mt.log_tab);
#pragma unroll
            for (int j = 0; j < VEC; ++j) logR[j] = col_tl.nL[j] + log10k;
          }
          T rg[VEC];
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            x[j] = rk.range(s0 + j);
            sv[j] = epa::cal_power_sample<T>(in.v[j], s0 + j, rk, nspread, col.nL[j], guard, x[j]);
            xok[j] = !(mask_range && !(in.v[j] == in.v[j]));
            rg[j] = xok[j] ? (T)x[j] : epa::M<T>::nan();
            if (a.range_max_out && xok[j]) xmax = fmax(xmax, (double)rg[j]);  // as stored (T)
          }
          if (sv_out && !kDenoise) epa::store_vec<T, VEC>(sv_out + off, sv);
          if (range_out) epa::store_vec<T, VEC>(range_out + off, rg);
        } else {
          epa::load_vec<T, VEC>(svp + off, sv);
          if (rgp) {
            T rg[VEC];
            epa::load_vec<T, VEC>(rgp + off, rg);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              x[j] = (double)rg[j];
              xok[j] = true;  // NaN coordinates drop out in the bin search
            }
          } else if (a.coef) {
            const epa::CoefRow cr = a.coef[row];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              x[j] = epa::row_range(cr, s0 + j);
              if (range_as_stored) x[j] = (double)(T)x[j];  // what the echo_range array of dtype T would hold
              xok[j] = true;
            }
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              x[j] = 0.0;
              xok[j] = true;
            }
          }
        }
        if (kDenoise) {
          {
            // clean/api.py:425-430 (noise of the ping block + transmission loss) and :485-487
            const T nb = (T)a.noise[(size_t)c * a.n_pblocks + (p + a.noise_phase) / a.noise_ping_num];
            const T na2 = (T)a.alpha2[row];
            const T snr = (T)a.snr;
            T sn[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
              const T xr = xok[j] ? (T)x[j] : epa::M<T>::nan();  // echo_range is NaN where masked
              const T lg = kFromRaw ? logR[j] : epa::fast_log10(xr >= (T)1 ? xr : (T)1, mt.log_tab);
              const T tl = (T)20 * (xr >= (T)1 ? lg : (T)0) + na2 * xr;
              sn[j] = nb + tl;
              const T lin = epa::lin_from_db(sv[j], mt.exp2_tab) - epa::lin_from_db(sn[j], mt.exp2_tab);
              T corr = lin > (T)0 ? (T)10 * epa::fast_log10(lin, mt.log_tab) : epa::M<T>::nan();
              const bool keep = corr - sn[j] > snr;
              sv[j] = keep ? corr : epa::M<T>::nan();
              vlin[j] = keep ? lin : epa::M<T>::nan();
              if (a.mm_keys) {  // fmin / fmax ignore NaN operands
                mm[0] = fmin(mm[0], (double)sn[j]);
                mm[1] = fmax(mm[1], (double)sn[j]);
                mm[2] = fmin(mm[2], (double)sv[j]);
                mm[3] = fmax(mm[3], (double)sv[j]);
              }
            }
            if (sv_noise_out) epa::store_vec<T, VEC>(sv_noise_out + off, sn);
            if (sv_out) epa::store_vec<T, VEC>(sv_out + off, sv);
          }
        }
        T a2 = (T)0;
        if (OP == OP_NOISE) a2 = (T)a.alpha2[row];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          T v;
          if (OP == OP_MVBS) {
            // _log2lin, compute.py:14-27 (the denoised source already holds the linear value)
            v = kDenoise ? vlin[j] : epa::lin_from_db(sv[j], mt.exp2_tab);
          } else {
            // clean/api.py:397-401: where(R >= 1, R, 1) also maps NaN ranges to 1
            const T xr = (T)x[j];
            const T lg = kFromRaw ? logR[j] : epa::fast_log10(xr >= (T)1 ? xr : (T)1, mt.log_tab);
            const T tl = (T)20 * (xr >= (T)1 ? lg : (T)0) + a2 * xr;
            v = epa::lin_from_db(sv[j] - tl, mt.exp2_tab);
          }
          if (phys) {
            // fast path: the column is still inside the bin it was in for the previous ping
            const bool same = xok[j] && (closed_right ? (x[j] > blo[j] && x[j] <= bhi[j])
                                                      : (x[j] >= blo[j] && x[j] < bhi[j]));
            if (!same) {
              const int rb = xok[j] ? epa::range_bin_index(x[j], a.range_bin, a.inv_range_bin,
                                                            n_rbins, closed_right)
                                    : -1;
              if (rb != acc_rb[j]) {
                if (acc_rb[j] >= 0 && acc_cnt[j] > 0u && !extra) {
                  atomic_add(asum + acc_rb[j], acc_sum[j]);
                  atomicAdd(acnt + acc_rb[j], acc_cnt[j]);
                }
                acc_rb[j] = rb;
                acc_sum[j] = (T)0;
                acc_cnt[j] = 0u;
              }
              if (rb >= 0) {
                blo[j] = (double)rb * a.range_bin;
                bhi[j] = (double)(rb + 1) * a.range_bin;
              } else {
                blo[j] = 1.0;
                bhi[j] = 0.0;
              }
            }
          }
          if (acc_rb[j] >= 0 && (!skipna || v == v)) {
            acc_sum[j] += v;
            acc_cnt[j] += 1u;
          }
        }
      }
      if (!extra) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (acc_rb[j] >= 0 && acc_cnt[j] > 0u) {
            atomic_add(asum + acc_rb[j], acc_sum[j]);
            atomicAdd(acnt + acc_rb[j], acc_cnt[j]);
          }
        }
      }
    }
  }
  }
  if (a.range_max_out) {  // nanmax(echo_range) by-product: order-preserving u64 key, decoded by the launcher
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) xmax = fmax(xmax, __shfl_down(xmax, o, 64));
    if ((threadIdx.x & 63) == 0 && xmax > -__builtin_inf()) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(xmax);
      atomicMax(reinterpret_cast<unsigned long long*>(a.range_max_out),
                (b >> 63) ? ~b : (b | 0x8000000000000000ull));
    }
  }
  if (a.mm_keys) {
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
        atomicMin(a.mm_keys + 0, key(mm[0]));
        atomicMax(a.mm_keys + 1, key(mm[1]));
      }
      if (mm[2] <= mm[3]) {
        atomicMin(a.mm_keys + 2, key(mm[2]));
        atomicMax(a.mm_keys + 3, key(mm[3]));
      }
    }
  }
  if (extra) return;

  if (!use_lds) return;  // accumulated straight into global partials; a finalize kernel follows
  __syncthreads();
  if (a.nparts > 1) {
    for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
      const uint32_t n = lcnt[i];
      if (n > 0u) {
        atomic_add(gsum + i, lsum[i]);
        atomicAdd(gcnt + i, n);
      }
    }
    return;
  }
  if (OP == OP_MVBS) {
    T* out = reinterpret_cast<T*>(a.out) + cell0;
    for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
      const uint32_t n = lcnt[i];
      const T s = lsum[i];
      out[i] = n > 0u ? (T)10 * epa::M<T>::log10(s / (T)n) : (T)a.fill_value;  // _lin2log
      if (gsum) gsum[i] = s;
      if (gcnt) gcnt[i] = n;
    }
  } else {
    if (a.edge_sum_out && (tb == 0 || tb == a.n_tbins - 1)) {
      const size_t e0 = ((size_t)(tb == 0 ? 0 : 1) * a.C + c) * n_rbins;
      for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
        a.edge_sum_out[e0 + i] = (double)lsum[i];
        a.edge_cnt_out[e0 + i] = lcnt[i];
      }
    }
    T best = (T)__builtin_inf();
    bool any = false;
    for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
      const uint32_t n = lcnt[i];
      if (n > 0u) {
        const T db = (T)10 * epa::M<T>::log10(lsum[i] / (T)n);
        if (db == db) {
          best = fmin(best, db);
          any = true;
        }
      }
    }
    __syncthreads();  // all reads of lsum done before it is reused as scratch
    T m = block_nanmin<T>(best, any, lsum);
    if (threadIdx.x == 0) {
      double r = (double)m;
      if (a.noise_max == a.noise_max) r = (r < a.noise_max) ? r : a.noise_max;  // api.py:418-422
      reinterpret_cast<double*>(a.out)[(size_t)c * a.n_tbins + tb] = r;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(epa::kBlock) void mvbs_finalize_kernel(const T* __restrict__ sum,
                                                                    const uint32_t* __restrict__ cnt,
                                                                    size_t n, T fill,
                                                                    T* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t k = cnt[i];
    out[i] = k > 0u ? (T)10 * epa::M<T>::log10(sum[i] / (T)k) : fill;
  }
}

// noise, multi-part path: one workgroup per (channel, ping block) row of merged partials
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void noise_rowmin_kernel(const T* __restrict__ sum,
                                                                   const uint32_t* __restrict__ cnt,
                                                                   int n_rbins, double noise_max,
                                                                   double* __restrict__ out) {
  __shared__ T scratch[8];
  const size_t base = (size_t)blockIdx.x * n_rbins;
  T best = (T)__builtin_inf();
  bool any = false;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    const uint32_t n = cnt[base + i];
    if (n > 0u) {
      const T db = (T)10 * epa::M<T>::log10(sum[base + i] / (T)n);
      if (db == db) {
        best = fmin(best, db);
        any = true;
      }
    }
  }
  T m = block_nanmin<T>(best, any, scratch);
  if (threadIdx.x == 0) {
    double r = (double)m;
    if (noise_max == noise_max) r = (r < noise_max) ? r : noise_max;
    out[blockIdx.x] = r;
  }
}

// echo_range block min for compute_MVBS_index_binning (api.py:232-238): one wave per output cell
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void block_nanmin_kernel(const T* __restrict__ x, int C,
                                                                   int P, int S, int ping_num,
                                                                   int rsn, int Pb, int Sb,
                                                                   T* __restrict__ out) {
  const long long cells = (long long)C * Pb * Sb;
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long cell = wave0; cell < cells; cell += nwaves) {
    const int sb = (int)(cell % Sb);
    const int pbk = (int)((cell / Sb) % Pb);
    const int c = (int)(cell / ((long long)Sb * Pb));
    const int p0 = pbk * ping_num, p1 = min(P, p0 + ping_num);
    const int s0 = sb * rsn, s1 = min(S, s0 + rsn);
    const int w = s1 - s0;
    const int n = (p1 - p0) * w;
    T m = (T)__builtin_inf();
    int any = 0;
    for (int e = lane; e < n; e += 64) {
      const int p = p0 + e / w, s = s0 + e % w;
      const T v = x[((size_t)c * P + p) * S + s];
      if (v == v) {
        m = fmin(m, v);
        any = 1;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      m = fmin(m, __shfl_down(m, o, 64));
      any |= __shfl_down(any, o, 64);
    }
    if (lane == 0) out[cell] = any ? m : epa::M<T>::nan();
  }
}

}  // namespace

// fused_sv_mvbs.hip
int epa_fused_fast_path(const void* raw, int raw_is_i16, const int32_t* n_valid, const double* coef,
                        int C, int P, int S, double nspread,
                        unsigned cal_flags, const int32_t* bin_start, int n_tbins, double range_bin,
                        int n_rbins, unsigned bin_flags, double fill_value, void* sv_out,
                        void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                        size_t lds_bytes, unsigned cnt_off, unsigned long long* rmax_key,
                        unsigned long long* rstat, const double* dscale, const double* doffset, void* depth_out,
                        hipStream_t st);

// chain_fast.hip
int epa_mvbs_rows_fast_path(const void* sv, const double* coef, int C, int P, int S, const int32_t* bin_start,
                            int n_tbins, double range_bin, int n_rbins, int as_stored, double fill_value,
                            void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype, size_t lds_bytes,
                            unsigned cnt_off, hipStream_t st);
int epa_chain_fast_pass1(const float* raw, const double* coef, const double* alpha2, int C, int P, int S,
                         double nspread, int ping_num, int rsn, int ping_phase, double noise_max, void* sv_out,
                         double* noise_out, double* edge_sum_out, uint32_t* edge_cnt_out,
                         unsigned long long* rmax_key, unsigned long long* rstat, int dtype, hipStream_t st);
int epa_chain_fast_pass2(const float* raw, const double* coef, const double* alpha2, const double* noise, int C,
                         int P, int S, double nspread, int ping_num, int ping_phase, double snr,
                         const int32_t* bin_start, int n_tbins, double range_bin, int n_rbins, double fill_value,
                         void* noise_out, void* corr_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                         size_t lds_acc_bytes, unsigned cnt_off, unsigned long long* mm_keys, hipStream_t st);

namespace {

struct Plan {
  int nparts, use_lds, vec;
  size_t lds_bytes;
  unsigned cnt_off, tab_off;
};

template <typename T>
Plan make_plan(int C, int P, int S, int n_tbins, int n_rbins, bool aligned16) {
  Plan pl;
  pl.vec = (S % 4 == 0 && aligned16) ? 4 : 1;
  if (const char* e = getenv("EPA_REDUCE_VEC")) {  // tuning aid
    const int v = atoi(e);
    if ((v == 1) || (v == 2 && S % 2 == 0 && aligned16) || (v == 4 && S % 4 == 0 && aligned16)) pl.vec = v;
  }
  const size_t sum_bytes = ((size_t)n_rbins * sizeof(T) + 15) & ~(size_t)15;
  const size_t need = sum_bytes + (size_t)n_rbins * sizeof(uint32_t);
  pl.use_lds = need <= 128 * 1024;
  const size_t acc_bytes = pl.use_lds ? (need < 64 ? 64 : need) : 64;
  pl.cnt_off = (unsigned)sum_bytes;
  pl.tab_off = (unsigned)((acc_bytes + 15) & ~(size_t)15);
  pl.lds_bytes = pl.tab_off + epa::kMathTabBytes;
  const long long groups = (long long)C * n_tbins;
  pl.nparts = 1;
  if (groups < 1024) {
    const long long avg = n_tbins > 0 ? (P + n_tbins - 1) / n_tbins : P;
    long long want = (2048 + groups - 1) / groups;
    long long maxp = (avg + 7) / 8;
    if (maxp < 1) maxp = 1;
    pl.nparts = (int)(want < maxp ? want : maxp);
    if (pl.nparts < 1) pl.nparts = 1;
  }
  return pl;
}

template <typename T, int SRC, int OP>
int launch_reduce(ReduceArgs& a, const Plan& pl, hipStream_t st) {
  // fused source: one extra ping-bin slot for the pings that fall in no time bin (Sv only)
  const long long tslots = (long long)a.n_tbins + ((SRC != SRC_SV && a.bin_start) ? 1 : 0);
  const dim3 grid((unsigned)(tslots * a.nparts), (unsigned)a.C);
  const dim3 block(epa::kBlock);
  a.use_lds = pl.use_lds;
  a.cnt_off = pl.cnt_off;
  a.tab_off = pl.tab_off;
#define EPA_RL(V)                                                                                 \
  do {                                                                                            \
    auto kern = block_reduce_kernel<T, SRC, OP, V>;                                               \
    if (pl.lds_bytes > 64 * 1024) {                                                               \
      EPA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize,               \
                                        (int)pl.lds_bytes));                                      \
    }                                                                                             \
    hipLaunchKernelGGL(kern, grid, block, pl.lds_bytes, st, a);                                   \
  } while (0)
  if (pl.vec == 4) EPA_RL(4); else if (pl.vec == 2) EPA_RL(2); else EPA_RL(1);
#undef EPA_RL
  return epa::check_launch("block_reduce_kernel");
}

inline bool al16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
int zero_partials(void* sum_out, uint32_t* cnt_out, size_t cells, hipStream_t st) {
  EPA_CHECK_HIP(hipMemsetAsync(sum_out, 0, cells * sizeof(T), st));
  EPA_CHECK_HIP(hipMemsetAsync(cnt_out, 0, cells * sizeof(uint32_t), st));
  return EPA_OK;
}

template <typename T, int SRC>
int run_mvbs(ReduceArgs& a, hipStream_t st) {
  Plan pl = make_plan<T>(a.C, a.P, a.S, a.n_tbins, a.n_rbins,
                               al16(a.raw) && al16(a.sv) && al16(a.range) && al16(a.sv_out) &&
                                   al16(a.range_out));
  if (SRC == SRC_RAW && (a.range_max_out || a.raw_i16)) pl.nparts = 1;  // single-stage kernel only
  a.nparts = pl.nparts;
  const size_t cells = (size_t)a.C * a.n_tbins * a.n_rbins;
  const bool two_stage = pl.nparts > 1 || !pl.use_lds;
  if (SRC == SRC_RAW && !two_stage && pl.vec == 4 && !a.ping_perm && !a.range_out &&
      a.cal_flags == (EPA_FLAG_GUARD_POS | EPA_FLAG_MASK_RANGE) && a.bin_flags == EPA_BIN_SKIPNA &&
      (a.raw_i16 || !getenv("EPA_NO_FAST_PATH")) && (a.range_stats_filled = a.range_stats_out != nullptr, true))
    return epa_fused_fast_path(a.raw_i16 ? (const void*)a.raw_i16 : (const void*)a.raw, a.raw_i16 != nullptr,
                               a.n_valid, reinterpret_cast<const double*>(a.coef), a.C, a.P, a.S,
                               a.nspread, a.cal_flags, a.bin_start, a.n_tbins, a.range_bin,
                               a.n_rbins, a.bin_flags, a.fill_value, a.sv_out, a.out, a.sum_out,
                               a.cnt_out, sizeof(T) == 8 ? EPA_F64 : EPA_F32, pl.tab_off,
                               pl.cnt_off, reinterpret_cast<unsigned long long*>(a.range_max_out),
                               reinterpret_cast<unsigned long long*>(a.range_stats_out), a.dscale, a.doffset,
                               a.depth_out, st);
  if (a.dscale) {
    epa::set_error("epa_sv_mvbs_fused_depth: binning on depth is served by the default configuration only (sorted "
                   "pings, S %% 4 == 0, aligned arrays, range grid within LDS)");
    return EPA_EUNSUPPORTED;
  }
  if (SRC == SRC_RAW_DENOISE && !two_stage && pl.vec == 4 && !a.ping_perm && !a.range_out &&
      a.cal_flags == (EPA_FLAG_GUARD_POS | EPA_FLAG_MASK_RANGE) && a.bin_flags == EPA_BIN_SKIPNA &&
      !getenv("EPA_NO_FAST_PATH"))
    return epa_chain_fast_pass2(a.raw, reinterpret_cast<const double*>(a.coef), a.alpha2, a.noise, a.C, a.P, a.S,
                                a.nspread, a.noise_ping_num, a.noise_phase, a.snr, a.bin_start, a.n_tbins, a.range_bin,
                                a.n_rbins, a.fill_value, a.sv_noise_out, a.sv_out, a.out, a.sum_out, a.cnt_out,
                                sizeof(T) == 8 ? EPA_F64 : EPA_F32, pl.tab_off, pl.cnt_off, a.mm_keys, st);
  if (SRC == SRC_SV && !a.range && a.coef && !two_stage && pl.vec == 4 && !a.ping_perm &&
      (a.bin_flags & ~EPA_BIN_RANGE_AS_STORED) == EPA_BIN_SKIPNA && !getenv("EPA_NO_FAST_PATH"))
    return epa_mvbs_rows_fast_path(a.sv, reinterpret_cast<const double*>(a.coef), a.C, a.P, a.S, a.bin_start, a.n_tbins,
                                   a.range_bin, a.n_rbins, (a.bin_flags & EPA_BIN_RANGE_AS_STORED) ? 1 : 0,
                                   a.fill_value, a.out, a.sum_out, a.cnt_out, sizeof(T) == 8 ? EPA_F64 : EPA_F32,
                                   pl.tab_off, pl.cnt_off, st);
  if (a.raw_i16) {
    epa::set_error("epa_sv_mvbs_fused_i16: int16 ingest is served by the default configuration only "
                   "(guard + masked range, skipna, left-closed bins, sorted pings, S %% 4 == 0, no "
                   "echo_range output, range grid within LDS)");
    return EPA_EUNSUPPORTED;
  }
  if (two_stage) {
    EPA_CHECK_ARG(a.sum_out && a.cnt_out,
                  "binned reduction: this shape (C*n_tbins=%lld, n_rbins=%d) needs the sum_out/"
                  "cnt_out workspaces", (long long)a.C * a.n_tbins, a.n_rbins);
    int rc = zero_partials<T>(a.sum_out, a.cnt_out, cells, st);
    if (rc) return rc;
  }
  int rc = launch_reduce<T, SRC, OP_MVBS>(a, pl, st);
  if (rc) return rc;
  if (two_stage) {
    const int grid = (int)((cells + epa::kBlock - 1) / epa::kBlock < 4096
                               ? (cells + epa::kBlock - 1) / epa::kBlock : 4096);
    hipLaunchKernelGGL(mvbs_finalize_kernel<T>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const T*)a.sum_out, (const uint32_t*)a.cnt_out, cells, (T)a.fill_value,
                       (T*)a.out);
    return epa::check_launch("mvbs_finalize_kernel");
  }
  return EPA_OK;
}

int check_bins(const char* fn, const int32_t* bin_start, int n_tbins, double range_bin, int n_rbins) {
  EPA_CHECK_ARG(bin_start != nullptr, "%s: bin_start is NULL", fn);
  EPA_CHECK_ARG(n_tbins > 0 && n_rbins > 0, "%s: n_tbins=%d n_rbins=%d must be positive", fn,
                n_tbins, n_rbins);
  EPA_CHECK_ARG(range_bin > 0, "%s: range_bin must be positive", fn);
  return EPA_OK;
}

}  // namespace

namespace {
// keys [min, max, min, max] -> doubles; untouched slots (min key ~0, max key 0) -> NaN
__global__ void decode_minmax_kernel(double* p) {
  const int i = threadIdx.x;
  const unsigned long long k = reinterpret_cast<unsigned long long*>(p)[i];
  const bool none = (i & 1) ? k == 0ull : k == ~0ull;
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  p[i] = none ? __builtin_nan("") : __longlong_as_double(b);
}
__global__ void init_minmax_kernel(unsigned long long* p) { p[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull; }

// The by-products of a calibration pass -- the maximum of echo_range (an order-preserving key the kernels raise with
// atomics) and {min key, NaN count} -- are set up by ONE single-lane launch before the pass and decoded by ONE after
// it (round 4: three memsets and two kernels before; every launch in front of a 9-ms kernel costs the API route its
// dispatch gap).  mode 0: the maximum only; 1: statistics -> {nanmin, nanmax, NaN count} (as_f32: the values the
// float32 echo_range array holds -- rounding is monotone); 2: the kernel that ran leaves none: NaN count -1.
__global__ void init_range_out_kernel(double* rmax, double* st) {
  if (rmax) *reinterpret_cast<unsigned long long*>(rmax) = 0ull;  // key 0 = nothing seen
  if (st) {
    unsigned long long* u = reinterpret_cast<unsigned long long*>(st);
    u[0] = ~0ull;  // min key: nothing seen
    u[1] = 0ull;   // NaN count
    u[2] = 0ull;
  }
}
__global__ void decode_range_out_kernel(double* rmax, double* st, int mode, int as_f32) {
  const unsigned long long km = *reinterpret_cast<unsigned long long*>(rmax);
  // inverse of the order-preserving key; key 0 = nothing seen -> NaN
  const unsigned long long bm = (km >> 63) ? (km & 0x7fffffffffffffffull) : ~km;
  const double mx = km == 0ull ? __builtin_nan("") : __longlong_as_double(bm);
  *rmax = mx;
  if (mode == 1) {
    const unsigned long long k = reinterpret_cast<unsigned long long*>(st)[0];
    const unsigned long long n = reinterpret_cast<unsigned long long*>(st)[1];
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double lo = k == ~0ull ? __builtin_nan("") : __longlong_as_double(b), hi = mx;
    if (as_f32) {
      lo = (double)(float)lo;
      hi = (double)(float)hi;
    }
    st[0] = lo;
    st[1] = hi;
    st[2] = (double)n;
  } else if (mode == 2) {
    st[0] = st[1] = __builtin_nan("");
    st[2] = -1.0;
  }
}

int init_range_outputs(double* rmax, double* st, hipStream_t stream) {
  if (!rmax && !st) return EPA_OK;
  hipLaunchKernelGGL(init_range_out_kernel, dim3(1), dim3(1), 0, stream, rmax, st);
  return epa::check_launch("init_range_out_kernel");
}
int decode_range_outputs(double* rmax, double* st, bool filled, int as_f32, hipStream_t stream) {
  hipLaunchKernelGGL(decode_range_out_kernel, dim3(1), dim3(1), 0, stream, rmax, st, st ? (filled ? 1 : 2) : 0, as_f32);
  return epa::check_launch("decode_range_out_kernel");
}
}  // namespace

static int fused_entry(const float* raw, const int16_t* raw_i16, const int32_t* n_valid,
                       const double* coef, int C, int P, int S, int cal_type, unsigned cal_flags,
                       const int32_t* bin_start, const int32_t* ping_perm, int n_tbins,
                       double range_bin, int n_rbins, unsigned bin_flags, double fill_value,
                       void* sv_out, void* range_out, void* mvbs_out, void* sum_out,
                       uint32_t* cnt_out, double* range_max_out, double* range_stats_out, int dtype,
                       epa_stream_t stream, const double* dscale = nullptr, const double* doffset = nullptr,
                       void* depth_out = nullptr);

extern "C" int epa_sv_mvbs_fused(const float* raw, const double* coef, int C, int P, int S,
                                 int cal_type, unsigned cal_flags, const int32_t* bin_start,
                                 const int32_t* ping_perm, int n_tbins, double range_bin,
                                 int n_rbins, unsigned bin_flags, double fill_value, void* sv_out,
                                 void* range_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out,
                                 double* range_max_out, double* range_stats_out, int dtype,
                                 epa_stream_t stream) {
  EPA_CHECK_ARG(raw != nullptr, "epa_sv_mvbs_fused: NULL array argument");
  EPA_CHECK_ARG(!range_stats_out || range_max_out, "epa_sv_mvbs_fused: range_stats_out needs range_max_out");
  return fused_entry(raw, nullptr, nullptr, coef, C, P, S, cal_type, cal_flags, bin_start, ping_perm,
                     n_tbins, range_bin, n_rbins, bin_flags, fill_value, sv_out, range_out, mvbs_out,
                     sum_out, cnt_out, range_max_out, range_stats_out, dtype, stream);
}

extern "C" int epa_sv_mvbs_fused_i16(const int16_t* raw, const int32_t* n_valid, const double* coef,
                                     int C, int P, int S, int cal_type, const int32_t* bin_start,
                                     int n_tbins, double range_bin, int n_rbins, double fill_value,
                                     void* sv_out, void* mvbs_out, void* sum_out, uint32_t* cnt_out,
                                     double* range_max_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(raw && n_valid, "epa_sv_mvbs_fused_i16: NULL array argument");
  return fused_entry(nullptr, raw, n_valid, coef, C, P, S, cal_type,
                     EPA_FLAG_GUARD_POS | EPA_FLAG_MASK_RANGE, bin_start, nullptr, n_tbins, range_bin,
                     n_rbins, EPA_BIN_SKIPNA, fill_value, sv_out, nullptr, mvbs_out, sum_out, cnt_out,
                     range_max_out, nullptr, dtype, stream);
}

extern "C" int epa_sv_mvbs_fused_depth(const float* raw, const double* coef, const double* depth_scale,
                                       const double* depth_offset, int C, int P, int S, int cal_type,
                                       const int32_t* bin_start, int n_tbins, double range_bin, int n_rbins,
                                       double fill_value, void* sv_out, void* depth_out, void* mvbs_out, void* sum_out,
                                       uint32_t* cnt_out, double* depth_stats_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(raw && depth_scale && depth_offset && depth_stats_out, "epa_sv_mvbs_fused_depth: NULL array argument");
  EPA_CHECK_ARG(!depth_out || sv_out, "epa_sv_mvbs_fused_depth: depth_out needs sv_out");
  // depth_stats_out[32] is the slot of the maximum's key while the kernel runs (decoded into [1]): 256 bytes from the
  // other keys -- every workgroup sends atomics to these words, and two busy words in one cache line cost the launch
  // 15 % (round 6: identical code, 2.37 against 2.05 ms per 0.8 G samples)
  return fused_entry(raw, nullptr, nullptr, coef, C, P, S, cal_type, EPA_FLAG_GUARD_POS | EPA_FLAG_MASK_RANGE, bin_start,
                     nullptr, n_tbins, range_bin, n_rbins, EPA_BIN_SKIPNA, fill_value, sv_out, nullptr, mvbs_out, sum_out,
                     cnt_out, depth_stats_out + 32, depth_stats_out, dtype, stream, depth_scale, depth_offset, depth_out);
}

static int fused_entry(const float* raw, const int16_t* raw_i16, const int32_t* n_valid,
                       const double* coef, int C, int P, int S, int cal_type, unsigned cal_flags,
                       const int32_t* bin_start, const int32_t* ping_perm, int n_tbins,
                       double range_bin, int n_rbins, unsigned bin_flags, double fill_value,
                       void* sv_out, void* range_out, void* mvbs_out, void* sum_out,
                       uint32_t* cnt_out, double* range_max_out, double* range_stats_out, int dtype,
                       epa_stream_t stream, const double* dscale, const double* doffset, void* depth_out) {
  EPA_CHECK_ARG(coef && mvbs_out, "epa_sv_mvbs_fused: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_sv_mvbs_fused: C=%d P=%d S=%d", C, P, S);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_mvbs_fused: bad cal_type");
  if (int rc = check_bins("epa_sv_mvbs_fused", bin_start, n_tbins, range_bin, n_rbins)) return rc;
  ReduceArgs a{};
  a.raw = raw; a.raw_i16 = raw_i16; a.n_valid = n_valid;
  a.dscale = dscale; a.doffset = doffset; a.depth_out = depth_out;
  a.coef = reinterpret_cast<const epa::CoefRow*>(coef);
  a.C = C; a.P = P; a.S = S;
  a.nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  a.cal_flags = cal_flags;
  a.bin_start = bin_start; a.ping_perm = ping_perm; a.n_tbins = n_tbins; a.ping_num = 0;
  a.bin_mode = BIN_PHYS; a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.n_rbins = n_rbins; a.range_sample_num = 1; a.bin_flags = bin_flags;
  a.fill_value = fill_value; a.noise_max = __builtin_nan("");
  a.sv_out = sv_out; a.range_out = range_out; a.out = mvbs_out; a.sum_out = sum_out; a.cnt_out = cnt_out;
  a.range_max_out = range_max_out;
  a.range_stats_out = range_stats_out;
  EPA_CHECK_ARG(dtype == EPA_F64 || dtype == EPA_F32, "epa_sv_mvbs_fused: bad dtype %d", dtype);
  if (int rc0 = init_range_outputs(range_max_out, range_stats_out, (hipStream_t)stream)) return rc0;
  const int rc = dtype == EPA_F64 ? run_mvbs<double, SRC_RAW>(a, (hipStream_t)stream)
                                  : run_mvbs<float, SRC_RAW>(a, (hipStream_t)stream);
  epa::note_range_stats_filled(rc == EPA_OK && range_stats_out && a.range_stats_filled ? 1 : 0);
  if (rc == EPA_OK && range_max_out) {
    return decode_range_outputs(range_max_out, range_stats_out, a.range_stats_filled != 0, dtype == EPA_F32 ? 1 : 0,
                                (hipStream_t)stream);
  }
  return rc;
}

extern "C" int epa_mvbs(const void* sv, const void* range, const double* coef, int C, int P, int S,
                        const int32_t* bin_start, const int32_t* ping_perm, int n_tbins,
                        double range_bin, int n_rbins, unsigned bin_flags, double fill_value,
                        void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                        epa_stream_t stream) {
  EPA_CHECK_ARG(sv && mvbs_out, "epa_mvbs: NULL array argument");
  EPA_CHECK_ARG(range || coef, "epa_mvbs: either range or coef must be given");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0, "epa_mvbs: C=%d P=%d S=%d", C, P, S);
  if (int rc = check_bins("epa_mvbs", bin_start, n_tbins, range_bin, n_rbins)) return rc;
  ReduceArgs a{};
  a.sv = sv; a.range = range; a.coef = reinterpret_cast<const epa::CoefRow*>(coef);
  a.C = C; a.P = P; a.S = S;
  a.bin_start = bin_start; a.ping_perm = ping_perm; a.n_tbins = n_tbins;
  a.bin_mode = BIN_PHYS; a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.n_rbins = n_rbins; a.range_sample_num = 1; a.bin_flags = bin_flags;
  a.fill_value = fill_value; a.noise_max = __builtin_nan("");
  a.out = mvbs_out; a.sum_out = sum_out; a.cnt_out = cnt_out;
  if (dtype == EPA_F64) return run_mvbs<double, SRC_SV>(a, (hipStream_t)stream);
  if (dtype == EPA_F32) return run_mvbs<float, SRC_SV>(a, (hipStream_t)stream);
  epa::set_error("epa_mvbs: bad dtype %d", dtype);
  return EPA_EINVAL;
}

extern "C" int epa_mvbs_finalize(const void* sum, const uint32_t* cnt, size_t n, double fill_value,
                                 void* out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(sum && cnt && out, "epa_mvbs_finalize: NULL array argument");
  if (n == 0) return EPA_OK;
  const size_t blocks = (n + epa::kBlock - 1) / epa::kBlock;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  if (dtype == EPA_F64)
    hipLaunchKernelGGL(mvbs_finalize_kernel<double>, dim3(grid), dim3(epa::kBlock), 0,
                       (hipStream_t)stream, (const double*)sum, cnt, n, fill_value, (double*)out);
  else if (dtype == EPA_F32)
    hipLaunchKernelGGL(mvbs_finalize_kernel<float>, dim3(grid), dim3(epa::kBlock), 0,
                       (hipStream_t)stream, (const float*)sum, cnt, n, (float)fill_value, (float*)out);
  else {
    epa::set_error("epa_mvbs_finalize: bad dtype %d", dtype);
    return EPA_EINVAL;
  }
  return epa::check_launch("mvbs_finalize_kernel");
}

namespace {
template <typename T>
int run_index(const void* sv, const void* range, int C, int P, int S, int ping_num, int rsn,
              void* mvbs_out, void* range_min_out, hipStream_t st) {
  const int Pb = (P + ping_num - 1) / ping_num, Sb = (S + rsn - 1) / rsn;
  ReduceArgs a{};
  a.sv = sv; a.C = C; a.P = P; a.S = S;
  a.bin_start = nullptr; a.n_tbins = Pb; a.ping_num = ping_num;
  a.bin_mode = BIN_INDEX; a.range_bin = 1.0; a.inv_range_bin = 1.0; a.n_rbins = Sb;
  a.range_sample_num = rsn; a.bin_flags = EPA_BIN_SKIPNA;
  a.fill_value = __builtin_nan(""); a.noise_max = __builtin_nan("");
  a.out = mvbs_out;
  Plan pl = make_plan<T>(C, P, S, Pb, Sb, al16(sv));
  pl.nparts = 1;  // a ping block is at most ping_num pings; keep the single-stage path
  a.nparts = 1;
  EPA_CHECK_ARG(pl.use_lds, "epa_mvbs_index: %d range blocks exceed the LDS budget", Sb);
  int rc = launch_reduce<T, SRC_SV, OP_MVBS>(a, pl, st);
  if (rc) return rc;
  if (range && range_min_out) {
    const long long cells = (long long)C * Pb * Sb;
    const long long blocks = (cells + 3) / 4;
    const int grid = (int)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(block_nanmin_kernel<T>, dim3(grid), dim3(epa::kBlock), 0, st,
                       (const T*)range, C, P, S, ping_num, rsn, Pb, Sb, (T*)range_min_out);
    return epa::check_launch("block_nanmin_kernel");
  }
  return EPA_OK;
}

template <typename T>
int run_noise(const void* sv, const void* range, const double* coef, const double* alpha2, int C,
              int P, int S, int ping_num, int rsn, int ping_phase, double noise_max, double* noise_out,
              double* edge_sum_out, uint32_t* edge_cnt_out, hipStream_t st) {
  const int Pb = (P + ping_phase + ping_num - 1) / ping_num, Sb = (S + rsn - 1) / rsn;
  ReduceArgs a{};
  a.sv = sv; a.range = range; a.coef = reinterpret_cast<const epa::CoefRow*>(coef);
  a.alpha2 = alpha2;
  a.C = C; a.P = P; a.S = S;
  a.bin_start = nullptr; a.n_tbins = Pb; a.ping_num = ping_num; a.ping_phase = ping_phase;
  a.bin_mode = BIN_INDEX; a.range_bin = 1.0; a.inv_range_bin = 1.0; a.n_rbins = Sb;
  a.range_sample_num = rsn; a.bin_flags = EPA_BIN_SKIPNA;
  a.fill_value = __builtin_nan(""); a.noise_max = noise_max;
  a.out = noise_out;
  a.edge_sum_out = edge_sum_out; a.edge_cnt_out = edge_cnt_out;
  Plan pl = make_plan<T>(C, P, S, Pb, Sb, al16(sv) && al16(range));
  pl.nparts = 1;
  a.nparts = 1;
  EPA_CHECK_ARG(pl.use_lds, "epa_noise_estimate: %d range blocks exceed the LDS budget", Sb);
  return launch_reduce<T, SRC_SV, OP_NOISE>(a, pl, st);
}
}  // namespace

extern "C" int epa_mvbs_index(const void* sv, const void* range, int C, int P, int S, int ping_num,
                              int range_sample_num, void* mvbs_out, void* range_min_out, int dtype,
                              epa_stream_t stream) {
  EPA_CHECK_ARG(sv && mvbs_out, "epa_mvbs_index: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0 && range_sample_num > 0,
                "epa_mvbs_index: sizes must be positive");
  if (dtype == EPA_F64)
    return run_index<double>(sv, range, C, P, S, ping_num, range_sample_num, mvbs_out, range_min_out,
                             (hipStream_t)stream);
  if (dtype == EPA_F32)
    return run_index<float>(sv, range, C, P, S, ping_num, range_sample_num, mvbs_out, range_min_out,
                            (hipStream_t)stream);
  epa::set_error("epa_mvbs_index: bad dtype %d", dtype);
  return EPA_EINVAL;
}

extern "C" int epa_noise_estimate(const void* sv, const void* range, const double* coef,
                                  const double* alpha2, int C, int P, int S, int ping_num,
                                  int range_sample_num, int ping_phase, double noise_max, double* noise_out,
                                  double* edge_sum_out, uint32_t* edge_cnt_out, int dtype,
                                  epa_stream_t stream) {
  EPA_CHECK_ARG(sv && alpha2 && noise_out, "epa_noise_estimate: NULL array argument");
  EPA_CHECK_ARG(range || coef, "epa_noise_estimate: either range or coef must be given");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0 && range_sample_num > 0,
                "epa_noise_estimate: sizes must be positive");
  EPA_CHECK_ARG(ping_phase >= 0 && ping_phase < ping_num, "epa_noise_estimate: ping_phase %d not in [0, %d)",
                ping_phase, ping_num);
  EPA_CHECK_ARG((edge_sum_out == nullptr) == (edge_cnt_out == nullptr),
                "epa_noise_estimate: edge_sum_out and edge_cnt_out come together");
  if (dtype == EPA_F64)
    return run_noise<double>(sv, range, coef, alpha2, C, P, S, ping_num, range_sample_num, ping_phase,
                             noise_max, noise_out, edge_sum_out, edge_cnt_out, (hipStream_t)stream);
  if (dtype == EPA_F32)
    return run_noise<float>(sv, range, coef, alpha2, C, P, S, ping_num, range_sample_num, ping_phase, noise_max,
                            noise_out, edge_sum_out, edge_cnt_out, (hipStream_t)stream);
  epa::set_error("epa_noise_estimate: bad dtype %d", dtype);
  return EPA_EINVAL;
}

// merged (sum, count) rows of noise blocks -> noise value per row (clean/api.py:402-422: mean -> dB -> min over
// the range blocks -> clamp); the counts come as doubles because they travel through an all-reduce with the sums
namespace {
__global__ __launch_bounds__(epa::kBlock) void noise_rows_f64_kernel(const double* __restrict__ sum,
                                                                     const double* __restrict__ cnt, int n_rbins,
                                                                     double noise_max, double* __restrict__ out) {
  __shared__ double scratch[8];
  const size_t base = (size_t)blockIdx.x * n_rbins;
  double best = __builtin_inf();
  bool any = false;
  for (int i = threadIdx.x; i < n_rbins; i += epa::kBlock) {
    const double n = cnt[base + i];
    if (n > 0.0) {
      const double db = 10.0 * ::log10(sum[base + i] / n);
      if (db == db) {
        best = fmin(best, db);
        any = true;
      }
    }
  }
  const double m = block_nanmin<double>(best, any, scratch);
  if (threadIdx.x == 0) {
    double r = m;
    if (noise_max == noise_max) r = (r < noise_max) ? r : noise_max;
    out[blockIdx.x] = r;
  }
}
}  // namespace

extern "C" int epa_noise_finalize(const double* sum, const double* cnt, int rows, int n_rblocks, double noise_max,
                                  double* noise_out, epa_stream_t stream) {
  EPA_CHECK_ARG(sum && cnt && noise_out, "epa_noise_finalize: NULL array argument");
  EPA_CHECK_ARG(rows > 0 && n_rblocks > 0, "epa_noise_finalize: sizes must be positive");
  hipLaunchKernelGGL(noise_rows_f64_kernel, dim3(rows), dim3(epa::kBlock), 0, (hipStream_t)stream, sum, cnt,
                     n_rblocks, noise_max, noise_out);
  return epa::check_launch("noise_rows_f64_kernel");
}

// ---- fused chain: compute_Sv + estimate_background_noise, then remove_background_noise + compute_MVBS ----
namespace {
template <typename T>
int run_sv_noise(const float* raw, const double* coef, const double* alpha2, int C, int P, int S,
                 double nspread, unsigned cal_flags, int ping_num, int rsn, int ping_phase, double noise_max,
                 void* sv_out, void* range_out, double* noise_out, double* edge_sum_out, uint32_t* edge_cnt_out,
                 double* range_max_out, double* range_stats_out, int* stats_filled, hipStream_t st) {
  const int Pb = (P + ping_phase + ping_num - 1) / ping_num, Sb = (S + rsn - 1) / rsn;
  ReduceArgs a{};
  a.raw = raw; a.coef = reinterpret_cast<const epa::CoefRow*>(coef); a.alpha2 = alpha2;
  a.C = C; a.P = P; a.S = S; a.nspread = nspread; a.cal_flags = cal_flags;
  a.bin_start = nullptr; a.n_tbins = Pb; a.ping_num = ping_num; a.ping_phase = ping_phase;
  a.bin_mode = BIN_INDEX; a.range_bin = 1.0; a.inv_range_bin = 1.0; a.n_rbins = Sb;
  a.range_sample_num = rsn; a.bin_flags = EPA_BIN_SKIPNA;
  a.fill_value = __builtin_nan(""); a.noise_max = noise_max;
  a.sv_out = sv_out; a.range_out = range_out; a.out = noise_out; a.range_max_out = range_max_out;
  a.edge_sum_out = edge_sum_out; a.edge_cnt_out = edge_cnt_out;
  Plan pl = make_plan<T>(C, P, S, Pb, Sb, al16(raw) && al16(sv_out) && al16(range_out));
  pl.nparts = 1;
  a.nparts = 1;
  EPA_CHECK_ARG(pl.use_lds, "epa_sv_noise_fused: %d range blocks exceed the LDS budget", Sb);
  if (pl.vec == 4 && !range_out && cal_flags == (EPA_FLAG_GUARD_POS | EPA_FLAG_MASK_RANGE) &&
      !getenv("EPA_NO_FAST_PATH")) {
    *stats_filled = range_stats_out != nullptr;
    return epa_chain_fast_pass1(raw, coef, alpha2, C, P, S, nspread, ping_num, rsn, ping_phase, noise_max, sv_out,
                                noise_out, edge_sum_out, edge_cnt_out,
                                reinterpret_cast<unsigned long long*>(range_max_out),
                                reinterpret_cast<unsigned long long*>(range_stats_out),
                                sizeof(T) == 8 ? EPA_F64 : EPA_F32, st);
  }
  return launch_reduce<T, SRC_RAW, OP_NOISE>(a, pl, st);
}
}  // namespace

extern "C" int epa_sv_noise_fused(const float* raw, const double* coef, const double* alpha2, int C, int P,
                                  int S, int cal_type, unsigned cal_flags, int ping_num,
                                  int range_sample_num, int ping_phase, double noise_max, void* sv_out,
                                  void* range_out, double* noise_out, double* edge_sum_out, uint32_t* edge_cnt_out,
                                  double* range_max_out, double* range_stats_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(raw && coef && alpha2 && noise_out, "epa_sv_noise_fused: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0 && range_sample_num > 0,
                "epa_sv_noise_fused: sizes must be positive");
  EPA_CHECK_ARG(ping_phase >= 0 && ping_phase < ping_num, "epa_sv_noise_fused: ping_phase %d not in [0, %d)",
                ping_phase, ping_num);
  EPA_CHECK_ARG((edge_sum_out == nullptr) == (edge_cnt_out == nullptr),
                "epa_sv_noise_fused: edge_sum_out and edge_cnt_out come together");
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_noise_fused: bad cal_type");
  const double nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  EPA_CHECK_ARG(dtype == EPA_F64 || dtype == EPA_F32, "epa_sv_noise_fused: bad dtype %d", dtype);
  EPA_CHECK_ARG(!range_stats_out || range_max_out, "epa_sv_noise_fused: range_stats_out needs range_max_out");
  if (int rc0 = init_range_outputs(range_max_out, range_stats_out, (hipStream_t)stream)) return rc0;
  int filled = 0;
  const int rc = dtype == EPA_F64
      ? run_sv_noise<double>(raw, coef, alpha2, C, P, S, nspread, cal_flags, ping_num, range_sample_num, ping_phase,
                             noise_max, sv_out, range_out, noise_out, edge_sum_out, edge_cnt_out, range_max_out,
                             range_stats_out, &filled, (hipStream_t)stream)
      : run_sv_noise<float>(raw, coef, alpha2, C, P, S, nspread, cal_flags, ping_num, range_sample_num, ping_phase,
                            noise_max, sv_out, range_out, noise_out, edge_sum_out, edge_cnt_out, range_max_out,
                            range_stats_out, &filled, (hipStream_t)stream);
  epa::note_range_stats_filled(rc == EPA_OK && range_stats_out && filled ? 1 : 0);
  if (rc == EPA_OK && range_max_out) {
    // (the fast kernel tracks the values as stored: no rounding left to do)
    return decode_range_outputs(range_max_out, range_stats_out, filled != 0, 0, (hipStream_t)stream);
  }
  return rc;
}

extern "C" int epa_denoise_mvbs(const void* sv, const void* range, const double* coef, const double* alpha2,
                                const double* noise, int C, int P, int S, int ping_num,
                                double snr_threshold, const int32_t* bin_start, const int32_t* ping_perm,
                                int n_tbins, double range_bin, int n_rbins, unsigned bin_flags,
                                double fill_value, void* sv_noise_out, void* sv_corrected_out,
                                void* mvbs_out, void* sum_out, uint32_t* cnt_out, int dtype,
                                epa_stream_t stream) {
  EPA_CHECK_ARG(sv && alpha2 && noise && mvbs_out, "epa_denoise_mvbs: NULL array argument");
  EPA_CHECK_ARG(range || coef, "epa_denoise_mvbs: either range or coef must be given");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0, "epa_denoise_mvbs: sizes must be positive");
  if (int rc = check_bins("epa_denoise_mvbs", bin_start, n_tbins, range_bin, n_rbins)) return rc;
  ReduceArgs a{};
  a.sv = sv; a.range = range; a.coef = reinterpret_cast<const epa::CoefRow*>(coef);
  a.alpha2 = alpha2; a.noise = noise; a.noise_ping_num = ping_num; a.n_pblocks = (P + ping_num - 1) / ping_num;
  a.snr = snr_threshold;
  a.C = C; a.P = P; a.S = S;
  a.bin_start = bin_start; a.ping_perm = ping_perm; a.n_tbins = n_tbins;
  a.bin_mode = BIN_PHYS; a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.n_rbins = n_rbins; a.range_sample_num = 1; a.bin_flags = bin_flags;
  a.fill_value = fill_value; a.noise_max = __builtin_nan("");
  a.sv_out = sv_corrected_out; a.sv_noise_out = sv_noise_out;
  a.out = mvbs_out; a.sum_out = sum_out; a.cnt_out = cnt_out;
  if (dtype == EPA_F64) return run_mvbs<double, SRC_SV_DENOISE>(a, (hipStream_t)stream);
  if (dtype == EPA_F32) return run_mvbs<float, SRC_SV_DENOISE>(a, (hipStream_t)stream);
  epa::set_error("epa_denoise_mvbs: bad dtype %d", dtype);
  return EPA_EINVAL;
}

extern "C" int epa_sv_denoise_mvbs(const float* raw, const double* coef, const double* alpha2,
                                   const double* noise, int C, int P, int S, int cal_type, unsigned cal_flags,
                                   int ping_num, int ping_phase, double snr_threshold, const int32_t* bin_start,
                                   const int32_t* ping_perm, int n_tbins, double range_bin, int n_rbins,
                                   unsigned bin_flags, double fill_value, void* sv_noise_out,
                                   void* sv_corrected_out, void* range_out, void* mvbs_out, void* sum_out,
                                   uint32_t* cnt_out, double* minmax_out, int dtype, epa_stream_t stream) {
  EPA_CHECK_ARG(raw && coef && alpha2 && noise && mvbs_out, "epa_sv_denoise_mvbs: NULL array argument");
  EPA_CHECK_ARG(C > 0 && P > 0 && S > 0 && ping_num > 0, "epa_sv_denoise_mvbs: sizes must be positive");
  EPA_CHECK_ARG(ping_phase >= 0 && ping_phase < ping_num, "epa_sv_denoise_mvbs: ping_phase %d not in [0, %d)",
                ping_phase, ping_num);
  EPA_CHECK_ARG(cal_type == EPA_CAL_SV || cal_type == EPA_CAL_TS, "epa_sv_denoise_mvbs: bad cal_type");
  if (int rc = check_bins("epa_sv_denoise_mvbs", bin_start, n_tbins, range_bin, n_rbins)) return rc;
  ReduceArgs a{};
  a.raw = raw; a.coef = reinterpret_cast<const epa::CoefRow*>(coef);
  a.alpha2 = alpha2; a.noise = noise; a.noise_ping_num = ping_num; a.noise_phase = ping_phase;
  a.n_pblocks = (P + ping_phase + ping_num - 1) / ping_num;
  a.snr = snr_threshold;
  a.C = C; a.P = P; a.S = S;
  a.nspread = cal_type == EPA_CAL_SV ? 20.0 : 40.0;
  a.cal_flags = cal_flags;
  a.bin_start = bin_start; a.ping_perm = ping_perm; a.n_tbins = n_tbins;
  a.bin_mode = BIN_PHYS; a.range_bin = range_bin; a.inv_range_bin = 1.0 / range_bin;
  a.n_rbins = n_rbins; a.range_sample_num = 1; a.bin_flags = bin_flags;
  a.fill_value = fill_value; a.noise_max = __builtin_nan("");
  a.sv_out = sv_corrected_out; a.sv_noise_out = sv_noise_out; a.range_out = range_out;
  a.out = mvbs_out; a.sum_out = sum_out; a.cnt_out = cnt_out;
  a.mm_keys = reinterpret_cast<unsigned long long*>(minmax_out);
  EPA_CHECK_ARG(dtype == EPA_F64 || dtype == EPA_F32, "epa_sv_denoise_mvbs: bad dtype %d", dtype);
  if (minmax_out) {
    hipLaunchKernelGGL(init_minmax_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, a.mm_keys);
    if (int rc = epa::check_launch("init_minmax_kernel")) return rc;
  }
  const int rc = dtype == EPA_F64 ? run_mvbs<double, SRC_RAW_DENOISE>(a, (hipStream_t)stream)
                                  : run_mvbs<float, SRC_RAW_DENOISE>(a, (hipStream_t)stream);
  if (rc == EPA_OK && minmax_out) {
    hipLaunchKernelGGL(decode_minmax_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, minmax_out);
    return epa::check_launch("decode_minmax_kernel");
  }
  return rc;
}
