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
#include "fast_math.h"
#include "log_tab_data.h"
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

// ---- the same pass as ONE-PIECE workgroups (round 5) ------------------------------------------------------------------
// sv_power_kernel above keeps a workgroup on one range chunk while it strides over the rows (the cached column logs
// then never change).  Measured on the traffic mix alone (scripts/probes/hbm_walk5_probe.hip, profiles/r05_walk_probes.txt,
// every box of the pool): a workgroup that LOOPS over strided pieces streams 4 B read + 8 B written per sample at
// 5.0-5.4 TB/s, a workgroup that takes ONE 1024-sample piece and ends at 6.2 TB/s.  Nothing in compute_Sv ties a
// workgroup to more than one piece -- only the cached n log10(s - d) did; here it is evaluated per sample instead
// (table-driven, ~22 instructions, from the 2-KB table in device memory: a workgroup that lives for four samples per lane
// cannot build it in LDS), and a workgroup is (row, chunk), chunk fastest, the XCDs walking contiguous eighths.
// Same arithmetic otherwise (cal_power_sample); the logarithm is fast_log10_lean's (4e-16) instead of ocml's: Sv moves by
// less than 1e-14 dB against the kernel above.  {min, max, NaN count} of the echo_range go to three ordered keys with
// atomics a wavefront issues only when its value would change the key (almost never after the first workgroups).
__device__ __forceinline__ unsigned long long ordered_key(double v) {
  const unsigned long long b = __double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

constexpr int kStatSlots = 4096;  // {max key, min key, NaN count} per slot; a wavefront uses slot (row mod kStatSlots)
constexpr unsigned long long kDKeyNone = ~0ull;

// n log10(s - d) of a sample: from the channel's table when every row of the channel has the same d (EK files: d is the
// TVG correction in samples, a constant of the channel), else evaluated here.
template <typename T>
__device__ __forceinline__ T piece_nlog(double sd, T nspread) {
  if (sizeof(T) == 4) return nspread * epa::M<T>::log10((T)sd);
  if (__builtin_expect(sd >= 1.0, 1)) return nspread * (T)epa::fast_log10_lean(sd, epa::kLogTabGlobal);
  return nspread * epa::log10_noinline<T>((T)sd);
}

// {min, max} of the rows' d per channel as ordered keys (a NaN d, or rows that differ in ra -- the table would not depend
// on it, but a channel whose sample interval changes is not the plain case -- poison the channel: keys {0, ~0})
__global__ __launch_bounds__(epa::kBlock) void d_span_kernel(const epa::CoefRow* __restrict__ coef, int P,
                                                             unsigned long long* __restrict__ dkeys) {
  const int c = blockIdx.y;
  double lo = __builtin_inf(), hi = -__builtin_inf();
  bool bad = false;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const double d = coef[(size_t)c * P + p].d;
    bad |= !(d == d);
    lo = fmin(lo, d);
    hi = fmax(hi, d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = fmin(lo, __shfl_down(lo, o, 64));
    hi = fmax(hi, __shfl_down(hi, o, 64));
  }
  const bool any_bad = __ballot(bad) != 0ull;
  if ((threadIdx.x & 63) == 0) {
    if (any_bad) {
      atomicMin(dkeys + 2 * c, 0ull);
      atomicMax(dkeys + 2 * c + 1, kDKeyNone);
    } else if (lo <= hi) {
      atomicMin(dkeys + 2 * c, ordered_key(lo));
      atomicMax(dkeys + 2 * c + 1, ordered_key(hi));
    }
  }
}
__global__ void d_keys_init_kernel(unsigned long long* dkeys, int C) {
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    dkeys[2 * i] = kDKeyNone;
    dkeys[2 * i + 1] = 0ull;
  }
}
// table[c][s] = nspread * log10(s - d_c), the value ColumnLog::update caches (same out-of-line ocml log10: the same bits)
template <typename T>
__global__ __launch_bounds__(epa::kBlock) void nl_table_kernel(const unsigned long long* __restrict__ dkeys, int S,
                                                               T nspread, T* __restrict__ table) {
  const int c = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long klo = dkeys[2 * c], khi = dkeys[2 * c + 1];
  if (klo != khi || s >= S) return;  // (not one d for the channel: the table is not used)
  const double d = __longlong_as_double((long long)((klo >> 63) ? (klo & 0x7fffffffffffffffull) : ~klo));
  table[(size_t)c * S + s] = nspread * epa::log10_noinline<T>((T)((double)s - d));
}

template <typename T, bool RANGE, bool STATS>
__global__ __launch_bounds__(epa::kBlock) void sv_power_piece_kernel(const float* __restrict__ raw,
                                                                     const epa::CoefRow* __restrict__ coef, int S, int P,
                                                                     int chunks_per_row, T nspread, unsigned flags,
                                                                     T* __restrict__ out, T* __restrict__ range_out,
                                                                     unsigned long long* __restrict__ keys,
                                                                     const unsigned long long* __restrict__ dkeys,
                                                                     const T* __restrict__ table, int xcd_map) {
  using LM = epa::LaneMap<T>;
  constexpr int NSEG = LM::NSEG, LEN = LM::LEN;
  const bool guard = flags & EPA_FLAG_GUARD_POS;
  const bool mask_range = flags & EPA_FLAG_MASK_RANGE;
  const int b = xcd_map ? epa::xcd_contiguous((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int row = b / chunks_per_row, chunk = b - row * chunks_per_row;
  const epa::RowK<T> rk(coef[row]);
  bool tabled = false;  // (scalar)
  const T* __restrict__ trow = nullptr;
  if (sizeof(T) == 8 && table) {
    const int c = row / P;
    tabled = dkeys[2 * c] == dkeys[2 * c + 1];
    trow = table + (size_t)c * S;
  }
  unsigned long long* kslot = STATS ? keys + 3 * (row & (kStatSlots - 1)) : nullptr;
  T rall[STATS ? NSEG * LEN : 1];  // STATS: every range value of the lane (for the wavefronts that need them one by one)
  bool lane_nan = false;           // STATS: one of the lane's range values is NaN
  // STATS: the raw samples next to the wavefront's 256, requested first (scalar loads of lines the neighbours stream
  // anyway): is there a valid sample before / after this wavefront in the row?  (see the end)
  const int w_s0 = chunk * 1024 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 256;
  float before = __builtin_nanf(""), after = __builtin_nanf("");
  if (STATS && mask_range) {
    if (w_s0 > 0) before = raw[(size_t)row * S + w_s0 - 1];
    if (w_s0 + 256 < S) after = raw[(size_t)row * S + w_s0 + 256];
  }
#pragma unroll
  for (int g = 0; g < NSEG; ++g) {
    const int s0 = LM::first(chunk * 1024, g);
    if (s0 >= S) {
      if (STATS) {
#pragma unroll
        for (int j = 0; j < LEN; ++j) rall[g * LEN + j] = epa::M<T>::nan();
      }
      continue;
    }
    const size_t off = (size_t)row * S + s0;
    epa::RawVec<LEN> in;
    in.load(raw + off);
    T nl[LEN];
    if (tabled) epa::load_vec<T, LEN>(trow + s0, nl);
    T o[LEN], rg[LEN];
#pragma unroll
    for (int j = 0; j < LEN; ++j) {
      const double r = rk.range(s0 + j);
      if (!tabled) nl[j] = piece_nlog<T>((double)(s0 + j) - rk.d, nspread);
      o[j] = epa::cal_power_sample<T>(in.v[j], s0 + j, rk, nspread, nl[j], guard, r);
      if (RANGE || STATS) rg[j] = (mask_range && !(in.v[j] == in.v[j])) ? epa::M<T>::nan() : (T)r;
      if (STATS) {
        rall[g * LEN + j] = rg[j];
        lane_nan |= !(rg[j] == rg[j]);
      }
    }
    epa::store_vec<T, LEN>(out + off, o);
    if (RANGE) epa::store_vec<T, LEN>(range_out + off, rg);
  }
  if (STATS) {
    // {min, max, NaN count} of the echo_range.  One of kStatSlots slots per row (a single key sees an atomic from
    // nearly every row of a file whose sound speed drifts: 30 ms of serialised atomics, measured).  A workgroup lives
    // for four samples per lane: a per-sample min / max / NaN ballot and a wavefront reduction cost it a third of its
    // time (round 5: +35 %; with the reduction skipped behind a read of the slot's keys still 1.5-2 ms per 4 G samples --
    // the read waits for a cache line the others' atomics keep busy; two atomics per wavefront without a return value:
    // 0.8 ms, round 6).  What is left to skip is the atomics themselves.  The range grows with the sample number
    // (ra, rb > 0), so a COMPLETE wavefront without a NaN has its smallest value in lane 0's first sample and its
    // largest in lane 63's last (one ballot, two v_readlane) -- and it can hold the ROW's smallest value only if no valid
    // sample precedes it in the row, the largest only if none follows: the two neighbouring raw samples, requested
    // at the start, say so (with the range masked by NaN inputs; else the row's ends are its first and last sample).
    // By induction over the wavefronts of a row some wavefront that sends holds a value <= (>=) any valid one.  A row
    // of full-length pings then costs two atomics, not two per wavefront.  Every other wavefront (a NaN inside: the
    // padded tail of one ping in ten; the end of a row; a row that does not grow) goes through its values one by one
    // and always sends.
    unsigned long long* k = kslot;
    const unsigned long long nanlanes = __ballot(lane_nan);
    const bool whole = w_s0 + 256 <= S;  // (scalar) no idle lane
    double lo, hi;
    unsigned nn = 0u;
    bool send_lo = true, send_hi = true;  // (scalar)
    if (nanlanes == 0ull && whole && rk.ra > 0.0 && rk.rb > 0.0) {
      const double first = (double)rall[0], last = (double)rall[NSEG * LEN - 1];
      lo = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(first), 0), __builtin_amdgcn_readlane(__double2loint(first), 0));
      hi = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(last), 63), __builtin_amdgcn_readlane(__double2loint(last), 63));
      send_lo = mask_range ? !(before == before) : w_s0 == 0;        // (NaN, or the row starts here)
      send_hi = mask_range ? !(after == after) : w_s0 + 256 >= S;    // (NaN, or the row ends here)
    } else {
      lo = __builtin_inf();
      hi = -__builtin_inf();
#pragma unroll
      for (int i = 0; i < NSEG * LEN; ++i) {
        const double x = (double)rall[i];
        lo = fmin(lo, x);  // fmin / fmax ignore a NaN operand
        hi = fmax(hi, x);
        const int s_i = LM::first(chunk * 1024, i / LEN) + i % LEN;  // (an idle lane's placeholder is not a value)
        nn += (unsigned)__builtin_popcountll(__ballot(s_i < S && !(x == x)));
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_down(lo, o, 64));
        hi = fmax(hi, __shfl_down(hi, o, 64));
      }
    }
    if ((threadIdx.x & 63) == 0) {
      if (send_hi && hi > -__builtin_inf()) atomicMax(k + 0, ordered_key(hi));
      if (send_lo && lo < __builtin_inf()) atomicMin(k + 1, ordered_key(lo));
      if (nn > 0u) atomicAdd(k + 2, (unsigned long long)nn);
    }
  }
}

__global__ __launch_bounds__(epa::kBlock) void piece_keys_init_kernel(unsigned long long* keys) {
  for (int i = threadIdx.x; i < kStatSlots; i += blockDim.x) {
    keys[3 * i + 0] = 0ull;   // max key: nothing seen
    keys[3 * i + 1] = ~0ull;  // min key: nothing seen
    keys[3 * i + 2] = 0ull;   // NaN count
  }
}
__global__ __launch_bounds__(epa::kBlock) void piece_keys_decode_kernel(const unsigned long long* keys, double* stats) {
  __shared__ unsigned long long smax[4], smin[4], scnt[4];
  unsigned long long mx = 0ull, mn = ~0ull, cnt = 0ull;
  for (int i = threadIdx.x; i < kStatSlots; i += blockDim.x) {
    mx = keys[3 * i] > mx ? keys[3 * i] : mx;
    mn = keys[3 * i + 1] < mn ? keys[3 * i + 1] : mn;
    cnt += keys[3 * i + 2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_down(mx, o, 64), b2 = __shfl_down(mn, o, 64), c2 = __shfl_down(cnt, o, 64);
    mx = a > mx ? a : mx;
    mn = b2 < mn ? b2 : mn;
    cnt += c2;
  }
  if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = mx; smin[threadIdx.x >> 6] = mn; scnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { mx = smax[w] > mx ? smax[w] : mx; mn = smin[w] < mn ? smin[w] : mn; }
    cnt = scnt[0] + scnt[1] + scnt[2] + scnt[3];
    auto unkey = [](unsigned long long k) {
      return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
    };
    stats[0] = mn == ~0ull ? __builtin_nan("") : unkey(mn);
    stats[1] = mx == 0ull ? __builtin_nan("") : unkey(mx);
    stats[2] = (double)cnt;
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
  // one-piece workgroups (EPA_K1_PIECES=0: the strided-rows kernel; development knob)
  const char* pe = getenv("EPA_K1_PIECES");  // (read per call: a test compares the two kernels in one process)
  const bool pieces_off = pe && pe[0] == '0';
  // (fp64 with the echo_range array written as well -- 20 B per sample -- or with its statistics as a by-product are the
  //  variants the strided-rows kernel serves faster: 13.6-13.9 against 14.2-15.4 ms and 8.7-9.2 against 10.3-10.7 ms per
  //  4 G samples; a workgroup that lives for four samples per lane pays for the statistics every time.  fp32 and plain
  //  fp64 Sv / TS take the pieces: profiles/r05_k1_pieces_ab.txt)
  // (EPA_K1_F64_STATS_PIECES=1: the pieces for fp64 with the statistics, too -- round 6 measured them again with the
  //  statistics' cost cut (no key reads, atomics from the row-end wavefronts only): 9.5-10.7 against 8.9-9.3 ms per 4 G
  //  samples for the strided rows, which keep the case; fp32 gained, 6.7 -> 6.1-6.3.  profiles/r06_k1_pieces_ab.txt)
  const char* pf = getenv("EPA_K1_F64_STATS_PIECES");
  const bool f64_stats_pieces = pf && pf[0] == '1';
  if (vec && !pieces_off && rows * chunks_per_row < (1ll << 31) &&
      !(sizeof(T) == 8 && (range_out || (stats_out && !f64_stats_pieces)))) {
    const dim3 pgrid((unsigned)(rows * chunks_per_row));
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(part);
    const int xm = epa::xcd_map_enabled() ? 1 : 0;
    // fp64: the per-channel table of n log10(s - d) and the {min, max} keys of d, in stream-ordered scratch memory
    // (freed behind the kernel).  No memory pool on the device: the strided-rows kernel serves the call.
    void* scratch = nullptr;
    T* table = nullptr;
    unsigned long long* dkeys = nullptr;
    if (sizeof(T) == 8) {
      const size_t kbytes = ((size_t)2 * C * sizeof(unsigned long long) + 255) & ~(size_t)255;
      if (hipMallocAsync(&scratch, kbytes + (size_t)C * S * sizeof(T), st) != hipSuccess) {
        (void)hipGetLastError();
        scratch = nullptr;
      }
      if (scratch) {
        dkeys = reinterpret_cast<unsigned long long*>(scratch);
        table = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(scratch) + kbytes);
        hipLaunchKernelGGL(d_keys_init_kernel, dim3(1), dim3(64), 0, st, dkeys, C);
        if (int rc = epa::check_launch("d_keys_init_kernel")) { (void)hipFreeAsync(scratch, st); return rc; }
        hipLaunchKernelGGL(d_span_kernel, dim3((unsigned)((P + 4 * epa::kBlock - 1) / (4 * epa::kBlock) < 256 ? (P + 4 * epa::kBlock - 1) / (4 * epa::kBlock) : 256), (unsigned)C),
                           dim3(epa::kBlock), 0, st, cf, P, dkeys);
        if (int rc = epa::check_launch("d_span_kernel")) { (void)hipFreeAsync(scratch, st); return rc; }
        hipLaunchKernelGGL((nl_table_kernel<T>), dim3((unsigned)((S + epa::kBlock - 1) / epa::kBlock), (unsigned)C),
                           dim3(epa::kBlock), 0, st, dkeys, S, nspread, table);
        if (int rc = epa::check_launch("nl_table_kernel")) { (void)hipFreeAsync(scratch, st); return rc; }
      }
    }
    if (sizeof(T) == 4 || scratch) {
      if (stats_out) {
        hipLaunchKernelGGL(piece_keys_init_kernel, dim3(1), dim3(epa::kBlock), 0, st, keys);
        if (int rc = epa::check_launch("piece_keys_init_kernel")) {
          if (scratch) (void)hipFreeAsync(scratch, st);
          return rc;
        }
      }
#define EPA_PIECE(R, ST)                                                                                                 \
  hipLaunchKernelGGL((sv_power_piece_kernel<T, R, ST>), pgrid, dim3(epa::kBlock), 0, st, raw, cf, S, P, chunks_per_row, \
                     nspread, flags, (T*)out, (T*)range_out, keys, dkeys, table, xm)
      if (stats_out) { if (range_out) EPA_PIECE(true, true); else EPA_PIECE(false, true); }
      else { if (range_out) EPA_PIECE(true, false); else EPA_PIECE(false, false); }
#undef EPA_PIECE
      const int rc = epa::check_launch("sv_power_piece_kernel");
      if (scratch) (void)hipFreeAsync(scratch, st);
      if (rc) return rc;
      if (stats_out) {
        hipLaunchKernelGGL(piece_keys_decode_kernel, dim3(1), dim3(epa::kBlock), 0, st, keys, stats_out);
        return epa::check_launch("piece_keys_decode_kernel");
      }
      return EPA_OK;
    }
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
  {  // one-piece workgroups (reduce_util.hip) where the shape takes 16-byte accesses
    const int rc = epa_rows_piece_launch((flags & EPA_FLAG_MASK_RANGE) ? raw : nullptr, nullptr, coef, nullptr, nullptr, rows,
                                         S, range_out, out_dtype, nullptr, nullptr, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
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
