// Round 6 (hbm_walk7): the walks of hbm_walk2 / hbm_walk3 side by side on the SAME buffers, over several allocation sets
// of one process -- which walk keeps its rate where the driver's placement of the output array costs the bin-owning
// walk 5-10 % (profiles/r06_stream_pairs.txt)?  Walk A = the shipped kernels' (a workgroup owns 20 rows of 1024 columns,
// prefetch a row ahead); rows G = 20 / 4 / 2: a workgroup takes G consecutive FULL rows (+ LDS bins, + global atomics of
// the sums / and counts); piece: one 1024-sample piece per workgroup.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o scripts/probes/bin/hbm_walk7_probe scripts/probes/hbm_walk7_probe.hip
// (below: hbm_walk3_probe's kernels)  Round 5, second step: can a streaming walk that does NOT tie a workgroup to a time bin's 20 rows carry the bins?
// (hbm_walk2_probe: every walk in which a workgroup owns 20 rows streams the 4 B read + 8 B written per sample at 5.1-5.4
//  TB/s, a walk of one-piece workgroups at 6.0-6.2.)  Here a workgroup takes G consecutive FULL rows (contiguous memory),
// adds every sample to one of NR range bins in LDS and hands its bins to the time bin's global accumulators with atomics
// -- a mock of a fused kernel that owns no bin.  MODE 0: the walk alone; 1: + LDS bins; 2: + global f64 atomics of the sums;
// 3: + u32 atomics of the counts.  PIECE: a workgroup = one 1024-sample piece of one row (G = 1/4).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o echopype_amd/lib/hbm_walk3_probe scripts/probes/hbm_walk3_probe.hip
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NR = 787, RB = 20;  // range bins per time bin, rows per time bin

template <int MODE>
__device__ __forceinline__ void pair(const float* ip, double* op, int s, double* lsum, unsigned* lcnt) {
  const f2 v = *reinterpret_cast<const f2*>(ip + s);
  d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
  __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + s));
  if (MODE >= 1) {
    const int b0 = (s * 787) >> 12, b1 = ((s + 1) * 787) >> 12;  // 5.2 samples per bin
    if (b0 == b1) {
      unsafeAtomicAdd(lsum + b0, o.x + o.y);
      atomicAdd(lcnt + b0, 2u);
    } else {
      unsafeAtomicAdd(lsum + b0, o.x);
      unsafeAtomicAdd(lsum + b1, o.y);
      atomicAdd(lcnt + b0, 1u);
      atomicAdd(lcnt + b1, 1u);
    }
  }
}

template <int MODE, int PIECE>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ in, double* __restrict__ out, int S, int G,
                                            double* __restrict__ acc, unsigned* __restrict__ cnt, int xcd) {
  __shared__ double lsum[NR];
  __shared__ unsigned lcnt[NR];
  int b = blockIdx.x;
  if (xcd) {
    const int n = gridDim.x, per = n >> 3;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  if (MODE >= 1) {
    for (int i = threadIdx.x; i < NR; i += 256) {
      lsum[i] = 0.0;
      lcnt[i] = 0u;
    }
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sA = wave * 256 + 2 * lane, sB = sA + 128;
  size_t row0;
  if (PIECE) {  // one 1024-sample piece of one row
    const int nch = S / 1024;
    row0 = (size_t)(b / nch);
    const int c = b - (int)row0 * nch;
    pair<MODE>(in + row0 * S, out + row0 * S, c * 1024 + sA, lsum, lcnt);
    pair<MODE>(in + row0 * S, out + row0 * S, c * 1024 + sB, lsum, lcnt);
  } else {
    row0 = (size_t)b * G;
    for (int r = 0; r < G; ++r)
      for (int c = 0; c < S; c += 1024) {
        pair<MODE>(in + (row0 + r) * S, out + (row0 + r) * S, c + sA, lsum, lcnt);
        pair<MODE>(in + (row0 + r) * S, out + (row0 + r) * S, c + sB, lsum, lcnt);
      }
  }
  if (MODE >= 2) {
    __syncthreads();
    const size_t tb = row0 / RB;
    for (int i = threadIdx.x; i < NR; i += 256) {
      if (lcnt[i] != 0u) {
        unsafeAtomicAdd(acc + tb * NR + i, lsum[i]);
        if (MODE >= 3) atomicAdd(cnt + tb * NR + i, lcnt[i]);
      }
    }
  }
}


__global__ __launch_bounds__(256) void walk_a(const float* __restrict__ in, double* __restrict__ out, int S, int R, int xcd) {
  const int nch = S / 1024;
  int b = blockIdx.x;
  if (xcd) {
    const int n = gridDim.x, per = n >> 3;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  const int g = b / nch, c = b - g * nch;
  const size_t base = (size_t)g * R * S + (size_t)c * 1024;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sA = wave * 256 + 2 * lane, sB = sA + 128;
  f2 a = *reinterpret_cast<const f2*>(in + base + sA), bb = *reinterpret_cast<const f2*>(in + base + sB);
  for (int r = 0; r < R; ++r) {
    const f2 ca = a, cb = bb;
    if (r + 1 < R) {
      a = *reinterpret_cast<const f2*>(in + base + (size_t)(r + 1) * S + sA);
      bb = *reinterpret_cast<const f2*>(in + base + (size_t)(r + 1) * S + sB);
    }
    d2 oa = {(double)ca.x * 1.5 + 1.0, (double)ca.y * 1.5 + 1.0}, ob = {(double)cb.x * 1.5 + 1.0, (double)cb.y * 1.5 + 1.0};
    __builtin_nontemporal_store(oa, reinterpret_cast<d2*>(out + base + (size_t)r * S + sA));
    __builtin_nontemporal_store(ob, reinterpret_cast<d2*>(out + base + (size_t)r * S + sB));
  }
}

template <typename F>
double timed(F launch, double bytes) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  return bytes / best / 1e9;
}

int main() {
  const int S = 4096, P = 200000, NSET = 6;
  double *acc;
  unsigned* cnt;
  (void)hipMalloc(&acc, (size_t)(P / RB + 1) * NR * 8);
  (void)hipMalloc(&cnt, (size_t)(P / RB + 1) * NR * 4);
  (void)hipMemset(acc, 0, (size_t)(P / RB + 1) * NR * 8);
  (void)hipMemset(cnt, 0, (size_t)(P / RB + 1) * NR * 4);
  const double bytes = (double)P * S * 12.0;
  printf("set   walk A   rows20+lds  rows4+sum  rows4+sum+cnt  rows2+sum  rows1+sum  piece  piece+sum   (TB/s; one allocation set per line)\n");
  for (int k = 0; k < NSET; ++k) {  // (the sets stay allocated: every set its own placement)
    float* in;
    double* out;
    (void)hipMalloc(&in, (size_t)P * S * 4);
    (void)hipMalloc(&out, (size_t)P * S * 8);
    (void)hipMemset(in, 0, (size_t)P * S * 4);
    (void)hipMemset(out, 0, (size_t)P * S * 8);
    const double a = timed([&] { hipLaunchKernelGGL(walk_a, dim3((P / RB) * (S / 1024)), dim3(256), 0, 0, in, out, S, RB, 1); }, bytes);
    const double r20 = timed([&] { hipLaunchKernelGGL((walk<1, 0>), dim3(P / 20), dim3(256), 0, 0, in, out, S, 20, acc, cnt, 1); }, bytes);
    const double r4 = timed([&] { hipLaunchKernelGGL((walk<2, 0>), dim3(P / 4), dim3(256), 0, 0, in, out, S, 4, acc, cnt, 1); }, bytes);
    const double r4c = timed([&] { hipLaunchKernelGGL((walk<3, 0>), dim3(P / 4), dim3(256), 0, 0, in, out, S, 4, acc, cnt, 1); }, bytes);
    const double r2 = timed([&] { hipLaunchKernelGGL((walk<2, 0>), dim3(P / 2), dim3(256), 0, 0, in, out, S, 2, acc, cnt, 1); }, bytes);
    const double r1 = timed([&] { hipLaunchKernelGGL((walk<2, 0>), dim3(P), dim3(256), 0, 0, in, out, S, 1, acc, cnt, 1); }, bytes);
    const double pc = timed([&] { hipLaunchKernelGGL((walk<0, 1>), dim3(P * (S / 1024)), dim3(256), 0, 0, in, out, S, 1, acc, cnt, 1); }, bytes);
    const double pcs = timed([&] { hipLaunchKernelGGL((walk<2, 1>), dim3(P * (S / 1024)), dim3(256), 0, 0, in, out, S, 1, acc, cnt, 1); }, bytes);
    printf("%3d   %6.3f   %9.3f  %9.3f  %13.3f  %9.3f  %9.3f  %5.3f  %9.3f\n", k, a, r20, r4, r4c, r2, r1, pc, pcs);
    fflush(stdout);
  }
  return 0;
}
