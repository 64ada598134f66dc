// Round 5, second step: can a streaming walk that does NOT tie a workgroup to a time bin's 20 rows carry the bins?
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

template <int MODE, int PIECE>
void run(const char* name, const float* in, double* out, int P, int S, int G, double* acc, unsigned* cnt, int xcd) {
  const int grid = PIECE ? P * (S / 1024) : P / G;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<MODE, PIECE>), dim3(grid), dim3(256), 0, 0, in, out, S, G, acc, cnt, xcd);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-22s mode %d G %2d xcd %d wgs %7d : %7.3f ms  %6.3f TB/s\n", name, MODE, PIECE ? 0 : G, xcd, grid, best,
         (double)P * S * 12.0 / best / 1e9);
  fflush(stdout);
}

int main() {
  const int S = 4096, P = 200000;
  float* in;
  double *out, *acc;
  unsigned* cnt;
  (void)hipMalloc(&in, (size_t)P * S * 4);
  (void)hipMalloc(&out, (size_t)P * S * 8);
  (void)hipMalloc(&acc, (size_t)(P / RB + 1) * NR * 8);
  (void)hipMalloc(&cnt, (size_t)(P / RB + 1) * NR * 4);
  (void)hipMemset(in, 0, (size_t)P * S * 4);
  (void)hipMemset(out, 0, (size_t)P * S * 8);
  (void)hipMemset(acc, 0, (size_t)(P / RB + 1) * NR * 8);
  (void)hipMemset(cnt, 0, (size_t)(P / RB + 1) * NR * 4);
  for (int xcd : {0, 1}) {
    for (int G : {1, 2, 4, 5, 10, 20}) {
      run<0, 0>("rows", in, out, P, S, G, acc, cnt, xcd);
      run<1, 0>("rows+lds", in, out, P, S, G, acc, cnt, xcd);
      run<2, 0>("rows+lds+sum", in, out, P, S, G, acc, cnt, xcd);
      run<3, 0>("rows+lds+sum+cnt", in, out, P, S, G, acc, cnt, xcd);
    }
    run<0, 1>("piece", in, out, P, S, 1, acc, cnt, xcd);
    run<1, 1>("piece+lds", in, out, P, S, 1, acc, cnt, xcd);
    run<2, 1>("piece+lds+sum", in, out, P, S, 1, acc, cnt, xcd);
    run<3, 1>("piece+lds+sum+cnt", in, out, P, S, 1, acc, cnt, xcd);
  }
  return 0;
}
