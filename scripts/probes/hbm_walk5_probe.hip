// Round 5, last step: the three walks of a (pings x 4096) array side by side on ONE box, XCD-contiguous order, 256-lane
// workgroups of A|B pairs, 4 B read + 8 B written per sample (boxes of the pool differ by 15 % on some of them):
//   A  workgroup = 20 rows, 1024-column chunks outermost (the shipped fused kernel's walk)
//   C  workgroup = 20 rows, rows outermost (the row-major form: one contiguous run per workgroup)
//   P  workgroup = one 1024-sample piece (no loop)
// hipcc --offload-arch=gfx950 -O3 -o echopype_amd/lib/hbm_walk5_probe scripts/probes/hbm_walk5_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pair(const float* ip, double* op, int s) {
  const f2 v = *reinterpret_cast<const f2*>(ip + s);
  d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
  __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + s));
}
template <int MODE>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ in, double* __restrict__ out, int S, int G) {
  int b = blockIdx.x;
  const int n = gridDim.x, per = n >> 3;
  if (b < per * 8) b = (b & 7) * per + (b >> 3);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sA = wave * 256 + 2 * lane, sB = sA + 128;
  if (MODE == 2) {
    const int nch = S / 1024;
    const size_t row = b / nch;
    const int c = (b - (int)row * nch) * 1024;
    pair(in + row * S, out + row * S, c + sA);
    pair(in + row * S, out + row * S, c + sB);
    return;
  }
  const size_t row0 = (size_t)b * G;
  if (MODE == 0) {
    for (int c = 0; c < S; c += 1024)
      for (int r = 0; r < G; ++r) {
        pair(in + (row0 + r) * S, out + (row0 + r) * S, c + sA);
        pair(in + (row0 + r) * S, out + (row0 + r) * S, c + sB);
      }
  } else {
    for (int r = 0; r < G; ++r)
      for (int c = 0; c < S; c += 1024) {
        pair(in + (row0 + r) * S, out + (row0 + r) * S, c + sA);
        pair(in + (row0 + r) * S, out + (row0 + r) * S, c + sB);
      }
  }
}
template <int MODE>
void run(const char* name, const float* in, double* out, int P, int S, int G) {
  const int grid = MODE == 2 ? P * (S / 1024) : P / G;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f, worst = 0.f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<MODE>), dim3(grid), dim3(256), 0, 0, in, out, S, G);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
  }
  printf("%-34s wgs %7d : %7.3f .. %7.3f ms  %6.3f TB/s (best)\n", name, grid, best, worst, (double)P * S * 12.0 / best / 1e9);
  fflush(stdout);
}
int main() {
  const int S = 4096, P = 200000;
  float* in;
  double* out;
  (void)hipMalloc(&in, (size_t)P * S * 4);
  (void)hipMalloc(&out, (size_t)P * S * 8);
  (void)hipMemset(in, 0, (size_t)P * S * 4);
  (void)hipMemset(out, 0, (size_t)P * S * 8);
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  printf("%s, %d CUs, memory clock %d kHz, bus %d bit\n", prop.name, prop.multiProcessorCount, prop.memoryClockRate, prop.memoryBusWidth);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("A chunks outermost (shipped walk)", in, out, P, S, 20);
    run<1>("C rows outermost (row-major)", in, out, P, S, 20);
    run<2>("P one piece per workgroup", in, out, P, S, 1);
  }
  return 0;
}
