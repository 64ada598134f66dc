// Round 5, third step: the row-major walk (a workgroup streams its 20 rows whole) with WIDER workgroups -- fewer columns
// per lane would let a lane keep lane-private column sums (16 columns per lane at 256 lanes is too many).
//   NT lanes per workgroup, every lane 4 samples (A|B pairs) per 4*NT-sample step, S / (4 NT) steps per row, G rows.
// hipcc --offload-arch=gfx950 -O3 -o echopype_amd/lib/hbm_walk4_probe scripts/probes/hbm_walk4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void pair(const float* ip, double* op, int s) {
  const f2 v = *reinterpret_cast<const f2*>(ip + s);
  d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
  __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + s));
}

template <int NT, int WPS>  // WPS: waves per SIMD the launch bounds ask for (register budget of the real kernel)
__global__ __launch_bounds__(NT) void walk(const float* __restrict__ in, double* __restrict__ out, int S, int G, int xcd) {
  extern __shared__ unsigned char pad[];
  if (xcd == 99) pad[threadIdx.x] = 0;
  int b = blockIdx.x;
  if (xcd) {
    const int n = gridDim.x, per = n >> 3;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sA = wave * 256 + 2 * lane, sB = sA + 128;
  const size_t row0 = (size_t)b * G;
  for (int r = 0; r < G; ++r)
    for (int c = 0; c < S; c += 4 * NT) {
      pair(in + (row0 + r) * S, out + (row0 + r) * S, c + sA);
      pair(in + (row0 + r) * S, out + (row0 + r) * S, c + sB);
    }
}

template <int NT>
void run(const float* in, double* out, int P, int S, int G, int pad) {
  const int grid = P / G;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(walk<NT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<NT, 4>), dim3(grid), dim3(NT), pad, 0, in, out, S, G, 1);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("rows NT %4d G %3d pad %6d wgs %7d : %7.3f ms  %6.3f TB/s\n", NT, G, pad, grid, best, (double)P * S * 12.0 / best / 1e9);
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
  // pad caps the workgroups per CU (160 KB LDS): 16 waves per CU = 4 per SIMD = the real kernel's occupancy
  for (int G : {20}) {
    run<256>(in, out, P, S, G, 0);
    run<256>(in, out, P, S, G, 36000);   // 4 workgroups of 256 per CU
    run<512>(in, out, P, S, G, 0);
    run<512>(in, out, P, S, G, 60000);   // 2 workgroups of 512 per CU
    run<512>(in, out, P, S, G, 36000);   // 4 per CU (8 waves per SIMD)
    run<1024>(in, out, P, S, G, 0);
    run<1024>(in, out, P, S, G, 60000);  // 2 workgroups of 1024 per CU (8 waves per SIMD); 1 per CU is not reachable by LDS
  }
  return 0;
}
