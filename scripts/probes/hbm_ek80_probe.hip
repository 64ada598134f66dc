// Streaming ceiling of the EK80 complex traffic: two float32 planes of 4 sectors per sample in (2 x 16 B), one double
// out (8 B) -- development aid.  Variants: samples per lane and their stride (the FFT kernel: 8 samples at a stride of
// 256; a plain streamer: 1 sample per lane per step), workgroups per tile / per row.
// hipcc --offload-arch=gfx950 -O3 -o echopype_amd/lib/hbm_ek80_probe scripts/probes/hbm_ek80_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: one workgroup per 2048-sample tile, lane j takes samples j + 256 i, all 16 loads first (the FFT kernel)
// MODE 1: the same tile, but load - sum - store sample by sample
// MODE 2: persistent workgroups, grid-stride over 256-sample pieces (one sample per lane per step)
// MODE 3: as 0 with only the real plane read twice (one input stream)
template <int MODE>
__global__ __launch_bounds__(256) void ek80_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                   double* __restrict__ out, size_t n) {
  const int j = threadIdx.x;
  if (MODE == 2) {
    for (size_t s = (size_t)blockIdx.x * 256 + j; s < n; s += (size_t)gridDim.x * 256) {
      const f4 a = *reinterpret_cast<const f4*>(re + s * 4), b = *reinterpret_cast<const f4*>(im + s * 4);
      out[s] = (double)(a.x + a.y + a.z + a.w) * 1.5 + (double)(b.x + b.y + b.z + b.w);
    }
    return;
  }
  const size_t t0 = (size_t)blockIdx.x * 2048;
  if (MODE == 1) {
    for (int i = 0; i < 8; ++i) {
      const size_t s = t0 + j + 256 * i;
      if (s < n) {
        const f4 a = *reinterpret_cast<const f4*>(re + s * 4), b = *reinterpret_cast<const f4*>(im + s * 4);
        out[s] = (double)(a.x + a.y + a.z + a.w) * 1.5 + (double)(b.x + b.y + b.z + b.w);
      }
    }
    return;
  }
  f4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t s = t0 + j + 256 * i;
    a[i] = *reinterpret_cast<const f4*>(re + s * 4);
    b[i] = *reinterpret_cast<const f4*>((MODE == 3 ? re : im) + s * 4);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const size_t s = t0 + j + 256 * i;
    out[s] = (double)(a[i].x + a[i].y + a[i].z + a[i].w) * 1.5 + (double)(b[i].x + b[i].y + b[i].z + b[i].w);
  }
}

template <int MODE>
void run(const char* name, const float* re, const float* im, double* out, size_t n, int grid) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((ek80_kernel<MODE>), dim3(grid), dim3(256), 0, 0, re, im, out, n);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-58s wgs %8d : %7.3f ms  %6.3f TB/s\n", name, grid, best, (double)n * 40.0 / best / 1e9);
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)2048 * 200000;  // 0.41 G samples: 13 GB in, 3.3 GB out
  float *re, *im;
  double* out;
  (void)hipMalloc(&re, n * 16);
  (void)hipMalloc(&im, n * 16);
  (void)hipMalloc(&out, n * 8);
  (void)hipMemset(re, 0, n * 16);
  (void)hipMemset(im, 0, n * 16);
  run<0>("tile per workgroup, 16 loads then 8 stores (FFT kernel)", re, im, out, n, (int)(n / 2048));
  run<1>("tile per workgroup, sample by sample", re, im, out, n, (int)(n / 2048));
  run<3>("tile per workgroup, one input plane read twice", re, im, out, n, (int)(n / 2048));
  for (int g : {2048, 8192, 65536}) run<2>("persistent, one sample per lane per step", re, im, out, n, g);
  return 0;
}
