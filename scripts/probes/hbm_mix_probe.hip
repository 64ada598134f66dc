// Streaming-ceiling probe for the headline traffic mix (read 4 B, write 8 B per sample) -- development aid.
// Variants: bytes per lane, workgroup -> memory mapping (linear / XCD-contiguous), contiguous run per workgroup,
// store flavour.   hipcc --offload-arch=gfx950 -O3 -o echopype_amd/lib/hbm_mix_probe scripts/probes/hbm_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int STORE>
__device__ __forceinline__ void st2(double* p, double a, double b) {
  d2 v = {a, b};
  if (STORE == 0) *reinterpret_cast<d2*>(p) = v;
  else if (STORE == 1) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p));
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// run = samples of one contiguous piece handled by a workgroup at a time; pieces are dealt to workgroups either
// linearly (piece i -> workgroup i % nwg) or so that each XCD (workgroup id % 8) owns one contiguous eighth
template <int VEC, int STORE, bool XCD>
__global__ __launch_bounds__(256) void mix_kernel(const float* __restrict__ in, double* __restrict__ out, size_t n,
                                                  int run) {
  const size_t npieces = n / run;
  const int nwg = gridDim.x;
  for (size_t it = blockIdx.x;; it += nwg) {
    size_t piece;
    if (XCD) {
      const int xcd = blockIdx.x & 7;
      const size_t per = npieces / 8;
      const size_t k = (it >> 3);  // this workgroup's sequence number among its XCD's workgroups, strided
      if (k >= per) break;
      piece = (size_t)xcd * per + k;
    } else {
      if (it >= npieces) break;
      piece = it;
    }
    const float* ip = in + piece * run;
    double* op = out + piece * run;
    for (int s = threadIdx.x * VEC; s < run; s += 256 * VEC) {
      if (VEC == 2) {
        const f2 v = *reinterpret_cast<const f2*>(ip + s);
        st2<STORE>(op + s, (double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0);
      } else {
        const f4 v = *reinterpret_cast<const f4*>(ip + s);
        st2<STORE>(op + s, (double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0);
        st2<STORE>(op + s + 2, (double)v.z * 1.5 + 1.0, (double)v.w * 1.5 + 1.0);
      }
    }
  }
}

template <int VEC, int STORE, bool XCD>
void run_variant(const char* name, const float* in, double* out, size_t n, int run, int nwg) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((mix_kernel<VEC, STORE, XCD>), dim3(nwg), dim3(256), 0, 0, in, out, n, run);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-34s run %6d  wgs %7d : %7.3f ms  %6.3f TB/s\n", name, run, nwg, best, n * 12.0 / best / 1e9);
  fflush(stdout);
}

int main() {
  const size_t n = (size_t)800 << 20;
  float* in;
  double* out;
  hipMalloc(&in, n * 4);
  hipMalloc(&out, n * 8);
  hipMemset(in, 0, n * 4);
  hipMemset(out, 0, n * 8);
  for (int run : {2048, 8192, 32768, 131072}) {
    for (int nwg : {2048, 8192, 65536}) {
      run_variant<2, 1, false>("vec2 nt linear", in, out, n, run, nwg);
      run_variant<2, 1, true>("vec2 nt xcd-contiguous", in, out, n, run, nwg);
      run_variant<4, 1, false>("vec4 nt linear", in, out, n, run, nwg);
      run_variant<4, 1, true>("vec4 nt xcd-contiguous", in, out, n, run, nwg);
    }
  }
  run_variant<2, 0, false>("vec2 plain linear", in, out, n, 8192, 8192);
  run_variant<2, 2, false>("vec2 sc0 sc1 nt linear", in, out, n, 8192, 8192);
  run_variant<4, 0, false>("vec4 plain linear", in, out, n, 8192, 8192);
  run_variant<4, 2, false>("vec4 sc0 sc1 nt linear", in, out, n, 8192, 8192);
  run_variant<4, 2, true>("vec4 sc0 sc1 nt xcd", in, out, n, 8192, 8192);
  return 0;
}
