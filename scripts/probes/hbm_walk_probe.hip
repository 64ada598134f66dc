// How the ORDER in which workgroups walk a (pings x range) array changes the streaming rate of the headline traffic mix
// (read 4 B, write 8 B per sample) -- development aid.  Rows of S samples; a "group" = R consecutive rows (a time bin).
//   A  workgroup = group, column chunks of 1024 outermost, rows inside        (the fused kernel's walk)
//   B  workgroup = (group, chunk): 1024 columns x R rows                      (4 x more, smaller workgroups)
//   C  workgroup = group, rows outermost (one contiguous run of R x S samples)
//   E  workgroup = one row                                                    (no grouping at all)
// hipcc --offload-arch=gfx950 -O3 -o echopype_amd/lib/hbm_walk_probe scripts/probes/hbm_walk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void piece(const float* ip, double* op, int n) {  // n <= 1024 samples, 4 per lane as A|B pairs
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sA = wave * 256 + 2 * lane, sB = sA + 128;
  if (sA < n) {
    const f2 v = *reinterpret_cast<const f2*>(ip + sA);
    d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
    __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + sA));
  }
  if (sB < n) {
    const f2 v = *reinterpret_cast<const f2*>(ip + sB);
    d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
    __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + sB));
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void walk_kernel(const float* __restrict__ in, double* __restrict__ out, int P, int S,
                                                   int R, int xcd) {
  const int nchunks = (S + 1023) / 1024;
  int b = blockIdx.x;
  if (MODE == 0) {  // A
    const size_t r0 = (size_t)b * R;
    for (int c = 0; c < nchunks; ++c)
      for (int r = 0; r < R; ++r)
        piece(in + (r0 + r) * S + c * 1024, out + (r0 + r) * S + c * 1024, min(1024, S - c * 1024));
  } else if (MODE == 1) {  // B: chunk fastest
    const int g = b / nchunks, c = b - g * nchunks;
    const size_t r0 = (size_t)g * R;
    for (int r = 0; r < R; ++r)
      piece(in + (r0 + r) * S + c * 1024, out + (r0 + r) * S + c * 1024, min(1024, S - c * 1024));
  } else if (MODE == 2) {  // C
    const size_t r0 = (size_t)b * R;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < nchunks; ++c)
        piece(in + (r0 + r) * S + c * 1024, out + (r0 + r) * S + c * 1024, min(1024, S - c * 1024));
  } else if (MODE == 3) {  // E
    for (int c = 0; c < nchunks; ++c)
      piece(in + (size_t)b * S + c * 1024, out + (size_t)b * S + c * 1024, min(1024, S - c * 1024));
  } else if (MODE == 4) {  // F: one row per workgroup, plain 512-sample steps (one pair per lane)
    const float* ip = in + (size_t)b * S;
    double* op = out + (size_t)b * S;
    for (int s = threadIdx.x * 2; s < S; s += 512) {
      const f2 v = *reinterpret_cast<const f2*>(ip + s);
      d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
      __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + s));
    }
  } else if (MODE == 5) {  // G: persistent workgroups, rows dealt round-robin, A|B pieces
    for (int row = b; row < P; row += gridDim.x)
      for (int c = 0; c < nchunks; ++c)
        piece(in + (size_t)row * S + c * 1024, out + (size_t)row * S + c * 1024, min(1024, S - c * 1024));
  } else {  // H: persistent workgroups, rows dealt round-robin, plain 512-sample steps
    for (int row = b; row < P; row += gridDim.x) {
      const float* ip = in + (size_t)row * S;
      double* op = out + (size_t)row * S;
      for (int s = threadIdx.x * 2; s < S; s += 512) {
        const f2 v = *reinterpret_cast<const f2*>(ip + s);
        d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
        __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + s));
      }
    }
  }
}

template <int MODE>
void run(const char* name, const float* in, double* out, int P, int S, int R) {
  const int nchunks = (S + 1023) / 1024;
  const int grid = MODE == 0 || MODE == 2 ? P / R : MODE == 1 ? P / R * nchunks : MODE >= 5 ? R : P;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk_kernel<MODE>), dim3(grid), dim3(256), 0, 0, in, out, P, S, R, 0);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-46s S %5d R %3d wgs %7d : %7.3f ms  %6.3f TB/s\n", name, S, R, grid, best, (double)P * S * 12.0 / best / 1e9);
  fflush(stdout);
}

int main() {
  for (int S : {4000, 2000}) {
    const int P = S == 4000 ? 200000 : 400000;
    float* in;
    double* out;
    (void)hipMalloc(&in, (size_t)P * S * 4);
    (void)hipMalloc(&out, (size_t)P * S * 8);
    (void)hipMemset(in, 0, (size_t)P * S * 4);
    (void)hipMemset(out, 0, (size_t)P * S * 8);
    for (int R : {20, 5}) {
      run<0>("A group, chunks outermost (fused kernel)", in, out, P, S, R);
      run<1>("B (group, chunk) workgroups", in, out, P, S, R);
      run<2>("C group, rows outermost", in, out, P, S, R);
    }
    run<3>("E one row per workgroup", in, out, P, S, 1);
    run<4>("F one row per workgroup, 512-sample steps", in, out, P, S, 1);
    for (int g : {2048, 8192, 65536}) {
      run<5>("G persistent, A|B pieces", in, out, P, S, g);
      run<6>("H persistent, 512-sample steps", in, out, P, S, g);
    }
    (void)hipFree(in);
    (void)hipFree(out);
  }
  return 0;
}
