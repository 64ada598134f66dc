// Round 5: the walk-order probe at the headline's ALIGNED row length (S = 4096: 16-KB input rows, 32-KB output rows) and
// with wider workgroups -- which walk of a (pings x range) array streams the 4 B read + 8 B written per sample fastest when
// a workgroup has to own R = 20 consecutive rows of a column range (a time bin's accumulators)?  Development aid.
//   W<NT>  workgroup of NT lanes = (group of R rows, NT*4 consecutive columns); rows inside; A|B pair pieces per wave
//   L<NT>  the same, lanes laid out linearly (lane -> 2 consecutive samples, step NT*2): two steps per row piece
// hipcc --offload-arch=gfx950 -O3 -o echopype_amd/lib/hbm_walk2_probe scripts/probes/hbm_walk2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void pair(const float* ip, double* op, int s) {
  const f2 v = *reinterpret_cast<const f2*>(ip + s);
  d2 o = {(double)v.x * 1.5 + 1.0, (double)v.y * 1.5 + 1.0};
  __builtin_nontemporal_store(o, reinterpret_cast<d2*>(op + s));
}

template <int NT, int LAYOUT, int PF>
__global__ __launch_bounds__(NT) void walk(const float* __restrict__ in, double* __restrict__ out, int S, int R, int xcd,
                                          int chunk_major) {
  constexpr int W = NT * 4;  // columns of a workgroup
  const int nch = S / W;
  int b = blockIdx.x;
  if (xcd) {  // contiguous eighth per XCD
    const int n = gridDim.x, per = n >> 3;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  int g, c;
  if (chunk_major) { c = b / (gridDim.x / nch); g = b - c * (gridDim.x / nch); }
  else { g = b / nch; c = b - g * nch; }
  const size_t base = (size_t)g * R * S + (size_t)c * W;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (LAYOUT == 0) {  // A|B pairs: wave covers 256 consecutive samples as two 128-sample instructions
    const int sA = wave * 256 + 2 * lane, sB = sA + 128;
    if (PF) {  // software prefetch of the next row, as the kernel does
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
    } else {
      for (int r = 0; r < R; ++r) {
        pair(in + base + (size_t)r * S, out + base + (size_t)r * S, sA);
        pair(in + base + (size_t)r * S, out + base + (size_t)r * S, sB);
      }
    }
  } else {  // linear: lane -> 2 consecutive samples, the workgroup steps NT*2 samples
    for (int r = 0; r < R; ++r) {
      pair(in + base + (size_t)r * S, out + base + (size_t)r * S, threadIdx.x * 2);
      pair(in + base + (size_t)r * S, out + base + (size_t)r * S, NT * 2 + threadIdx.x * 2);
    }
  }
}

template <int NT, int LAYOUT, int PF>
void run(const char* name, const float* in, double* out, int P, int S, int R, int xcd, int chunk_major) {
  const int grid = (P / R) * (S / (NT * 4));
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<NT, LAYOUT, PF>), dim3(grid), dim3(NT), 0, 0, in, out, S, R, xcd, chunk_major);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("%-28s NT %4d S %5d R %3d xcd %d cm %d wgs %7d : %7.3f ms  %6.3f TB/s\n", name, NT, S, R, xcd, chunk_major, grid,
         best, (double)(P / R * R) * S * 12.0 / best / 1e9);
  fflush(stdout);
}

int main() {
  for (int S : {4096, 2048}) {
    const int P = S == 4096 ? 200000 : 400000;
    float* in;
    double* out;
    (void)hipMalloc(&in, (size_t)P * S * 4);
    (void)hipMalloc(&out, (size_t)P * S * 8);
    (void)hipMemset(in, 0, (size_t)P * S * 4);
    (void)hipMemset(out, 0, (size_t)P * S * 8);
    for (int R : {20, 1}) {
      for (int xcd : {0, 1}) {
        run<256, 0, 0>("W256 pairs", in, out, P, S, R, xcd, 0);
        run<256, 0, 1>("W256 pairs prefetch", in, out, P, S, R, xcd, 0);
        run<256, 1, 0>("L256 linear", in, out, P, S, R, xcd, 0);
        run<512, 0, 0>("W512 pairs", in, out, P, S, R, xcd, 0);
        run<512, 1, 0>("L512 linear", in, out, P, S, R, xcd, 0);
        run<1024, 0, 0>("W1024 pairs", in, out, P, S, R, xcd, 0);
        run<1024, 0, 1>("W1024 pairs prefetch", in, out, P, S, R, xcd, 0);
        run<1024, 1, 0>("L1024 linear", in, out, P, S, R, xcd, 0);
      }
      run<256, 0, 0>("W256 pairs chunk-major", in, out, P, S, R, 0, 1);
      run<256, 0, 0>("W256 pairs chunk-major", in, out, P, S, R, 1, 1);
    }
    (void)hipFree(in);
    (void)hipFree(out);
  }
  return 0;
}
