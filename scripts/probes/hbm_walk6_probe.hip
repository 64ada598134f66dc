// Round 6: is the ceiling of bin-owning walks (5.0-5.4 TB/s against 6.2 for loop-free workgroups, hbm_walk2/3 probes) the
// in-order retirement of vector-memory operations?  On gfx9-family parts loads AND stores share one counter (vmcnt): a
// wavefront that waits for the row it requested one iteration ago also waits for the stores it issued BEFORE that request,
// i.e. for the acknowledgement of the previous row's 1 KiB -- one iteration of slack.  Here the rows are requested D
// iterations ahead (D = 0: load, use, store per iteration; 1: what the kernels do; 2-4: deeper), everything else as in
// hbm_walk2_probe's "W256 pairs": a workgroup owns R = 20 rows of 1024 columns, 4 B read + 8 B written per sample.
// hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/hbm_walk6_probe scripts/probes/hbm_walk6_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int D, int NT_STORE>
__global__ __launch_bounds__(256) void walk(const float* __restrict__ in, double* __restrict__ out, int S, int R, int xcd) {
  constexpr int W = 1024;
  const int nch = S / W;
  int b = blockIdx.x;
  if (xcd) {  // contiguous eighth per XCD
    const int n = gridDim.x, per = n >> 3;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  const int g = b / nch, c = b - g * nch;
  const size_t base = (size_t)g * R * S + (size_t)c * W;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sA = wave * 256 + 2 * lane, sB = sA + 128;
  const float* ip = in + base;
  double* op = out + base;
  auto st = [&](d2 v, double* p) {
    if (NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p));
    else *reinterpret_cast<d2*>(p) = v;
  };
  if (D == 0) {
    for (int r = 0; r < R; ++r) {
      const f2 ca = *reinterpret_cast<const f2*>(ip + (size_t)r * S + sA), cb = *reinterpret_cast<const f2*>(ip + (size_t)r * S + sB);
      st(d2{(double)ca.x * 1.5 + 1.0, (double)ca.y * 1.5 + 1.0}, op + (size_t)r * S + sA);
      st(d2{(double)cb.x * 1.5 + 1.0, (double)cb.y * 1.5 + 1.0}, op + (size_t)r * S + sB);
    }
    return;
  }
  constexpr int DD = D > 0 ? D : 1;
  f2 qa[DD], qb[DD];  // rows r + 1 .. r + D in flight
#pragma unroll
  for (int k = 0; k < DD; ++k) {
    const int rr = k < R ? k : R - 1;
    qa[k] = *reinterpret_cast<const f2*>(ip + (size_t)rr * S + sA);
    qb[k] = *reinterpret_cast<const f2*>(ip + (size_t)rr * S + sB);
  }
  for (int r0 = 0; r0 < R; r0 += DD) {
#pragma unroll
    for (int k = 0; k < DD; ++k) {  // (unrolled by the depth: slot k is refilled in place, no register rotation)
      const int r = r0 + k;
      if (r >= R) break;
      const f2 ca = qa[k], cb = qb[k];
      if (r + DD < R) {
        qa[k] = *reinterpret_cast<const f2*>(ip + (size_t)(r + DD) * S + sA);
        qb[k] = *reinterpret_cast<const f2*>(ip + (size_t)(r + DD) * S + sB);
      }
      st(d2{(double)ca.x * 1.5 + 1.0, (double)ca.y * 1.5 + 1.0}, op + (size_t)r * S + sA);
      st(d2{(double)cb.x * 1.5 + 1.0, (double)cb.y * 1.5 + 1.0}, op + (size_t)r * S + sB);
    }
  }
}

template <int D, int NT_STORE>
void run(const float* in, double* out, int P, int S, int R, int xcd) {
  const int grid = (P / R) * (S / 1024);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<D, NT_STORE>), dim3(grid), dim3(256), 0, 0, in, out, S, R, xcd);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  printf("rows ahead %d  nt-store %d  S %5d R %3d xcd %d wgs %7d : %7.3f ms  %6.3f TB/s\n", D, NT_STORE, S, R, xcd, grid, best,
         (double)(P / R * R) * S * 12.0 / best / 1e9);
  fflush(stdout);
}

template <int D>
double rate(const float* in, double* out, int P, int S, int R) {
  const int grid = (P / R) * (S / 1024);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((walk<D, 1>), dim3(grid), dim3(256), 0, 0, in, out, S, R, 1);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  return (double)(P / R * R) * S * 12.0 / best / 1e9;
}

int main(int argc, char**) {
  if (argc > 1) {  // several allocation sets of one process: the placement of a set decides more than any variant
    const int S = 4096, P = 200000;
    printf("set   rows ahead: 0      1      2      4      6   | R = 1 (loop-free)   (TB/s, R = 20 rows per workgroup)\n");
    for (int k = 0; k < 8; ++k) {
      float* in;
      double* out;
      (void)hipMalloc(&in, (size_t)P * S * 4);
      (void)hipMalloc(&out, (size_t)P * S * 8);
      (void)hipMemset(in, 0, (size_t)P * S * 4);
      (void)hipMemset(out, 0, (size_t)P * S * 8);
      printf("%3d            %6.3f %6.3f %6.3f %6.3f %6.3f   | %6.3f\n", k, rate<0>(in, out, P, S, 20), rate<1>(in, out, P, S, 20),
             rate<2>(in, out, P, S, 20), rate<4>(in, out, P, S, 20), rate<6>(in, out, P, S, 20), rate<0>(in, out, P, S, 1));
      fflush(stdout);
    }
    return 0;
  }

  for (int S : {4096, 2048}) {
    const int P = S == 4096 ? 200000 : 400000;
    float* in;
    double* out;
    (void)hipMalloc(&in, (size_t)P * S * 4);
    (void)hipMalloc(&out, (size_t)P * S * 8);
    (void)hipMemset(in, 0, (size_t)P * S * 4);
    (void)hipMemset(out, 0, (size_t)P * S * 8);
    for (int R : {20, 40, 1}) {
      for (int xcd : {1}) {
        run<0, 1>(in, out, P, S, R, xcd);
        run<1, 1>(in, out, P, S, R, xcd);
        run<2, 1>(in, out, P, S, R, xcd);
        run<3, 1>(in, out, P, S, R, xcd);
        run<4, 1>(in, out, P, S, R, xcd);
        run<6, 1>(in, out, P, S, R, xcd);
        run<1, 0>(in, out, P, S, R, xcd);
        run<4, 0>(in, out, P, S, R, xcd);
      }
    }
    (void)hipFree(in);
    (void)hipFree(out);
  }
  return 0;
}
