// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (operands in registers, no memory).
// Prints TFLOP/s for 1 and 2 waves per SIMD and NACC independent accumulators, plus in-kernel clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(double* out, int iters, unsigned long long* clk) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int NACC>
void run(int blocks, int threads, int iters) {
  double* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * threads * 8); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, 100, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64;
  double flops = waves * iters * NACC * 2048.0;
  double cyc_per_mfma = (double)h[0] / ((double)iters * NACC);
  printf("blocks=%d threads=%d nacc=%d: %.2f TFLOP/s  (%.3f ms)  cycles/MFMA(per wave)=%.1f  clock=%.0f MHz\n", blocks,
         threads, NACC, flops / ms / 1e9, ms, cyc_per_mfma, (double)h[0] / (double)h[1] * 100.0);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<4>(256, 256, 20000);    // 1 wave / SIMD
    run<16>(256, 256, 5000);
    run<4>(512, 256, 20000);    // 2 waves / SIMD
    run<16>(512, 256, 5000);
    run<16>(1024, 256, 5000);   // 4 waves / SIMD
    run<1>(256, 256, 40000);    // dependent chain: latency
  }
  return 0;
}
