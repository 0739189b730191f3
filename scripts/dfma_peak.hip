// Microbenchmark: sustained v_fma_f64 (vector fp64 FMA) rate on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NCH>
__global__ __launch_bounds__(256) void dfma_loop(double* out, int iters) {
  double a[NCH];
  const double b = 1.0000001, c = 1e-9 * threadIdx.x;
  for (int i = 0; i < NCH; ++i) a[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) a[i] = fma(a[i], b, c);
  }
  double s = 0;
  for (int i = 0; i < NCH; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(int blocks, int iters) {
  double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(dfma_loop<NCH>, dim3(blocks), dim3(256), 0, 0, out, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(dfma_loop<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 256 * iters * NCH * 2.0;
  double waves_per_simd = (double)blocks * 4 / 1024.0;
  double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * iters * NCH);
  printf("blocks=%d (%.0f waves/SIMD) chains=%d: %.2f TFLOP/s, %.2f cycles per wave64 v_fma_f64 per SIMD (at 2.4 GHz)\n", blocks, waves_per_simd, NCH, flops / ms / 1e9, cyc);
  hipFree(out);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) { run<8>(256, 20000); run<8>(1024, 20000); run<16>(2048, 10000); run<4>(2048, 20000); }
  return 0;
}
