// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD, at 2.4 GHz) of the VALU ops kfill's exp/sqrt use.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHAINS 8
#define OPK(NAME, ASM)                                                                              \
  __global__ __launch_bounds__(256) void NAME(double* out, int iters) {                             \
    double a[CHAINS];                                                                               \
    int e = (threadIdx.x & 3) - 1;                                                                  \
    for (int i = 0; i < CHAINS; ++i) a[i] = 1.0 + i * 1e-3 + threadIdx.x * 1e-6;                    \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(e));  \
    }                                                                                               \
    double s = 0;                                                                                   \
    for (int i = 0; i < CHAINS; ++i) s += a[i];                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
  }
OPK(k_fma, "v_fma_f64 %0, %0, %0, %0")
OPK(k_mul, "v_mul_f64 %0, %0, %0")
OPK(k_add, "v_add_f64 %0, %0, %0")
OPK(k_max, "v_max_f64 %0, %0, %0")
OPK(k_rsq, "v_rsq_f64 %0, %0")
OPK(k_rcp, "v_rcp_f64 %0, %0")
OPK(k_sqrt, "v_sqrt_f64 %0, %0")
OPK(k_ldexp, "v_ldexp_f64 %0, %0, %1")
OPK(k_rndne, "v_rndne_f64 %0, %0")
OPK(k_fract, "v_fract_f64 %0, %0")
OPK(k_mov64, "v_mov_b64 %0, %0")
OPK(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, %0")
// 32-bit ops on the low dword of the chain register
#define OPK32(NAME, ASM)                                                                            \
  __global__ __launch_bounds__(256) void NAME(double* out, int iters) {                             \
    int a[CHAINS];                                                                                  \
    double dsrc = 1.5 + threadIdx.x;                                                                \
    for (int i = 0; i < CHAINS; ++i) a[i] = i + threadIdx.x;                                        \
    for (int it = 0; it < iters; ++it) {                                                            \
      _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(dsrc)); \
    }                                                                                               \
    int s = 0;                                                                                      \
    for (int i = 0; i < CHAINS; ++i) s += a[i];                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                 \
  }
OPK32(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1")
OPK32(k_and32, "v_and_b32 %0, 31, %0")
OPK32(k_lshladd32, "v_lshl_add_u32 %0, %0, 3, %0")
OPK32(k_cndmask, "v_cndmask_b32 %0, %0, %0, vcc")
OPK32(k_fma32, "v_fma_f32 %0, %0, %0, %0")
OPK32(k_rsq32, "v_rsq_f32 %0, %0")
template <class K> void run(const char* name, K kern) {
  const int blocks = 2048, iters = 4000;
  double* out; (void)hipMalloc(&out, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = (double)blocks * 4 / 1024.0;
  printf("%-16s %6.2f cycles per wave64 instruction per SIMD\n", name, ms * 1e-3 * 2.4e9 / (waves_per_simd * iters * CHAINS));
  (void)hipFree(out);
}
int main() {
  run("v_fma_f64", k_fma); run("v_fma_f64", k_fma); run("v_mul_f64", k_mul); run("v_add_f64", k_add); run("v_max_f64", k_max);
  run("v_rsq_f64", k_rsq); run("v_rcp_f64", k_rcp); run("v_sqrt_f64", k_sqrt); run("v_ldexp_f64", k_ldexp);
  run("v_rndne_f64", k_rndne); run("v_fract_f64", k_fract); run("v_mov_b64", k_mov64); run("v_lshl_add_u64", k_lshladd64);
  run("v_cvt_i32_f64", k_cvt_i32_f64); run("v_and_b32", k_and32); run("v_lshl_add_u32", k_lshladd32);
  run("v_cndmask_b32", k_cndmask); run("v_fma_f32", k_fma32); run("v_rsq_f32", k_rsq32);
  return 0;
}
