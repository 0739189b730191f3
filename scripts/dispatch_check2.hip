// Round 3 follow-up of dispatch_check.hip: beside a chip-filling GEMM-like kernel (2 workgroups per CU, 69.6 KB of LDS and 256
// VGPRs each) whose workgroups retire at STAGGERED times (as real GEMM tiles do), how long does a chain of small dependent
// kernels on a high-priority stream take, as a function of the small kernel's LDS footprint?  160 KB per CU - 69.6 KB = 90.4 KB
// is what one retiring GEMM workgroup leaves free.  hipcc --offload-arch=gfx950 -O2 dispatch_check2.hip -o dispatch_check2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(256))) void hog(double* out, long long cycles) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  // 0.5x .. 1.5x of `cycles`, by block: retirements spread over the whole round instead of coming in lock-step
  const long long mine = cycles / 2 + (cycles * (long long)((blockIdx.x * 2654435761u) >> 22)) / 1024;
  const long long t0 = __builtin_amdgcn_s_memtime();
  double a = lds[threadIdx.x];
  while (__builtin_amdgcn_s_memtime() - t0 < mine) a = a * 1.0000001 + 1e-9;
  if (a == 1.2345) out[0] = a;
}
__global__ __launch_bounds__(256) void tiny(double* out, long long cycles) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = 1.0;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  double a = lds[threadIdx.x];
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) a = a * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) out[1] = lds[5] + a;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double* out; (void)hipMalloc(&out, 64);
  (void)hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  (void)hipFuncSetAttribute((const void*)tiny, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t s_main, s_hi, s_lo;
  (void)hipStreamCreateWithPriority(&s_main, hipStreamNonBlocking, lo);
  (void)hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo);
  (void)hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi);
  const int lds_kb[8] = {146, 130, 100, 92, 89, 77, 68, 16};
  // tiny kernel: ~20 us of work (2000 ticks of the 100 MHz counter); chain of 32 dependent launches; alone: 32 x ~25 us
  for (int wgs = 1; wgs <= 16; wgs *= 16)
    for (int pr = 0; pr < 2; ++pr)
      for (int li = 0; li < 8; ++li) {
        hipStream_t ss = pr ? s_hi : s_lo;
        double best1 = 1e9, best2 = 0, alone = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          (void)hipDeviceSynchronize();
          double t0 = now_ms();
          for (int c = 0; c < 32; ++c) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), lds_kb[li] * 1024, ss, out, 2000LL);
          (void)hipStreamSynchronize(ss);
          double t1 = now_ms();
          if (t1 - t0 < alone) alone = t1 - t0;
          (void)hipDeviceSynchronize();
          t0 = now_ms();
          hipLaunchKernelGGL(hog, dim3(2 * cus * 16), dim3(256), 69632, s_main, out, 40000LL);  // 16 rounds of ~0.2-0.6 ms
          for (int c = 0; c < 32; ++c) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), lds_kb[li] * 1024, ss, out, 2000LL);
          (void)hipStreamSynchronize(ss);
          t1 = now_ms();
          (void)hipStreamSynchronize(s_main);
          const double t2 = now_ms();
          if (t1 - t0 < best1) { best1 = t1 - t0; best2 = t2 - t0; }
        }
        printf("chain of 32 kernels x %2d WGs, %-4s stream, LDS %3d KB: alone %.3f ms, beside the hog %.3f ms (hog done %.3f ms)\n", wgs,
               pr ? "HIGH" : "low", lds_kb[li], alone, best1, best2);
      }
  return 0;
}
