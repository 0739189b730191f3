// Does a CU-masked stream give a small kernel a reserved place beside a chip-filling kernel?  (gfx950, ordinary user)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
__global__ __launch_bounds__(256) void hog(double* out, long long cycles) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  double a = lds[threadIdx.x];
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) a = a * 1.0000001 + 1e-9;
  if (a == 1.2345) out[0] = a;
}
__global__ __launch_bounds__(256) void tiny(double* out) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = 1.0;
  __syncthreads();
  if (threadIdx.x == 0) out[1] = lds[5];
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double* out; (void)hipMalloc(&out, 64);
  (void)hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  (void)hipFuncSetAttribute((const void*)tiny, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("CUs: %d\n", cus);
    const int words = (cus + 31) / 32;
  std::vector<uint32_t> m_main(words, 0), m_side(words, 0);
  // bit i = CU i/8 of XCC i%8 (scripts/cumask_map.hip): the lowest 16 bits are 2 CUs on every XCD
  for (int i = 0; i < cus; ++i) { if (i < 16) m_side[i / 32] |= 1u << (i % 32); else m_main[i / 32] |= 1u << (i % 32); }
  hipStream_t s_main_plain, s_side_plain, s_main_mask, s_side_mask;
  (void)hipStreamCreateWithFlags(&s_main_plain, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s_side_plain, hipStreamNonBlocking);
  hipError_t e1 = hipExtStreamCreateWithCUMask(&s_main_mask, words, m_main.data());
  hipError_t e2 = hipExtStreamCreateWithCUMask(&s_side_mask, words, m_side.data());
  printf("hipExtStreamCreateWithCUMask: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  if (e1 != hipSuccess || e2 != hipSuccess) return 1;
  for (int mode = 0; mode < 2; ++mode) {
    hipStream_t sm = mode ? s_main_mask : s_main_plain, ss = mode ? s_side_mask : s_side_plain;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipDeviceSynchronize();
      const double t0 = now_ms();
      hipLaunchKernelGGL(hog, dim3(2 * cus * 4), dim3(256), 70 * 1024, sm, out, 2400000LL);  // 4 rounds of ~1 ms tiles
      hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 130 * 1024, ss, out);
      (void)hipStreamSynchronize(ss);
      const double t1 = now_ms();
      (void)hipStreamSynchronize(sm);
      const double t2 = now_ms();
      printf("%s: tiny kernel done after %.3f ms, hog done after %.3f ms\n", mode ? "CU-masked streams" : "plain streams    ", t1 - t0, t2 - t0);
    }
  }
  return 0;
}
