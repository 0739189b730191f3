// Cost of a cross-stream dependency on MI355X: a chain of tiny kernels that alternates between two streams through
// hipEventRecord / hipStreamWaitEvent, against the same chain on ONE stream -- for the event flag combinations HIP offers.
//   hipcc --offload-arch=gfx950 -O2 -o xstream_latency scripts/xstream_latency.hip && ./xstream_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void tiny(double* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0; }
__global__ void busy(double* p, int iters) {
  double a = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
  p[threadIdx.x] = a;
}
int main() {
  double* d;
  CK(hipMalloc(&d, 1 << 20));
  CK(hipMemset(d, 0, 1 << 20));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s[2];
  CK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, hi));
  const int hops = 200;
  struct V { const char* name; unsigned flags; };
  V vs[] = {{"DisableTiming", hipEventDisableTiming},
            {"DisableTiming|ReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
            {"DisableTiming|DisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
            {"DisableTiming|ReleaseToSystem", hipEventDisableTiming | hipEventReleaseToSystem},
            {"default (timing)", 0}};
  for (int kern = 0; kern < 2; ++kern) {
    auto launch = [&](hipStream_t st) {
      if (kern == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d);
      else hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, st, d, 4000);   // ~20 us of dependent FMAs
    };
    // one stream
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < hops; ++i) launch(s[0]);
      CK(hipDeviceSynchronize());
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("kernel %-5s one stream                          : %7.2f us per kernel\n", kern ? "20us" : "tiny", us / hops);
    }
    for (const V& v : vs) {
      std::vector<hipEvent_t> ev(hops);
      for (auto& e : ev) CK(hipEventCreateWithFlags(&e, v.flags));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < hops; ++i) {
          hipStream_t st = s[i & 1], nx = s[(i + 1) & 1];
          launch(st);
          CK(hipEventRecord(ev[i], st));
          CK(hipStreamWaitEvent(nx, ev[i], 0));
        }
        CK(hipDeviceSynchronize());
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rep) printf("kernel %-5s ping-pong, %-34s: %7.2f us per kernel\n", kern ? "20us" : "tiny", v.name, us / hops);
      }
      for (auto& e : ev) CK(hipEventDestroy(e));
    }
  }
  return 0;
}
