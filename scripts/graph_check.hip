// Does stream capture into a hipGraph work across the kinds of streams the context uses (priority streams, CU-masked streams),
// with event fork / join?  hipcc --offload-arch=gfx950 -O2 graph_check.hip -o graph_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k(double* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i] += 1.0; }
int run(int use_masked, int use_memcpy) {
  double* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithPriority(&s0, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi));
  if (use_masked) {
    std::vector<uint32_t> mask(8, 0xffffffffu); mask[0] = 0;
    CK(hipExtStreamCreateWithCUMask(&s2, 8, mask.data()));
  } else CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
  hipEvent_t f, j1, j2; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
  CK(hipEventRecord(f, s0)); CK(hipStreamWaitEvent(s1, f, 0)); CK(hipStreamWaitEvent(s2, f, 0));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, d, 0);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d, 1);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s2, d, 2);
  if (use_memcpy) CK(hipMemcpyAsync(d + 16, d + 32, 64, hipMemcpyDeviceToDevice, s1));
  if (use_memcpy) CK(hipMemsetAsync(d + 64, 0, 64, s2));
  CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2)); CK(hipStreamWaitEvent(s0, j1, 0)); CK(hipStreamWaitEvent(s0, j2, 0));
  hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
  size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n));
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int it = 0; it < 3; ++it) CK(hipGraphLaunch(ex, s0));
  CK(hipDeviceSynchronize());
  double h[3]; CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
  printf("masked=%d memcpy=%d: nodes=%zu results %.0f %.0f %.0f (expect 3 3 3)\n", use_masked, use_memcpy, n, h[0], h[1], h[2]);
  return 0;
}
// variants that the recorded panel loop produces: a forked stream with NO work of its own (fork wait, then the join record),
// a wait on an event that was last recorded OUTSIDE the capture, the same event recorded twice inside the capture
int run2(int empty_branch, int stale_wait, int rerecord) {
  double* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t f, j, st, rr; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&st, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&rr, hipEventDisableTiming));
  CK(hipEventRecord(st, s1)); CK(hipDeviceSynchronize());   // recorded outside any capture
  printf("empty_branch=%d stale_wait=%d rerecord=%d: ", empty_branch, stale_wait, rerecord); fflush(stdout);
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
  CK(hipEventRecord(f, s0)); CK(hipStreamWaitEvent(s1, f, 0));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, d, 0);
  if (stale_wait) CK(hipStreamWaitEvent(s0, st, 0));
  if (!empty_branch) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s1, d, 1);
  if (rerecord) { CK(hipEventRecord(rr, s0)); CK(hipStreamWaitEvent(s1, rr, 0)); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s0, d, 2);
                  CK(hipEventRecord(rr, s0)); CK(hipStreamWaitEvent(s1, rr, 0)); }
  CK(hipEventRecord(j, s1)); CK(hipStreamWaitEvent(s0, j, 0));
  hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
  size_t n = 0; CK(hipGraphGetNodes(g, nullptr, &n));
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s0)); CK(hipDeviceSynchronize());
  printf("ok, %zu nodes\n", n);
  return 0;
}
struct Big { const double* g[8]; int t[8][4]; int a[8]; long long b, c; int d, e; int sr[128]; int pf[129]; };
__global__ void kbig(const Big u, double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[3] += (double)u.pf[128] + (double)u.sr[5]; }
int run3() {
  double* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t s0; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  Big u; for (int i = 0; i < 128; ++i) u.sr[i] = i; for (int i = 0; i < 129; ++i) u.pf[i] = 2 * i;
  printf("kernel with a %zu-byte by-value argument in a capture: ", sizeof(Big)); fflush(stdout);
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
  hipLaunchKernelGGL(kbig, dim3(1), dim3(64), 0, s0, u, d);
  hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s0)); CK(hipGraphLaunch(ex, s0)); CK(hipDeviceSynchronize());
  double h; CK(hipMemcpy(&h, d + 3, 8, hipMemcpyDeviceToHost));
  printf("ok, result %.0f (expect 522)\n", h);
  return 0;
}
// the pattern that makes hipStreamEndCapture SEGFAULT on ROCm 7.2 (argument 98; run it alone): a forked, otherwise empty
// side stream waits for an event it has just recorded ITSELF
int run4() {
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t f, e, j; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  printf("side stream waits for its own event inside a capture: "); fflush(stdout);
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
  CK(hipEventRecord(f, s0)); CK(hipStreamWaitEvent(s1, f, 0));
  CK(hipEventRecord(e, s1)); CK(hipStreamWaitEvent(s1, e, 0));
  CK(hipEventRecord(j, s1)); CK(hipStreamWaitEvent(s0, j, 0));
  hipGraph_t g; CK(hipStreamEndCapture(s0, &g));
  printf("survived\n");
  return 0;
}
int main(int argc, char** argv) {
  if (argc > 1 && atoi(argv[1]) == 98) return run4();
  if (argc > 1 && atoi(argv[1]) == 99) return run3();
  if (argc > 1) { int v = atoi(argv[1]); return run2(v & 1, (v >> 1) & 1, (v >> 2) & 1); }
  for (int m = 0; m < 2; ++m) for (int c = 0; c < 2; ++c) { fflush(stdout); if (run(m, c)) return 1; }
  return 0;
}
