// Semantics check of the 64-bit DPP forms the leaf kernel relies on (gfx950): row_newbcast:N = lane N of each 16-lane row.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const double* in, double* out) {
  const int t = threadIdx.x;
  double a = in[t], b = in[t + 64], c = in[t + 128];
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(a), "v"(b));
  double p;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(p) : "v"(a));
  out[t] = c;
  out[t + 64] = p;
}
int main() {
  double h[192], r[128];
  for (int i = 0; i < 192; ++i) h[i] = 1.0 + 0.37 * i + 0.001 * i * i;
  double *d, *o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 64; ++t) {
    const int row = t / 16;
    const double want_c = __builtin_fma(h[row * 16 + 3], h[64 + t], h[128 + t]);
    const double want_p = h[row * 16 + 5];
    if (r[t] != want_c || r[64 + t] != want_p) { if (bad < 5) printf("lane %d: fmac %.17g want %.17g; mov %.17g want %.17g\n", t, r[t], want_c, r[64 + t], want_p); ++bad; }
  }
  printf("dpp row_newbcast check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
