// How do the bits of a hipExtStreamCreateWithCUMask mask map to physical CUs on gfx950?  Counts distinct (xcc, se, sh, cu).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <set>
__global__ __launch_bounds__(256) void where(unsigned* out, long long cycles) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = 0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID
    out[blockIdx.x] = (xcc & 15) << 16 | (hw & 0xffff);
  }
}
int main() {
  unsigned* out; (void)hipMalloc(&out, 4096 * 4);
  (void)hipFuncSetAttribute((const void*)where, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  const int cus = 256, words = 8;
  auto run = [&](const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    (void)hipMemsetAsync(out, 0xff, 4096 * 4, s);
    hipLaunchKernelGGL(where, dim3(1024), dim3(256), 140 * 1024, s, out, 240000LL);  // 1 WG per CU at a time
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(1024); (void)hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost);
    std::set<unsigned> cuset; std::set<unsigned> xccs;
    for (unsigned v : h) { cuset.insert(((v >> 16) << 16) | (v & 0xff00)); xccs.insert(v >> 16); }
    int bits = 0; for (auto w : mask) bits += __builtin_popcount(w);
    printf("%-28s bits=%3d -> distinct CUs used %3zu on %zu XCCs\n", name, bits, cuset.size(), xccs.size());
    (void)hipStreamDestroy(s);
  };
  std::vector<uint32_t> all(words, 0xffffffffu), lo(words, 0), hi(words, 0), ev(words, 0x55555555u), m16(words, 0), first32(words, 0), w0(words, 0);
  for (int i = 0; i < 128; ++i) lo[i / 32] |= 1u << (i % 32);
  for (int i = 128; i < 256; ++i) hi[i / 32] |= 1u << (i % 32);
  for (int i = 0; i < cus; ++i) if (i % 16 != 0) m16[i / 32] |= 1u << (i % 32);
  first32[0] = 0xffffffffu;
  for (int i = 0; i < 8; ++i) w0[0] |= 1u << i;
  run("all 256", all); run("bits 0..127", lo); run("bits 128..255", hi); run("even bits", ev); run("all but i%16==0", m16);
  run("bits 0..31", first32); run("bits 0..7", w0);
  std::vector<uint32_t> inv16(words, 0); for (int i = 0; i < cus; ++i) if (i % 16 == 0) inv16[i / 32] |= 1u << (i % 32);
  run("only i%16==0", inv16);
  return 0;
}
