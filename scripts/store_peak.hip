// Microbenchmark: sustained HBM store / load+store bandwidth on gfx950 for an 8 GiB fp64 matrix (the kfill ceiling).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// each workgroup writes TILE x TILE doubles of a row-major n x n matrix (the kfill store pattern), 16 B per lane
template <int TILE>
__global__ __launch_bounds__(256) void tile_store(double* __restrict__ out, int64_t n, double v) {
  const int tiles = (int)(n / TILE);
  const int64_t r0 = (int64_t)(blockIdx.x / tiles) * TILE, c0 = (int64_t)(blockIdx.x % tiles) * TILE;
  constexpr int LPR = TILE / 2;          // lanes per row (2 doubles each)
  constexpr int RPP = 256 / LPR;         // rows per pass
  const int t = threadIdx.x;
  for (int r = t / LPR; r < TILE; r += RPP) {
    double2 x = {v + r, v + t};
    *(double2*)(out + (r0 + r) * n + c0 + 2 * (t % LPR)) = x;
  }
}
template <int TILE>
__global__ __launch_bounds__(256) void tile_store_nt(double* __restrict__ out, int64_t n, double v) {
  const int tiles = (int)(n / TILE);
  const int64_t r0 = (int64_t)(blockIdx.x / tiles) * TILE, c0 = (int64_t)(blockIdx.x % tiles) * TILE;
  constexpr int LPR = TILE / 2;
  constexpr int RPP = 256 / LPR;
  const int t = threadIdx.x;
  for (int r = t / LPR; r < TILE; r += RPP) {
    double* p = out + (r0 + r) * n + c0 + 2 * (t % LPR);
    __builtin_nontemporal_store(v + r, p);
    __builtin_nontemporal_store(v + t, p + 1);
  }
}
// 64 rows x 128 columns per workgroup: every wave stores whole 1 KiB row segments
__global__ __launch_bounds__(256) void tile_store_64x128(double* __restrict__ out, int64_t n, double v) {
  const int tiles = (int)(n / 128);
  const int64_t r0 = (int64_t)(blockIdx.x / tiles) * 64, c0 = (int64_t)(blockIdx.x % tiles) * 128;
  const int t = threadIdx.x;
  for (int r = t / 64; r < 64; r += 4) {
    double2 x = {v + r, v + t};
    *(double2*)(out + (r0 + r) * n + c0 + 2 * (t % 64)) = x;
  }
}
__global__ __launch_bounds__(256) void flat_store_nt(double* __restrict__ out, int64_t n2, double v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    __builtin_nontemporal_store(v, out + 2 * i);
    __builtin_nontemporal_store(v + 1, out + 2 * i + 1);
  }
}
// the store pattern of kfill v3: MODE 0 = direct quadrant stores (4 rows x 256 B per wave instruction, 16 B lanes),
// MODE 1 = mirror stores (2 rows x 256 B per wave instruction, 8 B lanes), MODE 2 = both (lower tile + its mirror)
template <int MODE>
__global__ __launch_bounds__(256) void kfill_pattern(double* __restrict__ out, int64_t n, double v) {
  int ti, tj;
  const int tiles = (int)(n / 64);
  if (MODE == 2) {
    const int w = blockIdx.x;
    ti = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= w) ++ti;
    while (ti * (ti + 1) / 2 > w) --ti;
    tj = w - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.x / tiles; tj = blockIdx.x % tiles;
  }
  const int64_t i0 = (int64_t)ti * 64, j0 = (int64_t)tj * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, g = lane >> 4, q = lane & 15;
  if (MODE == 0 || MODE == 2) {
    const int c0 = wn * 32 + 2 * q;
    for (int mi = 0; mi < 2; ++mi)
      for (int vv = 0; vv < 4; ++vv) {
        const int r = wm * 32 + mi * 16 + g + 4 * vv;
        *(double2*)(out + (i0 + r) * n + j0 + c0) = double2{v + r, v + t};
      }
  }
  if (MODE == 1 || (MODE == 2 && ti != tj)) {
    const int tx = t & 31, ty = t >> 5;
    for (int half = 0; half < 2; ++half)
      for (int qq = 0; qq < 8; ++qq) {
        const int c = ty + 8 * qq;
        out[(j0 + c) * n + i0 + 32 * half + tx] = v + c;
      }
  }
}
// flat grid-stride store: every lane 16 B, fully linear
__global__ __launch_bounds__(256) void flat_store(double2* __restrict__ out, int64_t n2, double v) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) out[i] = double2{v, v + 1};
}
__global__ __launch_bounds__(256) void flat_copy(const double2* __restrict__ in, double2* __restrict__ out, int64_t n2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void flat_read(const double2* __restrict__ in, double* __restrict__ out, int64_t n2) {
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) { double2 x = in[i]; s += x.x + x.y; }
  if (s == 1.2345) out[0] = s;
}
template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  const int64_t n = 32768; const size_t bytes = (size_t)n * n * 8;
  double *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  float ms;
  ms = timeit([&] { hipMemsetAsync(a, 0, bytes, 0); }, 5);
  printf("hipMemsetAsync 8 GiB            : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  for (int g : {2048, 8192, 65536}) {
    ms = timeit([&] { hipLaunchKernelGGL(flat_store, dim3(g), dim3(256), 0, 0, (double2*)a, (int64_t)(n * n / 2), 1.0); }, 5);
    printf("flat_store grid=%6d          : %.3f ms  %.2f TB/s\n", g, ms, bytes / ms / 1e9);
  }
  ms = timeit([&] { hipLaunchKernelGGL(tile_store<64>, dim3((n / 64) * (n / 64)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("tile_store<64>  (512 B rows)    : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(tile_store<128>, dim3((n / 128) * (n / 128)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("tile_store<128> (1 KiB rows)    : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(tile_store<256>, dim3((n / 256) * (n / 256)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("tile_store<256> (2 KiB rows)    : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(tile_store_nt<64>, dim3((n / 64) * (n / 64)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("tile_store_nt<64>               : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(tile_store_nt<128>, dim3((n / 128) * (n / 128)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("tile_store_nt<128>              : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(tile_store_64x128, dim3((n / 64) * (n / 128)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("tile_store 64x128               : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(flat_store_nt, dim3(65536), dim3(256), 0, 0, a, (int64_t)(n * n / 2), 1.0); }, 5);
  printf("flat_store_nt grid=65536        : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(kfill_pattern<0>, dim3((n / 64) * (n / 64)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("kfill direct pattern            : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(kfill_pattern<1>, dim3((n / 64) * (n / 64)), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("kfill mirror pattern            : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(kfill_pattern<2>, dim3((n / 64) * (n / 64 + 1) / 2), dim3(256), 0, 0, a, n, 1.0); }, 5);
  printf("kfill lower+mirror pattern      : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(flat_read, dim3(8192), dim3(256), 0, 0, (const double2*)a, b, (int64_t)(n * n / 2)); }, 5);
  printf("flat_read  8 GiB                : %.3f ms  %.2f TB/s\n", ms, bytes / ms / 1e9);
  ms = timeit([&] { hipLaunchKernelGGL(flat_copy, dim3(8192), dim3(256), 0, 0, (const double2*)a, (double2*)b, (int64_t)(n * n / 2)); }, 5);
  printf("flat_copy  8+8 GiB              : %.3f ms  %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
  return 0;
}
