// Deterministic column reductions over the (training point x evaluation point) matrices.
//
//   v != nullptr : out[j] = sum_i B[i][j] * v[i]      posterior mean  k_j^T alpha       (gp.py:137)
//   v == nullptr : out[j] = sum_i B[i][j]^2           k_j^T K^-1 k_j = |L^-1 k_j|^2     (gp.py:142-144, 253-255)
//
// Fixed summation order (row chunks in order, then chunks in order) so that results -- and the
// arg-max / arg-min selections built on them -- do not depend on scheduling.
#include "gpx_internal.h"

namespace {

// Few-row blocks (the transposed products of the backward substitution: 1024 x 1024 .. 4096 x 4096) are latency-bound: with
// 128-row chunks a 1024-row block was 32 workgroups walking 128 rows each (31 us for 8 MB, profiles/r02_potrs.txt); 16-row
// chunks make it 256 workgroups.  Large reductions (rows >= 16 * 2048 / column blocks) are not affected.
constexpr int ROWS_PER_CHUNK_MIN = 16;

__global__ __launch_bounds__(256) void colreduce_kernel(const double* __restrict__ B, int64_t ld, int64_t rows,
                                                        int64_t chunk, const double* __restrict__ v,
                                                        double* __restrict__ partial, int64_t pcols) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > rows) r1 = rows;
  if (j >= pcols) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const double* p = B + r0 * ld + j;
  int64_t i = r0;
  if (v) {
    for (; i + 4 <= r1; i += 4, p += 4 * ld) {
      s0 = fma(p[0], v[i], s0);
      s1 = fma(p[ld], v[i + 1], s1);
      s2 = fma(p[2 * ld], v[i + 2], s2);
      s3 = fma(p[3 * ld], v[i + 3], s3);
    }
    for (; i < r1; ++i, p += ld) s0 = fma(p[0], v[i], s0);
  } else {
    for (; i + 4 <= r1; i += 4, p += 4 * ld) {
      double a = p[0], b = p[ld], c = p[2 * ld], d = p[3 * ld];
      s0 = fma(a, a, s0);
      s1 = fma(b, b, s1);
      s2 = fma(c, c, s2);
      s3 = fma(d, d, s3);
    }
    for (; i < r1; ++i, p += ld) s0 = fma(p[0], p[0], s0);
  }
  partial[(int64_t)blockIdx.y * pcols + j] = (s0 + s1) + (s2 + s3);
}

// out[j] = sum of the partials, or (subtract) out[j] -= sum: the transposed-GEMV update of the triangular sweeps in one pass
__global__ __launch_bounds__(256) void colreduce_final_kernel(const double* __restrict__ partial, int64_t nchunk,
                                                              int64_t pcols, double* __restrict__ out, int subtract) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= pcols) return;
  // four interleaved partial sums, combined in a fixed order: the loads of a round are independent (a single running sum
  // waits for every load in turn: 19 us for 8 partials)
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  const double* p = partial + j;
  int64_t c = 0;
  for (; c + 4 <= nchunk; c += 4, p += 4 * pcols) {
    const double a = p[0], b = p[pcols], d = p[2 * pcols], e = p[3 * pcols];
    s0 += a;
    s1 += b;
    s2 += d;
    s3 += e;
  }
  for (; c < nchunk; ++c, p += pcols) s0 += p[0];
  const double s = (s0 + s1) + (s2 + s3);
  out[j] = subtract ? out[j] - s : s;
}

__global__ __launch_bounds__(256) void sum_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

inline void plan(int64_t rows, int64_t pcols, int64_t* chunk, int64_t* nchunk) {
  int64_t colblocks = (pcols + 255) / 256;
  int64_t want = 2048 / colblocks;
  if (want < 1) want = 1;
  int64_t maxc = (rows + ROWS_PER_CHUNK_MIN - 1) / ROWS_PER_CHUNK_MIN;
  if (maxc < 1) maxc = 1;
  if (want > maxc) want = maxc;
  *chunk = (rows + want - 1) / want;
  if (*chunk < 1) *chunk = 1;
  *nchunk = (rows + *chunk - 1) / *chunk;
  if (*nchunk < 1) *nchunk = 1;
}

}  // namespace

// Partial sums a reduction of at most `rows` x `pcols` needs.  Callers size ONE buffer for a family of launches (the
// recursive sweeps reduce sub-blocks of every shape), so this is a bound that is monotone in both arguments, not the exact
// need of the (rows, pcols) launch: chunks <= 2048 / column blocks and <= rows / ROWS_PER_CHUNK_MIN, columns <= 256 per
// column block.  (The exact product is NOT monotone: 5120 x 4096 needs more than 9216 x 9216.)
int64_t colreduce_partial_elems(int64_t rows, int64_t pcols) {
  const int64_t by_rows = ((rows + ROWS_PER_CHUNK_MIN - 1) / ROWS_PER_CHUNK_MIN + 1) * pcols;
  const int64_t cap = (int64_t)2048 * 256 + pcols;
  return by_rows < cap ? by_rows : cap;
}

int launch_colreduce(gpx_ctx* ctx, const double* B, int64_t ld, int64_t rows, int64_t pcols, const double* v,
                     double* out, double* d_partial, int subtract) {
  int64_t chunk, nchunk;
  plan(rows, pcols, &chunk, &nchunk);
  GPX_ARG(nchunk * pcols <= colreduce_partial_elems(rows, pcols), "colreduce: partial-sum plan exceeds its own bound");
  ProfScope ps(ctx, GPX_PROF_REDUCE, 2.0 * (double)rows * pcols, 8.0 * (double)rows * pcols);
  dim3 grid((unsigned)((pcols + 255) / 256), (unsigned)nchunk);
  hipLaunchKernelGGL(colreduce_kernel, grid, dim3(256), 0, ctx->stream, B, ld, rows, chunk, v, d_partial, pcols);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(grid.x), dim3(256), 0, ctx->stream, d_partial, nchunk, pcols, out,
                     subtract);
  GPX_HIP(hipGetLastError());
  return 0;
}

namespace {

// out[r] = sum_c B[r][c] * v[c]  (v != nullptr)  or  sum_c B[r][c]^2  (v == nullptr)  or  sum_c v[c] B[r][c]^2
// (v != nullptr, weighted_squares): one workgroup per row, every thread a fixed strided subset of the columns,
// fixed-shape tree over the 256 partial sums -> bit-reproducible
__global__ __launch_bounds__(256) void rowreduce_kernel(const double* __restrict__ B, int64_t ld, int64_t cols,
                                                        const double* __restrict__ v, double* __restrict__ out,
                                                        int weighted_squares) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  const double* row = B + (int64_t)blockIdx.x * ld;
  double s0 = 0.0, s1 = 0.0;
  if (v && weighted_squares) {
    for (int64_t c = 2 * t; c < cols; c += 512) {
      const double2 b = *reinterpret_cast<const double2*>(row + c);
      const double2 w = *reinterpret_cast<const double2*>(v + c);
      s0 = fma(b.x * w.x, b.x, s0);
      s1 = fma(b.y * w.y, b.y, s1);
    }
  } else if (v) {
    for (int64_t c = 2 * t; c < cols; c += 512) {
      const double2 b = *reinterpret_cast<const double2*>(row + c);
      const double2 w = *reinterpret_cast<const double2*>(v + c);
      s0 = fma(b.x, w.x, s0);
      s1 = fma(b.y, w.y, s1);
    }
  } else {
    for (int64_t c = 2 * t; c < cols; c += 512) {
      const double2 b = *reinterpret_cast<const double2*>(row + c);
      s0 = fma(b.x, b.x, s0);
      s1 = fma(b.y, b.y, s1);
    }
  }
  red[t] = s0 + s1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  if (t == 0) out[blockIdx.x] = red[0];
}

}  // namespace

// rows x cols (cols even, v padded to cols) -> out[rows]
int launch_rowreduce(gpx_ctx* ctx, const double* B, int64_t ld, int64_t rows, int64_t cols, const double* v,
                     double* out, int weighted_squares) {
  if (rows <= 0) return 0;
  GPX_ARG(cols % 2 == 0 && ld % 2 == 0, "rowreduce: even column count and leading dimension");
  ProfScope ps(ctx, GPX_PROF_REDUCE, 2.0 * (double)rows * cols, 8.0 * (double)rows * cols);
  hipLaunchKernelGGL(rowreduce_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->stream, B, ld, cols, v, out,
                     weighted_squares);
  GPX_HIP(hipGetLastError());
  return 0;
}

int launch_sum(gpx_ctx* ctx, const double* x, int64_t n, double* d_out) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, ctx->stream, x, n, d_out);
  GPX_HIP(hipGetLastError());
  return 0;
}

// test hook (host logic only, no device work): chunk count a rows x pcols reduction launches with, and the buffer bound
extern "C" int gpx_dbg_colreduce_plan(int64_t rows, int64_t pcols, int64_t* nchunk, int64_t* bound_elems) {
  if (rows < 1 || pcols < 1 || !nchunk || !bound_elems) return -1;
  int64_t chunk;
  plan(rows, pcols, &chunk, nchunk);
  *bound_elems = colreduce_partial_elems(rows, pcols);
  return 0;
}
