// Cholesky factorisation, triangular solves and log-determinant for gfx950.
//
// Replaces numpy.linalg.pinv (SVD, gp.py:181,400; experimentalDesign.py:268,280,826) and
// numpy.linalg.slogdet (LU, gp.py:434) by K = L L^T.  The factorisation is recursive
// (A11 -> L11; A21 <- A21 L11^-T; A22 <- A22 - L21 L21^T; recurse on A22) so that every flop
// outside the 128x128 leaves lands in the MFMA GEMM kernel with the largest possible inner
// dimension (the top-level SYRK has K = N/2), which keeps the read-modify-write traffic on C
// far below the HBM roof.  Leaves: one workgroup factors a 128x128 diagonal block in LDS and
// also inverts it, so that all triangular solves against a leaf are GEMMs with the inverse.
#include "gpx_internal.h"
#include <math.h>

namespace {

constexpr int NB = GPX_TILE;  // 128
constexpr int LS = NB + 1;    // LDS row stride (conflict-free column walks)

// ---- leaf: potf2 + trtri of one 128x128 diagonal block -------------------------------------------
// S (LDS): lower triangle = block being factored; strict upper + pad column = inverse, stored as
// X[i][c] (i >= c) at S[c][i+1].
__global__ __launch_bounds__(256) void leaf_kernel(double* __restrict__ A, int64_t ld, double* __restrict__ inv,
                                                   int64_t base_index, int64_t n_valid, int* __restrict__ info) {
  __shared__ double S[NB * LS];
  const int t = threadIdx.x;
  for (int idx = t; idx < NB * NB; idx += 256) {
    int i = idx >> 7, j = idx & 127;
    if (j <= i) S[i * LS + j] = A[(int64_t)i * ld + j];
  }
  __syncthreads();
  for (int j = 0; j < NB; ++j) {
    if (t == 0) {
      double dj = S[j * LS + j];
      if (!(dj > 0.0)) {  // non-positive or NaN pivot: record the first one, keep going with 1.0
        if (base_index + j < n_valid) atomicCAS(info, 0, (int)(base_index + j + 1));
        dj = 1.0;
      }
      S[j * LS + j] = sqrt(dj);
    }
    __syncthreads();
    const double piv = S[j * LS + j];
    if (t > j && t < NB) S[t * LS + j] = S[t * LS + j] / piv;
    __syncthreads();
    // trailing update: column c = j+1+(t&127), two threads per column interleave the rows
    const int c = j + 1 + (t & 127);
    if (c < NB) {
      const double lcj = S[c * LS + j];
      for (int i = c + (t >> 7); i < NB; i += 2) S[i * LS + c] = fma(-S[i * LS + j], lcj, S[i * LS + c]);
    }
    __syncthreads();
  }
  // write L back (zero the strict upper part of the diagonal block)
  for (int idx = t; idx < NB * NB; idx += 256) {
    int i = idx >> 7, j = idx & 127;
    A[(int64_t)i * ld + j] = (j <= i) ? S[i * LS + j] : 0.0;
  }
  // inverse: thread c owns column c of X = L^-1
  if (t < NB) {
    const int c = t;
    S[c * LS + c + 1] = 1.0 / S[c * LS + c];
  }
  __syncthreads();
  for (int i = 1; i < NB; ++i) {
    if (t < i) {  // c = t < i
      const int c = t;
      double s = 0.0;
      for (int k = c; k < i; ++k) s = fma(S[i * LS + k], S[c * LS + k + 1], s);
      S[c * LS + i + 1] = -s / S[i * LS + i];
    }
    // column c only reads its own earlier entries and L: no barrier needed between rows
  }
  __syncthreads();
  for (int idx = t; idx < NB * NB; idx += 256) {
    int i = idx >> 7, c = idx & 127;
    inv[idx] = (c <= i) ? S[c * LS + i + 1] : 0.0;
  }
}

// ---- TRSV pieces (potrs) ---------------------------------------------------------------------------
// y_k <- op(invL_kk) y_k for one 128 block; one workgroup of 128 threads
__global__ __launch_bounds__(128) void trsv_diag_kernel(const double* __restrict__ inv, double* __restrict__ y,
                                                        int transposed) {
  __shared__ double ys[NB];
  const int t = threadIdx.x;
  ys[t] = y[t];
  __syncthreads();
  double s = 0.0;
  if (!transposed) {
    for (int c = 0; c <= t; ++c) s = fma(inv[t * NB + c], ys[c], s);
  } else {
    for (int i = t; i < NB; ++i) s = fma(inv[i * NB + t], ys[i], s);
  }
  y[t] = s;
}

// forward: y[r] -= L[r][k0:k0+128] . yk   for rows r in [r0, r0+rows); one wave per row (8 rows per wave)
__global__ __launch_bounds__(256) void trsv_fwd_update_kernel(const double* __restrict__ L, int64_t ld,
                                                              const double* __restrict__ yk, double* __restrict__ y,
                                                              int64_t rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double2 v = reinterpret_cast<const double2*>(yk)[lane];
  const int64_t rbase = ((int64_t)blockIdx.x * 4 + wave) * 8;
  for (int q = 0; q < 8; ++q) {
    const int64_t r = rbase + q;
    if (r >= rows) break;
    const double2 l = reinterpret_cast<const double2*>(L + r * ld)[lane];
    double s = fma(l.x, v.x, l.y * v.y);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[r] -= s;
  }
}

// backward: z[j] -= sum_i L[i][j] * zk[i], i in the 128 rows of block k, j in [0, cols)
__global__ __launch_bounds__(256) void trsv_bwd_update_kernel(const double* __restrict__ Lrow, int64_t ld,
                                                              const double* __restrict__ zk, double* __restrict__ z,
                                                              int64_t cols) {
  __shared__ double zs[NB];
  if (threadIdx.x < NB) zs[threadIdx.x] = zk[threadIdx.x];
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= cols) return;
  double s = 0.0;
#pragma unroll 8
  for (int i = 0; i < NB; ++i) s = fma(Lrow[(int64_t)i * ld + j], zs[i], s);
  z[j] -= s;
}

__global__ __launch_bounds__(256) void logdet_kernel(const double* __restrict__ L, int64_t ld, int64_t n,
                                                     double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += log(L[i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = 2.0 * red[0];
}

inline int64_t split(int64_t n) { return (n / NB / 2) * NB; }  // n multiple of 128, n > 128

}  // namespace

int launch_leaf(gpx_ctx* ctx, double* A, int64_t ld, double* inv, int64_t base_index, int64_t n_valid) {
  ProfScope ps(ctx, GPX_PROF_LEAF, 2.0 * NB * NB * NB / 3.0, 0.0);
  hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), 0, ctx->stream, A, ld, inv, base_index, n_valid, ctx->d_info);
  GPX_HIP(hipGetLastError());
  return 0;
}

// X (m x n) <- X * L^-T, L lower n x n at (L, ldl) whose leaf inverses are invd[(row/128)] blocks
int chol_trsm_right(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* X, int64_t ldx,
                    int64_t m, int64_t n) {
  if (m == 0 || n == 0) return 0;
  if (n == NB) return launch_gemm(ctx, X, ldx, invd, NB, X, ldx, m, NB, NB, true, false, false);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(chol_trsm_right(ctx, L, ldl, invd, X, ldx, m, n1));
  // X2 -= X1 * L21^T
  GPX_TRY(launch_gemm(ctx, X, ldx, L + n1 * ldl, ldl, X + n1, ldx, m, n2, n1, true, true, false));
  return chol_trsm_right(ctx, L + n1 * ldl + n1, ldl, invd + (n1 / NB) * NB * NB, X + n1, ldx, m, n2);
}

// B (n x m) <- L^-1 B
int chol_trsm_left(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* B, int64_t ldb,
                   int64_t n, int64_t m) {
  if (m == 0 || n == 0) return 0;
  if (n == NB) return launch_gemm(ctx, invd, NB, B, ldb, B, ldb, NB, m, NB, false, false, false);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(chol_trsm_left(ctx, L, ldl, invd, B, ldb, n1, m));
  // B2 -= L21 * W1
  GPX_TRY(launch_gemm(ctx, L + n1 * ldl, ldl, B, ldb, B + n1 * ldb, ldb, n2, m, n1, false, true, false));
  return chol_trsm_left(ctx, L + n1 * ldl + n1, ldl, invd + (n1 / NB) * NB * NB, B + n1 * ldb, ldb, n2, m);
}

static int potrf_rec(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base, int64_t n_valid) {
  if (n == NB) return launch_leaf(ctx, A, ld, invd, base, n_valid);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(potrf_rec(ctx, A, ld, n1, invd, base, n_valid));
  double* A21 = A + n1 * ld;
  GPX_TRY(chol_trsm_right(ctx, A, ld, invd, A21, ld, n2, n1));
  GPX_TRY(launch_gemm(ctx, A21, ld, A21, ld, A21 + n1, ld, n2, n2, n1, true, true, true));
  return potrf_rec(ctx, A21 + n1, ld, n2, invd + (n1 / NB) * NB * NB, base + n1, n_valid);
}

int chol_potrf(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t n_valid) {
  GPX_ARG(n > 0 && n % NB == 0, "potrf: padded order must be a positive multiple of 128");
  GPX_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
  return potrf_rec(ctx, A, ld, n, invd, 0, n_valid);
}

int chol_trsv(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* y, int64_t n, bool transposed) {
  GPX_ARG(n % NB == 0, "trsv: padded length must be a multiple of 128");
  const int64_t nblk = n / NB;
  ProfScope ps(ctx, GPX_PROF_TRSV, (double)n * n, 4.0 * (double)n * n);
  if (!transposed) {
    for (int64_t k = 0; k < nblk; ++k) {
      double* yk = y + k * NB;
      hipLaunchKernelGGL(trsv_diag_kernel, dim3(1), dim3(128), 0, ctx->stream, invd + k * NB * NB, yk, 0);
      const int64_t rows = n - (k + 1) * NB;
      if (rows > 0)
        hipLaunchKernelGGL(trsv_fwd_update_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, ctx->stream,
                           L + (k + 1) * NB * ld + k * NB, ld, yk, y + (k + 1) * NB, rows);
    }
  } else {
    for (int64_t k = nblk - 1; k >= 0; --k) {
      double* zk = y + k * NB;
      hipLaunchKernelGGL(trsv_diag_kernel, dim3(1), dim3(128), 0, ctx->stream, invd + k * NB * NB, zk, 1);
      const int64_t cols = k * NB;
      if (cols > 0)
        hipLaunchKernelGGL(trsv_bwd_update_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream,
                           L + k * NB * ld, ld, zk, y, cols);
    }
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

int launch_logdet(gpx_ctx* ctx, const double* L, int64_t ld, int64_t n, double* d_out) {
  ProfScope ps(ctx, GPX_PROF_REDUCE, 0.0, 8.0 * (double)n);
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, ctx->stream, L, ld, n, d_out);
  GPX_HIP(hipGetLastError());
  return 0;
}
