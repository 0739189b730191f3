// f1: gradient of the integrated posterior variance w.r.t. the design (training) points -- gfx950.
//
// Replaces costFunctionGP_IVAR.derivative (experimentalDesign.py:168-172) -> GP.evaluateVarianceDerivative
// (gp.py:282-341), which builds an (N*d x M) matrix with O(N*d) dense N x N products in Python and then averages its
// columns.  With beta = K^-1 K(X,Z) (N x M) and S = beta beta^T (N x N) the averaged gradient collapses to
//
//   G[a,l] = -2 s / cl_l^2 * [ sum_n beta[a,n] (z_n,l - x_a,l) k(z_n, x_a)  +  sum_i S[a,i] (x_a,l - x_i,l) k(x_a, x_i) ],
//   dIVAR/dx_{a,l} = G[a,l] / M
//
// (squared-exponential kernel; the extra factor s is the reference's own convention: kernels.py:177 multiplies the
// kernel value, which already contains signalSize, by signalSize again -- kept for parity).  beta comes from two
// triangular solves against the factor (W = L^-1 K(X,Z), beta^T = W^T L^-1), S from one MFMA GEMM, and the two sums
// from one fused row kernel that recomputes the kernel values from the coordinates.
#include "gpx_internal.h"
#include <math.h>

namespace {

// one workgroup per design point a: T[a][l] = sum_n Bm[a][n] k(z_n,x_a)(z_n,l - x_a,l) + sum_i S[a][i] k0(x_a,x_i)(x_a,l - x_i,l)
__global__ __launch_bounds__(256) void ivar_grad_row_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                            const double* __restrict__ Z, int64_t m,
                                                            const double* __restrict__ Bm, int64_t ldb,
                                                            const double* __restrict__ S, int64_t lds_,
                                                            double scale_out, double* __restrict__ grad) {
  __shared__ double red[256];
  __shared__ double xa[GPX_MAXD];
  const int64_t a = blockIdx.x;
  const int d = kp.d;
  const int t = threadIdx.x;
  if (t < d) xa[t] = X[a * d + t];
  __syncthreads();
  double acc[GPX_MAXD];
#pragma unroll
  for (int l = 0; l < GPX_MAXD; ++l) acc[l] = 0.0;
  // evaluation points
  for (int64_t j = t; j < m; j += 256) {
    double r2 = 0.0;
    double dz[GPX_MAXD];
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l) {
      if (l < d) {
        dz[l] = Z[j * d + l] - xa[l];
        const double e = dz[l] * kp.scale[l];
        r2 = fma(e, e, r2);
      }
    }
    const double w = Bm[a * ldb + j] * kp.sig * exp(-0.5 * r2);
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l)
      if (l < d) acc[l] = fma(w, dz[l], acc[l]);
  }
  // other design points
  for (int64_t i = t; i < n; i += 256) {
    double r2 = 0.0;
    double dx[GPX_MAXD];
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l) {
      if (l < d) {
        dx[l] = xa[l] - X[i * d + l];
        const double e = dx[l] * kp.scale[l];
        r2 = fma(e, e, r2);
      }
    }
    const double w = S[a * lds_ + i] * kp.sig * exp(-0.5 * r2);
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l)
      if (l < d) acc[l] = fma(w, dx[l], acc[l]);
  }
  // deterministic tree reduction per coordinate
  for (int l = 0; l < d; ++l) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < GPX_MAXD; ++q)
      if (q == l) v = acc[q];
    red[t] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) red[t] += red[t + w];
      __syncthreads();
    }
    if (t == 0) grad[a * d + l] = scale_out * kp.sig * kp.scale[l] * kp.scale[l] * red[0];
    __syncthreads();
  }
}

struct Scratch {
  gpx_ctx* ctx;
  std::vector<std::pair<void*, int64_t>> bufs;
  explicit Scratch(gpx_ctx* c) : ctx(c) {}
  int get(int64_t bytes, void** out) {
    int r = gpx_dev_alloc(ctx, bytes, out);
    if (r == 0) bufs.push_back({*out, bytes});
    return r;
  }
  ~Scratch() {
    (void)hipDeviceSynchronize();
    for (auto& b : bufs) gpx_dev_release(ctx, b.first, b.second);
  }
};

}  // namespace

extern "C" {

int gpx_ivar_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                  const gpx_mat* Z, double* grad) {
  GPX_ARG(ctx && L && X && Z && grad, "NULL argument");
  GPX_ARG(L->factored && L->aux, "matrix has not been factored by gpx_potrf");
  GPX_ARG(kind == GPX_K_SE, "ivar_grad: only the squared-exponential kernel has a point derivative here "
                            "(the reference defines it for SE and 1-D Mehler only, kernels.py:146-181, 295-324)");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_ARG(X->cols == d && X->pcols == d && Z->cols == d && Z->pcols == d, "point sets must be unpadded (n x d)");
  const int64_t n = L->rows, np = L->prows, m = Z->rows;
  GPX_ARG(X->rows == n && m > 0, "X does not match the factor / no integration points");
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z));
  const int64_t mp = gpx_round_up(m, GPX_TILE);
  Scratch sc(ctx);
  void *pW, *pWt, *pS, *pg;
  GPX_TRY(sc.get(np * mp * 8, &pW));
  GPX_TRY(sc.get(mp * np * 8, &pWt));
  GPX_TRY(sc.get(np * np * 8, &pS));
  GPX_TRY(sc.get(n * d * 8, &pg));
  double* W = (double*)pW;
  double* Wt = (double*)pWt;
  // W = L^-1 K(X,Z)
  GPX_TRY(launch_kfill(ctx, kp, X->p, n, Z->p, m, 0, nullptr, 0, 0.0, W, np, mp, mp));
  GPX_TRY(chol_trsm_left(ctx, L->p, L->ld, L->aux, W, mp, np, mp));
  // beta^T = W^T L^-1   (mp x np)
  GPX_TRY(launch_transpose(ctx, W, np, mp, mp, Wt, np));
  GPX_TRY(chol_trsm_right_n(ctx, L->p, L->ld, L->aux, Wt, np, mp, np));
  // beta (np x mp) back in W;  S = beta beta^T
  GPX_TRY(launch_transpose(ctx, Wt, mp, np, np, W, mp));
  GPX_TRY(launch_gemm(ctx, W, mp, W, mp, (double*)pS, np, np, np, mp, true, false, false));
  {
    ProfScope ps(ctx, GPX_PROF_REDUCE, 0.0, 8.0 * ((double)n * m + (double)n * n));
    hipLaunchKernelGGL(ivar_grad_row_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, kp, X->p, n, Z->p, m,
                       (const double*)W, mp, (const double*)pS, np, -2.0 / (double)m, (double*)pg);
  }
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(grad, pg, (size_t)(n * d * 8), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

}  // extern "C"
