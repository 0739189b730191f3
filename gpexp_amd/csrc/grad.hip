// f1: gradients of the posterior variance w.r.t. point locations -- gfx950.
//
// Replaces GP.evaluateVarianceDerivative (gp.py:282-341: an (N*d x M) matrix built from O(N*d) dense N x N products in
// Python), GP.evaluateVarianceDerivWRTnewpt (gp.py:261-280) and costFunctionGP_IVAR.derivative (experimentalDesign.py:
// 168-179: the column mean of the former), for the two kernels the reference differentiates: the squared exponential
// (kernels.py:146-181) and the 1-D Mehler kernel (kernels.py:295-324), with or without a heteroscedastic noise model
// (gp.py:314-320).  Nothing N x N ever goes to the host.
//
// Notation: dk(u, p)[l] = d k(u, p) / d u_l in the REFERENCE's convention --
//     SE        -s (u_l - p_l) / cl_l^2 * k(u, p)   with k already containing s: the doubled signalSize of kernels.py:177
//     Mehler1D  -(t^2 u - t p) / (1 - t^2) * k(u, p)
// beta = K^-1 K(X, Z) (N x M, two triangular solves against the factor), nd[j][l] = d noise(x_j) / d x_jl (optional).
// With c_jl[i] = dk(x_j, x_i)[l] + [x_i == x_j] nd[j][l]   (gp.py:310, 316: `indUse` marks coincident training points)
// and  T_j[m][l] = -dk(z_m, x_j)[l]                        (gp.py:312)
//     d var(z_m) / d x_jl = beta[j][m] * ( -2 T_j[m][l] + 2 sum_i c_jl[i] beta[i][m] - c_jl[j] beta[j][m] )     (gp.py:322-338)
// The sum over i is one MFMA GEMM per coordinate (A_l = [c_jl[i]] is N x N, C_l = A_l beta); its column mean -- the IVAR
// gradient -- collapses to one GEMM S = beta beta^T and a fused row kernel:
//     d IVAR / d x_jl = 1/M ( -2 sum_m beta[j][m] T_j[m][l] + 2 sum_i c_jl[i] S[j][i] - c_jl[j] S[j][j] ).
#include "gpx_internal.h"
#include <math.h>
#include <stdlib.h>

namespace {

// k(a, b) and, through `dk`, the reference's point derivative (SE any d; Mehler d == 1)
__device__ __forceinline__ double kval(const KParams& kp, const double* __restrict__ a, const double* __restrict__ b) {
  if (kp.kind == GPX_K_MEHLER) {
    double pa = 0.0, pb = 0.0, cr = 0.0;
    for (int k = 0; k < kp.d; ++k) {
      pa = fma(kp.c1[k] * a[k], a[k], pa);
      pb = fma(kp.c1[k] * b[k], b[k], pb);
      cr = fma(kp.c2[k] * a[k], b[k], cr);
    }
    return kp.sig * exp(-(pa + pb - cr));
  }
  double r2 = 0.0;
  for (int k = 0; k < kp.d; ++k) {
    const double e = (a[k] - b[k]) * kp.scale[k];  // difference first (kernels.py:121-122)
    r2 = fma(e, e, r2);
  }
  return kp.sig * exp(-0.5 * r2);
}

// d k(u, p) / d u_l given kv = k(u, p)
__device__ __forceinline__ double dk(const KParams& kp, double ul, double pl, int l, double kv) {
  if (kp.kind == GPX_K_MEHLER) return -(2.0 * kp.c1[l] * ul - kp.c2[l] * pl) * kv;
  return -kp.sig * (ul - pl) * kp.scale[l] * kp.scale[l] * kv;
}

// coincident training points: np.linalg.norm(pp - p) < 1e-10 (gp.py:308)
__device__ __forceinline__ bool same_point(const double* __restrict__ a, const double* __restrict__ b, int d) {
  double s = 0.0;
  for (int k = 0; k < d; ++k) s = fma(a[k] - b[k], a[k] - b[k], s);
  return sqrt(s) < 1e-10;
}

// ---- IVAR gradient: one workgroup per design point a ------------------------------------------------------------------
// grad[a][l] = 1/M ( 2 sum_m Bm[a][m] dk(z_m, x_a)[l] + 2 sum_i c_al[i] S[a][i] - c_al[a] S[a][a] )
// S (n x n, row stride ld, n a multiple of 32): the strictly upper triangle <- the transpose of the lower one.  One 32 x 32
// tile pair per workgroup (blockIdx.x enumerates the tiles on / below the diagonal); a diagonal tile mirrors inside itself.
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ S, int64_t ld, int nt) {
  __shared__ double tile[32][33];
  // tile (ti, tj), tj <= ti, from the linear index: ti (ti + 1) / 2 + tj
  int ti = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((int64_t)(ti + 1) * (ti + 2) / 2 <= (int64_t)blockIdx.x) ++ti;
  while ((int64_t)ti * (ti + 1) / 2 > (int64_t)blockIdx.x) --ti;
  const int tj = (int)((int64_t)blockIdx.x - (int64_t)ti * (ti + 1) / 2);
  if (ti >= nt) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t i0 = (int64_t)ti * 32, j0 = (int64_t)tj * 32;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = S[(i0 + r) * ld + j0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    // element (j0 + r, i0 + tx) = S[i0 + tx][j0 + r]; on a diagonal tile only the part above the diagonal is written
    if (ti != tj || tx > r) S[(j0 + r) * ld + i0 + tx] = tile[tx][r];
  }
}

__global__ __launch_bounds__(256) void ivar_grad_row_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                            const double* __restrict__ Z, int64_t m,
                                                            const double* __restrict__ Bm, int64_t ldb,
                                                            const double* __restrict__ S, int64_t lds_,
                                                            const double* __restrict__ nd, double inv_m,
                                                            double* __restrict__ grad) {
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
  // evaluation points: -2 beta T = 2 beta dk(z, x_a)
  for (int64_t j = t; j < m; j += 256) {
    const double* z = Z + j * d;
    const double w = 2.0 * Bm[a * ldb + j] * kval(kp, z, xa);
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l)
      if (l < d) acc[l] += dk(kp, z[l], xa[l], l, w);
  }
  // other design points: 2 c_al[i] S[a][i]; the diagonal entry counts once (2 c S - c S)
  for (int64_t i = t; i < n; i += 256) {
    const double* xi = X + i * d;
    const double sai = (i == a ? 1.0 : 2.0) * S[a * lds_ + i];
    const double w = sai * kval(kp, xa, xi);
    const bool dup = nd != nullptr && (i == a || same_point(xa, xi, d));
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l)
      if (l < d) {
        acc[l] += dk(kp, xa[l], xi[l], l, w);
        if (dup) acc[l] += sai * nd[a * d + l];
      }
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
    if (t == 0) grad[a * d + l] = inv_m * red[0];
    __syncthreads();
  }
}

// The squared exponential's derivative is LINEAR in the coordinate difference: dk(u, p)[l] = c_l (u_l - p_l) k(u, p) with
// c_l = -s / cl_l^2.  So a sum of w_j dk(u_j, p)[l] over pairs is c_l * sum_j w_j (u_jl - p_l): per pair the differences the
// distance needs anyway, one exp and d fused multiply-adds -- instead of d derivative evaluations, which the generic kernels
// above (kept for the Mehler kernel) unroll to GPX_MAXD = 32 predicated ones: 10.7 ms of gpx_ivar_grad's 120 at N = 8192,
// M = 32768, d = 8 (3.4e8 pairs) against 1.3 ms of fp64 issue.  DMAX = d rounded up to a power of two.
template <int DMAX>
__device__ __forceinline__ double se_pair(const KParams& kp, const double* __restrict__ u, const double (&p)[DMAX],
                                          double (&diff)[DMAX]) {
  double r2 = 0.0;
#pragma unroll
  for (int l = 0; l < DMAX; ++l) {
    diff[l] = 0.0;
    if (l < kp.d) {
      diff[l] = u[l] - p[l];
      const double e = diff[l] * kp.scale[l];
      r2 = fma(e, e, r2);
    }
  }
  return kp.sig * exp(-0.5 * r2);
}

// c_l * s_l for every coordinate, reduced over the workgroup in the fixed tree order, written to out[0 .. d)
template <int DMAX>
__device__ __forceinline__ void se_finish(const KParams& kp, const double (&s1)[DMAX], double extra_w, const double* extra,
                                          double scale_out, double* red, double* __restrict__ out) {
  const int t = threadIdx.x;
#pragma unroll
  for (int l = 0; l < DMAX; ++l) {
    if (l < kp.d) {   // (uniform)
      double v = -kp.sig * kp.scale[l] * kp.scale[l] * s1[l];
      if (extra) v += extra_w * extra[l];
      red[t] = v;
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
      }
      if (t == 0) out[l] = scale_out * red[0];
      __syncthreads();
    }
  }
}

template <int DMAX>
__global__ __launch_bounds__(256) void ivar_grad_row_se_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                               const double* __restrict__ Z, int64_t m,
                                                               const double* __restrict__ Bm, int64_t ldb,
                                                               const double* __restrict__ S, int64_t lds_,
                                                               const double* __restrict__ nd, double inv_m,
                                                               double* __restrict__ grad, int64_t a0) {
  // a0: design point of workgroup 0 -- Bm, S, nd and grad hold the rows a0 .. only (gpx_ivar_grad_rows; 0 = all points)
  __shared__ double red[256];
  const int64_t a = a0 + blockIdx.x;
  Bm -= a0 * ldb;
  S -= a0 * lds_;
  grad -= a0 * kp.d;
  const int d = kp.d, t = threadIdx.x;
  double xa[DMAX], s1[DMAX], diff[DMAX];
#pragma unroll
  for (int l = 0; l < DMAX; ++l) {
    xa[l] = l < d ? X[a * d + l] : 0.0;
    s1[l] = 0.0;
  }
  // evaluation points: 2 beta[a][j] dk(z_j, x_a)
  for (int64_t j = t; j < m; j += 256) {
    const double w = 2.0 * Bm[a * ldb + j] * se_pair<DMAX>(kp, Z + j * d, xa, diff);
#pragma unroll
    for (int l = 0; l < DMAX; ++l) s1[l] = fma(w, diff[l], s1[l]);
  }
  // other design points: 2 S[a][i] dk(x_a, x_i) (the diagonal entry once); diff = x_i - x_a = -(u - p)
  double dups = 0.0;
  for (int64_t i = t; i < n; i += 256) {
    const double sai = (i == a ? 1.0 : 2.0) * S[a * lds_ + i];
    const double w = sai * se_pair<DMAX>(kp, X + i * d, xa, diff);
#pragma unroll
    for (int l = 0; l < DMAX; ++l) s1[l] = fma(-w, diff[l], s1[l]);
    if (nd != nullptr) {
      double q = 0.0;
#pragma unroll
      for (int l = 0; l < DMAX; ++l) q = fma(diff[l], diff[l], q);
      if (i == a || sqrt(q) < 1e-10) dups += sai;   // coincident training points (gp.py:308)
    }
  }
  se_finish<DMAX>(kp, s1, dups, nd ? nd + a * d : nullptr, inv_m, red, grad + a * d);
}

// out[m][l] = -2 sum_j dk(z_m, x_j)[l] beta[j][m] from beta^T (row m contiguous over j): one workgroup per evaluation point
template <int DMAX>
__global__ __launch_bounds__(256) void var_grad_newpt_se_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                                const double* __restrict__ Z,
                                                                const double* __restrict__ betaT, int64_t ldt, int64_t col0,
                                                                double* __restrict__ out) {
  __shared__ double red[256];
  const int64_t mm = blockIdx.x;
  const int d = kp.d, t = threadIdx.x;
  double zs[DMAX], s1[DMAX], diff[DMAX];
#pragma unroll
  for (int l = 0; l < DMAX; ++l) {
    zs[l] = l < d ? Z[(col0 + mm) * d + l] : 0.0;
    s1[l] = 0.0;
  }
  for (int64_t j = t; j < n; j += 256) {
    const double w = -2.0 * betaT[mm * ldt + j] * se_pair<DMAX>(kp, X + j * d, zs, diff);   // diff = x_j - z = -(u - p)
#pragma unroll
    for (int l = 0; l < DMAX; ++l) s1[l] = fma(-w, diff[l], s1[l]);
  }
  se_finish<DMAX>(kp, s1, 0.0, nullptr, 1.0, red, out + (col0 + mm) * d);
}

// ---- full matrix: A_l[j][i] = c_jl[i] (np x np, zero outside the n x n block) ----------------------------------------
__global__ __launch_bounds__(256) void dcov_kernel(KParams kp, const double* __restrict__ X, int64_t n, int l,
                                                   const double* __restrict__ nd, double* __restrict__ A, int64_t lda,
                                                   int64_t np) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (i >= np) return;
  double v = 0.0;
  if (i < n && j < n) {
    const int d = kp.d;
    const double* xj = X + j * d;
    const double* xi = X + i * d;
    v = dk(kp, xj[l], xi[l], l, kval(kp, xj, xi));
    if (nd != nullptr && (i == j || same_point(xj, xi, d))) v += nd[j * d + l];
  }
  A[j * lda + i] = v;
}

// out[j][m] = beta[j][m] * ( 2 dk(z_m, x_j)[l] (+ 2 dkb[j][l]) + 2 C[j][m] - A[j][j] beta[j][m] )  for coordinate l
__global__ __launch_bounds__(256) void var_grad_finish_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                              const double* __restrict__ Z, int64_t m, int l,
                                                              const double* __restrict__ beta, int64_t ldb,
                                                              const double* __restrict__ Cl,
                                                              const double* __restrict__ A, int64_t lda,
                                                              const double* __restrict__ dkb,
                                                              double* __restrict__ out, int64_t ldo) {
  const int64_t mm = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (mm >= m || j >= n) return;
  const int d = kp.d;
  const double* z = Z + mm * d;
  const double* xj = X + j * d;
  const double b = beta[j * ldb + mm];
  double t2 = 2.0 * dk(kp, z[l], xj[l], l, kval(kp, z, xj));  // -2 T
  if (dkb) t2 += 2.0 * dkb[j * d + l];                          // gp.py:320: derivTotal[-1] -= noiseFunc.deriv(p)
  out[j * ldo + mm] = b * (t2 + 2.0 * Cl[j * ldb + mm] - A[j * lda + j] * b);
}

// B[j][m] += bias[j]   (gp.py:319: totEvals[:, zz] += noiseFunc(p))
__global__ __launch_bounds__(256) void add_row_bias_kernel(double* __restrict__ B, int64_t ldb, int64_t n, int64_t m,
                                                           const double* __restrict__ bias) {
  const int64_t mm = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = blockIdx.y;
  if (mm < m && j < n) B[j * ldb + mm] += bias[j];
}

// out[m][l] = -2 sum_j dk(z_m, x_j)[l] beta[j][m]: one workgroup per evaluation point (gp.py:261-280)
__global__ __launch_bounds__(256) void var_grad_newpt_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                             const double* __restrict__ Z, const double* __restrict__ beta,
                                                             int64_t ldb, int64_t col0, double* __restrict__ out) {
  __shared__ double red[256];
  __shared__ double zs[GPX_MAXD];
  const int64_t mm = blockIdx.x;  // column inside the chunk; global evaluation point col0 + mm
  const int d = kp.d, t = threadIdx.x;
  if (t < d) zs[t] = Z[(col0 + mm) * d + t];
  __syncthreads();
  double acc[GPX_MAXD];
#pragma unroll
  for (int l = 0; l < GPX_MAXD; ++l) acc[l] = 0.0;
  for (int64_t j = t; j < n; j += 256) {
    const double* xj = X + j * d;
    const double w = -2.0 * beta[j * ldb + mm] * kval(kp, zs, xj);
#pragma unroll
    for (int l = 0; l < GPX_MAXD; ++l)
      if (l < d) acc[l] += dk(kp, zs[l], xj[l], l, w);
  }
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
    if (t == 0) out[(col0 + mm) * d + l] = red[0];
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

int check_args(int kind, int d, const gpx_mat* L, const gpx_mat* X, const gpx_mat* Z, const gpx_fitc* fitc = nullptr) {
  GPX_ARG((L || fitc) && X && Z, "NULL argument");
  GPX_ARG(fitc || (L->factored && L->aux), "matrix has not been factored by gpx_potrf");
  GPX_ARG(kind == GPX_K_SE || (kind == GPX_K_MEHLER && d == 1),
          "point derivatives exist for the squared-exponential and the 1-D Mehler kernel only "
          "(as in the reference: kernels.py:146-181, 295-324)");
  GPX_ARG(X->cols == d && X->pcols == d && Z->cols == d && Z->pcols == d, "point sets must be unpadded (n x d)");
  GPX_ARG(X->rows == (fitc ? fitc_n(fitc) : L->rows) && Z->rows > 0, "X does not match the factor / no evaluation points");
  return 0;
}

// beta for the evaluation points Zc[0 .. mc): W = L^-1 (K(X, Zc) + bias), beta = L^-T W.  A, B: two buffers of np * mcp
// doubles; *out = beta^T (mcp x np, row stride np) when `transposed`, else beta (np x mcp, row stride mcp) -- in whichever of
// the two buffers it ends up in.  Large factors (T != NULL: mcp * chol_binv_order(np) doubles): both solves through the
// factor's explicit block inverses -- forward out of place (chol_trsm_left_oop, the IVAR solve of bench.py), backward as the
// right solve beta^T = W^T L^-1 (chol_trsm_right_n_leading) -- every product K >= 1024 on 128-tiles: 2 x 31 ms at N = 8192,
// M = 32768 where the in-place leaf recursions took 2 x 36.
// Wfwd != NULL: W = L^-1 K(X, Zc) is at hand (np x mcp; kept by gpx_ivar_keep at the same design) -- fill and forward solve are
// skipped, a third of the gradient's work.
int solve_beta(gpx_ctx* ctx, const KParams& kp, const gpx_mat* L, const gpx_mat* X, const double* Zc, int64_t mc,
               int64_t mcp, const double* d_bias, double* A, double* B, double* T, int transposed, double** out,
               const gpx_mat* Wfwd = nullptr, const gpx_fitc* fitc = nullptr, double* U = nullptr) {
  const int64_t n = fitc ? fitc_n(fitc) : L->rows, np = fitc ? fitc_np(fitc) : L->prows;
  double *cur, *other;   // cur: beta^T
  if (fitc != nullptr) {
    // FITC model: beta = P (K(X, Zc) + bias) with the Woodbury precision (fitc.hip); U = mcp x nup doubles of scratch
    GPX_ARG(U != nullptr && Wfwd == nullptr, "solve_beta: the FITC form needs its scratch");
    GPX_TRY(launch_kfill(ctx, kp, X->p, n, Zc, mc, 0, nullptr, 0, 0.0, A, np, mcp, mcp));
    if (d_bias) {
      dim3 grid((unsigned)((mc + 255) / 256), (unsigned)n);
      hipLaunchKernelGGL(add_row_bias_kernel, grid, dim3(256), 0, ctx->stream, A, mcp, n, mc, d_bias);
    }
    GPX_TRY(fitc_solve_beta_t(ctx, fitc, A, mcp, B, U));
    if (transposed) {
      *out = B;
      return 0;
    }
    *out = A;
    return launch_transpose(ctx, B, mcp, np, np, A, mcp);
  }
  if (Wfwd != nullptr) {
    GPX_ARG(!d_bias && Wfwd->prows == np && Wfwd->pcols == mcp, "solve_beta: the kept forward solve does not match");
    if (T != nullptr) {
      GPX_TRY(launch_transpose(ctx, Wfwd->p, np, mcp, Wfwd->ld, A, np));
      GPX_TRY(chol_trsm_right_n_leading(ctx, const_cast<gpx_mat*>(L), np, A, np, mcp, T));
      cur = A;
      other = B;
    } else {
      GPX_TRY(launch_transpose(ctx, Wfwd->p, np, mcp, Wfwd->ld, B, np));
      GPX_TRY(chol_trsm_right_n(ctx, L->p, L->ld, L->aux, B, np, mcp, np));
      cur = B;
      other = A;
    }
    if (transposed) {
      *out = cur;
      return 0;
    }
    *out = other;
    return launch_transpose(ctx, cur, mcp, np, np, other, mcp);
  }
  GPX_TRY(launch_kfill(ctx, kp, X->p, n, Zc, mc, 0, nullptr, 0, 0.0, A, np, mcp, mcp));
  if (d_bias) {
    dim3 grid((unsigned)((mc + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(add_row_bias_kernel, grid, dim3(256), 0, ctx->stream, A, mcp, n, mc, d_bias);
  }
  if (T != nullptr) {
    gpx_mat* Lm = const_cast<gpx_mat*>(L);   // (the block inverses are a cache inside the factor)
    GPX_TRY(chol_trsm_left_oop(ctx, Lm, A, mcp, B, mcp, mcp));
    GPX_TRY(launch_transpose(ctx, B, np, mcp, mcp, A, np));
    GPX_TRY(chol_trsm_right_n_leading(ctx, Lm, np, A, np, mcp, T));
    cur = A;
    other = B;
  } else {
    GPX_TRY(chol_trsm_left(ctx, L->p, L->ld, L->aux, A, mcp, np, mcp));
    GPX_TRY(launch_transpose(ctx, A, np, mcp, mcp, B, np));
    GPX_TRY(chol_trsm_right_n(ctx, L->p, L->ld, L->aux, B, np, mcp, np));
    cur = B;
    other = A;
  }
  if (transposed) {
    *out = cur;
    return 0;
  }
  *out = other;
  return launch_transpose(ctx, cur, mcp, np, np, other, mcp);
}

// scratch of the block-inverse solves (NULL below the order where they pay)
int solve_scratch(gpx_ctx* ctx, Scratch& sc, int64_t np, int64_t mcp, double** T) {
  *T = nullptr;
  if (np < 4096) return 0;
  void* p;
  GPX_TRY(sc.get(mcp * chol_binv_order(np) * 8, &p));
  *T = (double*)p;
  return 0;
}

// d rounded up to the instantiated register-array sizes
#define GPX_SE_DISPATCH(d_, CALL) \
  do {                            \
    if ((d_) <= 1) { CALL(1); }   \
    else if ((d_) <= 2) { CALL(2); } \
    else if ((d_) <= 4) { CALL(4); } \
    else if ((d_) <= 8) { CALL(8); } \
    else if ((d_) <= 16) { CALL(16); } \
    else { CALL(32); }            \
  } while (0)

int upload(gpx_ctx* ctx, Scratch& sc, const double* host, int64_t count, double** dev) {
  *dev = nullptr;
  if (!host) return 0;
  void* p;
  GPX_TRY(sc.get(count * 8, &p));
  GPX_HIP(hipMemcpyAsync(p, host, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
  *dev = (double*)p;
  return 0;
}

// evaluation points per chunk: beta and the transpose scratch / C_l (two np x mc matrices) + the output slab under ~12 GiB
int64_t grad_chunk(int64_t np) {
  int64_t budget = (int64_t)12 << 30;
  const char* e = getenv("GPX_CROSS_BYTES");
  if (e && atoll(e) > 0) budget = atoll(e);
  int64_t mc = budget / (3 * np * 8) / GPX_TILE * GPX_TILE;
  return mc < GPX_TILE ? GPX_TILE : mc;
}

}  // namespace

extern "C" {

int gpx_ivar_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                  const gpx_mat* Z, const double* noise_deriv, double* grad) {
  return gpx_ivar_grad_w(ctx, kind, d, hyp, nhyp, L, X, Z, noise_deriv, nullptr, grad);
}

int gpx_ivar_grad_w(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* Z, const double* noise_deriv, const gpx_mat* Wkept, double* grad) {
  GPX_ARG(ctx && grad, "NULL argument");
  GPX_ARG(Wkept == nullptr || ctx->live_mats.count(Wkept), "ivar_grad: the kept solve is not a live matrix of this context");
  GPX_TRY(check_args(kind, d, L, X, Z));
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z));
  const int64_t n = L->rows, np = L->prows, m = Z->rows;
  const int64_t mp = gpx_round_up(m, GPX_TILE);
  Scratch sc(ctx);
  void *pW, *pWt, *pS, *pg;
  GPX_TRY(sc.get(np * mp * 8, &pW));
  GPX_TRY(sc.get(mp * np * 8, &pWt));
  GPX_TRY(sc.get(np * np * 8, &pS));
  GPX_TRY(sc.get(n * d * 8, &pg));
  double* d_nd;
  GPX_TRY(upload(ctx, sc, noise_deriv, n * d, &d_nd));
  double *W, *T;
  GPX_TRY(solve_scratch(ctx, sc, np, mp, &T));
  GPX_ARG(Wkept == nullptr || (Wkept->prows == np && Wkept->pcols == mp && Wkept->rows == n && Wkept->cols == m),
          "ivar_grad: the kept solve has another shape");
  GPX_TRY(solve_beta(ctx, kp, L, X, Z->p, m, mp, nullptr, (double*)pW, (double*)pWt, T, 0, &W, Wkept));
  // S = beta beta^T: symmetric -- only the tiles on / below the diagonal are computed (N^2 M flops instead of 2 N^2 M), the
  // rest is mirrored (the row kernel reads whole rows).  A SMALL C under a long k range: N = 8192 is 2080 lower 128-tiles
  // for 512 resident workgroups -- 4.06 rounds, the last one nearly empty (36.5 ms = 60 TF/s at M = 32768) -- so the k range
  // goes in slices (launch_gemm_ksplit; the buffer the solve no longer needs holds the partials when it is large enough).
  {
    const int64_t t128 = (np / 128) * (np / 128 + 1) / 2;
    int64_t parts = 1;
    while (parts < 8 && t128 * parts < 8192 && mp % (2 * parts * 16) == 0 && mp / (2 * parts) >= 4096) parts *= 2;
    double* other = W == (double*)pW ? (double*)pWt : (double*)pW;
    while (parts > 1 && parts * np * np > np * mp) parts /= 2;
    if (parts > 1)
      GPX_TRY(launch_gemm_ksplit(ctx, W, mp, W, mp, (double*)pS, np, np, np, mp, true, parts, other, true));
    else
      GPX_TRY(launch_gemm(ctx, W, mp, W, mp, (double*)pS, np, np, np, mp, true, false, true));
  }
  {
    const int nt = (int)(np / 32);
    hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)((int64_t)nt * (nt + 1) / 2)), dim3(256), 0, ctx->stream, (double*)pS, np, nt);
    GPX_HIP(hipGetLastError());
  }
  {
    ProfScope ps(ctx, GPX_PROF_REDUCE, 0.0, 8.0 * ((double)n * m + (double)n * n));
    if (kind == GPX_K_SE) {
#define GPX_CALL(DM_)                                                                                                        \
  hipLaunchKernelGGL((ivar_grad_row_se_kernel<DM_>), dim3((unsigned)n), dim3(256), 0, ctx->stream, kp, X->p, n, Z->p, m,     \
                     (const double*)W, mp, (const double*)pS, np, (const double*)d_nd, 1.0 / (double)m, (double*)pg, (int64_t)0)
      GPX_SE_DISPATCH(d, GPX_CALL);
#undef GPX_CALL
    } else {
      hipLaunchKernelGGL(ivar_grad_row_kernel, dim3((unsigned)n), dim3(256), 0, ctx->stream, kp, X->p, n, Z->p, m,
                         (const double*)W, mp, (const double*)pS, np, (const double*)d_nd, 1.0 / (double)m, (double*)pg);
    }
  }
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(grad, pg, (size_t)(n * d * 8), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// The gradient for the design points from r0 on ONLY -- the batch loop pins the earlier ones by equal bounds
// (experimentalDesign.py:719-724), their entries are never used -- from the kept forward solve W = L^-1 K(X, Z), squared
// exponential, homoscedastic.  With T the rows from r0 on:
//     beta_T = L_TT^-T W_T                 (back substitution touches nothing above T:   (n - r0)^2 M flops)
//     S_T    = beta_T beta^T = (beta_T W^T) L^-1          (one (n - r0) x N x M product + a few-row right solve)
// and the row kernel for those rows: 2 (n - r0) N M flops where the full gradient needs the whole backward solve and the whole
// S = beta beta^T (2 N^2 M).  grad: (n - r0) x d on the host, point-major.
int gpx_ivar_grad_rows(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                       const gpx_mat* Z, const gpx_mat* W, int64_t r0, double* grad) {
  GPX_ARG(ctx && grad && W, "NULL argument");
  GPX_TRY(check_args(kind, d, L, X, Z));
  GPX_ARG(kind == GPX_K_SE, "ivar_grad_rows: squared-exponential kernel only");
  GPX_ARG(ctx->live_mats.count(W), "ivar_grad_rows: W is not a live matrix of this context");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z));
  const int64_t n = L->rows, np = L->prows, m = Z->rows, mp = gpx_round_up(m, GPX_TILE);
  GPX_ARG(W->rows == n && W->cols == m && W->prows == np && W->pcols == mp, "ivar_grad_rows: the kept solve has another shape");
  GPX_ARG(r0 > 0 && r0 < n && r0 % GPX_TILE == 0, "ivar_grad_rows: r0 must be a positive multiple of 128 below the number of points");
  const int64_t bp = np - r0, b = n - r0, ldw = W->ld;
  Scratch sc(ctx);
  void *pA, *pB, *pG, *pP = nullptr, *pT = nullptr, *pg;
  GPX_TRY(sc.get(mp * bp * 8, &pA));
  GPX_TRY(sc.get(bp * mp * 8, &pB));
  GPX_TRY(sc.get(bp * np * 8, &pG));
  GPX_TRY(sc.get(b * d * 8, &pg));
  double *A = (double*)pA, *B = (double*)pB, *G = (double*)pG;
  // beta_T^T = W_T^T L_TT^-1, then back to rows
  GPX_TRY(launch_transpose(ctx, W->p + r0 * ldw, bp, mp, ldw, A, bp));
  GPX_TRY(chol_trsm_right_n(ctx, L->p + r0 * (L->ld + 1), L->ld, L->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, A, bp, mp, bp));
  GPX_TRY(launch_transpose(ctx, A, mp, bp, bp, B, mp));
  // G = beta_T W^T: few rows under the long k range M -- in slices of it when 128-tiles alone cannot fill the chip
  {
    int64_t parts = 1;
    while (parts < 8 && (bp / 128) * (np / 128) * parts < 1024 && mp % (2 * parts * 16) == 0 && mp / (2 * parts) >= 4096) parts *= 2;
    if (parts > 1 && sc.get(parts * bp * np * 8, &pP) != 0) parts = 1;
    if (parts > 1)
      GPX_TRY(launch_gemm_ksplit(ctx, B, mp, W->p, ldw, G, np, bp, np, mp, false, parts, (double*)pP, true));
    else
      GPX_TRY(launch_gemm(ctx, B, mp, W->p, ldw, G, np, bp, np, mp, true, false, false));
  }
  // S_T = G L^-1
  if (np >= 4096) {
    GPX_TRY(sc.get(bp * chol_binv_order(np) * 8, &pT));
    GPX_TRY(chol_trsm_right_n_leading(ctx, const_cast<gpx_mat*>(L), np, G, np, bp, (double*)pT));
  } else {
    GPX_TRY(chol_trsm_right_n(ctx, L->p, L->ld, L->aux, G, np, bp, np));
  }
  {
    ProfScope ps(ctx, GPX_PROF_REDUCE, 0.0, 8.0 * ((double)b * m + (double)b * n));
#define GPX_CALL(DM_)                                                                                                        \
  hipLaunchKernelGGL((ivar_grad_row_se_kernel<DM_>), dim3((unsigned)b), dim3(256), 0, ctx->stream, kp, X->p, n, Z->p, m,     \
                     (const double*)B, mp, (const double*)G, np, (const double*)nullptr, 1.0 / (double)m, (double*)pg, r0)
    GPX_SE_DISPATCH(d, GPX_CALL);
#undef GPX_CALL
  }
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(grad, pg, (size_t)(b * d * 8), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// gpx_var_grad / gpx_fitc_var_grad: `fitc` != NULL takes beta from the FITC model's Woodbury precision instead of the factor L
static int var_grad_impl(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_fitc* fitc,
                         const gpx_mat* X, const gpx_mat* Z, const double* noise_deriv, const double* eval_bias,
                         const double* dk_bias, double* out) {
  GPX_ARG(ctx && out, "NULL argument");
  GPX_TRY(check_args(kind, d, L, X, Z, fitc));
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z));
  const int64_t n = fitc ? fitc_n(fitc) : L->rows, np = fitc ? fitc_np(fitc) : L->prows, M = Z->rows;
  GPX_ARG(np <= 65535, "var_grad: at most 65535 training points");
  const int64_t mcmax = grad_chunk(np);
  const int64_t mc_alloc = gpx_round_up(M < mcmax ? M : mcmax, GPX_TILE);
  Scratch sc(ctx);
  void *pW, *pWt, *pA, *pO;
  GPX_TRY(sc.get(np * mc_alloc * 8, &pW));
  GPX_TRY(sc.get(np * mc_alloc * 8, &pWt));   // transpose scratch, then C_l
  GPX_TRY(sc.get(np * np * 8, &pA));
  GPX_TRY(sc.get(n * mc_alloc * 8, &pO));
  double *d_nd, *d_eb, *d_db;
  GPX_TRY(upload(ctx, sc, noise_deriv, n * d, &d_nd));
  GPX_TRY(upload(ctx, sc, eval_bias, n, &d_eb));
  GPX_TRY(upload(ctx, sc, dk_bias, n * d, &d_db));
  double* T = nullptr;
  void* pU = nullptr;
  if (fitc) GPX_TRY(sc.get(mc_alloc * fitc_nup(fitc) * 8, &pU));
  else GPX_TRY(solve_scratch(ctx, sc, np, mc_alloc, &T));
  // The (N d) x M result goes to PAGEABLE host memory by contract: every strided copy blocks the calling thread.  So the
  // coordinates are pipelined -- coordinate l+1's kernels are queued BEFORE the copy of coordinate l is issued (two output
  // buffers, the copy on the low-priority stream behind an event), and the GPU computes while the host sits in the copy.
  void* pO_b;
  GPX_TRY(sc.get(n * mc_alloc * 8, &pO_b));
  double* const obuf[2] = {(double*)pO, (double*)pO_b};
  hipStream_t Ms = ctx->stream, Cs = ctx->streams[4];
  hipEvent_t evd[2] = {nullptr, nullptr};
  const bool piped = Cs != Ms && hipEventCreateWithFlags(&evd[0], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&evd[1], hipEventDisableTiming) == hipSuccess;
  struct Item { int64_t j0, mc, mcp; int l; };
  std::vector<Item> items;
  for (int64_t j0 = 0; j0 < M; j0 += mcmax) {
    const int64_t mc = (M - j0) < mcmax ? (M - j0) : mcmax;
    for (int l = 0; l < d; ++l) items.push_back({j0, mc, gpx_round_up(mc, GPX_TILE), l});
  }
  double *W = nullptr, *Cl = nullptr;
  auto compute = [&](const Item& it, int b) -> int {
    const double* Zc = Z->p + it.j0 * d;
    if (it.l == 0) {
      GPX_TRY(solve_beta(ctx, kp, L, X, Zc, it.mc, it.mcp, d_eb, (double*)pW, (double*)pWt, T, 0, &W, nullptr, fitc, (double*)pU));
      Cl = W == (double*)pW ? (double*)pWt : (double*)pW;   // the buffer the solve no longer needs
    }
    dim3 ga((unsigned)((np + 255) / 256), (unsigned)np);
    hipLaunchKernelGGL(dcov_kernel, ga, dim3(256), 0, ctx->stream, kp, X->p, n, it.l, (const double*)d_nd, (double*)pA, np, np);
    GPX_TRY(launch_gemm(ctx, (double*)pA, np, W, it.mcp, Cl, it.mcp, np, it.mcp, np, false, false, false));  // C_l = A_l beta
    dim3 gf((unsigned)((it.mc + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(var_grad_finish_kernel, gf, dim3(256), 0, ctx->stream, kp, X->p, n, Zc, it.mc, it.l, (const double*)W,
                       it.mcp, (const double*)Cl, (const double*)pA, np, (const double*)d_db, obuf[b], it.mc);
    GPX_HIP(hipGetLastError());
    if (piped) GPX_HIP(hipEventRecord(evd[b], Ms));
    return 0;
  };
  auto copy_out = [&](const Item& it, int b) -> int {
    hipStream_t st = piped ? Cs : Ms;
    if (piped) GPX_HIP(hipStreamWaitEvent(Cs, evd[b], 0));
    // rows j*d + l of the (N*d x M) result, columns [j0, j0+mc)
    GPX_HIP(hipMemcpy2DAsync(out + (int64_t)it.l * M + it.j0, (size_t)d * M * 8, obuf[b], (size_t)it.mc * 8, (size_t)it.mc * 8,
                             (size_t)n, hipMemcpyDeviceToHost, st));
    GPX_HIP(hipStreamSynchronize(st));   // the buffer is rewritten two coordinates on
    return 0;
  };
  int r = items.empty() ? 0 : compute(items[0], 0);
  for (size_t i = 0; i < items.size() && r == 0; ++i) {
    if (piped && i + 1 < items.size()) r = compute(items[i + 1], (int)((i + 1) & 1));
    if (r == 0) r = copy_out(items[i], (int)(i & 1));
    if (r == 0 && !piped && i + 1 < items.size()) r = compute(items[i + 1], (int)((i + 1) & 1));
  }
  (void)hipDeviceSynchronize();
  for (int b = 0; b < 2; ++b)
    if (evd[b]) (void)hipEventDestroy(evd[b]);
  return r;
}

static int var_grad_newpt_impl(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_fitc* fitc,
                        const gpx_mat* X, const gpx_mat* Z, double* out) {
  GPX_ARG(ctx && out, "NULL argument");
  GPX_TRY(check_args(kind, d, L, X, Z, fitc));
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z));
  const int64_t n = fitc ? fitc_n(fitc) : L->rows, np = fitc ? fitc_np(fitc) : L->prows, M = Z->rows;
  const int64_t mcmax = grad_chunk(np);
  const int64_t mc_alloc = gpx_round_up(M < mcmax ? M : mcmax, GPX_TILE);
  Scratch sc(ctx);
  void *pW, *pWt, *pO;
  GPX_TRY(sc.get(np * mc_alloc * 8, &pW));
  GPX_TRY(sc.get(np * mc_alloc * 8, &pWt));
  GPX_TRY(sc.get(M * d * 8, &pO));
  double* T = nullptr;
  void* pU = nullptr;
  if (fitc) GPX_TRY(sc.get(mc_alloc * fitc_nup(fitc) * 8, &pU));
  else GPX_TRY(solve_scratch(ctx, sc, np, mc_alloc, &T));
  for (int64_t j0 = 0; j0 < M; j0 += mcmax) {
    const int64_t mc = (M - j0) < mcmax ? (M - j0) : mcmax;
    const int64_t mcp = gpx_round_up(mc, GPX_TILE);
    double* beta;
    if (kind == GPX_K_SE) {   // beta^T: row m contiguous over the training points, and no transpose back
      GPX_TRY(solve_beta(ctx, kp, L, X, Z->p + j0 * d, mc, mcp, nullptr, (double*)pW, (double*)pWt, T, 1, &beta, nullptr, fitc,
                         (double*)pU));
#define GPX_CALL(DM_)                                                                                                      \
  hipLaunchKernelGGL((var_grad_newpt_se_kernel<DM_>), dim3((unsigned)mc), dim3(256), 0, ctx->stream, kp, X->p, n, Z->p,     \
                     (const double*)beta, np, j0, (double*)pO)
      GPX_SE_DISPATCH(d, GPX_CALL);
#undef GPX_CALL
    } else {
      GPX_TRY(solve_beta(ctx, kp, L, X, Z->p + j0 * d, mc, mcp, nullptr, (double*)pW, (double*)pWt, T, 0, &beta, nullptr, fitc,
                         (double*)pU));
      hipLaunchKernelGGL(var_grad_newpt_kernel, dim3((unsigned)mc), dim3(256), 0, ctx->stream, kp, X->p, n, Z->p,
                         (const double*)beta, mcp, j0, (double*)pO);
    }
    GPX_HIP(hipGetLastError());
  }
  GPX_HIP(hipMemcpyAsync(out, pO, (size_t)(M * d * 8), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_var_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                 const gpx_mat* Z, const double* noise_deriv, const double* eval_bias, const double* dk_bias,
                 double* out) {
  return var_grad_impl(ctx, kind, d, hyp, nhyp, L, nullptr, X, Z, noise_deriv, eval_bias, dk_bias, out);
}

int gpx_var_grad_newpt(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                       const gpx_mat* Z, double* out) {
  return var_grad_newpt_impl(ctx, kind, d, hyp, nhyp, L, nullptr, X, Z, out);
}

// The same two derivatives on a FITC model (round 6): the reference computes them from whatever `precisionMatrix` holds, for
// FITC the Woodbury precision (gp.py:194-206, 275, 322); beta = P K(X, Z) comes from the model's factors (fitc.hip), the rest is
// the dense path's kernels.  (kind, d, hyp): the kernel the model was fitted with.
int gpx_fitc_var_grad(gpx_ctx* ctx, const gpx_fitc* f, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X,
                      const gpx_mat* Z, const double* noise_deriv, const double* eval_bias, const double* dk_bias, double* out) {
  GPX_ARG(f != nullptr, "fitc model is NULL");
  return var_grad_impl(ctx, kind, d, hyp, nhyp, nullptr, f, X, Z, noise_deriv, eval_bias, dk_bias, out);
}

int gpx_fitc_var_grad_newpt(gpx_ctx* ctx, const gpx_fitc* f, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X,
                            const gpx_mat* Z, double* out) {
  GPX_ARG(f != nullptr, "fitc model is NULL");
  return var_grad_newpt_impl(ctx, kind, d, hyp, nhyp, nullptr, f, X, Z, out);
}

}  // extern "C"
