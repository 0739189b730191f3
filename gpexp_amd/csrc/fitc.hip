// f4: FITC sparse approximation -- gfx950.
//
// Replaces the reference's dense construction (gp.py:182-210, 401-426; gp_kernel_utilities.py:70-104), which forms the
// N x N matrices Q = Kfu Quu^-1 Kuf, K, G = diag(K - Q), and the Woodbury precision
//     P = Gi - Gi Kfu (Quu + Kuf Gi Kfu)^-1 Kuf Gi,          Gi = diag(1 / (g + 1e-12)),
// and then runs the dense N x N prediction loops on P.  Here nothing N x N is ever formed on the product path: with
// nu inducing points the model is  Lu = chol(Quu),  Kuf (nu x N),  W = Lu^-1 Kuf,  g = diag(K) - colsum(W^2),
// Ks = -Kuf Gi,  La = chol(Quu + Kuf Gi Kfu)  -- all built from the kfill / potrf / TRSM / GEMM / reduction kernels of
// the dense path -- and
//     P y      = Gi (y - Kfu A^-1 Kuf Gi y)                                   (two reductions + one nu x nu solve)
//     log|Q+G| = sum log g + log|A| - log|Quu|                                (determinant lemma)
//     var(z)   = k(z,z) - sum_i Gi_i k_zi^2 + |La^-1 Kuf Gi k_z|^2           (one NT GEMM + one right TRSM per chunk)
// Quu and K both carry the nugget, exactly as the reference builds them.  gpx_fitc_dense materialises Q + G and P for
// the reference's `covarianceMatrix` / `precisionMatrix` attributes (small N only; parity tests).
#include "gpx_internal.h"
#include <math.h>
#include <stdlib.h>
#include <vector>

struct gpx_fitc {
  KParams kp;
  int64_t n, nu, np, nup;
  double noise;
  gpx_mat* Lu;   // chol(Quu)                       nup x nup
  gpx_mat* Kuf;  // K(S, X)                         nup x np
  gpx_mat* W;    // Lu^-1 Kuf                       nup x np
  gpx_mat* Ks;   // -Kuf diag(ginv)                 nup x np
  gpx_mat* La;   // chol(Quu + Kuf Gi Kfu)          nup x nup
  double* g;     // diag(K - Q), 1 on the padding   np (device)
  double* ginv;  // 1 / (g + 1e-12), 0 on padding   np (device)
  double sumlogg;
};

namespace {

// g[i] = kd[i] + noise - qd[i], ginv[i] = 1/(g[i] + 1e-12) for i < n; padding: g = 1, ginv = 0
__global__ void fitc_g_kernel(const double* __restrict__ kd, const double* __restrict__ qd, double noise, int64_t n,
                              int64_t np, double* __restrict__ g, double* __restrict__ ginv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  if (i < n) {
    const double v = kd[i] + noise - qd[i];
    g[i] = v;
    ginv[i] = 1.0 / (v + 1e-12);
  } else {
    g[i] = 1.0;
    ginv[i] = 0.0;
  }
}

// out[r][c] = -in[r][c] * s[c]
__global__ __launch_bounds__(256) void scale_cols_neg_kernel(const double* __restrict__ in, int64_t ldi,
                                                             const double* __restrict__ s, double* __restrict__ out,
                                                             int64_t ldo, int64_t cols) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  const double2 v = *reinterpret_cast<const double2*>(in + r * ldi + c);
  const double2 w = *reinterpret_cast<const double2*>(s + c);
  *reinterpret_cast<double2*>(out + r * ldo + c) = double2{-v.x * w.x, -v.y * w.y};
}

__global__ void log_sum_kernel(const double* __restrict__ g, int64_t n, double* __restrict__ out) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double s = 0.0;
  for (int64_t i = t; i < n; i += 256) s += log(g[i]);
  red[t] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  if (t == 0) out[0] = red[0];
}

// t[i] = ginv[i] * (y[i] - t[i])
__global__ void fitc_coeff_kernel(const double* __restrict__ ginv, const double* __restrict__ y, double* __restrict__ t,
                                  int64_t np) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < np) t[i] = ginv[i] * (y[i] - t[i]);
}

__global__ void negate_kernel(double* __restrict__ x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = -x[i];
}

int factor_in_place(gpx_ctx* ctx, gpx_mat* K, const char* what) {
  if (!K->aux) {
    K->aux_bytes = K->prows * GPX_TILE * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, K->aux_bytes, &p));
    K->aux = (double*)p;
  }
  GPX_TRY(chol_potrf(ctx, K->p, K->ld, K->prows, K->aux, K->rows));
  int info = 0;
  GPX_HIP(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  K->binv_ib = 0;  // block inverses (chol_potrs) belong to the previous contents
  K->factored = (info == 0);
  if (info != 0) {
    gpx_set_error("fitc: %s is not positive definite (pivot %d <= 0)", what, info);
    return info;
  }
  return 0;
}

void fitc_release(gpx_ctx* ctx, gpx_fitc* f) {
  if (!f) return;
  (void)hipStreamSynchronize(ctx->stream);
  if (f->Lu) gpx_mat_free(ctx, f->Lu);
  if (f->Kuf) gpx_mat_free(ctx, f->Kuf);
  if (f->W) gpx_mat_free(ctx, f->W);
  if (f->Ks) gpx_mat_free(ctx, f->Ks);
  if (f->La) gpx_mat_free(ctx, f->La);
  if (f->g) gpx_dev_release(ctx, f->g, f->np * 8);
  if (f->ginv) gpx_dev_release(ctx, f->ginv, f->np * 8);
  delete f;
}

// chunk of evaluation points handled at once (same budget as the dense posterior: GPX_CROSS_BYTES, default 16 GiB)
int64_t eval_chunk(int64_t np) {
  int64_t budget = (int64_t)16 << 30;
  const char* e = getenv("GPX_CROSS_BYTES");
  if (e && atoll(e) > 0) budget = atoll(e);
  int64_t mc = budget / (np * 8) / GPX_TILE * GPX_TILE;
  if (mc < GPX_TILE) mc = GPX_TILE;
  return mc;
}

struct Tmp {  // scratch buffers released on scope exit (after a stream sync)
  gpx_ctx* ctx;
  std::vector<std::pair<void*, int64_t>> bufs;
  explicit Tmp(gpx_ctx* c) : ctx(c) {}
  int get(int64_t bytes, double** out) {
    void* p;
    int r = gpx_dev_alloc(ctx, bytes, &p);
    if (r == 0) {
      bufs.push_back({p, bytes});
      *out = (double*)p;
    }
    return r;
  }
  ~Tmp() {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : bufs) gpx_dev_release(ctx, b.first, b.second);
  }
};

// Bt[m][i] *= s[i] (row-major m x cols, row stride ld)
__global__ __launch_bounds__(256) void scale_cols_kernel(double* __restrict__ Bt, int64_t ld, const double* __restrict__ s,
                                                         int64_t cols) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r = blockIdx.y;
  if (c >= cols) return;
  double2 v = *reinterpret_cast<double2*>(Bt + r * ld + c);
  const double2 w = *reinterpret_cast<const double2*>(s + c);
  v.x *= w.x;
  v.y *= w.y;
  *reinterpret_cast<double2*>(Bt + r * ld + c) = v;
}

}  // namespace

int64_t fitc_n(const gpx_fitc* f) { return f->n; }
int64_t fitc_np(const gpx_fitc* f) { return f->np; }
int64_t fitc_nup(const gpx_fitc* f) { return f->nup; }

// beta^T = (P B)^T for B = np x mcp right-hand sides (row stride mcp; K(X, Z) of one chunk of evaluation points) under the
// Woodbury precision P = Gi - Ks^T A^-1 Ks (Ks = -Kuf Gi, A = La La^T): the point-derivative routines of the reference read
// `precisionMatrix`, which for a FITC model is exactly this P (gp.py:194-206, 275, 322).  Nothing N x N is formed:
//   Bt = B^T (mcp x np);  U = Bt Ks^T (mcp x nup);  U <- U La^-T La^-1 = (A^-1 Ks B)^T;  Bt <- Bt Gi - U Ks.
// Bt: mcp x np doubles (row stride np) = the result; U: mcp x nup doubles of scratch.
int fitc_solve_beta_t(gpx_ctx* ctx, const gpx_fitc* f, const double* B, int64_t mcp, double* Bt, double* U) {
  const int64_t np = f->np, nup = f->nup;
  GPX_TRY(launch_transpose(ctx, B, np, mcp, mcp, Bt, np));
  GPX_TRY(launch_gemm(ctx, Bt, np, f->Ks->p, f->Ks->ld, U, nup, mcp, nup, np, true, false, false));
  GPX_TRY(chol_trsm_right(ctx, f->La->p, f->La->ld, f->La->aux, U, nup, mcp, nup));
  GPX_TRY(chol_trsm_right_n(ctx, f->La->p, f->La->ld, f->La->aux, U, nup, mcp, nup));
  dim3 grid((unsigned)((np / 2 + 255) / 256), (unsigned)mcp);
  hipLaunchKernelGGL(scale_cols_kernel, grid, dim3(256), 0, ctx->stream, Bt, np, f->ginv, np);
  GPX_TRY(launch_gemm(ctx, U, nup, f->Ks->p, f->Ks->ld, Bt, np, mcp, np, nup, false, true, false));
  GPX_HIP(hipGetLastError());
  return 0;
}

namespace {
}  // namespace

extern "C" {

int gpx_fitc_free(gpx_ctx* ctx, gpx_fitc* f) {
  GPX_ARG(ctx != nullptr, "ctx is NULL");
  fitc_release(ctx, f);
  return 0;
}

int gpx_fitc_fit(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X, const gpx_mat* S,
                 double noise, gpx_fitc** out) {
  GPX_ARG(ctx && X && S && out, "NULL argument");
  gpx_fitc* f = new gpx_fitc();
  f->Lu = f->Kuf = f->W = f->Ks = f->La = nullptr;
  f->g = f->ginv = nullptr;
  int r = gpx_make_kparams(kind, d, hyp, nhyp, &f->kp);
  if (r != 0) {
    delete f;
    return r;
  }
  if (!(X->cols == d && X->pcols == d && S->cols == d && S->pcols == d && S->rows > 0 && X->rows > 0)) {
    delete f;
    gpx_set_error("fitc: X and the inducing points must be non-empty unpadded (n x d) point sets");
    return -1;
  }
  if ((r = gpx_kparams_sets(ctx, &f->kp, X, S)) != 0) {
    delete f;
    return r;
  }
  const KParams& kp = f->kp;
  f->n = X->rows;
  f->nu = S->rows;
  f->noise = noise;
  Tmp tmp(ctx);
  do {
    // Lu = chol(K(S,S) + noise I)
    if ((r = gpx_mat_new(ctx, f->nu, f->nu, 1, &f->Lu)) != 0) break;
    f->nup = f->Lu->prows;
    if ((r = launch_kfill(ctx, kp, S->p, f->nu, S->p, f->nu, 1, nullptr, 1, noise, f->Lu->p, f->nup, f->nup,
                          f->Lu->ld)) != 0) break;
    if ((r = factor_in_place(ctx, f->Lu, "K(inducing, inducing) + noise")) != 0) break;
    // Kuf, W = Lu^-1 Kuf
    if ((r = gpx_mat_new(ctx, f->nu, f->n, 1, &f->Kuf)) != 0) break;
    f->np = f->Kuf->pcols;
    if ((r = gpx_mat_new(ctx, f->nu, f->n, 1, &f->W)) != 0) break;
    if ((r = gpx_mat_new(ctx, f->nu, f->n, 1, &f->Ks)) != 0) break;
    if ((r = launch_kfill(ctx, kp, S->p, f->nu, X->p, f->n, 0, nullptr, 0, 0.0, f->Kuf->p, f->nup, f->np,
                          f->Kuf->ld)) != 0) break;
    // (out of place through Lu's 1024-order block inverses, every product K >= 1024 on 128-tiles: 7.9 ms against 9.3 for the
    // in-place leaf recursion at nu = 4096, N = 32768.  The solve consumes its right-hand side: a second fill of Kuf into Ks --
    // which is only written further down -- costs 0.19 ms, the copy it replaces 0.61.)
    if ((r = launch_kfill(ctx, kp, S->p, f->nu, X->p, f->n, 0, nullptr, 0, 0.0, f->Ks->p, f->nup, f->np,
                          f->Ks->ld)) != 0) break;
    if ((r = chol_trsm_left_oop(ctx, f->Lu, f->Ks->p, f->Ks->ld, f->W->p, f->W->ld, f->np)) != 0) break;
    // g = diag(K) + noise - colsum(W^2)
    double *qd, *kd, *part;
    void* pg;
    if ((r = tmp.get(f->np * 8, &qd)) != 0) break;
    if ((r = tmp.get(f->np * 8, &kd)) != 0) break;
    if ((r = tmp.get(colreduce_partial_elems(f->nup, f->np) * 8 + 8, &part)) != 0) break;
    if ((r = gpx_dev_alloc(ctx, f->np * 8, &pg)) != 0) break;
    f->g = (double*)pg;
    if ((r = gpx_dev_alloc(ctx, f->np * 8, &pg)) != 0) break;
    f->ginv = (double*)pg;
    if ((r = launch_colreduce(ctx, f->W->p, f->W->ld, f->nu, f->np, nullptr, qd, part)) != 0) break;
    if ((r = launch_kdiag(ctx, kp, X->p, f->n, kd)) != 0) break;
    hipLaunchKernelGGL(fitc_g_kernel, dim3((unsigned)((f->np + 255) / 256)), dim3(256), 0, ctx->stream, kd, qd, noise,
                       f->n, f->np, f->g, f->ginv);
    hipLaunchKernelGGL(log_sum_kernel, dim3(1), dim3(256), 0, ctx->stream, f->g, f->n, ctx->d_scal);
    if (hipMemcpyAsync(&f->sumlogg, ctx->d_scal, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { r = -2; break; }
    // Ks = -Kuf Gi;  A = Quu + Kuf Gi Kfu = Quu - Ks Kuf^T;  La = chol(A)
    {
      dim3 grid((unsigned)((f->np / 2 + 255) / 256), (unsigned)f->nup);
      hipLaunchKernelGGL(scale_cols_neg_kernel, grid, dim3(256), 0, ctx->stream, f->Kuf->p, f->Kuf->ld, f->ginv, f->Ks->p,
                         f->Ks->ld, f->np);
    }
    if ((r = gpx_mat_new(ctx, f->nu, f->nu, 1, &f->La)) != 0) break;
    if ((r = launch_kfill(ctx, kp, S->p, f->nu, S->p, f->nu, 1, nullptr, 1, noise, f->La->p, f->nup, f->nup,
                          f->La->ld)) != 0) break;
    // nu x nu under a k range of N: as slices of the k range when C alone cannot fill the chip with 128-tiles (gemm_f64.hip,
    // launch_gemm_ksplit: 11.2 -> 9.0 ms at nu = 4096, N = 32768; 1024 or 4096 tiles wanted instead of 2048: no faster)
    {
      const int64_t t128 = (f->nup / 128) * (f->nup / 128 + 1) / 2;
      int64_t parts = 1;
      while (parts < 16 && t128 * parts < 2048 && f->np % (2 * parts * 16) == 0 && f->np / (2 * parts) >= 4096) parts *= 2;
      double* P = nullptr;
      if (parts > 1 && tmp.get(parts * f->nup * f->nup * 8, &P) != 0) parts = 1;   // (no room for the partials: one launch)
      if (parts > 1)
        r = launch_gemm_ksplit(ctx, f->Ks->p, f->Ks->ld, f->Kuf->p, f->Kuf->ld, f->La->p, f->La->ld, f->nup, f->nup, f->np, true,
                               parts, P);
      else
        r = launch_gemm(ctx, f->Ks->p, f->Ks->ld, f->Kuf->p, f->Kuf->ld, f->La->p, f->La->ld, f->nup, f->nup, f->np, true, true,
                        true);
      if (r != 0) break;
    }
    if ((r = factor_in_place(ctx, f->La, "Quu + Kuf G^-1 Kfu")) != 0) break;
    if (hipGetLastError() != hipSuccess) { r = -2; break; }
  } while (0);
  if (r != 0) {
    if (r == -2) gpx_set_error("fitc_fit: HIP failure: %s", hipGetErrorString(hipGetLastError()));
    fitc_release(ctx, f);
    return r;
  }
  *out = f;
  return 0;
}

int gpx_fitc_shape(const gpx_fitc* f, int64_t* n, int64_t* nu) {
  GPX_ARG(f != nullptr, "fitc model is NULL");
  if (n) *n = f->n;
  if (nu) *nu = f->nu;
  return 0;
}

// coeff = P y (host arrays of length n); quad = y^T P y
int gpx_fitc_solve(gpx_ctx* ctx, const gpx_fitc* f, const double* y, double* coeff, double* quad) {
  GPX_ARG(ctx && f && y && coeff, "NULL argument");
  Tmp tmp(ctx);
  double *dy, *du, *dt, *part;
  GPX_TRY(tmp.get(f->np * 8, &dy));
  GPX_TRY(tmp.get(f->nup * 8, &du));
  GPX_TRY(tmp.get(f->np * 8, &dt));
  GPX_TRY(tmp.get(colreduce_partial_elems(f->nup, f->np) * 8 + 8, &part));
  GPX_HIP(hipMemsetAsync(dy, 0, (size_t)f->np * 8, ctx->stream));
  GPX_HIP(hipMemcpyAsync(dy, y, (size_t)f->n * 8, hipMemcpyHostToDevice, ctx->stream));
  GPX_HIP(hipMemsetAsync(du, 0, (size_t)f->nup * 8, ctx->stream));
  // u = Kuf Gi y = -(Ks y);  w = A^-1 u;  t = Kfu w;  coeff = Gi (y - t)
  GPX_TRY(launch_rowreduce(ctx, f->Ks->p, f->Ks->ld, f->nu, f->np, dy, du));
  hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((f->nup + 255) / 256)), dim3(256), 0, ctx->stream, du, f->nup);
  // both sweeps against chol(A) through its explicit 1024-order block inverses (cached in La by the first solve): the
  // leaf-level sweeps stream every 512-row diagonal block through one workgroup (0.80 ms of the 1.4 ms at nu = 4096)
  double* ps;
  GPX_TRY(tmp.get(chol_potrs_scratch_bytes(f->nup), &ps));
  GPX_TRY(chol_potrs(ctx, f->La, du, ps));
  GPX_TRY(launch_colreduce(ctx, f->Kuf->p, f->Kuf->ld, f->nu, f->np, du, dt, part));
  hipLaunchKernelGGL(fitc_coeff_kernel, dim3((unsigned)((f->np + 255) / 256)), dim3(256), 0, ctx->stream, f->ginv, dy, dt,
                     f->np);
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(coeff, dt, (size_t)f->n * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  if (quad) {
    double s = 0.0;
    for (int64_t i = 0; i < f->n; ++i) s += y[i] * coeff[i];
    *quad = s;
  }
  return 0;
}

// log det(Q + G) = sum log g + log det A - log det Quu
int gpx_fitc_logdet(gpx_ctx* ctx, const gpx_fitc* f, double* out) {
  GPX_ARG(ctx && f && out, "NULL argument");
  double la = 0.0, lu = 0.0;
  GPX_TRY(launch_logdet(ctx, f->La->p, f->La->ld, f->nu, ctx->d_scal));
  GPX_HIP(hipMemcpyAsync(&la, ctx->d_scal, 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  GPX_TRY(launch_logdet(ctx, f->Lu->p, f->Lu->ld, f->nu, ctx->d_scal));
  GPX_HIP(hipMemcpyAsync(&lu, ctx->d_scal, 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  *out = f->sumlogg + la - lu;
  return 0;
}

// mean[j] = k(z_j, X) . coeff (coeff nullable),  var[j] = k(z,z) - k_z^T P k_z (signed; nullable)
int gpx_fitc_posterior(gpx_ctx* ctx, const gpx_fitc* f, const gpx_mat* X, const double* coeff, const gpx_mat* Z,
                       double* mean, double* var) {
  GPX_ARG(ctx && f && X && Z, "NULL argument");
  GPX_ARG(X->rows == f->n && X->cols == f->kp.d && Z->cols == f->kp.d && Z->pcols == f->kp.d, "point sets do not match");
  GPX_ARG(mean == nullptr || coeff != nullptr, "coeff is required for the mean");
  const int64_t M = Z->rows, d = f->kp.d, np = f->np, nup = f->nup;
  if (M == 0) return 0;
  KParams kpz = f->kp;  // the evaluation points may reach beyond the training domain
  GPX_TRY(gpx_kparams_sets(ctx, &kpz, X, Z));
  const int64_t mcmax = eval_chunk(np);
  const int64_t mc_alloc = gpx_round_up(M < mcmax ? M : mcmax, GPX_TILE);
  Tmp tmp(ctx);
  const int64_t ldb = gpx_skew_ld(np), ldu = gpx_skew_ld(nup);
  double *B, *U, *dc = nullptr, *o1, *o2, *kd;
  GPX_TRY(tmp.get(mc_alloc * ldb * 8, &B));
  GPX_TRY(tmp.get(mc_alloc * ldu * 8, &U));
  GPX_TRY(tmp.get(mc_alloc * 8, &o1));
  GPX_TRY(tmp.get(mc_alloc * 8, &o2));
  GPX_TRY(tmp.get(mc_alloc * 8, &kd));
  if (mean) {
    GPX_TRY(tmp.get(np * 8, &dc));
    GPX_HIP(hipMemsetAsync(dc, 0, (size_t)np * 8, ctx->stream));
    GPX_HIP(hipMemcpyAsync(dc, coeff, (size_t)f->n * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  std::vector<double> h1((size_t)mc_alloc), h2((size_t)mc_alloc), hk((size_t)mc_alloc);
  for (int64_t j0 = 0; j0 < M; j0 += mcmax) {
    const int64_t mc = (M - j0) < mcmax ? (M - j0) : mcmax;
    const int64_t mcp = gpx_round_up(mc, GPX_TILE);
    const double* Zc = Z->p + j0 * d;
    GPX_TRY(launch_kfill(ctx, kpz, Zc, mc, X->p, f->n, 0, nullptr, 0, 0.0, B, mcp, np, ldb));
    if (mean) {
      GPX_TRY(launch_rowreduce(ctx, B, ldb, mc, np, dc, o1));
      GPX_HIP(hipMemcpyAsync(mean + j0, o1, (size_t)mc * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (var) {
      GPX_TRY(launch_rowreduce(ctx, B, ldb, mc, np, f->ginv, o1, 1));                                  // sum Gi k^2
      GPX_TRY(launch_gemm(ctx, B, ldb, f->Ks->p, f->Ks->ld, U, ldu, mcp, nup, np, true, false, false));  // -(Kuf Gi k)^T
      GPX_TRY(chol_trsm_right(ctx, f->La->p, f->La->ld, f->La->aux, U, ldu, mcp, nup));
      GPX_TRY(launch_rowreduce(ctx, U, ldu, mc, nup, nullptr, o2));
      GPX_TRY(launch_kdiag(ctx, f->kp, Zc, mc, kd));
      GPX_HIP(hipMemcpyAsync(h1.data(), o1, (size_t)mc * 8, hipMemcpyDeviceToHost, ctx->stream));
      GPX_HIP(hipMemcpyAsync(h2.data(), o2, (size_t)mc * 8, hipMemcpyDeviceToHost, ctx->stream));
      GPX_HIP(hipMemcpyAsync(hk.data(), kd, (size_t)mc * 8, hipMemcpyDeviceToHost, ctx->stream));
      GPX_HIP(hipStreamSynchronize(ctx->stream));
      for (int64_t j = 0; j < mc; ++j) var[j0 + j] = hk[(size_t)j] - h1[(size_t)j] + h2[(size_t)j];
    }
  }
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// Dense Q + G and P (host, n x n row-major, each nullable): the reference's covarianceMatrix / precisionMatrix attributes.
int gpx_fitc_dense(gpx_ctx* ctx, const gpx_fitc* f, double* cov, double* prec) {
  GPX_ARG(ctx && f, "NULL argument");
  const int64_t n = f->n, np = f->np, nup = f->nup;
  Tmp tmp(ctx);
  double *T, *C;
  const int64_t ldt = gpx_skew_ld(nup), ldc = gpx_skew_ld(np);
  GPX_TRY(tmp.get(np * ldt * 8, &T));
  GPX_TRY(tmp.get(np * ldc * 8, &C));
  std::vector<double> hv((size_t)np);
  for (int which = 0; which < 2; ++which) {
    double* dst = which == 0 ? cov : prec;
    if (!dst) continue;
    const double* dvec = which == 0 ? f->g : f->ginv;
    if (which == 0) {
      GPX_TRY(launch_transpose(ctx, f->W->p, nup, np, f->W->ld, T, ldt));  // W^T (np x nup)
    } else {
      double* Y;
      GPX_TRY(tmp.get(nup * f->Ks->ld * 8, &Y));
      GPX_TRY(gpx_copy2d(ctx, f->Ks->p, f->Ks->ld, Y, f->Ks->ld, nup, np));
      GPX_TRY(chol_trsm_left(ctx, f->La->p, f->La->ld, f->La->aux, Y, f->Ks->ld, nup, np));  // La^-1 Ks
      GPX_TRY(launch_transpose(ctx, Y, nup, np, f->Ks->ld, T, ldt));
    }
    GPX_TRY(launch_gemm(ctx, T, ldt, T, ldt, C, ldc, np, np, nup, true, false, false));  // T T^T
    GPX_HIP(hipMemcpy2DAsync(dst, (size_t)n * 8, C, (size_t)ldc * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToHost,
                             ctx->stream));
    GPX_HIP(hipMemcpyAsync(hv.data(), dvec, (size_t)np * 8, hipMemcpyDeviceToHost, ctx->stream));
    GPX_HIP(hipStreamSynchronize(ctx->stream));
    if (which == 0) {
      for (int64_t i = 0; i < n; ++i) dst[i * n + i] += hv[(size_t)i];  // Q + diag(g)
    } else {
      for (int64_t i = 0; i < n * n; ++i) dst[i] = -dst[i];
      for (int64_t i = 0; i < n; ++i) dst[i * n + i] += hv[(size_t)i];  // diag(ginv) - Y^T Y
    }
  }
  return 0;
}

}  // extern "C"
