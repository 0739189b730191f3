// Design-cost evaluators and the on-request dense inverses -- gfx950.
//
//  gpx_greedy_var        performGreedyVarExperimentalDesign (experimentalDesign.py:787-845): the reference
//                        re-inverts K(sel,sel) with pinv and loops over all M candidates in Python at every
//                        step; here the same conditional variances are carried incrementally as a partial
//                        pivoted Cholesky of K(C,C): one kernel row + one O(M*n) update per step.
//  gpx_greedy_ivar_step  one step of discrete greedy IVAR (composition oracle):
//                        IVAR(X u {c}) = [sum_z var(z|X) - sum_z cov(z,c|X)^2 / (var(c|X)+noise)] / nMC with
//                        cov(Z,C|X) = K(Z,C) - W_Z^T W_C as one MFMA GEMM instead of M refits.
//  gpx_posterior_cov     GP.evaluate(compvar=2) (gp.py:146-152).
//  gpx_potri             explicit K^-1 for the lazy GP.precisionMatrix attribute (gp.py:181).
#include "gpx_internal.h"
#include <math.h>
#include <stdlib.h>

namespace {

// ---- 32x32 tiled transpose: out[j][i] = in[i][j] ----------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ in, int64_t ldi,
                                                        double* __restrict__ out, int64_t ldo) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = in[(i0 + r) * ldi + j0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) out[(j0 + r) * ldo + i0 + tx] = tile[tx][r];
}

// upper triangle <- transpose of the lower one (32 x 32 tiles, grid = tiles x tiles; tiles above the diagonal exit)
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ P, int64_t ld) {
  if (blockIdx.x > blockIdx.y) return;
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;   // tile (i0, j0) on / below the diagonal
  for (int r = ty; r < 32; r += 8) tile[r][tx] = P[(i0 + r) * ld + j0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t gi = j0 + r, gj = i0 + tx;      // element (gi, gj) of the upper part = tile[tx][r]
    if (gj > gi) P[gi * ld + gj] = tile[tx][r];
  }
}

__global__ __launch_bounds__(256) void set_identity_kernel(double* __restrict__ A, int64_t n, int64_t ld) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < n) A[idx * ld + idx] = 1.0;
}

// ---- covariance value for a pair of points given in global memory (generic d) -------------------------
__device__ __forceinline__ double kpair(const KParams& kp, const double* __restrict__ a, const double* __restrict__ b) {
  double acc = 0.0;
  if (kp.kind == GPX_K_MEHLER) {
    double pa = 0.0, pb = 0.0, cr = 0.0;
    for (int k = 0; k < kp.d; ++k) {
      const double x = a[k], y = b[k];
      pa = fma(kp.c1[k] * x, x, pa);
      pb = fma(kp.c1[k] * y, y, pb);
      cr = fma(kp.c2[k] * x, y, cr);
    }
    return kp.sig * exp(-(pa + pb - cr));
  }
  for (int k = 0; k < kp.d; ++k) {
    const double e = (a[k] - b[k]) * kp.scale[k];  // difference first, as the reference (kernels.py:121-122)
    acc = fma(e, e, acc);
  }
  if (kp.kind == GPX_K_SE) return kp.sig * exp(-0.5 * acc);
  const double t = sqrt(acc);
  if (kp.kind == GPX_K_MATERN32) return kp.sig * (1.0 + t) * exp(-t);
  return kp.sig * (1.0 + t + acc * (1.0 / 3.0)) * exp(-t);
}

// ---- greedy variance ---------------------------------------------------------------------------------------
// One selection step.  sel[cur] holds the index s chosen for this step.  For every candidate c:
//   w_c = (k(c_s, c) - sum_{t<cur} W[t][s] W[t][c]) / sqrt(d_s);  W[cur][c] = w_c;  d_out[c] = d_in[c] - w_c^2
// A pivot with d_s <= tol*prior_s adds no information (the reference's pinv truncates it): w = 0.
__global__ __launch_bounds__(256) void greedy_row_kernel(KParams kp, const double* __restrict__ Cp, int64_t M,
                                                         const int64_t* __restrict__ sel, int cur,
                                                         double* __restrict__ W, int64_t ldw,
                                                         const double* __restrict__ prior,
                                                         const double* __restrict__ d_in, double* __restrict__ d_out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= M) return;
  const int64_t s = sel[cur];
  const double ds = d_in[s];
  double w = 0.0;
  if (ds > 1e-13 * prior[s]) {
    double dot = 0.0;
    for (int t = 0; t < cur; ++t) dot = fma(W[(int64_t)t * ldw + s], W[(int64_t)t * ldw + c], dot);
    w = (kpair(kp, Cp + s * kp.d, Cp + c * kp.d) - dot) / sqrt(ds);
  }
  W[(int64_t)cur * ldw + c] = w;
  d_out[c] = fma(-w, w, d_in[c]);
}

// first-maximum arg-max of d[c]*w[c] (np.argmax tie rule: lowest index), two stages, deterministic
struct VI {
  double v;
  int64_t i;
};
__device__ __forceinline__ VI vi_max(VI a, VI b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

__global__ __launch_bounds__(256) void argmax_stage1(const double* __restrict__ d, const double* __restrict__ w,
                                                     int64_t M, double* __restrict__ pv, int64_t* __restrict__ pi) {
  __shared__ double sv[256];
  __shared__ int64_t si[256];
  VI best;
  best.v = -INFINITY;
  best.i = INT64_MAX;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < M; c += (int64_t)gridDim.x * 256) {
    VI x;
    x.v = w ? d[c] * w[c] : d[c];
    x.i = c;
    best = vi_max(best, x);
  }
  sv[threadIdx.x] = best.v;
  si[threadIdx.x] = best.i;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) {
      VI a{sv[threadIdx.x], si[threadIdx.x]}, b{sv[threadIdx.x + h], si[threadIdx.x + h]};
      VI m = vi_max(a, b);
      sv[threadIdx.x] = m.v;
      si[threadIdx.x] = m.i;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pv[blockIdx.x] = sv[0];
    pi[blockIdx.x] = si[0];
  }
}

__global__ __launch_bounds__(256) void argmax_stage2(const double* __restrict__ pv, const int64_t* __restrict__ pi,
                                                     int nb, int64_t* __restrict__ sel, int slot) {
  __shared__ double sv[256];
  __shared__ int64_t si[256];
  VI best;
  best.v = -INFINITY;
  best.i = INT64_MAX;
  for (int b = threadIdx.x; b < nb; b += 256) best = vi_max(best, VI{pv[b], pi[b]});
  sv[threadIdx.x] = best.v;
  si[threadIdx.x] = best.i;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) {
      VI m = vi_max(VI{sv[threadIdx.x], si[threadIdx.x]}, VI{sv[threadIdx.x + h], si[threadIdx.x + h]});
      sv[threadIdx.x] = m.v;
      si[threadIdx.x] = m.i;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) sel[slot] = si[0] == INT64_MAX ? 0 : si[0];  // all-NaN scores: index 0 like np.argmax
}

__global__ __launch_bounds__(256) void keval_kernel(KParams kp, const double* __restrict__ A, int64_t sa,
                                                    const double* __restrict__ B, int64_t sb, int64_t n,
                                                    double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = kpair(kp, A + i * sa, B + i * sb);
}

// cost_j = | (S0 - q_j / (kcc_j - ssc_j + noise)) / nmc |
__global__ __launch_bounds__(256) void ivar_cost_kernel(const double* __restrict__ q, const double* __restrict__ kcc,
                                                        const double* __restrict__ ssc, double noise, double s0,
                                                        double inv_nmc, int64_t M, double* __restrict__ cost) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  cost[j] = fabs((s0 - q[j] / (kcc[j] - ssc[j] + noise)) * inv_nmc);
}

struct Scratch {  // pooled device buffers released together
  gpx_ctx* ctx;
  std::vector<std::pair<void*, int64_t>> bufs;
  explicit Scratch(gpx_ctx* c) : ctx(c) {}
  int get(int64_t bytes, void** out) {
    int r = gpx_dev_alloc(ctx, bytes, out);
    if (r == 0) bufs.push_back({*out, bytes});
    return r;
  }
  ~Scratch() {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : bufs) gpx_dev_release(ctx, b.first, b.second);
  }
};

int check_points(const gpx_mat* P, int d) {
  GPX_ARG(P && P->cols == d && P->pcols == d, "point sets must be unpadded (n x d)");
  return 0;
}

double pairwise_sum(std::vector<double>& v, int64_t m) {
  if (m == 0) return 0.0;
  while (m > 1) {
    int64_t h = (m + 1) / 2;
    for (int64_t i = 0; i + h < m; ++i) v[(size_t)i] += v[(size_t)(i + h)];
    m = h;
  }
  return v[0];
}


// ---- multi-pick greedy IVAR with resident state (gpx_givar_*) -------------------------------------------------------------
// cost_j = | (S0 - q_j / (v_j + noise)) / nmc |, first minimum (np.argmin tie rule: lowest index), two stages, deterministic
__global__ __launch_bounds__(256) void givar_cost_kernel(const double* __restrict__ q, const double* __restrict__ v, double noise,
                                                         double s0, double inv_nmc, int64_t M, double* __restrict__ cost) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  //  a candidate that has been picked with noise == 0 has v -> 0 and q -> 0: 0 / 0.  np.argmin would return the first
  // NaN, the first-minimum reduction below never selects one -- so such a candidate is taken out of the race explicitly: +inf
  // (also a denominator at round-off level or below: conditioning on the point again cannot reduce anything)
  const double den = v[j] + noise;
  double c = den > 1e-300 ? fabs((s0 - q[j] / den) * inv_nmc) : INFINITY;
  cost[j] = (c == c) ? c : INFINITY;
}

__device__ __forceinline__ VI vi_min(VI a, VI b) {
  if (b.v < a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

__global__ __launch_bounds__(256) void argmin_stage1(const double* __restrict__ c, int64_t M, double* __restrict__ pv,
                                                     int64_t* __restrict__ pi) {
  __shared__ double sv[256];
  __shared__ int64_t si[256];
  VI best;
  best.v = INFINITY;
  best.i = INT64_MAX;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < M; j += (int64_t)gridDim.x * 256) {
    VI x;
    x.v = c[j];
    x.i = j;
    best = vi_min(best, x);
  }
  sv[threadIdx.x] = best.v;
  si[threadIdx.x] = best.i;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) {
      VI a{sv[threadIdx.x], si[threadIdx.x]}, b{sv[threadIdx.x + h], si[threadIdx.x + h]};
      VI m = vi_min(a, b);
      sv[threadIdx.x] = m.v;
      si[threadIdx.x] = m.i;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pv[blockIdx.x] = sv[0];
    pi[blockIdx.x] = si[0];
  }
}

__global__ __launch_bounds__(256) void argmin_stage2(const double* __restrict__ pv, const int64_t* __restrict__ pi, int nb,
                                                     double* __restrict__ out_v, int64_t* __restrict__ out_i) {
  __shared__ double sv[256];
  __shared__ int64_t si[256];
  VI best;
  best.v = INFINITY;
  best.i = INT64_MAX;
  for (int b = threadIdx.x; b < nb; b += 256) {
    VI x{pv[b], pi[b]};
    best = vi_min(best, x);
  }
  sv[threadIdx.x] = best.v;
  si[threadIdx.x] = best.i;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) {
      VI a{sv[threadIdx.x], si[threadIdx.x]}, b{sv[threadIdx.x + h], si[threadIdx.x + h]};
      VI m = vi_min(a, b);
      sv[threadIdx.x] = m.v;
      si[threadIdx.x] = m.i;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *out_v = sv[0];
    *out_i = si[0];
  }
}

// v_j = k(c_j, c_j) - |W_C[:, j]|^2
__global__ __launch_bounds__(256) void givar_v_kernel(const double* __restrict__ kd, const double* __restrict__ ss, int64_t M,
                                                      int64_t Mp, double* __restrict__ v) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < Mp) v[j] = j < M ? kd[j] - ss[j] : 1.0;
}

// The pivot pack of candidate s (local index), everything a rank needs to condition ITS candidates on it:
//   [0] delta = v_s + noise   [1] |r|^2 (filled by givar_rr_kernel)   [2 .. 2+d) the point c_s
//   r[zp]   = cov(Z, c_s | design) / sqrt(delta)            (column s of G)
//   w[np]   = W_C[:, s]                                       (c_s in the whitened basis of the ORIGINAL training set)
//   uc[nsel] = U[t][s], t < cur                                (its coordinates along the earlier picks)
__global__ __launch_bounds__(256) void givar_pack_kernel(const double* __restrict__ G, const double* __restrict__ Wc,
                                                         const double* __restrict__ U, const double* __restrict__ v,
                                                         const double* __restrict__ Cp, int d, int64_t Mp, int64_t zp,
                                                         int64_t np, int64_t nsel, int cur, int64_t s, double noise,
                                                         double* __restrict__ buf) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double delta = v[s] + noise;
  const double rs = 1.0 / sqrt(delta);
  double* r = buf + 2 + d;
  double* w = r + zp;
  double* uc = w + np;
  if (i == 0) buf[0] = delta;
  if (i < d) buf[2 + i] = Cp[s * d + i];
  if (i < zp) r[i] = G[i * Mp + s] * rs;
  if (i < np) w[i] = Wc[i * Mp + s];
  if (i < nsel) uc[i] = i < cur ? U[i * Mp + s] : 0.0;
}

// buf[1] = sum_z r[z]^2 in a fixed order (one workgroup)
__global__ __launch_bounds__(256) void givar_rr_kernel(double* __restrict__ buf, int d, int64_t zp) {
  __shared__ double sh[256];
  const double* r = buf + 2 + d;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < zp; i += 256) acc = fma(r[i], r[i], acc);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) buf[1] = sh[0];
}

// u_j = (k(c_s, c_j) - w^T W_C[:, j] - sum_t uc[t] U[t][j]) / sqrt(delta);  U[cur][j] = u_j;  v_j -= u_j^2
__global__ __launch_bounds__(256) void givar_u_kernel(KParams kp, const double* __restrict__ Cp, int64_t M, int64_t Mp,
                                                      const double* __restrict__ buf, int64_t zp, int64_t np,
                                                      const double* __restrict__ hdot, double* __restrict__ U, int cur,
                                                      double* __restrict__ v) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  double u = 0.0;
  if (j < M) {
    const double* uc = buf + 2 + kp.d + zp + np;
    double acc = kpair(kp, buf + 2, Cp + j * kp.d) - hdot[j];
    for (int t = 0; t < cur; ++t) acc = fma(-uc[t], U[(int64_t)t * Mp + j], acc);
    u = acc / sqrt(buf[0]);
    v[j] = fma(-u, u, v[j]);
  }
  U[(int64_t)cur * Mp + j] = u;
}

// G[z][j] -= r[z] u[j] over a chunk of rows, and the chunk's share of q_j = sum_z G[z][j]^2 (fixed order: rows inside a
// chunk, then chunks -- givar_q_kernel)
__global__ __launch_bounds__(256) void givar_rank1_kernel(double* __restrict__ G, int64_t Mp, int64_t zp, int64_t chunk,
                                                          const double* __restrict__ r, const double* __restrict__ u,
                                                          double* __restrict__ partial) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t z0 = (int64_t)blockIdx.y * chunk;
  int64_t z1 = z0 + chunk;
  if (z1 > zp) z1 = zp;
  if (j >= Mp) return;
  const double uj = u[j];
  double s0 = 0.0, s1 = 0.0;
  double* p = G + z0 * Mp + j;
  int64_t z = z0;
  for (; z + 2 <= z1; z += 2, p += 2 * Mp) {
    const double a = fma(-r[z], uj, p[0]), b = fma(-r[z + 1], uj, p[Mp]);
    p[0] = a;
    p[Mp] = b;
    s0 = fma(a, a, s0);
    s1 = fma(b, b, s1);
  }
  for (; z < z1; ++z, p += Mp) {
    const double a = fma(-r[z], uj, p[0]);
    p[0] = a;
    s0 = fma(a, a, s0);
  }
  partial[(int64_t)blockIdx.y * Mp + j] = s0 + s1;
}

__global__ __launch_bounds__(256) void givar_q_kernel(const double* __restrict__ partial, int64_t nchunk, int64_t Mp,
                                                      double* __restrict__ q) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  double s = 0.0;
  for (int64_t c = 0; c < nchunk; ++c) s += partial[c * Mp + j];
  q[j] = s;
}

}  // namespace

int launch_transpose(gpx_ctx* ctx, const double* in, int64_t rows, int64_t cols, int64_t ldi, double* out,
                     int64_t ldo) {
  dim3 grid((unsigned)(cols / 32), (unsigned)(rows / 32));
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, ctx->stream, in, ldi, out, ldo);
  GPX_HIP(hipGetLastError());
  return 0;
}

extern "C" {

int gpx_kernel_eval(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const double* A, int64_t na,
                    const double* B, int64_t nb, double* out) {
  GPX_ARG(ctx && A && B && out, "NULL argument");
  GPX_ARG(na >= 1 && nb >= 1 && (na == nb || na == 1 || nb == 1), "evaluate needs paired or one-vs-n point sets");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  const int64_t n = na > nb ? na : nb;
  Scratch sc(ctx);
  void *pa, *pb, *po;
  GPX_TRY(sc.get(na * d * 8, &pa));
  GPX_TRY(sc.get(nb * d * 8, &pb));
  GPX_TRY(sc.get(n * 8, &po));
  GPX_HIP(hipMemcpyAsync(pa, A, (size_t)(na * d * 8), hipMemcpyHostToDevice, ctx->stream));
  GPX_HIP(hipMemcpyAsync(pb, B, (size_t)(nb * d * 8), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(keval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, kp, (const double*)pa,
                     (int64_t)(na == 1 ? 0 : d), (const double*)pb, (int64_t)(nb == 1 ? 0 : d), n, (double*)po);
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(out, po, (size_t)(n * 8), hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_posterior_cov(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                      const gpx_mat* Z, double* cov) {
  GPX_ARG(ctx && L && X && Z && cov, "NULL argument");
  GPX_ARG(L->factored && L->aux, "matrix has not been factored by gpx_potrf");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(check_points(X, d));
  GPX_TRY(check_points(Z, d));
  const int64_t n = L->rows, np = L->prows, m = Z->rows, mp = gpx_round_up(m > 0 ? m : 1, GPX_TILE);
  GPX_ARG(X->rows == n, "X does not match the factor");
  if (m == 0) return 0;
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z));
  Scratch sc(ctx);
  void *pW, *pWt, *pK;
  GPX_TRY(sc.get(np * mp * 8, &pW));
  GPX_TRY(sc.get(np * mp * 8, &pWt));
  GPX_TRY(sc.get(mp * mp * 8, &pK));
  GPX_TRY(launch_kfill(ctx, kp, X->p, n, Z->p, m, 0, nullptr, 0, 0.0, (double*)pW, np, mp, mp));
  GPX_TRY(chol_trsm_left(ctx, L->p, L->ld, L->aux, (double*)pW, mp, np, mp));
  GPX_TRY(launch_transpose(ctx, (double*)pW, np, mp, mp, (double*)pWt, np));
  GPX_TRY(launch_kfill(ctx, kp, Z->p, m, Z->p, m, 1, nullptr, 0, 0.0, (double*)pK, mp, mp, mp));
  GPX_TRY(launch_gemm(ctx, (double*)pWt, np, (double*)pWt, np, (double*)pK, mp, mp, mp, np, true, true, false));
  GPX_HIP(hipMemcpy2DAsync(cov, (size_t)m * 8, pK, (size_t)mp * 8, (size_t)m * 8, (size_t)m, hipMemcpyDeviceToHost,
                           ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// K^-1 = L^-T L^-1.  full == 0 (the C ABI's contract: lower triangle valid): L^-1 by the halving recursion (n^3/2 flops,
// chol_trtri), its transpose U, and the lower tiles of U U^T with the k range of every tile starting at its row block
// (n^3/3): 0.83 n^3 instead of the 3 n^3 of round 1 (triangular solve against a dense identity + full product).
// full != 0 (the mutual-information design updates the whole matrix in place): the product fills both triangles.
int gpx_potri_impl(gpx_ctx* ctx, const gpx_mat* L, gpx_mat** outP, int full) {
  GPX_ARG(ctx && L && outP, "NULL argument");
  GPX_ARG(L->factored && L->aux, "matrix has not been factored by gpx_potrf");
  const int64_t np = L->prows;
  gpx_mat* P = nullptr;
  GPX_TRY(gpx_mat_new(ctx, L->rows, L->cols, 1, &P));
  Scratch sc(ctx);
  void *pI, *pT, *ptmp;
  int r = 0;
  do {
    if ((r = sc.get(np * np * 8, &pI)) != 0) break;
    if ((r = sc.get(np * np * 8, &pT)) != 0) break;
    if ((r = sc.get((np / 2 + 64) * (np / 2 + 64) * 8, &ptmp)) != 0) break;
    if ((r = chol_trtri(ctx, L, (double*)pI, (double*)ptmp)) != 0) break;                           // L^-1
    if ((r = launch_transpose(ctx, (double*)pI, np, np, np, (double*)pT, np)) != 0) break;          // U = L^-T
    // lower triangle of U U^T with the structurally zero part of every tile's k range skipped (N^3/3); `full`: the upper one is
    // its mirror image (two passes over N^2 doubles) -- round 3 formed both triangles by a dense product (2 N^3)
    if ((r = launch_gemm_tri(ctx, (double*)pT, np, (double*)pT, np, P->p, P->ld, np, np, np, true, false, true, 3)) != 0) break;
    if (full) {
      hipLaunchKernelGGL(mirror_lower_kernel, dim3((unsigned)(np / 32), (unsigned)(np / 32)), dim3(256), 0, ctx->stream, P->p,
                         P->ld);
      if (hipGetLastError() != hipSuccess) r = -2;
    }
  } while (0);
  if (r != 0) {
    gpx_mat_free(ctx, P);
    if (r == -2) gpx_set_error("potri: HIP call failed");
    return r;
  }
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  *outP = P;
  return 0;
}

int gpx_potri(gpx_ctx* ctx, const gpx_mat* L, gpx_mat** outP) { return gpx_potri_impl(ctx, L, outP, 0); }

int gpx_greedy_var(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* Cm, const double* w,
                   const int64_t* keep, int64_t nkeep, int64_t nsel, int64_t* out_idx) {
  GPX_ARG(ctx && Cm && out_idx, "NULL argument");
  GPX_ARG(nkeep >= 0 && nsel >= nkeep && (nkeep == 0 || keep), "bad keep/nsel");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(check_points(Cm, d));
  const int64_t M = Cm->rows;
  GPX_ARG(M > 0, "no candidates");
  for (int64_t i = 0; i < nkeep; ++i) GPX_ARG(keep[i] >= 0 && keep[i] < M, "keep index out of range");
  if (nsel == 0) return 0;
  Scratch sc(ctx);
  const int nb = (int)(((M + 255) / 256) < 1024 ? ((M + 255) / 256) : 1024);
  void *pW, *pd0, *pd1, *pprior, *psel, *ppv, *ppi, *pw = nullptr;
  GPX_TRY(sc.get(nsel * M * 8, &pW));
  GPX_TRY(sc.get(M * 8, &pd0));
  GPX_TRY(sc.get(M * 8, &pd1));
  GPX_TRY(sc.get(M * 8, &pprior));
  GPX_TRY(sc.get(nsel * 8, &psel));
  GPX_TRY(sc.get(nb * 8, &ppv));
  GPX_TRY(sc.get(nb * 8, &ppi));
  if (w) {
    GPX_TRY(sc.get(M * 8, &pw));
    GPX_HIP(hipMemcpyAsync(pw, w, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  if (nkeep > 0) GPX_HIP(hipMemcpyAsync(psel, keep, (size_t)nkeep * 8, hipMemcpyHostToDevice, ctx->stream));
  GPX_TRY(launch_kdiag(ctx, kp, Cm->p, M, (double*)pprior));
  GPX_HIP(hipMemcpyAsync(pd0, pprior, (size_t)M * 8, hipMemcpyDeviceToDevice, ctx->stream));
  double* din = (double*)pd0;
  double* dout = (double*)pd1;
  const dim3 gridM((unsigned)((M + 255) / 256));
  {
    ProfScope ps(ctx, GPX_PROF_GREEDY, (double)M * nsel * nsel, 0.0);
    for (int64_t cur = 0; cur < nsel; ++cur) {
      if (cur >= nkeep) {
        hipLaunchKernelGGL(argmax_stage1, dim3(nb), dim3(256), 0, ctx->stream, din, (const double*)pw, M, (double*)ppv,
                           (int64_t*)ppi);
        hipLaunchKernelGGL(argmax_stage2, dim3(1), dim3(256), 0, ctx->stream, (const double*)ppv, (const int64_t*)ppi,
                           nb, (int64_t*)psel, (int)cur);
      }
      if (cur + 1 < nsel) {  // the last pick needs no further conditioning
        hipLaunchKernelGGL(greedy_row_kernel, gridM, dim3(256), 0, ctx->stream, kp, Cm->p, M, (const int64_t*)psel,
                           (int)cur, (double*)pW, M, (const double*)pprior, (const double*)din, dout);
        double* t = din;
        din = dout;
        dout = t;
      }
    }
  }
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(out_idx, psel, (size_t)nsel * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int gpx_greedy_ivar_step(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L,
                         const gpx_mat* X, const gpx_mat* Cm, const gpx_mat* Z, double noise, double* out_cost,
                         int64_t* out_best) {
  GPX_ARG(ctx && L && X && Cm && Z && out_best, "NULL argument");
  GPX_ARG(L->factored && L->aux, "matrix has not been factored by gpx_potrf");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(check_points(X, d));
  GPX_TRY(check_points(Cm, d));
  GPX_TRY(check_points(Z, d));
  const int64_t n = L->rows, np = L->prows, M = Cm->rows, nmc = Z->rows;
  GPX_ARG(X->rows == n, "X does not match the factor");
  GPX_ARG(M > 0 && nmc > 0, "need candidates and integration points");
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z, Cm));
  const int64_t zp = gpx_round_up(nmc, GPX_TILE);
  // candidate chunk: W_C (np x mc) + G (zp x mc) under ~24 GiB
  int64_t budget = (int64_t)24 << 30;
  const char* e = getenv("GPX_CROSS_BYTES");
  if (e && atoll(e) > 0) budget = atoll(e);
  int64_t mcmax = budget / ((np + zp) * 8) / GPX_TILE * GPX_TILE;
  if (mcmax < GPX_TILE) mcmax = GPX_TILE;
  const int64_t mc_alloc = gpx_round_up(M < mcmax ? M : mcmax, GPX_TILE);

  Scratch sc(ctx);
  void *pWz, *pWzt, *pWc, *pG, *pss, *pkd, *pq, *ppart, *pcost;
  GPX_TRY(sc.get(np * zp * 8, &pWz));
  GPX_TRY(sc.get(np * zp * 8, &pWzt));
  GPX_TRY(sc.get(np * mc_alloc * 8, &pWc));
  GPX_TRY(sc.get(zp * mc_alloc * 8, &pG));
  const int64_t wide = zp > mc_alloc ? zp : mc_alloc;
  GPX_TRY(sc.get(wide * 8, &pss));
  GPX_TRY(sc.get(wide * 8, &pkd));
  GPX_TRY(sc.get(wide * 8, &pq));
  GPX_TRY(sc.get(wide * 8, &pcost));
  int64_t part = colreduce_partial_elems(np, wide);
  int64_t part2 = colreduce_partial_elems(zp, wide);
  GPX_TRY(sc.get((part > part2 ? part : part2) * 8 + 8, &ppart));

  // ---- integration points: W_Z = L^-1 K(X,Z), var_z, S0 ----
  GPX_TRY(launch_kfill(ctx, kp, X->p, n, Z->p, nmc, 0, nullptr, 0, 0.0, (double*)pWz, np, zp, zp));
  GPX_TRY(chol_trsm_left(ctx, L->p, L->ld, L->aux, (double*)pWz, zp, np, zp));
  GPX_TRY(launch_colreduce(ctx, (double*)pWz, zp, n, zp, nullptr, (double*)pss, (double*)ppart));
  GPX_TRY(launch_kdiag(ctx, kp, Z->p, nmc, (double*)pkd));
  std::vector<double> hs((size_t)nmc), hk((size_t)nmc);
  GPX_HIP(hipMemcpyAsync(hs.data(), pss, (size_t)nmc * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipMemcpyAsync(hk.data(), pkd, (size_t)nmc * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_TRY(launch_transpose(ctx, (double*)pWz, np, zp, zp, (double*)pWzt, np));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  for (int64_t j = 0; j < nmc; ++j) hk[(size_t)j] -= hs[(size_t)j];
  const double s0 = pairwise_sum(hk, nmc);

  // ---- candidates, chunked ----
  std::vector<double> cost((size_t)M);
  for (int64_t j0 = 0; j0 < M; j0 += mcmax) {
    const int64_t mc = (M - j0) < mcmax ? (M - j0) : mcmax;
    const int64_t mcp = gpx_round_up(mc, GPX_TILE);
    const double* Cc = Cm->p + j0 * d;
    GPX_TRY(launch_kfill(ctx, kp, X->p, n, Cc, mc, 0, nullptr, 0, 0.0, (double*)pWc, np, mcp, mcp));
    GPX_TRY(chol_trsm_left(ctx, L->p, L->ld, L->aux, (double*)pWc, mcp, np, mcp));
    GPX_TRY(launch_colreduce(ctx, (double*)pWc, mcp, n, mcp, nullptr, (double*)pss, (double*)ppart));
    GPX_TRY(launch_kdiag(ctx, kp, Cc, mc, (double*)pkd));
    // G = K(Z,C) - W_Z^T W_C
    GPX_TRY(launch_kfill(ctx, kp, Z->p, nmc, Cc, mc, 0, nullptr, 0, 0.0, (double*)pG, zp, mcp, mcp));
    GPX_TRY(launch_gemm(ctx, (double*)pWzt, np, (double*)pWc, mcp, (double*)pG, mcp, zp, mcp, np, false, true, false));
    GPX_TRY(launch_colreduce(ctx, (double*)pG, mcp, nmc, mcp, nullptr, (double*)pq, (double*)ppart));
    hipLaunchKernelGGL(ivar_cost_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)pq, (const double*)pkd, (const double*)pss, noise, s0, 1.0 / (double)nmc, mc,
                       (double*)pcost);
    GPX_HIP(hipGetLastError());
    GPX_HIP(hipMemcpyAsync(cost.data() + j0, pcost, (size_t)mc * 8, hipMemcpyDeviceToHost, ctx->stream));
    GPX_HIP(hipStreamSynchronize(ctx->stream));
  }
  int64_t best = 0;
  for (int64_t j = 1; j < M; ++j)
    if (cost[(size_t)j] < cost[(size_t)best]) best = j;  // first minimum (np.argmin)
  *out_best = best;
  if (out_cost)
    for (int64_t j = 0; j < M; ++j) out_cost[j] = cost[(size_t)j];
  return 0;
}


// =====================================================================================================================
// Multi-pick greedy IVAR with RESIDENT state (composition of experimentalDesign.py:79-117 per SURVEY 8c)
// =====================================================================================================================
// gpx_greedy_ivar_step scores ONE pick from scratch: two N x (M + nMC) triangular solves (385 ms at C3), and a k-point design
// costs k x (refit + that).  Conditioning on a picked candidate c_s is a rank-one change of every quantity the score needs:
//   delta = var(c_s | D) + noise,   r = cov(Z, c_s | D) / sqrt(delta),   u = cov(c_s, C | D) / sqrt(delta)
//   cov(Z, C | D + s) = G - r u^T,   var(c_j | D + s) = v_j - u_j^2,   sum_z var(z | D + s) = S0 - |r|^2
// with cov(c_s, C | D) = k(c_s, C) - W_C[:, s]^T W_C - sum_t U[t][s] U[t]  (W_C = L^-1 K(X, C) of the ORIGINAL training set stays
// resident; the earlier picks enter through their own rows U[t]: a partial Cholesky, like gpx_greedy_var).  Per pick: one
// weighted column reduction over W_C (8 N M bytes) and one read-modify-write of G (16 nMC M bytes) -- no refit, no N^2 solve.
// The state is sharded by CANDIDATES for the multi-GPU form (every rank: its slice of C; the pivot pack travels): the per-candidate
// arithmetic does not depend on the sharding.
struct gpx_givar {
  KParams kp;
  int d;
  const gpx_mat* Cm;
  int64_t n, np, M, Mp, nmc, zp, nsel, cur, chunk, nchunk, pack;
  double noise, s0;
  double *Wc, *G, *U, *v, *q, *cost, *hdot, *part, *qpart, *buf, *red;
  int64_t* redi;
  int64_t part_elems;
};

int gpx_givar_end(gpx_ctx* ctx, gpx_givar* st) {
  if (!st) return 0;
  GPX_ARG(ctx != nullptr, "ctx is NULL");
  (void)hipDeviceSynchronize();
  if (st->Wc) gpx_dev_release(ctx, st->Wc, st->np * st->Mp * 8);
  if (st->G) gpx_dev_release(ctx, st->G, st->zp * st->Mp * 8);
  if (st->U) gpx_dev_release(ctx, st->U, st->nsel * st->Mp * 8);
  if (st->v) gpx_dev_release(ctx, st->v, st->Mp * 8);
  if (st->q) gpx_dev_release(ctx, st->q, st->Mp * 8);
  if (st->cost) gpx_dev_release(ctx, st->cost, st->Mp * 8);
  if (st->hdot) gpx_dev_release(ctx, st->hdot, st->Mp * 8);
  if (st->part) gpx_dev_release(ctx, st->part, st->part_elems * 8 + 8);
  if (st->qpart) gpx_dev_release(ctx, st->qpart, st->nchunk * st->Mp * 8);
  if (st->buf) gpx_dev_release(ctx, st->buf, st->pack * 8);
  if (st->red) gpx_dev_release(ctx, st->red, 1024 * 8);
  if (st->redi) gpx_dev_release(ctx, st->redi, 1024 * 8);
  delete st;
  return 0;
}

int64_t gpx_givar_pivot_elems(const gpx_givar* st) { return st ? st->pack : 0; }

int gpx_givar_begin(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* Cm, const gpx_mat* Z, double noise, int64_t nsel, gpx_givar** out) {
  GPX_ARG(ctx && L && X && Cm && Z && out, "NULL argument");
  GPX_ARG(L->factored && L->aux, "matrix has not been factored by gpx_potrf");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_TRY(check_points(X, d));
  GPX_TRY(check_points(Cm, d));
  GPX_TRY(check_points(Z, d));
  const int64_t n = L->rows, np = L->prows, M = Cm->rows, nmc = Z->rows;
  GPX_ARG(X->rows == n, "X does not match the factor");
  GPX_ARG(M > 0 && nmc > 0 && nsel >= 1, "need candidates, integration points and at least one pick");
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X, Z, Cm));
  const int64_t zp = gpx_round_up(nmc, GPX_TILE), Mp = gpx_round_up(M, GPX_TILE);
  gpx_givar* st = new gpx_givar();
  st->kp = kp; st->d = d; st->Cm = Cm; st->n = n; st->np = np; st->M = M; st->Mp = Mp; st->nmc = nmc; st->zp = zp;
  st->nsel = nsel; st->cur = 0; st->noise = noise; st->s0 = 0.0;
  st->chunk = 128; st->nchunk = (zp + st->chunk - 1) / st->chunk;
  st->pack = 2 + d + zp + np + nsel;
  st->Wc = st->G = st->U = st->v = st->q = st->cost = st->hdot = st->part = st->qpart = st->buf = st->red = nullptr;
  st->redi = nullptr;
  {
    const int64_t a = colreduce_partial_elems(np, Mp), b = colreduce_partial_elems(zp, Mp), c = colreduce_partial_elems(np, zp);
    st->part_elems = a > b ? (a > c ? a : c) : (b > c ? b : c);
  }
  int r = 0;
  void* p;
  void *pWz = nullptr, *pWzt = nullptr, *pss = nullptr, *pkd = nullptr;
  const int64_t wide = zp > Mp ? zp : Mp;
  do {
    if ((r = gpx_dev_alloc(ctx, np * Mp * 8, &p)) != 0) break; st->Wc = (double*)p;
    if ((r = gpx_dev_alloc(ctx, zp * Mp * 8, &p)) != 0) break; st->G = (double*)p;
    if ((r = gpx_dev_alloc(ctx, nsel * Mp * 8, &p)) != 0) break; st->U = (double*)p;
    if ((r = gpx_dev_alloc(ctx, Mp * 8, &p)) != 0) break; st->v = (double*)p;
    if ((r = gpx_dev_alloc(ctx, Mp * 8, &p)) != 0) break; st->q = (double*)p;
    if ((r = gpx_dev_alloc(ctx, Mp * 8, &p)) != 0) break; st->cost = (double*)p;
    if ((r = gpx_dev_alloc(ctx, Mp * 8, &p)) != 0) break; st->hdot = (double*)p;
    if ((r = gpx_dev_alloc(ctx, st->part_elems * 8 + 8, &p)) != 0) break; st->part = (double*)p;
    if ((r = gpx_dev_alloc(ctx, st->nchunk * Mp * 8, &p)) != 0) break; st->qpart = (double*)p;
    if ((r = gpx_dev_alloc(ctx, st->pack * 8, &p)) != 0) break; st->buf = (double*)p;
    if ((r = gpx_dev_alloc(ctx, 1024 * 8, &p)) != 0) break; st->red = (double*)p;
    if ((r = gpx_dev_alloc(ctx, 1024 * 8, &p)) != 0) break; st->redi = (int64_t*)p;
    if ((r = gpx_dev_alloc(ctx, np * zp * 8, &pWz)) != 0) break;
    if ((r = gpx_dev_alloc(ctx, np * zp * 8, &pWzt)) != 0) break;
    if ((r = gpx_dev_alloc(ctx, wide * 8, &pss)) != 0) break;
    if ((r = gpx_dev_alloc(ctx, wide * 8, &pkd)) != 0) break;
    // ---- integration points: W_Z = L^-1 K(X, Z), var_z, S0 -- the launches of gpx_greedy_ivar_step, in its order ----
    if ((r = launch_kfill(ctx, kp, X->p, n, Z->p, nmc, 0, nullptr, 0, 0.0, (double*)pWz, np, zp, zp)) != 0) break;
    if ((r = chol_trsm_left(ctx, L->p, L->ld, L->aux, (double*)pWz, zp, np, zp)) != 0) break;
    if ((r = launch_colreduce(ctx, (double*)pWz, zp, n, zp, nullptr, (double*)pss, st->part)) != 0) break;
    if ((r = launch_kdiag(ctx, kp, Z->p, nmc, (double*)pkd)) != 0) break;
    std::vector<double> hs((size_t)nmc), hk((size_t)nmc);
    if (hipMemcpyAsync(hs.data(), pss, (size_t)nmc * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { r = -2; break; }
    if (hipMemcpyAsync(hk.data(), pkd, (size_t)nmc * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { r = -2; break; }
    if ((r = launch_transpose(ctx, (double*)pWz, np, zp, zp, (double*)pWzt, np)) != 0) break;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { r = -2; break; }
    for (int64_t j = 0; j < nmc; ++j) hk[(size_t)j] -= hs[(size_t)j];
    st->s0 = pairwise_sum(hk, nmc);
    // ---- candidates: W_C, v, G = K(Z, C) - W_Z^T W_C, q ----
    if ((r = launch_kfill(ctx, kp, X->p, n, Cm->p, M, 0, nullptr, 0, 0.0, st->Wc, np, Mp, Mp)) != 0) break;
    if ((r = chol_trsm_left(ctx, L->p, L->ld, L->aux, st->Wc, Mp, np, Mp)) != 0) break;
    if ((r = launch_colreduce(ctx, st->Wc, Mp, n, Mp, nullptr, (double*)pss, st->part)) != 0) break;
    if ((r = launch_kdiag(ctx, kp, Cm->p, M, (double*)pkd)) != 0) break;
    hipLaunchKernelGGL(givar_v_kernel, dim3((unsigned)(Mp / 256 + 1)), dim3(256), 0, ctx->stream, (const double*)pkd,
                       (const double*)pss, M, Mp, st->v);
    if ((r = launch_kfill(ctx, kp, Z->p, nmc, Cm->p, M, 0, nullptr, 0, 0.0, st->G, zp, Mp, Mp)) != 0) break;
    if ((r = launch_gemm(ctx, (double*)pWzt, np, st->Wc, Mp, st->G, Mp, zp, Mp, np, false, true, false)) != 0) break;
    if ((r = launch_colreduce(ctx, st->G, Mp, nmc, Mp, nullptr, st->q, st->part)) != 0) break;
    if (hipMemsetAsync(st->U, 0, (size_t)(nsel * Mp * 8), ctx->stream) != hipSuccess) { r = -2; break; }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { r = -2; break; }
  } while (0);
  if (pWz) gpx_dev_release(ctx, pWz, np * zp * 8);
  if (pWzt) gpx_dev_release(ctx, pWzt, np * zp * 8);
  if (pss) gpx_dev_release(ctx, pss, wide * 8);
  if (pkd) gpx_dev_release(ctx, pkd, wide * 8);
  if (r != 0) {
    gpx_givar_end(ctx, st);
    if (r == -2) gpx_set_error("givar_begin: HIP call failed");
    return r;
  }
  *out = st;
  return 0;
}

// costs of this rank's candidates given the picks applied so far; the local first minimum (an empty slice cannot exist: M > 0);
// all_costs (host, M doubles) optional.  Blocking.
int gpx_givar_score(gpx_ctx* ctx, gpx_givar* st, double* best_cost, int64_t* best_idx, double* all_costs) {
  GPX_ARG(ctx && st && best_cost && best_idx, "NULL argument");
  const dim3 gM((unsigned)((st->M + 255) / 256));
  hipLaunchKernelGGL(givar_cost_kernel, gM, dim3(256), 0, ctx->stream, (const double*)st->q, (const double*)st->v, st->noise,
                     st->s0, 1.0 / (double)st->nmc, st->M, st->cost);
  const int nb = (int)(gM.x < 1024 ? gM.x : 1024);
  hipLaunchKernelGGL(argmin_stage1, dim3(nb), dim3(256), 0, ctx->stream, (const double*)st->cost, st->M, st->red, st->redi);
  hipLaunchKernelGGL(argmin_stage2, dim3(1), dim3(256), 0, ctx->stream, (const double*)st->red, (const int64_t*)st->redi, nb,
                     st->red + 1023, st->redi + 1023);
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipMemcpyAsync(best_cost, st->red + 1023, 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipMemcpyAsync(best_idx, st->redi + 1023, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (all_costs) GPX_HIP(hipMemcpyAsync(all_costs, st->cost, (size_t)st->M * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  // no finite cost on this rank's slice: the caller's merge sees (+inf, M) -- a rank with nothing to offer, as an empty slice
  if (*best_idx < 0 || *best_idx >= st->M || !(*best_cost < INFINITY)) {
    *best_idx = st->M;
    *best_cost = INFINITY;
  }
  return 0;
}

// owner of the winning candidate (local index s): its pivot pack into buf (gpx_givar_pivot_elems doubles), for the caller to
// broadcast.  Asynchronous.
int gpx_givar_pack(gpx_ctx* ctx, gpx_givar* st, int64_t s, gpx_mat* buf) {
  GPX_ARG(ctx && st && buf && s >= 0 && s < st->M, "bad arguments");
  GPX_ARG(ctx->live_mats.count(st->Cm), "givar: the candidate point set the state was begun with has been freed");   // 
  GPX_ARG(buf->bytes >= st->pack * 8, "pivot buffer too small");
  GPX_ARG(st->cur < st->nsel, "all picks applied");
  int64_t w = st->zp > st->np ? st->zp : st->np;
  if (st->nsel > w) w = st->nsel;
  if (st->d > w) w = st->d;
  hipLaunchKernelGGL(givar_pack_kernel, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)st->G,
                     (const double*)st->Wc, (const double*)st->U, (const double*)st->v, (const double*)st->Cm->p, st->d, st->Mp,
                     st->zp, st->np, st->nsel, (int)st->cur, s, st->noise, buf->p);
  hipLaunchKernelGGL(givar_rr_kernel, dim3(1), dim3(256), 0, ctx->stream, buf->p, st->d, st->zp);
  GPX_HIP(hipGetLastError());
  return 0;
}

// every rank: condition its candidates on the pick whose pack is in buf.  Blocking (S0 lives on the host).
int gpx_givar_apply(gpx_ctx* ctx, gpx_givar* st, const gpx_mat* buf) {
  GPX_ARG(ctx && st && buf && buf->bytes >= st->pack * 8, "bad arguments");
  GPX_ARG(ctx->live_mats.count(st->Cm), "givar: the candidate point set the state was begun with has been freed");
  GPX_ARG(st->cur < st->nsel, "all picks applied");
  const double* b = buf->p;
  const double* r = b + 2 + st->d;
  const double* w = r + st->zp;
  GPX_TRY(launch_colreduce(ctx, st->Wc, st->Mp, st->np, st->Mp, w, st->hdot, st->part));
  const dim3 gMp((unsigned)((st->Mp + 255) / 256));
  hipLaunchKernelGGL(givar_u_kernel, gMp, dim3(256), 0, ctx->stream, st->kp, (const double*)st->Cm->p, st->M, st->Mp, b, st->zp,
                     st->np, (const double*)st->hdot, st->U, (int)st->cur, st->v);
  hipLaunchKernelGGL(givar_rank1_kernel, dim3(gMp.x, (unsigned)st->nchunk), dim3(256), 0, ctx->stream, st->G, st->Mp, st->zp,
                     st->chunk, r, (const double*)(st->U + st->cur * st->Mp), st->qpart);
  hipLaunchKernelGGL(givar_q_kernel, gMp, dim3(256), 0, ctx->stream, (const double*)st->qpart, st->nchunk, st->Mp, st->q);
  GPX_HIP(hipGetLastError());
  double rr = 0.0;
  GPX_HIP(hipMemcpyAsync(&rr, b + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  st->s0 -= rr;
  st->cur += 1;
  return 0;
}

// nsel picks on one GPU: out_idx[nsel], out_cost[nsel] (the winner's cost at each pick), all_costs (optional, nsel x M)
int gpx_greedy_ivar(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* Cm, const gpx_mat* Z, double noise, int64_t nsel, int64_t* out_idx, double* out_cost,
                    double* all_costs) {
  GPX_ARG(out_idx != nullptr, "NULL argument");
  gpx_givar* st = nullptr;
  GPX_TRY(gpx_givar_begin(ctx, kind, d, hyp, nhyp, L, X, Cm, Z, noise, nsel, &st));
  gpx_mat* buf = nullptr;
  int r = gpx_mat_alloc(ctx, st->pack, 1, 0, &buf);
  for (int64_t t = 0; r == 0 && t < nsel; ++t) {
    double c = 0.0;
    int64_t s = 0;
    if ((r = gpx_givar_score(ctx, st, &c, &s, all_costs ? all_costs + t * st->M : nullptr)) != 0) break;
    if (s >= st->M) {
      gpx_set_error("greedy IVAR: pick %lld of %lld: no candidate has a finite cost (every remaining one is already in the design "
                    "with zero noise, or the state is not finite)", (long long)(t + 1), (long long)nsel);
      r = -1;
      break;
    }
    out_idx[t] = s;
    if (out_cost) out_cost[t] = c;
    if (t + 1 == nsel) break;
    if ((r = gpx_givar_pack(ctx, st, s, buf)) != 0) break;
    r = gpx_givar_apply(ctx, st, buf);
  }
  if (buf) gpx_mat_free(ctx, buf);
  gpx_givar_end(ctx, st);
  return r;
}

}  // extern "C"
