// Hyper-parameter gradient of the log marginal likelihood and the greedy mutual-information design -- gfx950.
//
//  gpx_lml_grad   replaces the reference's N^2 Python calls to derivativeWrtHypParams and the dense
//                 (N,N,d+2) derivMat (gp.py:444-466): with P = K^-1 and T = alpha alpha^T - P,
//                   g_key = 1/2 tr(T dK_key) = 1/2 sum_ij T_ij dK_key,ij
//                 one fused pass over P recomputes K_ij and the per-dimension squared distances from the point
//                 coordinates (staged in LDS) and reduces all d+2 traces at once: 8 N^2 bytes read, no dK stored.
//  gpx_mi_greedy  replaces two pinv's per candidate per step (experimentalDesign.py:259-285, 775-783):
//                 numerator var(c|A) carried as an incremental Cholesky row (pivot + noise), denominator
//                 var(c | S\c) = 1/[(K_SS + noise I)^-1]_cc - noise with S = all \ A, the inverse being
//                 down-dated by a rank-one kernel whenever a point moves from S to A.
#include "gpx_internal.h"
#include <math.h>

namespace {

constexpr int TS = 64;

// ---- lml_grad ------------------------------------------------------------------------------------------
// partial[block][q], q = 0..d-1: sum T_ij K0_ij e_k^2 (scaled differences), q = d: sum T_ij K0_ij, q = d+1: sum_i T_ii
__global__ __launch_bounds__(256) void lmlgrad_kernel(KParams kp, const double* __restrict__ X, int64_t n,
                                                      const double* __restrict__ P, int64_t ld,
                                                      const double* __restrict__ alpha,
                                                      double* __restrict__ partial) {
  extern __shared__ double sm[];
  const int d = kp.d;
  double* As = sm;               // [TS][d] raw coords of the row points (differences are taken first, then scaled:
  double* Bs = sm + TS * d;      // [TS][d] of the column points          kernels.py:121-122)
  double* red = Bs + TS * d;     // [4] per-wave partials
  const int t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * TS, j0 = (int64_t)blockIdx.x * TS;
  // T and dK are symmetric and only the lower triangle of K^-1 is valid: tiles above the diagonal contribute nothing,
  // tiles below it count twice, diagonal tiles read the mirrored entry
  const bool upper = blockIdx.x > blockIdx.y;
  const double weight = blockIdx.x == blockIdx.y ? 1.0 : 2.0;
  for (int idx = t; idx < TS * d; idx += 256) {
    int p = idx / d, k = idx - p * d;
    int64_t gi = i0 + p, gj = j0 + p;
    As[idx] = gi < n ? X[gi * d + k] : 0.0;
    Bs[idx] = gj < n ? X[gj * d + k] : 0.0;
  }
  __syncthreads();
  const int tx = t & 31, ty = t >> 5;
  double tk[16];  // T_ij * K0_ij of this thread's 8 rows x 2 columns
  double diag = 0.0;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int r = ty + 8 * a;
    const int64_t gi = i0 + r;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int cc = 2 * tx + c;
      const int64_t gj = j0 + cc;
      double v = 0.0;
      if (!upper && gi < n && gj < n) {
        double acc = 0.0;
        for (int k = 0; k < d; ++k) {
          const double e = (As[r * d + k] - Bs[cc * d + k]) * kp.scale[k];
          acc = fma(e, e, acc);
        }
        const double tij = alpha[gi] * alpha[gj] - (gi >= gj ? P[gi * ld + gj] : P[gj * ld + gi]);
        v = weight * tij * kp.sig * exp(-0.5 * acc);
        if (gi == gj) diag += tij;
      }
      tk[a * 2 + c] = v;
    }
  }
  const int lane = t & 63, wave = t >> 6;
  for (int q = 0; q <= d + 1; ++q) {
    double s = 0.0;
    if (q < d) {
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int r = ty + 8 * a;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const double e = (As[r * d + q] - Bs[(2 * tx + c) * d + q]) * kp.scale[q];
          s = fma(tk[a * 2 + c], e * e, s);
        }
      }
    } else if (q == d) {
#pragma unroll
      for (int a = 0; a < 16; ++a) s += tk[a];
    } else {
      s = diag;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (t == 0)
      partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (d + 2) + q] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

__global__ __launch_bounds__(256) void lmlgrad_final_kernel(const double* __restrict__ partial, int64_t nblocks,
                                                            int nq, double* __restrict__ out) {
  __shared__ double red[256];
  const int q = blockIdx.x;
  double s = 0.0;
  for (int64_t b = threadIdx.x; b < nblocks; b += 256) s += partial[b * nq + q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[q] = red[0];
}

// ---- MI greedy -------------------------------------------------------------------------------------------
__device__ __forceinline__ double kpair_se_like(const KParams& kp, const double* __restrict__ a,
                                                const double* __restrict__ b) {
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
  double acc = 0.0;
  for (int k = 0; k < kp.d; ++k) {
    const double e = (a[k] - b[k]) * kp.scale[k];  // difference first, as the reference (kernels.py:121-122)
    acc = fma(e, e, acc);
  }
  if (kp.kind == GPX_K_SE) return kp.sig * exp(-0.5 * acc);
  const double t = sqrt(acc);
  if (kp.kind == GPX_K_MATERN32) return kp.sig * (1.0 + t) * exp(-t);
  return kp.sig * (1.0 + t + acc * (1.0 / 3.0)) * exp(-t);
}

// conditioning on sel[cur] with noisy observations: pivot = d_s + noise
__global__ __launch_bounds__(256) void mi_row_kernel(KParams kp, const double* __restrict__ Cp, int64_t M,
                                                     const int64_t* __restrict__ sel, int cur, double noise,
                                                     double* __restrict__ W, int64_t ldw,
                                                     const double* __restrict__ d_in, double* __restrict__ d_out) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= M) return;
  const int64_t s = sel[cur];
  double dot = 0.0;
  for (int t = 0; t < cur; ++t) dot = fma(W[(int64_t)t * ldw + s], W[(int64_t)t * ldw + c], dot);
  const double w = (kpair_se_like(kp, Cp + s * kp.d, Cp + c * kp.d) - dot) / sqrt(d_in[s] + noise);
  W[(int64_t)cur * ldw + c] = w;
  d_out[c] = fma(-w, w, d_in[c]);
}

// P <- P - p_s p_s^T / P_ss on the alive rows/cols; row/col s are left untouched (dead afterwards)
__global__ __launch_bounds__(256) void mi_downdate_kernel(double* __restrict__ P, int64_t ld, int64_t M,
                                                          const int64_t* __restrict__ sel, int slot) {
  const int64_t s = sel[slot];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j >= M || i == s || j == s) return;
  P[i * ld + j] -= P[i * ld + s] * P[s * ld + j] / P[s * ld + s];
}

__global__ __launch_bounds__(256) void mi_mark_kernel(int* __restrict__ alive, const int64_t* __restrict__ sel,
                                                      int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) alive[sel[slot]] = 0;
}

__global__ __launch_bounds__(256) void mi_ratio_kernel(const double* __restrict__ P, int64_t ld,
                                                       const double* __restrict__ dnum,
                                                       const int* __restrict__ alive, double noise, int64_t M,
                                                       double* __restrict__ ratio) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= M) return;
  ratio[c] = alive[c] ? dnum[c] / (1.0 / P[c * ld + c] - noise) : -INFINITY;
}

struct VI {
  double v;
  int64_t i;
};
__device__ __forceinline__ VI vi_max(VI a, VI b) {
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
// single-block first-max arg-max; also stores the winning value
__global__ __launch_bounds__(1024) void argmax_block_kernel(const double* __restrict__ x, int64_t M,
                                                            int64_t* __restrict__ sel, int slot,
                                                            double* __restrict__ val) {
  __shared__ double sv[1024];
  __shared__ int64_t si[1024];
  VI best{-INFINITY, INT64_MAX};
  for (int64_t c = threadIdx.x; c < M; c += 1024) best = vi_max(best, VI{x[c], c});
  sv[threadIdx.x] = best.v;
  si[threadIdx.x] = best.i;
  __syncthreads();
  for (int h = 512; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) {
      VI m = vi_max(VI{sv[threadIdx.x], si[threadIdx.x]}, VI{sv[threadIdx.x + h], si[threadIdx.x + h]});
      sv[threadIdx.x] = m.v;
      si[threadIdx.x] = m.i;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sel[slot] = si[0] == INT64_MAX ? 0 : si[0];
    if (val) val[slot] = sv[0];
  }
}

__global__ __launch_bounds__(256) void fill_int_kernel(int* __restrict__ a, int64_t n, int v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = v;
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
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& b : bufs) gpx_dev_release(ctx, b.first, b.second);
  }
};

}  // namespace

extern "C" {

int gpx_lml_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                 const double* alpha, double* grad) {
  GPX_ARG(ctx && L && X && alpha && grad, "NULL argument");
  GPX_ARG(L->factored && L->aux, "matrix has not been factored by gpx_potrf");
  GPX_ARG(kind == GPX_K_SE, "lml_grad: only the squared-exponential kernel has hyper-parameter derivatives "
                            "(the reference raises for the others, kernels.py:93-97)");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_ARG(X->cols == d && X->pcols == d && X->rows == L->rows, "X does not match the factor");
  const int64_t n = L->rows;
  gpx_mat* P = nullptr;
  GPX_TRY(gpx_potri(ctx, L, &P));
  int r = 0;
  {
    Scratch sc(ctx);
    void *pal, *ppart, *pout;
    const int64_t tiles = gpx_round_up(n, TS) / TS;
    const int nq = d + 2;
    do {
      if ((r = sc.get(n * 8, &pal)) != 0) break;
      if ((r = sc.get(tiles * tiles * nq * 8, &ppart)) != 0) break;
      if ((r = sc.get(nq * 8, &pout)) != 0) break;
      if (hipMemcpyAsync(pal, alpha, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { r = -2; break; }
      {
        ProfScope ps(ctx, GPX_PROF_REDUCE, 0.0, 8.0 * (double)n * n);
        dim3 grid((unsigned)tiles, (unsigned)tiles);
        size_t sh = (size_t)(2 * TS * d + 4) * sizeof(double);
        hipLaunchKernelGGL(lmlgrad_kernel, grid, dim3(256), sh, ctx->stream, kp, X->p, n, P->p, P->ld,
                           (const double*)pal, (double*)ppart);
        hipLaunchKernelGGL(lmlgrad_final_kernel, dim3(nq), dim3(256), 0, ctx->stream, (const double*)ppart,
                           tiles * tiles, nq, (double*)pout);
      }
      std::vector<double> h((size_t)nq);
      if (hipMemcpyAsync(h.data(), pout, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
          hipStreamSynchronize(ctx->stream) != hipSuccess) { r = -2; break; }
      // scaled difference e_k = D_k / cl_k  ->  dK/d cl_k = K0 D_k^2 / cl_k^3 = K0 e_k^2 / cl_k
      for (int k = 0; k < d; ++k) grad[k] = 0.5 * h[(size_t)k] / hyp[k];
      grad[d] = 0.5 * h[(size_t)d] / hyp[d];   // dK/d signalSize = K0 / signalSize
      grad[d + 1] = 0.5 * h[(size_t)d + 1];    // dK/d noise = I (caller scales by 2*noise, gp.py:463-464)
    } while (0);
  }
  gpx_mat_free(ctx, P);
  if (r == -2) gpx_set_error("lml_grad: HIP call failed: %s", hipGetErrorString(hipGetLastError()));
  return r;
}

int gpx_mi_greedy(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* Cm, double noise,
                  int64_t nsel, int64_t start, int64_t* out_idx, double* out_ratio) {
  GPX_ARG(ctx && Cm && out_idx, "NULL argument");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_ARG(Cm->cols == d && Cm->pcols == d, "candidates must be an unpadded (M x d) point set");
  const int64_t M = Cm->rows;
  GPX_ARG(nsel >= 1 && nsel <= M && start >= 0 && start < M, "bad nsel/start");
  GPX_ARG(M <= 65535, "mi_greedy supports at most 65535 candidates");
  // Sigma = K(C,C) + noise I, full inverse
  gpx_mat* S = nullptr;
  GPX_TRY(gpx_kfill(ctx, kind, d, hyp, nhyp, Cm, nullptr, &noise, 1, &S));
  int r = gpx_potrf(ctx, S);
  gpx_mat* P = nullptr;
  if (r == 0) r = gpx_potri_impl(ctx, S, &P, 1);  // the down-dates touch both triangles
  gpx_mat_free(ctx, S);
  if (r != 0) return r;
  {
    Scratch sc(ctx);
    void *pW, *pd0, *pd1, *psel, *palive, *pratio, *pval;
    do {
      if ((r = sc.get(nsel * M * 8, &pW)) != 0) break;
      if ((r = sc.get(M * 8, &pd0)) != 0) break;
      if ((r = sc.get(M * 8, &pd1)) != 0) break;
      if ((r = sc.get(nsel * 8, &psel)) != 0) break;
      if ((r = sc.get(nsel * 8, &pval)) != 0) break;
      if ((r = sc.get(M * 4, &palive)) != 0) break;
      if ((r = sc.get(M * 8, &pratio)) != 0) break;
      const dim3 gM((unsigned)((M + 255) / 256));
      if (hipMemcpyAsync(psel, &start, 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { r = -2; break; }
      hipLaunchKernelGGL(fill_int_kernel, gM, dim3(256), 0, ctx->stream, (int*)palive, M, 1);
      if ((r = launch_kdiag(ctx, kp, Cm->p, M, (double*)pd0)) != 0) break;
      double* din = (double*)pd0;
      double* dout = (double*)pd1;
      ProfScope ps(ctx, GPX_PROF_GREEDY, 2.0 * (double)M * M * nsel, 16.0 * (double)M * M * nsel);
      for (int64_t cur = 0; cur + 1 < nsel; ++cur) {
        // move sel[cur] from S to A
        hipLaunchKernelGGL(mi_row_kernel, gM, dim3(256), 0, ctx->stream, kp, Cm->p, M, (const int64_t*)psel, (int)cur,
                           noise, (double*)pW, M, (const double*)din, dout);
        { double* t = din; din = dout; dout = t; }
        hipLaunchKernelGGL(mi_downdate_kernel, dim3(gM.x, (unsigned)M), dim3(256), 0, ctx->stream, P->p, P->ld, M,
                           (const int64_t*)psel, (int)cur);
        hipLaunchKernelGGL(mi_mark_kernel, dim3(1), dim3(64), 0, ctx->stream, (int*)palive, (const int64_t*)psel,
                           (int)cur);
        hipLaunchKernelGGL(mi_ratio_kernel, gM, dim3(256), 0, ctx->stream, (const double*)P->p, P->ld,
                           (const double*)din, (const int*)palive, noise, M, (double*)pratio);
        hipLaunchKernelGGL(argmax_block_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)pratio, M,
                           (int64_t*)psel, (int)(cur + 1), (double*)pval);
      }
      if (hipGetLastError() != hipSuccess) { r = -2; break; }
      if (hipMemcpyAsync(out_idx, psel, (size_t)nsel * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { r = -2; break; }
      if (out_ratio && nsel > 1 &&
          hipMemcpyAsync(out_ratio, (double*)pval + 1, (size_t)(nsel - 1) * 8, hipMemcpyDeviceToHost, ctx->stream) !=
              hipSuccess) { r = -2; break; }
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) { r = -2; break; }
    } while (0);
  }
  gpx_mat_free(ctx, P);
  if (r == -2) gpx_set_error("mi_greedy: HIP call failed: %s", hipGetErrorString(hipGetLastError()));
  return r;
}

}  // extern "C"
