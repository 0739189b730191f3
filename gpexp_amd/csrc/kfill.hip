// kfill: dense covariance assembly  K[i][j] = k(A_i, B_j) (+ nugget on the diagonal)  -- gfx950.
//
// Replaces the reference's N-iteration Python row loop (gp_kernel_utilities.py:56-60, one
// Kernel.evaluate + np.tile per row) and the column loops that build kernelvals
// (gp.py:132-135, 246-249; experimentalDesign.py:829-831).
//
// HBM-bound by intent: algorithmic bytes = 8*rows*cols written + 8*(rows+cols)*d read.  What it takes to get
// there in fp64 (no hardware exp/sqrt):
//   * 64x64 tiles; point coordinates staged once per tile into LDS pre-multiplied by the length scale; each thread
//     keeps its two B points in registers, A points are LDS broadcasts; 16-byte stores forming 512-byte row segments.
//   * symmetric fill: only tiles on/below the diagonal are COMPUTED; each is also written transposed through a
//     padded LDS image (coalesced mirror stores) -- half the exp/sqrt work for the same 8*N^2 bytes.
//   * exp: one rndne + two-constant Cody-Waite reduction + degree-13 polynomial + v_ldexp (about 20 fp64
//     instructions, no special-case branches: arguments here are finite and <= ~1); sqrt: v_rsq_f64 + two coupled
//     Newton steps + residual correction.  Both stay within 2 ulp (tests: 1e-13 against the oracle / the reference).
#include "gpx_internal.h"
#include <math.h>

namespace {

constexpr int TM = 64;  // tile rows
constexpr int TN = 64;  // tile cols
constexpr int TP = TN + 1;  // padded stride of the transpose image

__device__ __forceinline__ double fast_exp(double x) {
  // exp(x) = 2^n * exp(r), n = rint(x/ln2), |r| <= ln2/2; Taylor to r^13 (truncation 4e-18 relative)
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821614599e-10;             // 1/13!
  p = fma(p, r, 2.0876756987868098979e-09);          // 1/12!
  p = fma(p, r, 2.5052108385441718775e-08);          // 1/11!
  p = fma(p, r, 2.7557319223985890653e-07);          // 1/10!
  p = fma(p, r, 2.7557319223985892511e-06);          // 1/9!
  p = fma(p, r, 2.4801587301587301566e-05);          // 1/8!
  p = fma(p, r, 1.9841269841269841253e-04);          // 1/7!
  p = fma(p, r, 1.3888888888888889419e-03);          // 1/6!
  p = fma(p, r, 8.3333333333333332177e-03);          // 1/5!
  p = fma(p, r, 4.1666666666666664354e-02);          // 1/4!
  p = fma(p, r, 1.6666666666666665741e-01);          // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

__device__ __forceinline__ double fast_sqrt(double x) {  // x >= 0, not subnormal (squared scaled distances)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  return x > 0.0 ? g : 0.0;
}

template <int KIND>
__device__ __forceinline__ double kvalue(double acc, double sig) {
  if (KIND == GPX_K_SE) {
    return sig * fast_exp(-0.5 * acc);
  } else if (KIND == GPX_K_MATERN32) {
    const double t = fast_sqrt(acc);
    return sig * (1.0 + t) * fast_exp(-t);
  } else if (KIND == GPX_K_MATERN52) {
    const double t = fast_sqrt(acc);
    return sig * (1.0 + t + acc * (1.0 / 3.0)) * fast_exp(-t);
  } else {  // Mehler: acc = pa + pb - cross
    return sig * fast_exp(-acc);
  }
}

// stage one tile of points into LDS: P[p][k] = X[g0+p][k] * s_k  (+ Mehler norm in slot d)
template <int KIND, bool SIDE_A>
__device__ __forceinline__ void stage_points(const KParams& kp, int d, int dp, const double* __restrict__ X,
                                             int64_t n, int64_t g0, double* P) {
  const int t = threadIdx.x;
  for (int idx = t; idx < TM * d; idx += 256) {
    int p = idx / d, k = idx - p * d;
    int64_t g = g0 + p;
    double v = 0.0;
    if (g < n) {
      v = X[g * d + k];
      if (KIND == GPX_K_MEHLER) {
        if (SIDE_A) v *= kp.c2[k];
      } else {
        v *= kp.scale[k];
      }
    }
    P[p * dp + k] = v;
  }
  if (KIND == GPX_K_MEHLER) {
    if (t < TM) {
      int64_t g = g0 + t;
      double s = 0.0;
      if (g < n)
        for (int k = 0; k < d; ++k) {
          double v = X[g * d + k];
          s = fma(kp.c1[k] * v, v, s);
        }
      P[t * dp + d] = s;
    }
  }
}

// SYM: 1-D grid over the tiles on/below the diagonal, mirror-written; otherwise 2-D grid over all tiles.
template <int KIND, int DT, bool SYM>
__global__ __launch_bounds__(256) void kfill_kernel(KParams kp, const double* __restrict__ A, int64_t na,
                                                    const double* __restrict__ B, int64_t nb, int symmetric,
                                                    const double* __restrict__ nugget, int64_t nugget_len,
                                                    double nugget_scalar, double* __restrict__ out, int64_t ld) {
  extern __shared__ double sm[];
  const int d = DT > 0 ? DT : kp.d;
  const int dp = d + (KIND == GPX_K_MEHLER ? 1 : 0);
  double* As = sm;
  double* Bs = sm + TM * dp;
  double* Tr = Bs + TN * dp;  // [32][TP] transpose image (SYM only)
  int ti, tj;
  if (SYM) {
    const int w = blockIdx.x;
    ti = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= w) ++ti;
    while (ti * (ti + 1) / 2 > w) --ti;
    tj = w - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.y;
    tj = blockIdx.x;
  }
  const int64_t i0 = (int64_t)ti * TM, j0 = (int64_t)tj * TN;
  stage_points<KIND, true>(kp, d, dp, A, na, i0, As);
  stage_points<KIND, false>(kp, d, dp, B, nb, j0, Bs);
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = 2 * tx;
  const double sig = kp.sig;
  const int64_t gj0 = j0 + c0, gj1 = gj0 + 1;

  constexpr int DD = DT > 0 ? DT : 1;
  double b0[DD], b1[DD];
  double pb0 = 0.0, pb1 = 0.0;
  if (DT > 0) {
#pragma unroll
    for (int k = 0; k < DD; ++k) {
      b0[k] = Bs[c0 * dp + k];
      b1[k] = Bs[(c0 + 1) * dp + k];
    }
  }
  if (KIND == GPX_K_MEHLER) {
    pb0 = Bs[c0 * dp + d];
    pb1 = Bs[(c0 + 1) * dp + d];
  }

  double2 val[TM / 8];  // this thread's 8 rows x 2 columns
#pragma unroll
  for (int a = 0; a < TM / 8; ++a) {
    const int r = ty + 8 * a;
    double s0 = 0.0, s1 = 0.0;
    if (DT > 0) {
#pragma unroll
      for (int k = 0; k < DD; ++k) {
        const double av = As[r * dp + k];
        if (KIND == GPX_K_MEHLER) {
          s0 = fma(av, b0[k], s0);
          s1 = fma(av, b1[k], s1);
        } else {
          const double e0 = av - b0[k], e1 = av - b1[k];
          s0 = fma(e0, e0, s0);
          s1 = fma(e1, e1, s1);
        }
      }
    } else {
      for (int k = 0; k < d; ++k) {
        const double av = As[r * dp + k];
        const double bv0 = Bs[c0 * dp + k], bv1 = Bs[(c0 + 1) * dp + k];
        if (KIND == GPX_K_MEHLER) {
          s0 = fma(av, bv0, s0);
          s1 = fma(av, bv1, s1);
        } else {
          const double e0 = av - bv0, e1 = av - bv1;
          s0 = fma(e0, e0, s0);
          s1 = fma(e1, e1, s1);
        }
      }
    }
    if (KIND == GPX_K_MEHLER) {
      const double pa = As[r * dp + d];
      s0 = pa + pb0 - s0;
      s1 = pa + pb1 - s1;
    }
    const int64_t gi = i0 + r;
    double v0 = kvalue<KIND>(s0, sig), v1 = kvalue<KIND>(s1, sig);
    const bool rin = gi < na;
    if (!(rin && gj0 < nb)) v0 = (symmetric && gi == gj0) ? 1.0 : 0.0;
    if (!(rin && gj1 < nb)) v1 = (symmetric && gi == gj1) ? 1.0 : 0.0;
    if (symmetric && rin && nugget_len > 0) {
      const double nz = nugget_len == 1 ? nugget_scalar : nugget[gi];
      if (gi == gj0) v0 += nz;
      if (gi == gj1) v1 += nz;
    }
    val[a].x = v0;
    val[a].y = v1;
    *reinterpret_cast<double2*>(out + gi * ld + gj0) = val[a];
  }

  if (SYM && ti != tj) {
    // mirror: out[j0 + c][i0 + r] = V[r][c], two passes of 32 tile rows through the padded image
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 4; ++a) {  // rows r = ty + 8*(4*half + a)  ->  image row ty + 8a
        Tr[(ty + 8 * a) * TP + c0] = val[4 * half + a].x;
        Tr[(ty + 8 * a) * TP + c0 + 1] = val[4 * half + a].y;
      }
      __syncthreads();
      // thread (tx, ty): transposed rows c = ty + 8q (q = 0..7), columns r = 32*half + tx  -> 32 lanes x 8 B = 256 B
      // segments; two 32-lane groups of a wave cover two different rows
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = ty + 8 * q;
        out[(j0 + c) * ld + i0 + 32 * half + tx] = Tr[tx * TP + c];
      }
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void kdiag_kernel(KParams kp, const double* __restrict__ Z, int64_t m,
                                                    double* __restrict__ out) {
  int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double acc = 0.0;
  if (KIND == GPX_K_MEHLER) {
    // pa + pb - cross with a == b:  sum_k (2 c1_k - c2_k) z_k^2
    for (int k = 0; k < kp.d; ++k) {
      double z = Z[j * kp.d + k];
      acc += 2.0 * (kp.c1[k] * z) * z - (kp.c2[k] * z) * z;
    }
  }
  out[j] = kvalue<KIND>(acc, kp.sig);
}

template <int KIND, bool SYM>
int launch_kind(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb,
                int symmetric, const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out,
                int64_t prows, int64_t pcols, int64_t ld) {
  dim3 grid;
  if (SYM) {
    const int64_t t = prows / TM;
    grid = dim3((unsigned)(t * (t + 1) / 2));
  } else {
    grid = dim3((unsigned)(pcols / TN), (unsigned)(prows / TM));
  }
  const int dp = kp.d + (KIND == GPX_K_MEHLER ? 1 : 0);
  size_t sh = (size_t)(2 * TM * dp + (SYM ? 32 * TP : 0)) * sizeof(double);
#define GPX_KF(DT)                                                                                             \
  hipLaunchKernelGGL((kfill_kernel<KIND, DT, SYM>), grid, dim3(256), sh, ctx->stream, kp, A, na, B, nb, symmetric, \
                     d_nugget, nugget_len, nugget_scalar, out, ld)
  switch (kp.d) {
    case 1: GPX_KF(1); break;
    case 2: GPX_KF(2); break;
    case 3: GPX_KF(3); break;
    case 4: GPX_KF(4); break;
    case 8: GPX_KF(8); break;
    case 10: GPX_KF(10); break;
    default: GPX_KF(0); break;
  }
#undef GPX_KF
  GPX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int launch_kfill(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb,
                 int symmetric, const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out,
                 int64_t prows, int64_t pcols, int64_t ld) {
  GPX_ARG(prows % TM == 0 && pcols % TN == 0, "kfill: padded shape must be a multiple of 64");
  GPX_ARG(prows / TM <= 65535, "kfill: too many row tiles");
  ProfScope ps(ctx, GPX_PROF_KFILL, 0.0, 8.0 * (double)prows * (double)pcols + 8.0 * (double)(na + nb) * kp.d);
  // the mirrored (half-compute) path needs a square output whose row and column point sets coincide
  const bool sym = symmetric && prows == pcols && A == B && na == nb;
#define GPX_KIND(K_)                                                                                            \
  (sym ? launch_kind<K_, true>(ctx, kp, A, na, B, nb, 1, d_nugget, nugget_len, nugget_scalar, out, prows, pcols, ld) \
       : launch_kind<K_, false>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar, out, prows,  \
                                pcols, ld))
  switch (kp.kind) {
    case GPX_K_SE: return GPX_KIND(GPX_K_SE);
    case GPX_K_MATERN32: return GPX_KIND(GPX_K_MATERN32);
    case GPX_K_MATERN52: return GPX_KIND(GPX_K_MATERN52);
    case GPX_K_MEHLER: return GPX_KIND(GPX_K_MEHLER);
  }
#undef GPX_KIND
  gpx_set_error("kfill: unknown kernel kind %d", kp.kind);
  return -1;
}

int launch_kfill_offset(gpx_ctx* ctx, const KParams& kp, const double* X, int64_t n, int64_t row_off, int64_t col_off,
                        const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out, int64_t prows,
                        int64_t pcols, int64_t ld) {
  GPX_ARG(row_off == col_off, "kfill_offset: the sub-block must start on the diagonal");
  int64_t na = n - row_off;
  if (na < 0) na = 0;
  int64_t nbp = n - col_off;
  if (nbp < 0) nbp = 0;
  if (nbp > pcols) nbp = pcols;
  const double* nug = (d_nugget && nugget_len > 1) ? d_nugget + row_off : d_nugget;
  return launch_kfill(ctx, kp, X + row_off * kp.d, na, X + col_off * kp.d, nbp, 1, nug, nugget_len, nugget_scalar, out,
                      prows, pcols, ld);
}

int launch_kdiag(gpx_ctx* ctx, const KParams& kp, const double* Z, int64_t m, double* out) {
  if (m <= 0) return 0;
  dim3 grid((unsigned)((m + 255) / 256));
  switch (kp.kind) {
    case GPX_K_SE: hipLaunchKernelGGL(kdiag_kernel<GPX_K_SE>, grid, dim3(256), 0, ctx->stream, kp, Z, m, out); break;
    case GPX_K_MATERN32:
      hipLaunchKernelGGL(kdiag_kernel<GPX_K_MATERN32>, grid, dim3(256), 0, ctx->stream, kp, Z, m, out);
      break;
    case GPX_K_MATERN52:
      hipLaunchKernelGGL(kdiag_kernel<GPX_K_MATERN52>, grid, dim3(256), 0, ctx->stream, kp, Z, m, out);
      break;
    case GPX_K_MEHLER:
      hipLaunchKernelGGL(kdiag_kernel<GPX_K_MEHLER>, grid, dim3(256), 0, ctx->stream, kp, Z, m, out);
      break;
    default: gpx_set_error("kdiag: unknown kernel kind %d", kp.kind); return -1;
  }
  GPX_HIP(hipGetLastError());
  return 0;
}
