// kfill: dense covariance assembly  K[i][j] = k(A_i, B_j) (+ nugget on the diagonal)  -- gfx950.
//
// Replaces the reference's N-iteration Python row loop (gp_kernel_utilities.py:56-60, one
// Kernel.evaluate + np.tile per row) and the column loops that build kernelvals
// (gp.py:132-135, 246-249; experimentalDesign.py:829-831).
//
// HBM-bound by construction: every point coordinate is read once per 64x64 tile from L2 into LDS
// (pre-multiplied by the kernel's length-scale), each thread keeps its two B points in registers,
// A points are LDS broadcasts, and the tile leaves as 16-byte stores that form full 512-byte row
// segments per half-wave.  Algorithmic bytes: 8*rows*cols written + 8*(rows+cols)*d read.
#include "gpx_internal.h"
#include <math.h>

namespace {

constexpr int TM = 64;  // tile rows
constexpr int TN = 64;  // tile cols

template <int KIND>
__device__ __forceinline__ double kvalue(double acc, double sig) {
  if (KIND == GPX_K_SE) {
    return sig * exp(-0.5 * acc);
  } else if (KIND == GPX_K_MATERN32) {
    double t = sqrt(acc);
    return sig * (1.0 + t) * exp(-t);
  } else if (KIND == GPX_K_MATERN52) {
    double t = sqrt(acc);
    return sig * (1.0 + t + acc * (1.0 / 3.0)) * exp(-t);
  } else {  // Mehler: acc = pa + pb - cross
    return sig * exp(-acc);
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

template <int KIND, int DT>
__global__ __launch_bounds__(256) void kfill_kernel(KParams kp, const double* __restrict__ A, int64_t na,
                                                    const double* __restrict__ B, int64_t nb, int symmetric,
                                                    const double* __restrict__ nugget, int64_t nugget_len,
                                                    double nugget_scalar, double* __restrict__ out, int64_t ld) {
  extern __shared__ double sm[];
  const int d = DT > 0 ? DT : kp.d;
  const int dp = d + (KIND == GPX_K_MEHLER ? 1 : 0);
  double* As = sm;
  double* Bs = sm + TM * dp;
  const int64_t i0 = (int64_t)blockIdx.y * TM, j0 = (int64_t)blockIdx.x * TN;
  stage_points<KIND, true>(kp, d, dp, A, na, i0, As);
  stage_points<KIND, false>(kp, d, dp, B, nb, j0, Bs);
  __syncthreads();

  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = 2 * tx;
  const double sig = kp.sig;
  const int64_t gj0 = j0 + c0, gj1 = gj0 + 1;

  if (DT > 0) {
    constexpr int DD = DT > 0 ? DT : 1;
    double b0[DD], b1[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) {
      b0[k] = Bs[c0 * dp + k];
      b1[k] = Bs[(c0 + 1) * dp + k];
    }
    double pb0 = 0.0, pb1 = 0.0;
    if (KIND == GPX_K_MEHLER) {
      pb0 = Bs[c0 * dp + d];
      pb1 = Bs[(c0 + 1) * dp + d];
    }
#pragma unroll 2
    for (int a = 0; a < TM / 8; ++a) {
      const int r = ty + 8 * a;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < DD; ++k) {
        double av = As[r * dp + k];
        if (KIND == GPX_K_MEHLER) {
          s0 = fma(av, b0[k], s0);
          s1 = fma(av, b1[k], s1);
        } else {
          double e0 = av - b0[k], e1 = av - b1[k];
          s0 = fma(e0, e0, s0);
          s1 = fma(e1, e1, s1);
        }
      }
      if (KIND == GPX_K_MEHLER) {
        double pa = As[r * dp + d];
        s0 = pa + pb0 - s0;
        s1 = pa + pb1 - s1;
      }
      const int64_t gi = i0 + r;
      double v0 = kvalue<KIND>(s0, sig), v1 = kvalue<KIND>(s1, sig);
      const bool rin = gi < na;
      if (!(rin && gj0 < nb)) v0 = (symmetric && gi == gj0) ? 1.0 : 0.0;
      if (!(rin && gj1 < nb)) v1 = (symmetric && gi == gj1) ? 1.0 : 0.0;
      if (symmetric && rin && nugget_len > 0) {
        double nz = nugget_len == 1 ? nugget_scalar : nugget[gi];
        if (gi == gj0) v0 += nz;
        if (gi == gj1) v1 += nz;
      }
      double2 w;
      w.x = v0;
      w.y = v1;
      *reinterpret_cast<double2*>(out + gi * ld + gj0) = w;
    }
  } else {
    // generic runtime d: both operands from LDS
    for (int a = 0; a < TM / 8; ++a) {
      const int r = ty + 8 * a;
      double s0 = 0.0, s1 = 0.0;
      for (int k = 0; k < d; ++k) {
        double av = As[r * dp + k];
        double bv0 = Bs[c0 * dp + k], bv1 = Bs[(c0 + 1) * dp + k];
        if (KIND == GPX_K_MEHLER) {
          s0 = fma(av, bv0, s0);
          s1 = fma(av, bv1, s1);
        } else {
          double e0 = av - bv0, e1 = av - bv1;
          s0 = fma(e0, e0, s0);
          s1 = fma(e1, e1, s1);
        }
      }
      if (KIND == GPX_K_MEHLER) {
        double pa = As[r * dp + d];
        s0 = pa + Bs[c0 * dp + d] - s0;
        s1 = pa + Bs[(c0 + 1) * dp + d] - s1;
      }
      const int64_t gi = i0 + r;
      double v0 = kvalue<KIND>(s0, sig), v1 = kvalue<KIND>(s1, sig);
      const bool rin = gi < na;
      if (!(rin && gj0 < nb)) v0 = (symmetric && gi == gj0) ? 1.0 : 0.0;
      if (!(rin && gj1 < nb)) v1 = (symmetric && gi == gj1) ? 1.0 : 0.0;
      if (symmetric && rin && nugget_len > 0) {
        double nz = nugget_len == 1 ? nugget_scalar : nugget[gi];
        if (gi == gj0) v0 += nz;
        if (gi == gj1) v1 += nz;
      }
      double2 w;
      w.x = v0;
      w.y = v1;
      *reinterpret_cast<double2*>(out + gi * ld + gj0) = w;
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

template <int KIND>
int launch_kind(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb,
                int symmetric, const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out,
                int64_t prows, int64_t pcols, int64_t ld) {
  dim3 grid((unsigned)(pcols / TN), (unsigned)(prows / TM));
  const int dp = kp.d + (KIND == GPX_K_MEHLER ? 1 : 0);
  size_t sh = (size_t)2 * TM * dp * sizeof(double);
#define GPX_KF(DT)                                                                                           \
  hipLaunchKernelGGL((kfill_kernel<KIND, DT>), grid, dim3(256), sh, ctx->stream, kp, A, na, B, nb, symmetric, \
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
  switch (kp.kind) {
    case GPX_K_SE:
      return launch_kind<GPX_K_SE>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar, out,
                                   prows, pcols, ld);
    case GPX_K_MATERN32:
      return launch_kind<GPX_K_MATERN32>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar,
                                         out, prows, pcols, ld);
    case GPX_K_MATERN52:
      return launch_kind<GPX_K_MATERN52>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar,
                                         out, prows, pcols, ld);
    case GPX_K_MEHLER:
      return launch_kind<GPX_K_MEHLER>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar,
                                       out, prows, pcols, ld);
  }
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
