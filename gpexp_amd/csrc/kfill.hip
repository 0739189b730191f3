// kfill: dense covariance assembly  K[i][j] = k(A_i, B_j) (+ nugget on the diagonal)  -- gfx950.
//
// Replaces the reference's N-iteration Python row loop (gp_kernel_utilities.py:56-60, one
// Kernel.evaluate + np.tile per row) and the column loops that build kernelvals
// (gp.py:132-135, 246-249; experimentalDesign.py:829-831).
//
// HBM-bound by intent: algorithmic bytes = 8*rows*cols written + 8*(rows+cols)*d read.  What it takes to get
// there in fp64 (no hardware exp/sqrt):
//   * 64x64 tiles; point coordinates staged once per tile into LDS pre-multiplied by the length scale.  The Gram
//     term of the distance, a.b, is a K=d GEMM and runs on the fp64 MFMA pipe (|a-b|^2 = |a|^2 + |b|^2 - 2 a.b);
//     the VALU -- measured at ~5.4 cycles per wave64 fp64 FMA, the binding resource of the all-VALU version (71
//     instructions per element) -- only evaluates the kernel function.  Diagonal entries keep an exact zero distance.
//   * symmetric fill: only tiles on/below the diagonal are COMPUTED; each is also written transposed through a
//     padded LDS image (coalesced mirror stores) -- half the exp/sqrt work for the same 8*N^2 bytes.
//   * exp: one rndne + two-constant Cody-Waite reduction + degree-13 polynomial + v_ldexp (about 20 fp64
//     instructions, no special-case branches: arguments here are finite and <= ~1); sqrt: v_rsq_f64 + two coupled
//     Newton steps + residual correction.  Both stay within 2 ulp (tests: 1e-13 against the oracle / the reference).
#include "gpx_internal.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int TM = 64;  // tile rows
constexpr int TN = 64;  // tile cols
constexpr int TP = TN + 1;  // padded stride of the transpose image

// exp(x) = 2^n * 2^(j/32) * exp(r):  m = rint(x * 32/ln2), n = m >> 5, j = m & 31, r = x - m*ln2/32 (two-constant
// Cody-Waite), |r| <= ln2/64, so a degree-6 polynomial is exact to 3.5e-18; tab[j] = 2^(j/32) sits in LDS (32 entries
// = 32 distinct bank pairs: gathers are conflict-free).  11 fp64 ops + 5 integer ops instead of 19 fp64 ops.
__device__ __forceinline__ double fast_exp(double x, const double* __restrict__ tab) {
  const double m = rint(x * 46.166241308446828384);           // 32 / ln 2
  double r = fma(m, -2.16608493865351192653e-02, x);          // ln2/32, high part
  r = fma(m, -5.96317165397058692545e-12, r);                 // ln2/32, low part
  const int mi = (int)m;
  const double tj = tab[mi & 31];
  double p = 1.3888888888888889419e-03;                       // 1/720
  p = fma(p, r, 8.3333333333333332177e-03);                   // 1/120
  p = fma(p, r, 4.1666666666666664354e-02);                   // 1/24
  p = fma(p, r, 1.6666666666666665741e-01);                   // 1/6
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(tj * p, mi >> 5);
}

__device__ __forceinline__ double fast_sqrt(double xin) {  // xin >= 0 (squared scaled distances)
  // coincident points give exactly 0: clamp instead of a compare + two selects; sqrt(1e-300) = 1e-150 leaves every
  // kernel value bit-identical to the r = 0 result (1 + 1e-150 == 1, exp(-1e-150) == 1)
  const double x = fmax(xin, 1e-300);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}

template <int KIND>
__device__ __forceinline__ double kvalue(double acc, double sig, const double* __restrict__ tab) {
  if (KIND == GPX_K_SE) {
    return sig * fast_exp(-0.5 * acc, tab);
  } else if (KIND == GPX_K_MATERN32) {
    const double t = fast_sqrt(acc);
    return sig * (1.0 + t) * fast_exp(-t, tab);
  } else if (KIND == GPX_K_MATERN52) {
    const double t = fast_sqrt(acc);
    return sig * (1.0 + t + acc * (1.0 / 3.0)) * fast_exp(-t, tab);
  } else {  // Mehler: acc = pa + pb - cross
    return sig * fast_exp(-acc, tab);
  }
}

// 2^(j/32), j = 0..31 (correctly rounded)
__device__ const double kExp2Tab[32] = {
    1.0, 1.0218971486541166, 1.0442737824274138, 1.0671404006768237, 1.0905077326652577, 1.1143867425958924,
    1.1387886347566916, 1.1637248587775775, 1.189207115002721, 1.215247359980469, 1.241857812073484,
    1.2690509571917332, 1.2968395546510096, 1.3252366431597413, 1.3542555469368927, 1.383909881963832,
    1.4142135623730951, 1.4451808069770467, 1.4768261459394993, 1.5091644275934228, 1.5422108254079407,
    1.5759808451078865, 1.6104903319492543, 1.645755478153965, 1.681792830507429, 1.718619298122478,
    1.7562521603732995, 1.7947090750031072, 1.8340080864093424, 1.8741676341103, 1.9152065613971474,
    1.9571441241754002};

typedef double d4 __attribute__((ext_vector_type(4)));

// Stage one 64-point tile into LDS for the MFMA Gram product: P[p][k] = coordinate k of point p times its scale
// (Mehler: times c2 on the A side), zero for k >= d (K is padded to a multiple of 4), and nrm[p] = the point's own
// term of the exponent (sum of squared scaled coordinates, or sum c1 x^2 for Mehler).
template <int KIND, bool SIDE_A>
__device__ __forceinline__ void stage_points(const KParams& kp, int d, int dpad, int sl, const double* __restrict__ X,
                                             int64_t n, int64_t g0, double* P, double* nrm) {
  const int t = threadIdx.x;
  for (int idx = t; idx < TM * dpad; idx += 256) {
    const int p = idx / dpad, k = idx - p * dpad;
    const int64_t g = g0 + p;
    double v = 0.0;
    if (g < n && k < d) {
      v = X[g * d + k];
      if (KIND == GPX_K_MEHLER) {
        if (SIDE_A) v *= kp.c2[k];
      } else {
        v *= kp.scale[k];
      }
    }
    P[p * sl + k] = v;
  }
  if (t < TM) {
    const int64_t g = g0 + t;
    double s = 0.0;
    if (g < n)
      for (int k = 0; k < d; ++k) {
        const double v = X[g * d + k];
        if (KIND == GPX_K_MEHLER) {
          s = fma(kp.c1[k] * v, v, s);
        } else {
          const double w = v * kp.scale[k];
          s = fma(w, w, s);
        }
      }
    nrm[t] = s;
  }
}

// 64x64 tile per 256-thread workgroup; wave (wm, wn) owns a 32x32 quadrant = 2x2 fp64 MFMA 16x16 blocks.
// The Gram term c = a.b comes from v_mfma_f64_16x16x4_f64 (the matrix pipe is otherwise idle here and co-issues with
// the VALU); the exponent argument is |a|^2 + |b|^2 - 2c (SE / Matern) or pa + pb - c (Mehler).  MFMA column q of
// block ni is mapped to tile column 2q+ni, so a lane's two blocks are ADJACENT columns: 16-byte stores, 256-byte row
// segments per 16-lane group.
// SYM: 1-D grid over the tiles on/below the diagonal, mirror-written; otherwise 2-D grid over all tiles.
template <int KIND, bool SYM>
__global__ __launch_bounds__(256) void kfill_kernel(KParams kp, const double* __restrict__ A, int64_t na,
                                                    const double* __restrict__ B, int64_t nb, int symmetric,
                                                    const double* __restrict__ nugget, int64_t nugget_len,
                                                    double nugget_scalar, double* __restrict__ out, int64_t ld) {
  extern __shared__ double sm[];
  const int d = kp.d;
  const int dpad = (d + 3) & ~3;
  const int sl = dpad + 1;  // odd stride
  double* As = sm;
  double* Bs = As + TM * sl;
  double* pa = Bs + TN * sl;
  double* pb = pa + TM;
  double* tab = pb + TN;   // 2^(j/32) table
  double* Tr = tab + 32;   // [32][TP] transpose image (SYM only)
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
  stage_points<KIND, true>(kp, d, dpad, sl, A, na, i0, As, pa);
  stage_points<KIND, false>(kp, d, dpad, sl, B, nb, j0, Bs, pb);
  if (threadIdx.x < 32) tab[threadIdx.x] = kExp2Tab[threadIdx.x];
  __syncthreads();

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, q = lane & 15;

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* ap = As + (wm * 32 + q) * sl + g;           // A[row = mi*16 + q][k = 4s + g]
  const double* bp = Bs + (wn * 32 + 2 * q) * sl + g;       // B[col = 2q + ni][k = 4s + g]
  for (int ks = 0; ks < dpad; ks += 4) {
    const double a0 = ap[ks], a1 = ap[16 * sl + ks];
    const double b0 = bp[ks], b1 = bp[sl + ks];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }

  // this lane's elements: rows r(mi, v) = wm*32 + mi*16 + g + 4v, columns c0, c0+1 with c0 = wn*32 + 2q
  const int c0 = wn * 32 + 2 * q;
  const double pb0 = pb[c0], pb1 = pb[c0 + 1];
  const double sig = kp.sig;
  const int64_t gj0 = j0 + c0, gj1 = gj0 + 1;
  // interior tile: fully inside both point sets and (for the symmetric forms) not touching the diagonal
  const bool interior = (i0 + TM <= na) && (j0 + TN <= nb) && (!symmetric || (i0 + TM <= j0) || (j0 + TN <= i0));
  char* const otile = reinterpret_cast<char*>(out + i0 * ld + j0);
  const unsigned ooff = (unsigned)(((wm * 32 + g) * ld + c0) * 8);
  double2 val[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = wm * 32 + mi * 16 + g + 4 * v;
      const double par = pa[r];
      double s0, s1;
      if (KIND == GPX_K_MEHLER) {
        s0 = par + pb0 - acc[mi][0][v];
        s1 = par + pb1 - acc[mi][1][v];
      } else {  // squared scaled distance; tiny negative round-off is clamped
        s0 = fmax(fma(-2.0, acc[mi][0][v], par + pb0), 0.0);
        s1 = fmax(fma(-2.0, acc[mi][1][v], par + pb1), 0.0);
      }
      const int64_t gi = i0 + r;
      if (!interior && symmetric && KIND != GPX_K_MEHLER) {  // exact zero distance on the diagonal, as the reference has it
        if (gi == gj0) s0 = 0.0;
        if (gi == gj1) s1 = 0.0;
      }
      double v0 = kvalue<KIND>(s0, sig, tab), v1 = kvalue<KIND>(s1, sig, tab);
      if (!interior) {  // wave-uniform: only edge tiles (padding) and tiles crossing the diagonal (nugget) pay for this
        const bool rin = gi < na;
        if (!(rin && gj0 < nb)) v0 = (symmetric && gi == gj0) ? 1.0 : 0.0;
        if (!(rin && gj1 < nb)) v1 = (symmetric && gi == gj1) ? 1.0 : 0.0;
        if (symmetric && rin && nugget_len > 0) {
          const double nz = nugget_len == 1 ? nugget_scalar : nugget[gi];
          if (gi == gj0) v0 += nz;
          if (gi == gj1) v1 += nz;
        }
      }
      val[mi][v].x = v0;
      val[mi][v].y = v1;
      // address = (wave-uniform row-group base, SGPRs) + (per-lane 32-bit byte offset): no 64-bit VALU address math
      *reinterpret_cast<double2*>(otile + (int64_t)(mi * 16 + 4 * v) * ld * 8 + ooff) = val[mi][v];
    }

  if (SYM && ti != tj) {
    // mirror: out[j0 + c][i0 + r] = V[r][c]; the two row halves (wm = 0 / 1) go through the padded image in turn
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
      if (wm == half) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rl = mi * 16 + g + 4 * v;  // row inside the half
            Tr[rl * TP + c0] = val[mi][v].x;
            Tr[rl * TP + c0 + 1] = val[mi][v].y;
          }
      }
      __syncthreads();
      const int tx = t & 31, ty = t >> 5;
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const int c = ty + 8 * qq;
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
  out[j] = kvalue<KIND>(acc, kp.sig, kExp2Tab);
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
  const int dpad = (kp.d + 3) & ~3;
  const size_t sh = (size_t)(2 * TM * (dpad + 1) + 2 * TM + 32 + (SYM ? 32 * TP : 0)) * sizeof(double);
  hipLaunchKernelGGL((kfill_kernel<KIND, SYM>), grid, dim3(256), sh, ctx->stream, kp, A, na, B, nb, symmetric, d_nugget,
                     nugget_len, nugget_scalar, out, ld);
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
