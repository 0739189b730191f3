// fp64 MFMA GEMM for gfx950:  C = C - A*op(B)  or  C = A*op(B)   (row-major, 128x128 tiles)
//
// This is the one compute kernel behind the Cholesky trailing update (SYRK, lower tiles only), the
// recursive TRSM updates and the leaf "multiply by the inverted diagonal block" steps -- i.e. it
// carries the N^3/3 flops that replace numpy.linalg.pinv (gp.py:181,400) and the N^2*M flops that
// replace the per-point k^T P k loop (gp.py:142-144, 253-255).
//
// Tiling (64-wide wavefronts): a 256-thread workgroup owns a 128x128 tile of C; its four waves
// form a 2x2 grid of 64x64 sub-tiles, each a 4x4 array of v_mfma_f64_16x16x4_f64 accumulators
// (16 x 4 fp64 = 128 VGPRs).  K is consumed in steps of 16 through a double-buffered LDS stage:
// global loads for step t+1 are issued before the 64 MFMAs of step t and written to the other
// LDS buffer afterwards, one barrier per step.  LDS row stride is KB+2 doubles (== 2 mod 32), which
// makes the 16-row x 2-k ds_read_b64 fragment pattern of a 32-lane group hit 32 distinct 8-byte
// bank pairs; the NN operand uses stride 128+16 for the same reason.
//
// MFMA f64 16x16x4 operand maps (cdna_hip_programming.md 3): A: lane l holds A[l&15][l>>4];
// B: lane l holds B[l>>4][l&15]; C/D: reg v of lane l is C[(l>>4)+4v][l&15].
#include "gpx_internal.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KB = 16;
constexpr int SA = KB + 2;    // row stride (doubles) of [row][k] images
constexpr int SBN = BN + 16;  // row stride of the [k][n] image (NN operand)

template <bool BT>
struct Smem {
  double a[2][BM * SA];
  double b[2][BT ? BN * SA : KB * SBN];
};

template <bool BT, bool ACC, bool LOWER>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(const double* A, int64_t lda,
                                                          const double* B, int64_t ldb,
                                                          double* C, int64_t ldc, int nk) {
  const int bx = blockIdx.x, by = blockIdx.y;
  if (LOWER && bx > by) return;  // tile strictly above the diagonal
  __shared__ Smem<BT> sm;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;

  // global->register staging maps
  const int ar = t >> 3, ac = (t & 7) * 2;           // A (and B^T): rows ar+32*i, k offset ac
  const int br = t >> 6, bc = (t & 63) * 2;          // B (NN): k rows br+4*i, col offset bc
  const double* Ag = A + (m0 + ar) * lda + ac;
  const double* Bg = BT ? (B + (n0 + ar) * ldb + ac) : (B + (int64_t)br * ldb + n0 + bc);

  // staging registers (kept as named SSA values: arrays captured by lambdas end up in scratch)
  double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const int64_t a_rs = 32 * lda;                       // A row step between the 4 staged rows
  const int64_t b_rs = BT ? 32 * ldb : 4 * ldb;        // B^T: 32 rows apart; B (NN): 4 k-rows apart
  const int64_t b_ks = BT ? (int64_t)KB : (int64_t)KB * ldb;  // advance per k-step
#define GPX_GLOAD(kt_)                                                        \
  do {                                                                        \
    const double* ap_ = Ag + (int64_t)(kt_) * KB;                             \
    const double* bp_ = Bg + (int64_t)(kt_) * b_ks;                           \
    ra0 = *reinterpret_cast<const double2*>(ap_);                             \
    ra1 = *reinterpret_cast<const double2*>(ap_ + a_rs);                      \
    ra2 = *reinterpret_cast<const double2*>(ap_ + 2 * a_rs);                  \
    ra3 = *reinterpret_cast<const double2*>(ap_ + 3 * a_rs);                  \
    rb0 = *reinterpret_cast<const double2*>(bp_);                             \
    rb1 = *reinterpret_cast<const double2*>(bp_ + b_rs);                      \
    rb2 = *reinterpret_cast<const double2*>(bp_ + 2 * b_rs);                  \
    rb3 = *reinterpret_cast<const double2*>(bp_ + 3 * b_rs);                  \
  } while (0)
  double* const sa_w = &sm.a[0][ar * SA + ac];
  double* const sb_w = BT ? &sm.b[0][ar * SA + ac] : &sm.b[0][br * SBN + bc];
  constexpr int A_BUF = BM * SA;
  constexpr int B_BUF = BT ? BN * SA : KB * SBN;
  constexpr int B_WS = BT ? 32 * SA : 4 * SBN;
#define GPX_SSTORE(buf_)                                                               \
  do {                                                                                 \
    double* aw_ = sa_w + (buf_) * A_BUF;                                               \
    double* bw_ = sb_w + (buf_) * B_BUF;                                               \
    *reinterpret_cast<double2*>(aw_) = ra0;                                            \
    *reinterpret_cast<double2*>(aw_ + 32 * SA) = ra1;                                  \
    *reinterpret_cast<double2*>(aw_ + 64 * SA) = ra2;                                  \
    *reinterpret_cast<double2*>(aw_ + 96 * SA) = ra3;                                  \
    *reinterpret_cast<double2*>(bw_) = rb0;                                            \
    *reinterpret_cast<double2*>(bw_ + B_WS) = rb1;                                     \
    *reinterpret_cast<double2*>(bw_ + 2 * B_WS) = rb2;                                 \
    *reinterpret_cast<double2*>(bw_ + 3 * B_WS) = rb3;                                 \
  } while (0)

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int fr = lane & 15, fk = lane >> 4;
  const double* const as0 = &sm.a[0][(wm * 64 + fr) * SA + fk];
  const double* const bs0 = BT ? &sm.b[0][(wn * 64 + fr) * SA + fk] : &sm.b[0][fk * SBN + wn * 64 + fr];
#define GPX_COMPUTE(buf_)                                                                            \
  do {                                                                                               \
    const double* as = as0 + (buf_) * A_BUF;                                                         \
    const double* bs = bs0 + (buf_) * B_BUF;                                                         \
    _Pragma("unroll") for (int kk = 0; kk < KB / 4; ++kk) {                                          \
      double af[4], bf[4];                                                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = as[(i * 16) * SA + kk * 4];              \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
          bf[j] = BT ? bs[(j * 16) * SA + kk * 4] : bs[(kk * 4) * SBN + j * 16];                     \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)    \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);        \
    }                                                                                                \
  } while (0)

  GPX_GLOAD(0);
  GPX_SSTORE(0);
  __syncthreads();
  int kt = 0;
  for (; kt + 1 < nk; ++kt) {  // steady state: prefetch step kt+1 while the MFMAs of step kt run
    const int buf = kt & 1;
    GPX_GLOAD(kt + 1);
    GPX_COMPUTE(buf);
    GPX_SSTORE(buf ^ 1);
    __syncthreads();
  }
  GPX_COMPUTE(kt & 1);  // last step: nothing left to prefetch
#undef GPX_GLOAD
#undef GPX_SSTORE
#undef GPX_COMPUTE

  // epilogue: reg v of lane l -> C[(l>>4)+4v][l&15] within each 16x16 tile
  double* Cw = C + (m0 + wm * 64 + fk) * ldc + n0 + wn * 64 + fr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double* cp = Cw + (int64_t)(i * 16 + 4 * v) * ldc;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ACC)
          cp[j * 16] = cp[j * 16] - acc[i][j][v];
        else
          cp[j * 16] = acc[i][j][v];
      }
    }
}

}  // namespace

int launch_gemm(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                int64_t m, int64_t n, int64_t k, bool bt, bool accumulate, bool lower) {
  if (m == 0 || n == 0) return 0;
  GPX_ARG(m % BM == 0 && n % BN == 0 && k % KB == 0 && k > 0, "gemm: m,n must be multiples of 128 and k of 16");
  GPX_ARG(m / BM <= 65535, "gemm: too many row tiles");
  GPX_ARG((lda % 2) == 0 && (ldb % 2) == 0, "gemm: leading dimensions must be even (16-byte loads)");
  dim3 grid((unsigned)(n / BN), (unsigned)(m / BM));
  const int nk = (int)(k / KB);
  double tiles = lower ? 0.5 * (double)(m / BM) * ((double)(m / BM) + 1.0) : (double)(m / BM) * (double)(n / BN);
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * tiles * BM * BN * (double)k, 0.0);
#define GPX_G(BT_, ACC_, LOW_)                                                                                  \
  hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb, C, ldc, \
                     nk)
  if (bt) {
    if (accumulate) {
      if (lower) GPX_G(true, true, true); else GPX_G(true, true, false);
    } else {
      if (lower) GPX_G(true, false, true); else GPX_G(true, false, false);
    }
  } else {
    if (accumulate) {
      if (lower) GPX_G(false, true, true); else GPX_G(false, true, false);
    } else {
      if (lower) GPX_G(false, false, true); else GPX_G(false, false, false);
    }
  }
#undef GPX_G
  GPX_HIP(hipGetLastError());
  return 0;
}
