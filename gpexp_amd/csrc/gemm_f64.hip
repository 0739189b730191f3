// fp64 MFMA GEMM for gfx950:  C = C - A*op(B)  or  C = A*op(B)   (row-major, 128x128 tiles)
//
// This is the one compute kernel behind the Cholesky trailing update (SYRK, lower tiles only), the
// recursive TRSM updates and the leaf "multiply by the inverted diagonal block" steps -- i.e. it
// carries the N^3/3 flops that replace numpy.linalg.pinv (gp.py:181,400) and the N^2*M flops that
// replace the per-point k^T P k loop (gp.py:142-144, 253-255).
//
// Tiling (64-wide wavefronts): a 256-thread workgroup owns a 128x128 tile of C; its four waves
// form a 2x2 grid of 64x64 sub-tiles, each a 4x4 array of v_mfma_f64_16x16x4_f64 accumulators
// (16 x 4 fp64 = 128 VGPRs; measured issue rate of that MFMA: one per 64 cycles per SIMD, 77.5 TF/s
// chip-wide).  K is consumed in steps of 16 through a double-buffered LDS stage fed by a TWO-step-deep
// register prefetch (global loads for step t+2 are issued before the 64 MFMAs of step t; the registers
// of step t+1 are written to the other LDS buffer after them; one barrier per step), two workgroups per CU.
//
// LDS images: [row][k] with an ODD row stride (17 doubles): hipcc fuses the per-lane fragment reads into
// ds_read2_b64, which banks modulo 32 dwords over 16-lane groups -- 16 rows x 2 dwords then cover all
// 32 banks exactly once (an even stride measured 40 % of LDS cycles as bank conflicts).  The NN
// operand image is [k][n] with stride 128+16.
//
// Workgroup -> tile map: XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs (observed, used for
// speed only), so ids congruent mod 8 share an L2.  Tiles are grouped in 8x8 super-blocks (8 A-panels +
// 8 B-panels feed 64 tiles) and each super-block is executed by 64 ids of one residue class, i.e. by
// the 64 workgroups resident on one XCD.  SYRK enumerates only super-blocks on/below the diagonal.
//
// MFMA f64 16x16x4 operand maps (cdna_hip_programming.md 3): A: lane l holds A[l&15][l>>4];
// B: lane l holds B[l>>4][l&15]; C/D: reg v of lane l is C[(l>>4)+4v][l&15].
#include "gpx_internal.h"
#include <stdlib.h>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KB = 16;
constexpr int SA = KB + 1;    // ODD row stride (doubles) of [row][k] images
constexpr int SBN = BN + 16;  // row stride of the [k][n] image (NN operand)

template <bool BT>
struct Smem {
  double a[2][BM * SA];
  double b[2][BT ? BN * SA : KB * SBN];
};

struct Stage {  // one k-step of one thread's global->LDS staging traffic: 4 x 16 B of A, 4 x 16 B of B
  double2 a0, a1, a2, a3, b0, b1, b2, b3;
};

// decode a super-block index + position inside it into tile coordinates; false = nothing to do for this slot
template <bool LOWER>
__device__ __forceinline__ bool tile_of(int sblk, int within, int tiles_m, int tiles_n, int sb_cols, int sb_shift,
                                        int* by, int* bx) {
  const int sbm = (1 << sb_shift) - 1;
  int sr, sc;
  if (LOWER) {  // sblk-th super-block of the lower triangle, row-major: sr(sr+1)/2 <= sblk
    sr = (int)((sqrtf(8.0f * (float)sblk + 1.0f) - 1.0f) * 0.5f);
    while ((sr + 1) * (sr + 2) / 2 <= sblk) ++sr;
    while (sr * (sr + 1) / 2 > sblk) --sr;
    sc = sblk - sr * (sr + 1) / 2;
  } else {
    sr = sblk / sb_cols;
    sc = sblk - sr * sb_cols;
  }
  *by = (sr << sb_shift) + (within >> sb_shift);
  *bx = (sc << sb_shift) + (within & sbm);
  if (*by >= tiles_m || *bx >= tiles_n) return false;
  if (LOWER && *bx > *by) return false;  // tile strictly above the diagonal
  return true;
}

template <bool BT, bool ACC, int PF, int KU>
__device__ __forceinline__ void gemm_tile(Smem<BT>& sm, const double* A, int64_t lda, const double* B, int64_t ldb,
                                          double* C, int64_t ldc, int nk, int by, int bx) {
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;

  // global->register staging maps.  Addresses are (wave-uniform base in SGPRs) + (32-bit per-thread byte offset):
  // one VGPR per operand instead of eight 64-bit pointers.
  const int ar = t >> 3, ac = (t & 7) * 2;           // A (and B^T): rows ar+32*i, k offset ac
  const int br = t >> 6, bc = (t & 63) * 2;          // B (NN): k rows br+4*i, col offset bc
  const unsigned voff_a = (unsigned)((ar * lda + ac) * 8);
  const unsigned voff_b = BT ? (unsigned)((ar * ldb + ac) * 8) : (unsigned)((br * ldb + bc) * 8);
  const char* const Abase = reinterpret_cast<const char*>(A + m0 * lda);
  const char* const Bbase = reinterpret_cast<const char*>(BT ? (B + n0 * ldb) : (B + n0));
  const int64_t a_rs = 32 * lda * 8;                                   // bytes between the 4 staged A rows
  const int64_t b_rs = (BT ? 32 * ldb : 4 * ldb) * 8;                  // B^T: 32 rows apart; B (NN): 4 k-rows apart
  const int64_t a_ks = (int64_t)KB * 8;                                // bytes per k-step
  const int64_t b_ks = (BT ? (int64_t)KB : (int64_t)KB * ldb) * 8;

#define GPX_LD16(base_, voff_) (*reinterpret_cast<const double2*>((base_) + (voff_)))
#define GPX_GLOAD(S_, kt_)                                                \
  do {                                                                    \
    const char* ap_ = Abase + (int64_t)(kt_) * a_ks;                      \
    const char* bp_ = Bbase + (int64_t)(kt_) * b_ks;                      \
    S_.a0 = GPX_LD16(ap_, voff_a);                                        \
    S_.a1 = GPX_LD16(ap_ + a_rs, voff_a);                                 \
    S_.a2 = GPX_LD16(ap_ + 2 * a_rs, voff_a);                             \
    S_.a3 = GPX_LD16(ap_ + 3 * a_rs, voff_a);                             \
    S_.b0 = GPX_LD16(bp_, voff_b);                                        \
    S_.b1 = GPX_LD16(bp_ + b_rs, voff_b);                                 \
    S_.b2 = GPX_LD16(bp_ + 2 * b_rs, voff_b);                             \
    S_.b3 = GPX_LD16(bp_ + 3 * b_rs, voff_b);                             \
  } while (0)

  double* const sa_w = &sm.a[0][ar * SA + ac];
  double* const sb_w = BT ? &sm.b[0][ar * SA + ac] : &sm.b[0][br * SBN + bc];
  constexpr int A_BUF = BM * SA;
  constexpr int B_BUF = BT ? BN * SA : KB * SBN;
  constexpr int B_WS = BT ? 32 * SA : 4 * SBN;
  // [row][k] images have an odd stride: rows are only 8-byte aligned -> two 8-byte stores per 16-byte register pair
#define GPX_ST2(p_, v_)  \
  do {                   \
    (p_)[0] = (v_).x;    \
    (p_)[1] = (v_).y;    \
  } while (0)
#define GPX_SSTORE(S_, buf_)                                                                     \
  do {                                                                                           \
    double* aw_ = sa_w + (buf_) * A_BUF;                                                         \
    double* bw_ = sb_w + (buf_) * B_BUF;                                                         \
    GPX_ST2(aw_, S_.a0);                                                                         \
    GPX_ST2(aw_ + 32 * SA, S_.a1);                                                               \
    GPX_ST2(aw_ + 64 * SA, S_.a2);                                                               \
    GPX_ST2(aw_ + 96 * SA, S_.a3);                                                               \
    if (BT) {                                                                                    \
      GPX_ST2(bw_, S_.b0);                                                                       \
      GPX_ST2(bw_ + B_WS, S_.b1);                                                                \
      GPX_ST2(bw_ + 2 * B_WS, S_.b2);                                                            \
      GPX_ST2(bw_ + 3 * B_WS, S_.b3);                                                            \
    } else {                                                                                     \
      *reinterpret_cast<double2*>(bw_) = S_.b0;                                                  \
      *reinterpret_cast<double2*>(bw_ + B_WS) = S_.b1;                                           \
      *reinterpret_cast<double2*>(bw_ + 2 * B_WS) = S_.b2;                                       \
      *reinterpret_cast<double2*>(bw_ + 3 * B_WS) = S_.b3;                                       \
    }                                                                                            \
  } while (0)

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int fr = lane & 15, fk = lane >> 4;
  const double* const as0 = &sm.a[0][(wm * 64 + fr) * SA + fk];
  const double* const bs0 = BT ? &sm.b[0][(wn * 64 + fr) * SA + fk] : &sm.b[0][fk * SBN + wn * 64 + fr];
// KU = k-substeps (of 4) unrolled together: 4 lets the compiler hoist every fragment read of the step (most VGPRs),
// 2 keeps the depth-2 prefetch variant under the 256-VGPR cap without spills.
#define GPX_COMPUTE(buf_)                                                                            \
  do {                                                                                               \
    const double* as = as0 + (buf_) * A_BUF;                                                         \
    const double* bs = bs0 + (buf_) * B_BUF;                                                         \
    _Pragma("unroll 1") for (int kk0 = 0; kk0 < KB / 4; kk0 += KU) {                                 \
      _Pragma("unroll") for (int ku = 0; ku < KU; ++ku) {                                            \
        const int kk = kk0 + ku;                                                                     \
        double af[4], bf[4];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = as[(i * 16) * SA + kk * 4];            \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                \
            bf[j] = BT ? bs[(j * 16) * SA + kk * 4] : bs[(kk * 4) * SBN + j * 16];                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)  \
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);      \
      }                                                                                              \
    }                                                                                                \
  } while (0)

  if (PF == 2) {
    // prologue: steps 0 and 1 in flight, step 0 staged
    Stage P, Q;
    GPX_GLOAD(P, 0);
    if (nk > 1) GPX_GLOAD(Q, 1);
    GPX_SSTORE(P, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {  // steady state, two steps per trip so that the register sets stay static
      GPX_GLOAD(P, kt + 2);
      GPX_COMPUTE(0);
      GPX_SSTORE(Q, 1);
      __syncthreads();
      GPX_GLOAD(Q, kt + 3);
      GPX_COMPUTE(1);
      GPX_SSTORE(P, 0);
      __syncthreads();
    }
    // tail: 1..3 steps left; at entry LDS buffer 0 holds step kt and Q holds step kt+1 (if it exists)
    if (kt + 2 < nk) GPX_GLOAD(P, kt + 2);
    GPX_COMPUTE(0);
    if (kt + 1 < nk) {
      GPX_SSTORE(Q, 1);
      __syncthreads();
      GPX_COMPUTE(1);
      if (kt + 2 < nk) {
        GPX_SSTORE(P, 0);
        __syncthreads();
        GPX_COMPUTE(0);
      }
    }
  } else {
    Stage P;
    GPX_GLOAD(P, 0);
    GPX_SSTORE(P, 0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {  // prefetch step kt+1 while the MFMAs of step kt run
      const int buf = kt & 1;
      GPX_GLOAD(P, kt + 1);
      GPX_COMPUTE(buf);
      GPX_SSTORE(P, buf ^ 1);
      __syncthreads();
    }
    GPX_COMPUTE(kt & 1);
  }
#undef GPX_GLOAD
#undef GPX_LD16
#undef GPX_SSTORE
#undef GPX_ST2
#undef GPX_COMPUTE

  // epilogue: reg v of lane l -> C[(l>>4)+4v][l&15] within each 16x16 tile
  double* Cw = C + (m0 + wm * 64 + fk) * ldc + n0 + wn * 64 + fr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double* cp = Cw + (int64_t)(i * 16 + 4 * v) * ldc;
      if (ACC) {
        double c0 = cp[0], c1 = cp[16], c2 = cp[32], c3 = cp[48];
        cp[0] = c0 - acc[i][0][v];
        cp[16] = c1 - acc[i][1][v];
        cp[32] = c2 - acc[i][2][v];
        cp[48] = c3 - acc[i][3][v];
      } else {
        cp[0] = acc[i][0][v];
        cp[16] = acc[i][1][v];
        cp[32] = acc[i][2][v];
        cp[48] = acc[i][3][v];
      }
    }
}


// one workgroup per tile; ids congruent mod 8 are assumed to share an XCD (true for the first wave of workgroups)
template <bool BT, bool ACC, bool LOWER, int PF, int KU>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(const double* A, int64_t lda, const double* B, int64_t ldb,
                                                          double* C, int64_t ldc, int nk, int tiles_m, int tiles_n,
                                                          int sb_cols, int sb_shift) {
  __shared__ Smem<BT> sm;
  const int w = blockIdx.x;
  const int xcd = w & 7, q = w >> 3;
  const int sbs2 = 2 * sb_shift;
  int by, bx;
  if (!tile_of<LOWER>((q >> sbs2) * 8 + xcd, q & ((1 << sbs2) - 1), tiles_m, tiles_n, sb_cols, sb_shift, &by, &bx))
    return;
  gemm_tile<BT, ACC, PF, KU>(sm, A, lda, B, ldb, C, ldc, nk, by, bx);
}

// Persistent variant for large grids: 2 workgroups per CU stay resident; each reads the XCD it really runs on
// (HW_REG_XCC_ID) and pulls (super-block, tile) slots of THAT XCD from a per-XCD atomic counter.  After the first wave
// the dispatcher hands workgroup ids to whichever XCD frees a slot, which smears a super-block over several L2s
// (measured: 18-31 % L2 hit on large GEMMs against 70-81 % when every XCD works on one super-block at a time).
// Every workgroup leaves the loop as soon as its XCD's slots are exhausted: no spinning, no inter-workgroup waits.
// STATUS (round 1): correct, but opt-in (GPX_GEMM_PERSIST_MIN=<tiles>).  PMC FETCH_SIZE showed the hit rate does not
// recover with exact XCD placement alone (4096x8192x16384: 51 GB fetched with or without it, 69 GB requested): the
// first wave of a launch shares panels because it starts in lock-step; later tiles start whenever a slot frees and the
// workgroups drift apart by more than the ~16 k-steps a 4 MiB L2 can bridge.  Next step: a bounded (performance-only)
// per-XCD re-synchronisation every few hundred k-steps on top of this kernel.
template <bool BT, bool ACC, bool LOWER, int PF, int KU>
__global__ __launch_bounds__(256, 2) void gemm_f64_persistent(const double* A, int64_t lda, const double* B,
                                                              int64_t ldb, double* C, int64_t ldc, int nk,
                                                              int tiles_m, int tiles_n, int sb_cols, int sb_shift,
                                                              int nsb, int* __restrict__ counters) {
  __shared__ Smem<BT> sm;
  __shared__ int s_slot;
  const int xcd = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;  // HW_REG_XCC_ID[3:0]
  const int sbs2 = 2 * sb_shift;
  for (;;) {
    if (threadIdx.x == 0) s_slot = atomicAdd(&counters[xcd], 1);
    __syncthreads();
    const int slot = s_slot;
    __syncthreads();  // everyone has read the slot (and finished the previous tile's LDS reads) before it is reused
    const int sblk = (slot >> sbs2) * 8 + xcd;
    if (sblk >= nsb) break;
    int by, bx;
    if (!tile_of<LOWER>(sblk, slot & ((1 << sbs2) - 1), tiles_m, tiles_n, sb_cols, sb_shift, &by, &bx)) continue;
    gemm_tile<BT, ACC, PF, KU>(sm, A, lda, B, ldb, C, ldc, nk, by, bx);
  }
}

}  // namespace

int launch_gemm(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                int64_t m, int64_t n, int64_t k, bool bt, bool accumulate, bool lower) {
  if (m == 0 || n == 0) return 0;
  GPX_ARG(m % BM == 0 && n % BN == 0 && k % KB == 0 && k > 0, "gemm: m,n must be multiples of 128 and k of 16");
  GPX_ARG((lda % 2) == 0 && (ldb % 2) == 0, "gemm: leading dimensions must be even (16-byte loads)");
  GPX_ARG(!lower || m == n, "gemm: lower-only update needs a square C");
  const int tm = (int)(m / BM), tn = (int)(n / BN);
  // largest super-block edge (8,4,2,1 tiles) that still leaves >= 16 super-blocks, i.e. >= 2 per XCD
  int sb_shift = 3, sbr = 0, sbc = 0;
  int64_t nsb = 0;
  for (;; --sb_shift) {
    const int e = 1 << sb_shift;
    sbr = (tm + e - 1) / e;
    sbc = (tn + e - 1) / e;
    nsb = lower ? (int64_t)sbr * (sbr + 1) / 2 : (int64_t)sbr * sbc;
    if (nsb >= 16 || sb_shift == 0) break;
  }
  const int64_t wgs = (nsb + 7) / 8 * 8 * ((int64_t)1 << (2 * sb_shift));  // super-blocks dealt over 8 XCD classes
  GPX_ARG(wgs < ((int64_t)1 << 31), "gemm: grid too large");
  dim3 grid((unsigned)wgs);
  const int nk = (int)(k / KB);
  const double tiles = lower ? 0.5 * (double)tm * ((double)tm + 1.0) : (double)tm * (double)tn;
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * tiles * BM * BN * (double)k, 0.0);
  static int pf = -1, persist_min = -1;
  if (pf < 0) {
    const char* e = getenv("GPX_GEMM_PF");
    pf = e ? atoi(e) : 0;  // 0: default (per-variant best); 2: depth-2 prefetch; 11/12/14: depth-1 + inner unroll 1/2/4
    const char* e2 = getenv("GPX_GEMM_PERSIST_MIN");  // tiles from which the persistent kernel is used (0 = never)
    persist_min = e2 ? atoi(e2) : 0;   // opt-in: measured neutral/slightly slower, see the kernel's comment
  }
  const bool persist = persist_min > 0 && tiles >= (double)persist_min && nsb >= 16;
  int* counters = nullptr;
  if (persist) {
    counters = ctx->d_counters + 8 * (ctx->counter_slot++ % GPX_COUNTER_SLOTS);
    GPX_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int), ctx->stream));
    grid = dim3((unsigned)(2 * ctx->cus));
  }
#define GPX_G(BT_, ACC_, LOW_)                                                                                        \
  do {                                                                                                                \
    if (persist)                                                                                                      \
      hipLaunchKernelGGL((gemm_f64_persistent<BT_, ACC_, LOW_, 1, 4>), grid, dim3(256), 0, ctx->stream, A, lda, B,    \
                         ldb, C, ldc, nk, tm, tn, sbc, sb_shift, (int)nsb, counters);                                 \
    else if (pf == 2)                                                                                                 \
      hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_, 2, 2>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb,   \
                         C, ldc, nk, tm, tn, sbc, sb_shift);                                                          \
    else if (pf == 12)                                                                                                \
      hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_, 1, 2>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb,   \
                         C, ldc, nk, tm, tn, sbc, sb_shift);                                                          \
    else if (pf == 11)                                                                                                \
      hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_, 1, 1>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb,   \
                         C, ldc, nk, tm, tn, sbc, sb_shift);                                                          \
    else if (pf == 14)                                                                                                \
      hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_, 1, 4>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb,   \
                         C, ldc, nk, tm, tn, sbc, sb_shift);                                                          \
    else /* default: measured best per operand form (C4 step: NT full unroll 287 vs 295 ms; NN unroll-2 534 vs 600 ms) */ \
      hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_, 1, (BT_ ? 4 : 2)>), grid, dim3(256), 0, ctx->stream, A,    \
                         lda, B, ldb, C, ldc, nk, tm, tn, sbc, sb_shift);                                             \
  } while (0)
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
