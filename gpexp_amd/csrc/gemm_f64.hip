// fp64 MFMA GEMM for gfx950:  C = C - A*op(B)  or  C = A*op(B)   (row-major)
//
// This is the one compute kernel behind the Cholesky trailing update (SYRK, lower tiles only) and the recursive TRSM
// updates -- i.e. it carries the N^3/3 flops that replace numpy.linalg.pinv (gp.py:181,400) and the N^2*M flops that
// replace the per-point k^T P k loop (gp.py:142-144, 253-255).  (Products with the inverted 128x128 diagonal leaves have
// their own kernels in chol.hip.)
//
// Tiling (64-wide wavefronts): a 256-thread workgroup owns a TE x TE tile of C; its four waves form a 2x2
// grid, each wave an FI x FI array of v_mfma_f64_16x16x4_f64 accumulators (measured issue rate of that MFMA:
// one per 64 cycles per SIMD = 77.5 TF/s chip-wide).  Two instantiations:
//   TE = 128 (FI = 4, 128 accumulator VGPRs, 2 workgroups/CU)  the throughput kernel;
//   TE =  64 (FI = 2)                                          the latency kernel for launches with too few
//       128-tiles to fill the chip: 4x the workgroups, 1/4 of the serial MFMA chain each (a 128x128x128 tile
//       cannot finish in less than 8 k-steps x 64 MFMAs x 64 cycles = 13.6 us however empty the GPU is).
// K is consumed in steps of 16 through a double-buffered LDS stage fed by a register prefetch, on a hand-pinned
// schedule (see the loop): loads for step t+1 first, 48 of step t's 64 MFMAs, LDS store + the step's one barrier, the
// fragment reads of step t+1 underneath the remaining 16 MFMAs.
//
// LDS images: [row][k] with an ODD row stride (17 doubles): hipcc fuses the per-lane fragment reads into
// ds_read2_b64, which banks modulo 32 dwords over 16-lane groups -- 16 rows x 2 dwords then cover all
// 32 banks exactly once (an even stride measured 40 % of LDS cycles as bank conflicts).  The NN
// operand image is [k][n] with stride TE+16.
//
// Workgroup -> tile map: XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs (observed, used for
// speed only), so ids congruent mod 8 share an L2.  Tiles are grouped in 8x8 super-blocks (8 A-panels +
// 8 B-panels feed 64 tiles) and each super-block is executed by 64 ids of one residue class.  SYRK enumerates
// only super-blocks on/below the diagonal.
//
// MFMA f64 16x16x4 operand maps (cdna_hip_programming.md 3): A: lane l holds A[l&15][l>>4];
// B: lane l holds B[l>>4][l&15]; C/D: reg v of lane l is C[(l>>4)+4v][l&15].
#include "gpx_internal.h"
#include <stdlib.h>

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int KB = 16;      // k-step
constexpr int SA = KB + 1;  // ODD row stride (doubles) of [row][k] images

template <bool BT, int TE>
struct Smem {
  double a[2][TE * SA];
  double b[2][BT ? TE * SA : KB * (TE + 16)];
};

// decode a super-block index + position inside it into tile coordinates; false = nothing to do for this slot
// rot != 0 (triangular B operand: work grows with the column): the column super-block index is rotated from row to row so
// that every XCD sees every column class -- super-blocks are dealt to XCDs by index mod 8, so with 4 (or 8, 16) column
// super-blocks the unrotated map gives each XCD the same columns every time (XCDs 3 and 7 all the heavy ones)
template <bool LOWER>
__device__ __forceinline__ bool tile_of(int sblk, int within, int tiles_m, int tiles_n, int sb_cols, int sb_shift,
                                        int* by, int* bx, int rot = 0) {
  const int sbm = (1 << sb_shift) - 1;
  int sr, sc;
  if (LOWER) {
    // Super-blocks of the lower triangle: the STRICTLY lower ones first (row-major), the diagonal ones last.  A diagonal
    // super-block has 28 of its 64 slots exit at once; the XCD refills them with tiles of its next super-block, and from then
    // on its 64 resident workgroups belong to two super-blocks at different k phases -- the lock-step that makes them share
    // operand panels in L2 is gone for the rest of the launch (TCC hit rate 0.81 rectangular vs 0.58 lower, same shapes,
    // profiles/r04_l2_hitrate.txt).  With the diagonal ones at the end only the launch's tail pays.
    const int nsr = (tiles_m + sbm) >> sb_shift;   // super-block rows of the (main part of the) lower triangle
    const int noff = nsr * (nsr - 1) / 2;
    if (sblk < noff) {                             // sblk-th strictly-lower super-block: (sr-1) sr / 2 <= sblk, sr >= 1
      sr = (int)((sqrtf(8.0f * (float)sblk + 1.0f) + 1.0f) * 0.5f);
      while (sr * (sr + 1) / 2 <= sblk) ++sr;
      while ((sr - 1) * sr / 2 > sblk) --sr;
      sc = sblk - (sr - 1) * sr / 2;
    } else {
      sr = sc = sblk - noff;                       // (indices beyond the last diagonal block fall out below: by >= tiles_m)
    }
  } else {
    sr = sblk / sb_cols;
    sc = sblk - sr * sb_cols;
    if (rot) {
      if ((8 % sb_cols) == 0)
        sc = (sc + (sblk >> 3)) % sb_cols;   // a round of 8 super-blocks spans whole rows: rotate per round
      else if ((sb_cols & 7) == 0)
        sc = (sc + sr) % sb_cols;            // a row spans whole rounds: rotate per row
    }
  }
  *by = (sr << sb_shift) + (within >> sb_shift);
  *bx = (sc << sb_shift) + (within & sbm);
  if (!LOWER && rot == 2) {
    // Round 5, triangular operand B (tri == 2: the k range of column tile bx ends at its diagonal, so the tiles of a super-block
    // take 1 .. 8 units of time): column-major inside the super-block, LONGEST column first.  The slots an XCD's short tiles
    // free early are refilled in id order from its next super-block -- with its long tiles (longest-processing-time-first list
    // scheduling) instead of a row-major mix: the launch's makespan comes down from ~1.4x to ~1.15x of work / slots.
    *by = (sr << sb_shift) + (within & sbm);
    *bx = (sc << sb_shift) + (sbm - (within >> sb_shift));
  }
  if (!LOWER && rot == 4) {   // tri == 1: the k range of ROW tile by ends at its diagonal -- the last row tiles of a super-block first
    *by = (sr << sb_shift) + (sbm - (within >> sb_shift));
    *bx = (sc << sb_shift) + (within & sbm);
  }
  if (!LOWER && rot == 3) {   // tri == 4: the k range of column tile bx STARTS at its diagonal -- the first columns are the long ones
    *by = (sr << sb_shift) + (within & sbm);
    *bx = (sc << sb_shift) + (within >> sb_shift);
  }
  if (*by >= tiles_m || *bx >= tiles_n) return false;
  if (LOWER && *bx > *by) return false;  // tile strictly above the diagonal
  return true;
}

// SEGMENTED k range (SEGF::on): the k range is a concatenation of nseg segments of seg_nk k-steps each, segment s with its
// own operand base pointers segf.a(s) / segf.b(s) (the tile's first row of each operand; uniform scalars).  This is how the
// distributed factorisation applies several received panels -- which live in separate packed buffers -- in ONE pass over C
// (K = nseg * nb instead of nb: the C read-modify-write and the tile prologue / epilogue are paid once, not per panel).
// The dense instantiations (NoSeg) compile to exactly the code they had before: every use is behind `if constexpr`.
struct NoSeg {
  static constexpr bool on = false;
};

// (Variants measured and dropped: a 2-deep register prefetch; reading only two k-substeps' fragments at a time.)
template <bool BT, bool ACC, int TE, class SEGF = NoSeg>
__device__ __forceinline__ void gemm_tile(Smem<BT, TE>& sm, const double* A, int64_t lda, const double* B, int64_t ldb,
                                          double* C, int64_t ldc, int nk, int by, int bx, const SEGF& segf = SEGF()) {
  constexpr int SBN = TE + 16;     // row stride of the [k][n] image
  constexpr int FI = TE / 32;      // MFMA tiles per wave and dimension
  constexpr int NL = TE / 32;      // 16-byte staging loads per thread and operand
  constexpr int WS = TE / 2;       // wave sub-tile edge
  constexpr int BKR = 512 / TE;    // k-rows of the [k][n] image covered by one staging pass
  constexpr int A_BUF = TE * SA;
  constexpr int B_BUF = BT ? TE * SA : KB * SBN;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)by * TE, n0 = (int64_t)bx * TE;

  // global->register staging maps.  Addresses are (wave-uniform base in SGPRs) + (32-bit per-thread byte offset).
  const int ar = t >> 3, ac = (t & 7) * 2;                    // [row][k] images: rows ar + 32*i, k offset ac
  const int br = t / (TE / 2), bc = (t % (TE / 2)) * 2;       // [k][n] image: k rows br + BKR*i, col offset bc
  const unsigned voff_a = (unsigned)((ar * lda + ac) * 8);
  const unsigned voff_b = BT ? (unsigned)((ar * ldb + ac) * 8) : (unsigned)((br * ldb + bc) * 8);
  const char* const Abase = reinterpret_cast<const char*>(A + m0 * lda);
  const char* const Bbase = reinterpret_cast<const char*>(BT ? (B + n0 * ldb) : (B + n0));
  const int64_t a_rs = 32 * lda * 8;                          // bytes between staged A rows
  const int64_t b_rs = (BT ? 32 * ldb : BKR * ldb) * 8;
  const int64_t a_ks = (int64_t)KB * 8;                       // bytes per k-step
  const int64_t b_ks = (BT ? (int64_t)KB : (int64_t)KB * ldb) * 8;

  // Staging registers.  For the 128-tile they are NAMED scalars with the address chain written out: the equivalent
  // array form compiles to a measurably slower NN loop (IVAR 551 vs 534 ms at C4) -- hipcc scheduling lottery.
  double2 Pa0, Pa1, Pa2, Pa3, Pb0, Pb1, Pb2, Pb3;
  const int64_t b_rs2 = 2 * b_rs, b_rs3 = 3 * b_rs, a_rs2 = 2 * a_rs, a_rs3 = 3 * a_rs;
#define GPX_LD16(base_, voff_) (*reinterpret_cast<const double2*>((base_) + (voff_)))
#define GPX_GLOAD_AT(S_, ap_, bp_)                                                                    \
  do {                                                                                                \
    S_##a0 = GPX_LD16(ap_, voff_a);                                                                    \
    S_##a1 = GPX_LD16(ap_ + a_rs, voff_a);                                                             \
    if (NL == 4) {                                                                                    \
      S_##a2 = GPX_LD16(ap_ + a_rs2, voff_a);                                                          \
      S_##a3 = GPX_LD16(ap_ + a_rs3, voff_a);                                                          \
    }                                                                                                 \
    S_##b0 = GPX_LD16(bp_, voff_b);                                                                    \
    S_##b1 = GPX_LD16(bp_ + b_rs, voff_b);                                                             \
    if (NL == 4) {                                                                                    \
      S_##b2 = GPX_LD16(bp_ + b_rs2, voff_b);                                                          \
      S_##b3 = GPX_LD16(bp_ + b_rs3, voff_b);                                                          \
    }                                                                                                 \
  } while (0)
#define GPX_GLOAD(S_, kt_)                                                                            \
  do {                                                                                                \
    const char* ap_ = Abase + (int64_t)(kt_) * a_ks;                                                  \
    const char* bp_ = Bbase + (int64_t)(kt_) * b_ks;                                                  \
    GPX_GLOAD_AT(S_, ap_, bp_);                                                                        \
  } while (0)

  double* const sa_w = &sm.a[0][ar * SA + ac];
  double* const sb_w = BT ? &sm.b[0][ar * SA + ac] : &sm.b[0][br * SBN + bc];
  constexpr int B_WS = BT ? 32 * SA : BKR * SBN;
  // [row][k] images have an odd stride: rows are only 8-byte aligned -> two 8-byte stores per 16-byte register pair
#define GPX_ST2(p_, v_)  \
  do {                   \
    (p_)[0] = (v_).x;    \
    (p_)[1] = (v_).y;    \
  } while (0)
#define GPX_STB(p_, v_)                               \
  do {                                                \
    if (BT) {                                         \
      GPX_ST2(p_, v_);                                \
    } else {                                          \
      *reinterpret_cast<double2*>(p_) = (v_);         \
    }                                                 \
  } while (0)
#define GPX_SSTORE(S_, buf_)                                                                          \
  do {                                                                                                \
    double* aw_ = sa_w + (buf_) * A_BUF;                                                              \
    double* bw_ = sb_w + (buf_) * B_BUF;                                                              \
    GPX_ST2(aw_, S_##a0);                                                                              \
    GPX_ST2(aw_ + 32 * SA, S_##a1);                                                                    \
    if (NL == 4) {                                                                                    \
      GPX_ST2(aw_ + 64 * SA, S_##a2);                                                                  \
      GPX_ST2(aw_ + 96 * SA, S_##a3);                                                                  \
    }                                                                                                 \
    GPX_STB(bw_, S_##b0);                                                                              \
    GPX_STB(bw_ + B_WS, S_##b1);                                                                       \
    if (NL == 4) {                                                                                    \
      GPX_STB(bw_ + 2 * B_WS, S_##b2);                                                                 \
      GPX_STB(bw_ + 3 * B_WS, S_##b3);                                                                 \
    }                                                                                                 \
  } while (0)

  d4 acc[FI][FI];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int fr = lane & 15, fk = lane >> 4;
  const double* const as0 = &sm.a[0][(wm * WS + fr) * SA + fk];
  const double* const bs0 = BT ? &sm.b[0][(wn * WS + fr) * SA + fk] : &sm.b[0][fk * SBN + wn * WS + fr];
  {
    // Steady state, scheduled by hand (GPX_PIN pins it): the global loads of step kt+1 are issued first and are
    // only waited for after three quarters of this step's MFMAs (left alone, hipcc hoists the LDS store + barrier to
    // after the first 20 MFMAs, ~1 us after the loads were issued: a vmcnt stall on every step -- 13 % of the MFMA
    // pipe idle in SQ_VALU_MFMA_BUSY_CYCLES); the last quarter runs after the barrier from fragments already in
    // registers, so the barrier skew is covered too.  Every fragment read of a step precedes its barrier.
#define GPX_FRAGS(buf_, kk_, slot_)                                                                   \
  do {                                                                                                \
    const double* as = as0 + (buf_) * A_BUF;                                                          \
    const double* bs = bs0 + (buf_) * B_BUF;                                                          \
    _Pragma("unroll") for (int i = 0; i < FI; ++i) fa##slot_[i] = as[(i * 16) * SA + (kk_) * 4];      \
    _Pragma("unroll") for (int j = 0; j < FI; ++j)                                                    \
        fb##slot_[j] = BT ? bs[(j * 16) * SA + (kk_) * 4] : bs[((kk_) * 4) * SBN + j * 16];           \
  } while (0)
#define GPX_MFMAS(slot_)                                                                              \
  do {                                                                                                \
    _Pragma("unroll") for (int i = 0; i < FI; ++i) _Pragma("unroll") for (int j = 0; j < FI; ++j)     \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa##slot_[i], fb##slot_[j], acc[i][j], 0, 0, 0); \
  } while (0)
    // memory operations cannot cross this, and it consumes the accumulator every k-substep updates last: the LDS store
    // and the barrier stay behind the third substep's MFMAs, the fourth substep stays behind the barrier
#define GPX_PIN() asm volatile("" : "+v"(acc[FI - 1][FI - 1]) : : "memory")
    // segmented instantiations only: hipcc hoists most of the fourth substep's MFMAs above the barrier there (3 of 16 stayed
    // behind it: nothing left to cover the fragment reads, 60 instead of 72 TF/s) -- consuming EVERY accumulator keeps all 16
    // behind the pin.  The dense kernels keep their single-operand pin (their schedule is right as it is).
#define GPX_PIN_ALL()                                                                                            \
  do {                                                                                                           \
    if constexpr (SEGF::on && FI == 4)                                                                           \
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]),    \
                        "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]),    \
                        "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]),    \
                        "+v"(acc[3][3]) : : "memory");                                                          \
  } while (0)
    static_assert(KB == 16, "the hand schedule below is written for four k-substeps");
    double fa0[FI], fa1[FI], fa2[FI], fa3[FI], fb0[FI], fb1[FI], fb2[FI], fb3[FI];  // one slot per k-substep
    // one steady-state step: the loads of the NEXT step first, 48 of this step's MFMAs, LDS store + barrier, the fragment
    // reads of the next step underneath the remaining 16 MFMAs
#define GPX_STEP(LOADS_, buf_)                                                                        \
  do {                                                                                                \
    LOADS_;                                                                                           \
    GPX_MFMAS(0);                                                                                     \
    GPX_MFMAS(1);                                                                                     \
    GPX_MFMAS(2);                                                                                     \
    GPX_PIN();                                                                                        \
    GPX_SSTORE(P, (buf_) ^ 1);                                                                        \
    __syncthreads();                                                                                  \
    GPX_PIN();                                                                                        \
    GPX_FRAGS((buf_) ^ 1, 0, 0);                                                                      \
    GPX_FRAGS((buf_) ^ 1, 1, 1);                                                                      \
    GPX_FRAGS((buf_) ^ 1, 2, 2);                                                                      \
    GPX_PIN(); /* the reads above are issued BEFORE the last 16 MFMAs (hipcc would sink most of them below) */ \
    GPX_PIN_ALL();                                                                                    \
    GPX_MFMAS(3);                                                                                     \
    GPX_FRAGS((buf_) ^ 1, 3, 3);                                                                      \
  } while (0)
    if constexpr (SEGF::on) {
      // Segmented k range as a TWO-LEVEL loop: the inner loop over the k-steps of one segment is the dense loop (one basic
      // block, pointers advance by a constant), and the step that crosses into the next segment is the same step body once
      // more with the next segment's base pointers -- the software pipeline never drains between segments.  (With the
      // switch folded into ONE loop as a conditional, hipcc split the loop into several blocks and the kernel ran at 66
      // instead of 72 TF/s: twice the share of parked wave cycles in SQ_WAIT_ANY.)
      const char* ap = reinterpret_cast<const char*>(segf.a(0));
      const char* bp = reinterpret_cast<const char*>(segf.b(0));
      GPX_GLOAD_AT(P, ap, bp);
      GPX_SSTORE(P, 0);
      __syncthreads();
      GPX_FRAGS(0, 0, 0);
      GPX_FRAGS(0, 1, 1);
      GPX_FRAGS(0, 2, 2);
      GPX_FRAGS(0, 3, 3);
      int buf = 0;
      for (int sgi = 0; sgi < segf.nseg; ++sgi) {
        for (int j = 1; j < segf.seg_nk; ++j) {
          ap += a_ks;
          bp += b_ks;
          GPX_STEP(GPX_GLOAD_AT(P, ap, bp), buf);
          buf ^= 1;
        }
        if (sgi + 1 < segf.nseg) {
          ap = reinterpret_cast<const char*>(segf.a(sgi + 1));
          bp = reinterpret_cast<const char*>(segf.b(sgi + 1));
          GPX_STEP(GPX_GLOAD_AT(P, ap, bp), buf);
          buf ^= 1;
        }
      }
    } else {
    GPX_GLOAD(P, 0);
    GPX_SSTORE(P, 0);
    __syncthreads();
    // The loop is rotated: on entry to a step its first fragments are already in registers -- they were read right after
    // the previous barrier, underneath that step's last 16 MFMAs -- so no LDS latency is exposed at the top.
    GPX_FRAGS(0, 0, 0);
    GPX_FRAGS(0, 1, 1);
    GPX_FRAGS(0, 2, 2);
    GPX_FRAGS(0, 3, 3);
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
      const int buf = kt & 1;
      GPX_GLOAD(P, kt + 1);
      GPX_MFMAS(0);
      GPX_MFMAS(1);
      GPX_MFMAS(2);
      GPX_PIN();
      GPX_SSTORE(P, buf ^ 1);
      __syncthreads();
      GPX_PIN();
      GPX_FRAGS(buf ^ 1, 0, 0);
      GPX_FRAGS(buf ^ 1, 1, 1);
      GPX_FRAGS(buf ^ 1, 2, 2);
      GPX_PIN();  // the reads above are issued BEFORE the last 16 MFMAs (hipcc would sink most of them below)
      GPX_MFMAS(3);
      GPX_FRAGS(buf ^ 1, 3, 3);
    }
    }
    // last step: its fragments are loaded
    GPX_MFMAS(0);
    GPX_MFMAS(1);
    GPX_MFMAS(2);
    GPX_MFMAS(3);
#undef GPX_STEP
#undef GPX_FRAGS
#undef GPX_MFMAS
#undef GPX_PIN
#undef GPX_PIN_ALL
  }
#undef GPX_GLOAD
#undef GPX_GLOAD_AT
#undef GPX_LD16
#undef GPX_SSTORE
#undef GPX_ST2
#undef GPX_STB
#undef GPX_COMPUTE

  // epilogue: reg v of lane l -> C[(l>>4)+4v][l&15] within each 16x16 tile
  double* Cw = C + (m0 + wm * WS + fk) * ldc + n0 + wn * WS + fr;
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double* cp = Cw + (int64_t)(i * 16 + 4 * v) * ldc;
      if (ACC) {
        double cv[FI];
#pragma unroll
        for (int j = 0; j < FI; ++j) cv[j] = cp[j * 16];
#pragma unroll
        for (int j = 0; j < FI; ++j) cp[j * 16] = cv[j] - acc[i][j][v];
      } else {
#pragma unroll
        for (int j = 0; j < FI; ++j) cp[j * 16] = acc[i][j][v];
      }
    }
}

// one workgroup per tile; ids congruent mod 8 are assumed to share an XCD (true for the first wave of workgroups).
// TE == 128 launches may carry a TAIL: the last rows of C (from element row tail_m0) are cut into 64x64 tiles handled by
// the workgroups with the highest ids, i.e. the ones dispatched last.  All tiles of a launch take the same time, so a
// launch whose tile count is not a multiple of the resident workgroups ends with a mostly idle round; tiles with a
// quarter of the work shorten it.
// TRI (triangular-operand modes, see launch_gemm_tri) is a TEMPLATE parameter: the dense instantiations must stay exactly
// the hand-scheduled kernel -- with `tri` as a run-time argument the IVAR solve lost 2 % (490 -> 500 ms), hipcc's
// scheduling of the k-loop is that sensitive to what surrounds it.
template <bool BT, bool ACC, bool LOWER, int TE, int TRI>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(const double* A, int64_t lda, const double* B, int64_t ldb,
                                                          double* C, int64_t ldc, int nk, int tiles_m, int tiles_n,
                                                          int sb_cols, int sb_shift, int main_wgs, int tail_m0,
                                                          int tail_tn) {
  constexpr int tri = TRI;
  __shared__ Smem<BT, TE> sm;
  // (64-tile launches have no tail: tail_tn < 0 there marks a launch on the chain stream, gpx_chain_prio.  Compile-time guarded:
  // the 128-tile instantiations stay exactly the hand-scheduled kernel.)
  if (TE == 64 && tail_tn < 0) __builtin_amdgcn_s_setprio(3);
  const int w = blockIdx.x;
  if (TE == 128 && w >= main_wgs) {
    const int wt = w - main_wgs;
    const int by = wt / tail_tn, bx = wt - by * tail_tn;
    if (LOWER && 64 * bx > tail_m0 + 64 * by + 63) return;  // tile entirely above the diagonal
    int64_t k0 = 0;
    if (tri == 3) k0 = tail_m0 + 64 * by;  // both operands upper triangular: nothing below the row block's first column
    gemm_tile<BT, ACC, 64>(reinterpret_cast<Smem<BT, 64>&>(sm), A + (int64_t)tail_m0 * lda + k0, lda, B + k0, ldb,
                                 C + (int64_t)tail_m0 * ldc, ldc, nk - (int)(k0 / KB), by, bx);
    return;
  }
  const int xcd = w & 7, q = w >> 3;
  const int sbs2 = 2 * sb_shift;
  int by, bx;
  // (triangular operands: longest k range first inside a super-block, tile_of's rot 2 / 3 / 4)
  if (!tile_of<LOWER>((q >> sbs2) * 8 + xcd, q & ((1 << sbs2) - 1), tiles_m, tiles_n, sb_cols, sb_shift, &by, &bx,
                      tri == 2 ? 2 : (tri == 4 ? 3 : (tri == 1 ? 4 : 0))))
    return;
  // (Dealing single tiles of a triangular product to XCDs diagonally balances them too, but gives up the super-blocks'
  // operand reuse in L2: measured 11.9 ms against 13.1 dense for 28672 x 4096 x 4096 -- fabric-bound.)
  int nkt = nk;
  if (tri == 1) nkt = min(nk, (by + 1) * (TE / KB));
  if (tri == 2) nkt = min(nk, (bx + 1) * (TE / KB));
  if (tri == 3) {
    const int64_t k0 = (int64_t)by * TE;
    A += k0;
    B += k0;
    nkt = nk - by * (TE / KB);
  }
  if (tri == 4) {  // B (k x n, not transposed) lower triangular: column tile bx has nothing above row bx * TE
    const int64_t k0 = (int64_t)bx * TE;
    A += k0;
    B += k0 * ldb;
    nkt = nk - bx * (TE / KB);
  }
  gemm_tile<BT, ACC, TE>(sm, A, lda, B, ldb, C, ldc, nkt, by, bx);
}

// Batched form for many small independent products with constant strides (the block inverses of the triangular sweeps,
// chol.hip): blockIdx.y = batch entry, blockIdx.x = 64x64 tile of that entry's C, row-major.
template <bool BT, bool ACC>
__global__ __launch_bounds__(256, 2) void gemm_f64_batched_kernel(const double* A, int64_t lda, int64_t sa,
                                                                  const double* B, int64_t ldb, int64_t sb, double* C,
                                                                  int64_t ldc, int64_t sc, int nk, int tiles_n, int hiprio) {
  __shared__ Smem<BT, 64> sm;
  if (hiprio) __builtin_amdgcn_s_setprio(3);
  const int64_t b = blockIdx.y;
  const int by = blockIdx.x / tiles_n, bx = blockIdx.x - by * tiles_n;
  gemm_tile<BT, ACC, 64>(sm, A + b * sa, lda, B + b * sb, ldb, C + b * sc, ldc, nk, by, bx);
}

// Split-K form for a SMALL C under a LONG k range (FITC: nu x nu x N with nu = N / 8; the rows of a design batch against the
// kept factor): slice b of the k range (blockIdx.y) writes its own m x n partial product P_b = A[:, slice] B[:, slice]^T (row
// stride n, consecutive), and ksplit_sub_kernel subtracts the partials from C in slice order -- deterministic, unlike atomics.
// A 4096^2 lower C is 528 128-tiles for 512 resident workgroups (one full round + 16 tiles alone) or 2080 64-tiles, whose 8
// flop per operand byte out of L2 hold that kernel at 49 TF/s; four slices of 128-tiles: 2112 workgroups at 16 flop per byte.
template <bool LOWER>
__global__ __launch_bounds__(256, 2) void gemm_f64_ksplit_kernel(const double* A, int64_t lda, const double* B, int64_t ldb,
                                                                 double* P, int64_t n, int64_t sp, int nk_slice, int parts,
                                                                 int tiles_m, int tiles_n, int sb_cols, int sb_shift) {
  __shared__ Smem<true, 128> sm;
  // the jobs (super-block, slice) are dealt to the XCDs round-robin like the super-blocks of gemm_f64_kernel: with the plain
  // 2-D grid (tile = blockIdx.x, slice = blockIdx.y) XCD c owns the tile columns bx = c mod 8 of EVERY slice -- 80 lower tiles
  // for XCD 0, 52 for XCD 7 at 4096^2 -- and the launch took 15.0 ms where the 64-tile kernel takes 11.2
  const int w = blockIdx.x, xcd = w & 7, q = w >> 3, sbs2 = 2 * sb_shift;
  const int job = (q >> sbs2) * 8 + xcd, within = q & ((1 << sbs2) - 1);
  int sb, slice;
  if (LOWER) {   // all slices of the strictly lower super-blocks first, the diagonal ones (slots that exit at once) last
    const int nsr = (tiles_m + (1 << sb_shift) - 1) >> sb_shift, noff = nsr * (nsr - 1) / 2;
    if (job < parts * noff) {
      slice = job / noff;
      sb = job - slice * noff;
    } else {
      const int jj = job - parts * noff;
      slice = jj / nsr;
      sb = noff + jj - slice * nsr;
    }
  } else {
    const int nsb = ((tiles_m + (1 << sb_shift) - 1) >> sb_shift) * sb_cols;
    slice = job / nsb;
    sb = job - slice * nsb;
  }
  if (slice >= parts) return;
  int by, bx;
  if (!tile_of<LOWER>(sb, within, tiles_m, tiles_n, sb_cols, sb_shift, &by, &bx)) return;
  const int64_t k0 = (int64_t)slice * nk_slice * KB;
  gemm_tile<true, false, 128>(sm, A + k0, lda, B + k0, ldb, P + (int64_t)slice * sp, n, nk_slice, by, bx);
}

// C[i][j] -= sum_b P_b[i][j] (j <= i when lower; assign: C = the sum), partials added in slice order
__global__ __launch_bounds__(256) void ksplit_sub_kernel(const double* __restrict__ P, int64_t parts, int64_t m, int64_t n,
                                                         double* __restrict__ C, int64_t ldc, int lower, int assign) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= n || (lower && j > (i | 127))) return;   // (whole diagonal tiles: the partials hold them)
  double s = 0.0;
  for (int64_t b = 0; b < parts; ++b) s += P[(b * m + i) * n + j];
  C[i * ldc + j] = assign ? s : C[i * ldc + j] - s;
}

// ---- trailing update of the 2-D block-cyclic factorisation (dist.hip / gpexp_amd/dist.py) ------------------------------------
// C = the local matrix of rank (pr, pc) from local block (li_first, lj_first) on; local block (li, lj) is global block
// (I, J) = (li Pr + pr, lj Pc + pc).  For every tile with I > J (or I >= J), in ONE launch,
//     C_tile -= sum over segments s of  Arows_s(tile rows) * Brows_s(tile columns)^T
// where segment s is panel k_s in its packed buffer g[s] (layout: dist.hip, "piece p = [D | leaf inverses | rows]"):
//     rows of local block li      = block (li - li0[s][pr]) of piece pr            (contiguous over li)
//     rows of global block J      = block (J / Pr - li0[s][J % Pr]) of piece J % Pr
// with li0[s][p] = number of blocks I' <= k_s with I' % Pr == p.  Replaces one launch per panel and block column (K = nb,
// 64-tiles: 39 TF/s at C4 on one rank, profiles/r03_dist_w1_before.txt) by one launch per GROUP of panels over the whole
// local trailing matrix (K = nseg * nb).
constexpr int GPX_SC_MAX = 128;  // super-block columns (8 tiles each) one launch may span
struct Dist2Upd {
  const double* g[GPX_SEG_MAX];
  int li0[GPX_SEG_MAX][GPX_MAX_PR];
  int nseg, seg_nk;
  int Pr, Pc, pr, pc;
  int nb, gld;   // block size; row stride of the packed rows (gpx_g_ld)
  int li_first, lj_first;
  int below_diag;
  int64_t piece_stride, dsz;
  // Workgroup -> tile map.  The active tiles form a trapezoid (local block (li, lj) is active iff its global block row is on /
  // below its global block column); enumerating the enclosing rectangle and letting the rest exit deals the XCDs very unequal
  // shares -- super-blocks go to XCDs round-robin, and with 8 super-block columns each XCD owns ONE column of a staircase
  // (2 x 4 grid, rank 0: 992 tiles on XCD 0, 96 on XCD 7; measured 60 against 66 TF/s even on the 1 x 1 triangle).  So only the
  // super-blocks (8 x 8 tiles) that contain an active tile are enumerated, column by column: super-block column sc holds
  // rows sr_min[sc] .. nsr-1 and starts at index prefix[sc]; index s goes to XCD class s % 8.
  // Round 4: the FULLY active super-blocks first (all columns), the partial ones of the staircase last.  A partial super-block
  // has some of its 64 slots exit at once; the XCD refills them with tiles of its next super-block, and from then on its 64
  // resident workgroups belong to two super-blocks at different k phases -- the lock-step that lets them share operand panels
  // in L2 is gone for the rest of the launch (same effect, same cure as the SYRK map in tile_of<LOWER>).
  // Column sc: rows sr_min[sc] .. sr_full[sc]-1 are partial, sr_full[sc] .. nsr-1 full; prefix_f / prefix_p = running counts.
  int nsc, nsr, nfull;
  int sr_min[GPX_SC_MAX];
  int sr_full[GPX_SC_MAX];
  int prefix_f[GPX_SC_MAX + 1];
  int prefix_p[GPX_SC_MAX + 1];
};

template <int TE>
struct Dist2Seg {
  static constexpr bool on = true;
  const Dist2Upd& u;
  int nseg, seg_nk;
  int64_t arow;  // (li_first * nb + tile row) : row index inside piece pr counted from local block 0
  int pj;        // piece of this tile's column block
  int64_t brow;  // (J / Pr) * nb + column offset inside the block
  __device__ __forceinline__ const double* a(int s) const {
    return u.g[s] + (int64_t)u.pr * u.piece_stride + u.dsz + (arow - (int64_t)u.li0[s][u.pr] * u.nb) * u.gld;
  }
  __device__ __forceinline__ const double* b(int s) const {
    return u.g[s] + (int64_t)pj * u.piece_stride + u.dsz + (brow - (int64_t)u.li0[s][pj] * u.nb) * u.gld;
  }
};

template <int TE>
__global__ __launch_bounds__(256, 2) void dist2_update_kernel(const Dist2Upd u, double* C, int64_t ldc, int tiles_m,
                                                              int tiles_n) {
  __shared__ Smem<true, TE> sm;
  const int w = blockIdx.x;
  const int xcd = w & 7, q = w >> 3;
  const int s = (q >> 6) * 8 + xcd, within = q & 63;  // s-th active super-block, tile `within` of its 8 x 8
  if (s >= u.nfull + u.prefix_p[u.nsc]) return;
  const bool full = s < u.nfull;
  const int* pre = full ? u.prefix_f : u.prefix_p;
  const int sq = full ? s : s - u.nfull;
  int lo = 0, hi = u.nsc - 1;  // last column with pre[sc] <= sq  (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (pre[mid] <= sq) lo = mid; else hi = mid - 1;
  }
  const int sc = lo, sr = (full ? u.sr_full[sc] : u.sr_min[sc]) + (sq - pre[sc]);
  const int by = sr * 8 + (within >> 3), bx = sc * 8 + (within & 7);
  if (by >= tiles_m || bx >= tiles_n) return;
  const int tpb = u.nb / TE;  // tiles per block edge
  const int li = u.li_first + by / tpb, lj = u.lj_first + bx / tpb;
  const int I = li * u.Pr + u.pr, J = lj * u.Pc + u.pc;
  if (I < J || (u.below_diag && I == J)) return;
  Dist2Seg<TE> sg{u, u.nseg, u.seg_nk, (int64_t)u.li_first * u.nb + (int64_t)by * TE, J % u.Pr,
                  (int64_t)(J / u.Pr) * u.nb + (int64_t)(bx % tpb) * TE};
  gemm_tile<true, true, TE, Dist2Seg<TE>>(sm, nullptr, u.gld, nullptr, u.gld, C, ldc, u.nseg * u.seg_nk, by, bx, sg);
}

// (A persistent per-XCD work-queue variant of this kernel was measured in round 1 and removed: exact XCD placement
// alone does not bring the L2 hit rate back -- 4096x8192x16384: 51 GB fetched with or without it, 69 GB requested; the
// first wave of a launch shares panels because it starts in lock-step (70-81 % hit), later tiles drift apart by more
// than the ~16 k-steps a 4 MiB L2 can bridge (18-31 %) -- and the kernel is MFMA-bound either way.)

struct Plan {
  int te, tm, tn, sb_shift, sbc;
  int64_t nsb, wgs;
};

// largest super-block edge (8,4,2,1 tiles) that still leaves >= 16 super-blocks, i.e. >= 2 per XCD
Plan make_plan(int64_t m, int64_t n, bool lower, int te) {
  Plan p;
  p.te = te;
  p.tm = (int)(m / te);
  p.tn = (int)(n / te);
  int sbr = 0;
  for (p.sb_shift = 3;; --p.sb_shift) {
    const int e = 1 << p.sb_shift;
    sbr = (p.tm + e - 1) / e;
    p.sbc = (p.tn + e - 1) / e;
    p.nsb = lower ? (int64_t)sbr * (sbr + 1) / 2 : (int64_t)sbr * p.sbc;
    if (p.nsb >= 16 || p.sb_shift == 0) break;
  }
  p.wgs = (p.nsb + 7) / 8 * 8 * ((int64_t)1 << (2 * p.sb_shift));  // super-blocks dealt over 8 XCD classes
  return p;
}

}  // namespace

// C_b (m x n) = (accumulate ? C_b - A_b*op(B_b) : A_b*op(B_b)) for b = 0..batch-1 with element strides sa, sb, sc between
// consecutive entries; m, n multiples of 64, k of 16; C_b must not alias A_b or B_b
int launch_gemm_batched(gpx_ctx* ctx, const double* A, int64_t lda, int64_t sa, const double* B, int64_t ldb, int64_t sb,
                        double* C, int64_t ldc, int64_t sc, int64_t m, int64_t n, int64_t k, bool bt, bool accumulate,
                        int64_t batch) {
  if (m == 0 || n == 0 || batch == 0) return 0;
  GPX_ARG(m % 64 == 0 && n % 64 == 0 && k % KB == 0 && k > 0, "gemm_batched: m,n must be multiples of 64 and k of 16");
  GPX_ARG((lda % 2) == 0 && (ldb % 2) == 0 && batch <= 65535, "gemm_batched: even leading dimensions, at most 65535 entries");
  const int tn = (int)(n / 64);
  dim3 grid((unsigned)((m / 64) * tn), (unsigned)batch);
  const int nk = (int)(k / KB);
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * (double)m * (double)n * (double)k * (double)batch, 0.0);
#define GPX_B(BT_, ACC_)                                                                                          \
  hipLaunchKernelGGL((gemm_f64_batched_kernel<BT_, ACC_>), grid, dim3(256), 0, ctx->stream, A, lda, sa, B, ldb, sb, C, \
                     ldc, sc, nk, tn, gpx_chain_prio(ctx))
  if (bt) {
    if (accumulate) GPX_B(true, true); else GPX_B(true, false);
  } else {
    if (accumulate) GPX_B(false, true); else GPX_B(false, false);
  }
#undef GPX_B
  GPX_HIP(hipGetLastError());
  return 0;
}

// C -= A B^T (assign: C = A B^T) over a long k range as `parts` slices (see gemm_f64_ksplit_kernel); P: parts * m * n doubles
int launch_gemm_ksplit(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                       int64_t m, int64_t n, int64_t k, bool lower, int64_t parts, double* P, bool assign) {
  if (m == 0 || n == 0) return 0;
  GPX_ARG(P && parts > 0 && parts <= 65535 && m % 128 == 0 && n % 128 == 0 && k % (parts * KB) == 0 && k > 0,
          "gemm_ksplit: m,n multiples of 128, k a multiple of 16 * parts");
  GPX_ARG((lda % 2) == 0 && (ldb % 2) == 0 && (!lower || m == n) && m <= 65535, "gemm_ksplit: bad operands");
  const Plan p = make_plan(m, n, lower, 128);   // (super-block edge 8 / 4 / 2 tiles: within 3 % of each other at 4096^2 x 32768)
  const int64_t jobs = p.nsb * parts, wgs = (jobs + 7) / 8 * 8 * ((int64_t)1 << (2 * p.sb_shift));
  GPX_ARG(wgs < ((int64_t)1 << 31), "gemm_ksplit: grid too large");
  dim3 grid((unsigned)wgs);
  const int nks = (int)(k / parts / KB);
  {
    ProfScope ps(ctx, GPX_PROF_GEMM, (lower ? 1.0 : 2.0) * (double)m * (double)n * (double)k, 0.0);
    if (lower)
      hipLaunchKernelGGL((gemm_f64_ksplit_kernel<true>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb, P, n, m * n, nks,
                         (int)parts, p.tm, p.tn, p.sbc, p.sb_shift);
    else
      hipLaunchKernelGGL((gemm_f64_ksplit_kernel<false>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb, P, n, m * n, nks,
                         (int)parts, p.tm, p.tn, p.sbc, p.sb_shift);
  }
  hipLaunchKernelGGL(ksplit_sub_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream, P, parts, m, n,
                     C, ldc, lower ? 1 : 0, assign ? 1 : 0);
  GPX_HIP(hipGetLastError());
  return 0;
}

// The same for products too small for 128-tiles (a few rows against one block of a factor: m x n of 128 64-tiles under a serial
// k range of 1024 -- 55-60 us on half the chip): C = (assign ? 0 : C) -/+ A B^T as `parts` slices of the k range on 64-tiles
// (blockIdx.y = slice), the partials summed in slice order.  assign: C = A B^T (C may alias A: the partials are complete before
// the sum is written).  P: parts * m * n doubles.
int launch_gemm_ksplit_small(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                             int64_t m, int64_t n, int64_t k, bool assign, int64_t parts, double* P) {
  if (m == 0 || n == 0) return 0;
  GPX_ARG(P && parts > 0 && k % (parts * KB) == 0 && m <= 65535, "gemm_ksplit_small: bad arguments");
  const int64_t ks = k / parts;
  GPX_TRY(launch_gemm_batched(ctx, A, lda, ks, B, ldb, ks, P, n, m * n, m, n, ks, true, false, parts));
  hipLaunchKernelGGL(ksplit_sub_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)m), dim3(256), 0, ctx->stream, P, parts, m, n,
                     C, ldc, 0, assign ? 1 : 0);
  GPX_HIP(hipGetLastError());
  return 0;
}

int launch_gemm(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                int64_t m, int64_t n, int64_t k, bool bt, bool accumulate, bool lower) {
  return launch_gemm_tri(ctx, A, lda, B, ldb, C, ldc, m, n, k, bt, accumulate, lower, 0);
}

// tri: 0 = dense operands; 1 = A is lower triangular (k == m); 2 = B is a lower-triangular n x k matrix used transposed
// (bt, k == n); 4 = B is a lower-triangular k x n matrix used as it is (!bt, k == n: the k range of a column tile STARTS at
// its diagonal); 3 = lower C = U U^T with A = B = U upper triangular (bt, lower): the structurally zero part of the k range
// is skipped per tile
int launch_gemm_tri(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                    int64_t m, int64_t n, int64_t k, bool bt, bool accumulate, bool lower, int tri) {
  if (m == 0 || n == 0) return 0;
  GPX_ARG(tri == 0 || (!lower && ((tri == 1 && k == m) || (tri == 2 && bt && k == n) || (tri == 4 && !bt && k == n))) ||
              (tri == 3 && lower && bt && k == m),
          "gemm: bad triangular-operand mode");
  GPX_ARG(m % 128 == 0 && n % 128 == 0 && k % KB == 0 && k > 0, "gemm: m,n must be multiples of 128 and k of 16");
  GPX_ARG((lda % 2) == 0 && (ldb % 2) == 0, "gemm: leading dimensions must be even (16-byte loads)");
  GPX_ARG(!lower || m == n, "gemm: lower-only update needs a square C");
  // 64x64 tiles while the 128-tile count is below this (C4 potrf, round 1: 270 ms without, 251 ms at 256, 247 ms at 1024;
  // round 6, same question on today's factorisation: 256 / 512 -> +3.4 / +0.8 ms, profiles/r06_potrf_variants.txt)
  constexpr int small_max = 1024;
  const double tiles128 =
      lower ? 0.5 * (double)(m / 128) * ((double)(m / 128) + 1.0) : (double)(m / 128) * (double)(n / 128);
  // In-place leaf products (C aliases A or B, n or m == 128) are race-free only when ONE workgroup's tile spans the
  // whole 128-wide leaf: every read of the aliased operand then precedes that workgroup's own stores.
  const bool aliased = (C == A) || (C == B);
  const int te = (!aliased && tiles128 < (double)small_max) ? 64 : 128;
  // tail: the last `tr` 128-row bands of C as 64x64 tiles (see the kernel), about one chip-load of them.  Used for the
  // triangular (SYRK) launches only: their tile counts are never a multiple of the 512 resident workgroups (8192^2: 2080
  // tiles = 4.06 rounds, 53 -> 55 TF/s with the tail; 16384^2: 65.2 -> 66.2).  Rectangular launches measured 1-4 % SLOWER
  // with it: their tile counts divide evenly, and what looks like a drain there is the clock/fabric ramp after a stretch
  // of small kernels (the same launch repeated back to back goes 4.86 -> 3.84 ms over ~30 ms: round-2 probe, profiles/r02_*).
  constexpr int tail_target = 512;  // 64-tiles wanted in the tail
  int64_t tr = 0;
  if (te == 128 && lower && !aliased && tail_target > 0 && m >= 512) {
    const int64_t tm = m / 128, tn = n / 128;
    int64_t t64 = 0;
    while (tr < tm / 2 && t64 < tail_target) {
      ++tr;
      const int64_t band = tm - tr;  // 0-based index of the band added: 2 rows of 64-tiles
      t64 += lower ? (2 * band + 1) + (2 * band + 2) : 4 * tn;
    }
  }
  const int64_t m_main = m - 128 * tr;
  const Plan p = make_plan(m_main, n, lower, te);
  const int64_t tail_tn = lower ? m / 64 : n / 64;
  const int64_t tail_wgs = 2 * tr * tail_tn;
  GPX_ARG(p.wgs + tail_wgs < ((int64_t)1 << 31), "gemm: grid too large");
  dim3 grid((unsigned)(p.wgs + tail_wgs));
  const int nk = (int)(k / KB);
  const int hiprio = gpx_chain_prio(ctx);
  // algorithmic flops: a triangular operand halves the k range on average (+ the diagonal blocks)
  const double kflops = tri == 3 ? (double)k * (2.0 / 3.0) : (tri ? 0.5 * (double)k + 64.0 : (double)k);
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * tiles128 * 128.0 * 128.0 * kflops, 0.0);
#define GPX_KT(BT_, ACC_, LOW_, TE_, TRI_)                                                                          \
  hipLaunchKernelGGL((gemm_f64_kernel<BT_, ACC_, LOW_, TE_, TRI_>), grid, dim3(256), 0, ctx->stream, A, lda, B, ldb, C, \
                     ldc, nk, p.tm, p.tn, p.sbc, p.sb_shift, (int)p.wgs, (int)m_main, (TE_ == 64 && hiprio) ? -1 : (int)tail_tn)
  if (tri != 0) {  // the operand forms the explicit inverses use
#define GPX_GT(BT_, ACC_, LOW_, TRI_)        \
  do {                                       \
    if (te == 64)                            \
      GPX_KT(BT_, ACC_, LOW_, 64, TRI_);     \
    else                                     \
      GPX_KT(BT_, ACC_, LOW_, 128, TRI_);    \
  } while (0)
    if (tri == 1 && !bt && !accumulate) GPX_GT(false, false, false, 1);
    else if (tri == 1 && !bt && accumulate) GPX_GT(false, true, false, 1);
    else if (tri == 2 && !accumulate) GPX_GT(true, false, false, 2);
    else if (tri == 3 && !accumulate) GPX_GT(true, false, true, 3);
    else if (tri == 4 && !accumulate) GPX_GT(false, false, false, 4);
    else {
      gpx_set_error("gemm: triangular-operand mode %d is not instantiated for bt=%d accumulate=%d", tri, (int)bt, (int)accumulate);
      return -1;
    }
#undef GPX_GT
    GPX_HIP(hipGetLastError());
    return 0;
  }
#define GPX_K(BT_, ACC_, LOW_, TE_) GPX_KT(BT_, ACC_, LOW_, TE_, 0)
#define GPX_G(BT_, ACC_, LOW_)          \
  do {                                  \
    if (te == 64)                       \
      GPX_K(BT_, ACC_, LOW_, 64);       \
    else                                \
      GPX_K(BT_, ACC_, LOW_, 128);      \
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
#undef GPX_K
#undef GPX_KT
  GPX_HIP(hipGetLastError());
  return 0;
}

// A[lr0 : lr0+m, lc0 : lc0+n] (local matrix, rank (pr, pc) of a Pr x Pc grid, block size nb) -= the contributions of the nseg
// panels ks[s] held in the packed buffers g[s], for the local blocks on / below (below_diag: strictly below) the global
// diagonal.  See dist2_update_kernel.  m, n multiples of 128, lr0 and lc0 multiples of nb.
int launch_dist2_update(gpx_ctx* ctx, double* C, int64_t ldc, int64_t lr0, int64_t m, int64_t lc0, int64_t n, int64_t nb, int Pr,
                        int Pc, int pr, int pc, int64_t piece_stride, int64_t dsz, int nseg, const double* const* g,
                        const int64_t* ks, int below_diag) {
  if (m == 0 || n == 0 || nseg == 0) return 0;
  GPX_ARG(nseg > 0 && nseg <= GPX_SEG_MAX && Pr >= 1 && Pr <= GPX_MAX_PR && Pc >= 1, "dist2 update: too many segments / grid rows");
  GPX_ARG(m % 128 == 0 && n % 128 == 0 && nb % 128 == 0 && lr0 % nb == 0 && lc0 % nb == 0 && (ldc % 2) == 0, "dist2 update: alignment");
  Dist2Upd u;
  for (int s = 0; s < nseg; ++s) {
    u.g[s] = g[s];
    for (int p = 0; p < Pr; ++p) u.li0[s][p] = (int)(ks[s] + 1 <= p ? 0 : (ks[s] - p) / Pr + 1);  // blocks I' <= k with I' % Pr == p
  }
  u.nseg = nseg;
  u.seg_nk = (int)(nb / KB);
  u.Pr = Pr; u.Pc = Pc; u.pr = pr; u.pc = pc;
  u.nb = (int)nb;
  u.gld = (int)gpx_g_ld(nb);
  u.li_first = (int)(lr0 / nb);
  u.lj_first = (int)(lc0 / nb);
  u.below_diag = below_diag;
  u.piece_stride = piece_stride;
  u.dsz = dsz;
  // work estimate (tiles on / below the diagonal) for the tile-size choice and the profile
  double tiles128 = 0.0;
  for (int64_t lj = lc0 / nb; lj * nb < lc0 + n; ++lj) {
    const int64_t J = lj * Pc + pc;
    const int64_t cw = (lc0 + n - lj * nb) < nb ? (lc0 + n - lj * nb) : nb;
    for (int64_t li = lr0 / nb; li * nb < lr0 + m; ++li) {
      const int64_t I = li * Pr + pr;
      if (I < J || (below_diag && I == J)) continue;
      const int64_t rh = (lr0 + m - li * nb) < nb ? (lr0 + m - li * nb) : nb;
      tiles128 += (double)(rh / 128) * (double)(cw / 128);
    }
  }
  if (tiles128 == 0.0) return 0;
  constexpr int small_max = 512;  // 64-tiles while fewer 128-tiles than this have work
  const int te = tiles128 < (double)small_max ? 64 : 128;
  const int tm = (int)(m / te), tn = (int)(n / te), tpb = (int)(nb / te);
  u.nsc = (tn + 7) / 8;
  u.nsr = (tm + 7) / 8;
  GPX_ARG(u.nsc <= GPX_SC_MAX, "dist2 update: too many column super-blocks in one launch");
  int64_t nfull = 0, npart = 0;
  for (int sc = 0; sc < u.nsc; ++sc) {
    // first active local block row of this super-block column: the column's LEFTMOST block has the lowest global index
    const int64_t lj = lc0 / nb + (8 * sc) / tpb, J = lj * Pc + pc;
    int64_t li_min = (J + (below_diag ? 1 : 0) - pr + Pr - 1) / Pr;  // smallest li with li Pr + pr >= J (+1)
    if (J + (below_diag ? 1 : 0) - pr < 0) li_min = 0;
    int64_t by_min = (li_min - lr0 / nb) * tpb;   // first tile row (relative to the launch) that can be active
    if (by_min < 0) by_min = 0;
    int srm = (int)(by_min / 8);
    if (srm > u.nsr) srm = u.nsr;
    u.sr_min[sc] = srm;
    // first super-block row whose 8 x 8 tiles are ALL active: its top tile row against the column's RIGHTMOST block
    int64_t bx_right = 8 * (int64_t)sc + 7;
    if (bx_right > tn - 1) bx_right = tn - 1;
    const int64_t ljr = lc0 / nb + bx_right / tpb, Jr = ljr * Pc + pc;
    int64_t li_full = (Jr + (below_diag ? 1 : 0) - pr + Pr - 1) / Pr;   // smallest li with li Pr + pr >= Jr (+1)
    if (Jr + (below_diag ? 1 : 0) - pr < 0) li_full = 0;
    int64_t by_full = (li_full - lr0 / nb) * tpb;                        // first tile row with every column of the super-block active
    if (by_full < 0) by_full = 0;
    int srf = (int)((by_full + 7) / 8);
    if (srf < srm) srf = srm;
    if (srf > u.nsr) srf = u.nsr;
    u.sr_full[sc] = srf;
    u.prefix_p[sc] = (int)npart;
    u.prefix_f[sc] = (int)nfull;
    npart += srf - srm;
    nfull += u.nsr - srf;
  }
  u.prefix_p[u.nsc] = (int)npart;
  u.prefix_f[u.nsc] = (int)nfull;
  u.nfull = (int)nfull;
  const int64_t nsb = nfull + npart;
  if (nsb == 0) return 0;
  const int64_t wgs = (nsb + 7) / 8 * 8 * 64;
  GPX_ARG(wgs < ((int64_t)1 << 31), "dist2 update: grid too large");
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * tiles128 * 128.0 * 128.0 * (double)nb * nseg, 0.0);
  double* Cb = C + lr0 * ldc + lc0;
  if (te == 64)
    hipLaunchKernelGGL(dist2_update_kernel<64>, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, u, Cb, ldc, tm, tn);
  else
    hipLaunchKernelGGL(dist2_update_kernel<128>, dim3((unsigned)wgs), dim3(256), 0, ctx->stream, u, Cb, ldc, tm, tn);
  GPX_HIP(hipGetLastError());
  return 0;
}
