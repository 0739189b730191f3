// Cholesky factorisation, triangular solves and log-determinant for gfx950.
//
// Replaces numpy.linalg.pinv (SVD, gp.py:181,400; experimentalDesign.py:268,280,826) and
// numpy.linalg.slogdet (LU, gp.py:434) by K = L L^T.  From N = 8192 the factorisation is BLOCKED RIGHT-LOOKING with 4096-wide
// panels and one panel of look-ahead (potrf_blocked: panel solve through explicit 1024-order block inverses, trailing SYRK with
// K = 4096, the next diagonal block's chain on the high-priority side stream underneath the bulk of the update); below that,
// and inside every diagonal block, it is recursive (A11 -> L11; A21 <- A21 L11^-T; A22 <- A22 - L21 L21^T; recurse on A22) so
// that every flop outside the 128x128 leaves lands in the MFMA GEMM kernel with the largest possible inner dimension, which
// keeps the read-modify-write traffic on C far below the HBM roof.  Leaves: one workgroup factors a 128x128 diagonal block in
// LDS and also inverts it, so that all triangular solves against a leaf are GEMMs with the inverse.
#include "gpx_internal.h"
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

namespace {

constexpr int NB = GPX_TILE;  // 128
constexpr int LS = NB + 1;    // LDS row stride (conflict-free column walks)

// ---- leaf: potf2 + trtri of one 128x128 diagonal block -------------------------------------------
// Everything is done in 16x16 blocks held in LDS (S, row stride 129: odd, so the 16-row fragment reads of a
// 16-lane group are bank-conflict free).  Per 16-column panel p:
//   (1) wave 0 factors the diagonal block in registers (lane = row, pivots/columns broadcast with v_readlane)
//       and inverts it by forward substitution (lane = column of the inverse)         -> T[p]
//   (2) the blocks below it are multiplied by T[p]^T              (4 fp64 MFMAs per block, in place)
//   (3) the trailing blocks get  C -= A B^T                        (4 fp64 MFMAs per block)
// The inverse of the whole 128x128 factor is then built block column by block column, one wave per column,
// with every intermediate X block kept in MFMA accumulator registers: the C/D layout of
// v_mfma_f64_16x16x4_f64 (reg v of lane l = row (l>>4)+4v, col l&15) is exactly its B-operand layout for
// k-step v, so an accumulator feeds the next product without touching LDS.
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_d(double v, int srclane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

constexpr int LB = 16;          // block edge
constexpr int TS17 = LB + 1;    // row stride inside a 16 x 16 block (odd: conflict-free fragment reads)
constexpr int BSZ = LB * TS17;  // doubles per stored block
constexpr int LEAF_CB = LB * LB + 64 + LB;  // leaf_diag_fast's broadcast buffer

// LDS image of the leaf (round 3): only the 36 blocks on / below the diagonal, block (I, J) at sb_off(I, J), element (r, c)
// at r * 17 + c -- 78 336 bytes instead of the 149 KB of the full 128 x 129 image + 8 inverse blocks.  That is what lets
// the leaf START beside a chip-filling GEMM: a GEMM workgroup holds 68 KB of the CU's 160 KB of LDS, so a kernel that needs
// less than ~90 KB gets a slot whenever ONE GEMM workgroup retires (every microsecond somewhere on the chip), while the old
// leaf needed a whole CU and waited for the END of the big kernel (scripts/dispatch_check2.hip: 89 KB runs as if alone, 92 KB
// waits for the end) -- which is why round 2 reserved four CUs per XCD for the diagonal chain.  The diagonal block (p, p)
// holds A_pp until it is factored; L_pp then goes straight to global memory and the slot is reused for T_p = L_pp^-1.
__device__ __forceinline__ constexpr int sb_off(int I, int J) { return (I * (I + 1) / 2 + J) * BSZ; }

// (1) of the leaf, GENERAL form: factor the 16x16 diagonal block p (LDS block D = S + sb_off(p, p)) and invert the factor, one
// wave.  Since round 5 this is the RARE path -- leaf_diag_fast below factors every block whose pivots are sound -- so it is
// written for few registers, not for speed (rounds 1-4 kept rows and inverse columns in 64 + 32 VGPRs and set the whole
// kernel's allocation): rows and inverse columns stay in LDS (D and the 16 x 16 scratch X: X[c][q] at X[16 c + q]), lane q of
// every 16-lane row works on row q / column q (the four rows redundantly: same values to the same addresses), the loop over the
// pivots is not unrolled.  Column j of L_pp is final after step j: it goes to global memory (Ag = the block's first element
// there, zeros above the diagonal), and row j of the inverse T_p replaces row j of the block in LDS (dead by then).
//
// Pivot policy (piv_min, skip): a pivot <= piv_min is "bad".  skip == 0: it is replaced by 1.0 and its global index is
// reported (first one wins) -- the caller sees a non-positive-definite matrix.  skip != 0 (the rank-deficient retry of
// GP._factor): the point is DROPPED -- L_jj = 1, the rest of column j and row j of the inverse are 0, so every solve
// against the factor returns 0 in that component, exactly as if the point were not in the training set; this is what
// numpy.linalg.pinv (gp.py:181) makes of an exactly duplicated point.  Dropped pivots are counted in info[1].
__device__ __forceinline__ void leaf_diag(double* __restrict__ D, double* __restrict__ Ag, int64_t ld, int c0, int q, int lane,
                                       int64_t base_index, int64_t n_valid, int* __restrict__ info, double piv_min, int skip,
                                       double* __restrict__ X) {
#pragma unroll 1
  for (int c = 0; c < LB; ++c) X[LB * c + q] = (c == q) ? 1.0 : 0.0;
  int first_bad = LB, nbad = 0;  // wave-uniform
  double* const grow = Ag + (int64_t)q * ld;
#pragma unroll 1
  for (int j = 0; j < LB; ++j) {
    double piv = D[j * TS17 + j];
    const bool ok = piv > piv_min;
    first_bad = (!ok && first_bad == LB) ? j : first_bad;
    nbad += (!ok && base_index + c0 + j < n_valid) ? 1 : 0;
    piv = ok ? piv : 1.0;
    const double y = __builtin_amdgcn_rsq(piv);
    double g = piv * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, piv), h, g);                 // sqrt(piv)
    const double rs = (ok || !skip) ? h + h : 0.0;  // 1/sqrt(piv); 0 drops the point (see the pivot policy above)
    const double aj = (q == j) ? g : ((q > j) ? D[q * TS17 + j] * rs : 0.0);
    const double xj = X[LB * j + q] * rs;
    grow[j] = aj;
    if (q > j) D[q * TS17 + j] = aj;  // the scaled column, for the other rows' updates below
#pragma unroll 1
    for (int c = j + 1; c < LB; ++c) {
      const double l = D[c * TS17 + j];  // L[c][j]: written above by lane c of this wave (LDS operations execute in order)
      if (q >= c) D[q * TS17 + c] = fma(-l, aj, D[q * TS17 + c]);
      X[LB * c + q] = fma(-l, xj, X[LB * c + q]);
    }
    D[j * TS17 + q] = xj;  // row j of the inverse over row j of the block: (j, c < j) and (j, j) are dead, (j, c > j) never read
  }
  if (first_bad < LB && lane == 0 && base_index + c0 + first_bad < n_valid) {
    if (skip)
      atomicAdd(info + 1, nbad);
    else
      atomicCAS(info, 0, (int)(base_index + c0 + first_bad + 1));
  }
}

// (1'), round 5: the same 16x16 step in ONE THIRD of the instructions.  A single wave on a SIMD issues an fp64 VALU instruction
// every ~8.6 cycles whether it depends on the previous one or not (profiles/r01_dfma_peak.txt: 8.65 cycles per v_fma_f64 at one
// wave per SIMD and eight independent chains), so leaf_diag's 390 cycles per pivot were its ~40 instructions per pivot, not
// its dependency chain.  What is removed:
//   * the four 16-lane rows no longer work redundantly: lanes 0..15 hold the rows of the block (v[c] = A[q][c]), lanes 32..47
//     the columns of the inverse (v[c] = Y[c][q]), so ONE v_fma_f64 per (pivot, column) updates both (there were two);
//   * the column entries L[c][j] reach every lane as BROADCAST READS of a small LDS buffer (one ds_read_b128 per two columns,
//     issued beside the VALU) instead of one DPP move each; only the pivot and the entry the next pivot waits for come
//     through v_readlane;
//   * the square roots leave the loop: the elimination runs in LDL^T form (only w = 1/d_j on the chain: v_rcp_f64 + two
//     Newton steps), every lane keeps ITS pivot d_q, and 1/sqrt(d_q) is formed once, lane-parallel, after the last step;
//     L[q][j] = a_qj / sqrt(d_j), T[j][q] = Y[j][q] / sqrt(d_j);
//   * no selects for the pivot policy: the block is factored as if it were positive definite and NOTHING is stored until the
//     pivots have been checked; a pivot <= piv_min (or NaN) sends the caller to leaf_diag, which re-reads the untouched block.
// CB (LEAF_CB doubles): 16 x 16 (column j of the unscaled factor at CB[16 j + row]) + 64 junk slots (the lanes that hold no row
// write there) + 16 for the 1/sqrt broadcast; touched by wave 0 only -- LDS operations of one wave execute in order, no barrier.
__device__ __forceinline__ bool leaf_diag_fast(double* __restrict__ D, double* __restrict__ Ag, int64_t ld, int q, int lane,
                                               double piv_min, double* __restrict__ CB) {
  const bool rowlane = lane < LB, collane = (lane >> 4) == 2;
  double v[LB];
#pragma unroll
  for (int c = 0; c < LB; ++c) v[c] = rowlane ? ((c <= q) ? D[q * TS17 + c] : 0.0) : ((collane && c == q) ? 1.0 : 0.0);
  double* const cbw = rowlane ? CB + q : CB + LB * LB + lane;  // + 16 j per column
  const double2* const cbr = reinterpret_cast<const double2*>(CB);
  double ntz_prev = 0.0;
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    // the entries of column j-1 for the far updates of pivot j-1: all reads in flight before the first use (one wait, not eight)
    double2 u[LB / 2];
    if (j > 0) {
#pragma unroll
      for (int c2 = (j + 1) / 2; c2 < LB / 2; ++c2) u[c2] = cbr[(LB / 2) * (j - 1) + c2];
    }
    __builtin_amdgcn_sched_barrier(0);
    // v[j] is final: pivot, the entry below it, and the column out for the broadcast reads of the NEXT iteration
    const double piv = readlane_d(v[j], j);
    const double unext = readlane_d(v[j], j + 1 < LB ? j + 1 : j);
    cbw[LB * j] = v[j];
    double w = __builtin_amdgcn_rcp(piv);  // ~2^-24 relative (ISA manual: 2^29 ulp); w (1 + e + e^2) leaves e^3 = 2^-72
    double e = fma(-piv, w, 1.0);
    e = fma(e, e, e);
    w = fma(w, e, w);
    const double ntz = -(v[j] * w);  // rows: -a_qj / d_j;  inverse columns: -Y[j][q] / d_j
    if (j + 1 < LB) v[j + 1] = fma(ntz, unext, v[j + 1]);
    __builtin_amdgcn_sched_barrier(0);
    // the far updates of pivot j-1 (columns j+1 ..; column j got it as that iteration's `unext` update)
    if (j > 0) {
#pragma unroll
      for (int c2 = (j + 1) / 2; c2 < LB / 2; ++c2) {
        if (2 * c2 >= j + 1) v[2 * c2] = fma(ntz_prev, u[c2].x, v[2 * c2]);
        if (2 * c2 + 1 >= j + 1) v[2 * c2 + 1] = fma(ntz_prev, u[c2].y, v[2 * c2 + 1]);
      }
    }
    ntz_prev = ntz;
  }
  // every row lane's pivot d_q went into the broadcast buffer at step q; all of them must be sound before anything is stored
  const double dq = CB[(LB + 1) * q];
  if (__builtin_amdgcn_ballot_w64(!(dq > piv_min)) != 0) return false;
  double rs;
  {
    const double y = __builtin_amdgcn_rsq(dq);
    double g = dq * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, dq), h, g);  // sqrt(d_q)
    r = fma(-g, h + h, 1.0);        // 1/sqrt(d_q) corrected against the corrected root
    rs = fma(h + h, r, h + h);
  }
  CB[LB * LB + 64 + q] = rs;  // (the four lanes of a column write the same value)
  double2 r2[LB / 2];  // all eight broadcast reads in flight at once
  const double2* const rsr = reinterpret_cast<const double2*>(CB + LB * LB + 64);
#pragma unroll
  for (int c2 = 0; c2 < LB / 2; ++c2) r2[c2] = rsr[c2];
#pragma unroll
  for (int c2 = 0; c2 < LB / 2; ++c2) {  // zero above the diagonal by construction
    v[2 * c2] *= r2[c2].x;
    v[2 * c2 + 1] *= r2[c2].y;
  }
  if (rowlane) {
    double* const grow = Ag + (int64_t)q * ld;
#pragma unroll
    for (int c2 = 0; c2 < LB / 2; ++c2) *reinterpret_cast<double2*>(grow + 2 * c2) = double2{v[2 * c2], v[2 * c2 + 1]};
  }
  if (collane) {
#pragma unroll
    for (int c = 0; c < LB; ++c) D[c * TS17 + q] = v[c];
  }
  return true;
}

// The 28 blocks (I,K), 1 <= K <= I <= 7, of the trailing matrix live in REGISTERS (MFMA accumulator layout) from the
// first step to the step that makes their block column current (column-major enumeration below).
struct LeafBlk { int I, K; };
__device__ constexpr int kLeafI[28] = {1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};
__device__ constexpr int kLeafK[28] = {1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7};
__device__ __forceinline__ constexpr LeafBlk leaf_blk(int b) {  // b-th block, columns K = 1..7 holding rows I = K..7
  return LeafBlk{kLeafI[b], kLeafK[b]};
}

// Ownership with look-ahead: wave 0 only factors diagonal blocks; waves 1..3 own the 28 blocks, block 3s + (W-1) of the
// column-major enumeration in slot s (10 / 9 / 9 slots), so a wave's block column never decreases with s.
constexpr int leaf_nslots(int W) { return (28 - (W - 1) + 2) / 3; }
constexpr int leaf_first_slot(int W, int kmin) {  // first slot of wave W whose block column is >= kmin
  int s = 0;
  while (s < leaf_nslots(W) && kLeafK[3 * s + (W - 1)] < kmin) ++s;
  return s;
}

// C(slot) -= B_I B_K^T for slots [S0, S1) of wave W, from the scaled block column P; the MFMAs of different slots
// interleave (independent accumulators)
template <int W, int P, int S0, int S1, int NS>
__device__ __forceinline__ void leaf_update(const double* __restrict__ S, d4 (&blk)[NS], int g, int q) {
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
    for (int s = S0; s < S1; ++s) {
      const LeafBlk bk = leaf_blk(3 * s + (W - 1));
      const double a = -S[sb_off(bk.I, P) + q * TS17 + 4 * s4 + g];
      const double b = S[sb_off(bk.K, P) + q * TS17 + 4 * s4 + g];
      blk[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, blk[s], 0, 0, 0);
    }
  }
}

// step P's update split in two: PRIORITY = the blocks of column P+1 (they gate the next diagonal factor) and DEFERRED =
// everything to the right of it, which waves 1..3 run underneath wave 0's factorisation of diagonal block P+1
template <int W, int P, int NS>
__device__ __forceinline__ void leaf_update_priority(double* __restrict__ S, d4 (&blk)[NS], int g, int q) {
  constexpr int S0 = leaf_first_slot(W, P + 1), S1 = leaf_first_slot(W, P + 2);
  leaf_update<W, P, S0, S1, NS>(S, blk, g, q);
#pragma unroll
  for (int s = S0; s < S1; ++s) {  // column P+1 becomes current: back to LDS for the diagonal factor and the scaling
    const LeafBlk bk = leaf_blk(3 * s + (W - 1));
#pragma unroll
    for (int v = 0; v < 4; ++v) S[sb_off(bk.I, bk.K) + (g + 4 * v) * TS17 + q] = blk[s][v];
  }
}
template <int W, int P, int NS>
__device__ __forceinline__ void leaf_update_deferred(const double* __restrict__ S, d4 (&blk)[NS], int g, int q) {
  leaf_update<W, P, leaf_first_slot(W, P + 2), NS, NS>(S, blk, g, q);
}

// (2) of a step: the blocks below the diagonal of column p are multiplied by T[p]^T (all four waves)
__device__ __forceinline__ void leaf_scale(double* __restrict__ S, int p, int wave, int g, int q) {
  const double* Tp = S + sb_off(p, p);
  for (int I = p + 1 + wave; I < 8; I += 4) {
    double* B = S + sb_off(I, p);
    double af[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) af[s4] = B[q * TS17 + 4 * s4 + g];
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s4], Tp[q * TS17 + 4 * s4 + g], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) B[(g + 4 * v) * TS17 + q] = acc[v];
  }
}

// ---- the 128-order inverse, built ROW BLOCK BY ROW BLOCK underneath the factorisation (round 5) ---------------------------
// X = L^-1: X[R][K] = -T_R * sum_{J=K}^{R-1} L[R][J] X[J][K] (K < R), X[R][R] = T_R.  Rounds 1-4 built X column by column AFTER
// the factorisation (4.9 us of the leaf's 37).  But waves 1..3 idle while wave 0 factors a diagonal block (1.8 us per step), and
// everything row R needs but T_R is final one step earlier.  So each of waves 1..3 owns block columns of X (balanced over the
// last steps, where the rows are long: {0,5,7} / {1,4} / {2,3,6}), keeps them in registers in MFMA accumulator layout (= the
// B-operand layout of the next product), and inside the window of step P -- while wave 0 factors block P --
//   (a) finishes row P-1:   X[P-1][K] = -T_{P-1} * acc_K      (T_{P-1} has been in LDS since barrier [B] of step P-1)
//   (b) forms row P's sums: acc_K = sum_J L[P][J] X[J][K]     (block row P of L is final since step P-1)
// and stores every block of X as it completes.  Left for the end: row 7's finish, 4 MFMAs per column.
__device__ constexpr int inv_owner(int K) { return (K == 0 || K == 5 || K == 7) ? 1 : ((K == 1 || K == 4) ? 2 : 3); }

template <int W, int R>
__device__ __forceinline__ void inv_finish_row(const double* __restrict__ S, double* __restrict__ inv, d4 (&xb)[8][8],
                                               d4 (&acc)[8], int g, int q) {
#pragma unroll
  for (int K = 0; K <= R; ++K) {
    if (inv_owner(K) != W) continue;
    d4 r;
    if (K == R) {
#pragma unroll
      for (int v = 0; v < 4; ++v) r[v] = S[sb_off(R, R) + (g + 4 * v) * TS17 + q];
    } else {
      r = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double a = -S[sb_off(R, R) + q * TS17 + 4 * s4 + g];
        r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[K][s4], r, 0, 0, 0);
      }
    }
    xb[K][R - K] = r;
#pragma unroll
    for (int v = 0; v < 4; ++v) inv[(LB * R + g + 4 * v) * NB + LB * K + q] = r[v];
  }
}

template <int W, int P>
__device__ __forceinline__ void inv_row_sums(const double* __restrict__ S, d4 (&xb)[8][8], d4 (&acc)[8], int g, int q) {
  // the products of one J share the A fragments of L[P][J]; the accumulators of different K interleave
#pragma unroll
  for (int K = 0; K < P; ++K)
    if (inv_owner(K) == W) acc[K] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int J = 0; J < P; ++J) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double a = S[sb_off(P, J) + q * TS17 + 4 * s4 + g];
#pragma unroll
      for (int K = 0; K <= J; ++K)
        if (inv_owner(K) == W) acc[K] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[K][J - K][s4], acc[K], 0, 0, 0);
    }
  }
}

// block column PC of L (blocks (I, PC), I > PC: final once scaled) from LDS to global memory, by the 192 threads of waves 1..3
template <int PC>
__device__ __forceinline__ void leaf_store_column(const double* __restrict__ S, double* __restrict__ A, int64_t ld, int tt) {
  constexpr int R0 = LB * (PC + 1), NIT = ((NB - R0) * 8 + 191) / 192;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tt + 192 * i, r = R0 + (idx >> 3), c2 = idx & 7;
    if (r < NB) {
      const double* sp = S + sb_off(r >> 4, PC) + (r & 15) * TS17 + 2 * c2;
      *reinterpret_cast<double2*>(A + (int64_t)r * ld + LB * PC + 2 * c2) = double2{sp[0], sp[1]};
    }
  }
}

// The 35 lower blocks other than (0, 0), column by column (the columns step 0 needs first), and the 28 upper ones
__device__ constexpr int kLowI[35] = {1, 2, 3, 4, 5, 6, 7, 1, 2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};
__device__ constexpr int kLowJ[35] = {0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7};

// waves 1..3, while wave 0 factors block (0, 0): wave W brings blocks W-1, W+2, ... of that list into LDS -- two 16-byte loads per
// lane and block, all 24 in flight at once (one round trip to memory; rounds 1-4: every thread 32 loads in two batches, wave 0
// waiting with the others)
template <int W>
__device__ __forceinline__ void leaf_load_blocks(double* __restrict__ S, const double* __restrict__ A, int64_t ld, int lane) {
  constexpr int NBK = (35 - (W - 1) + 2) / 3;
  double2 v[NBK][2];
  const int rr = lane >> 3, c = 2 * (lane & 7);
#pragma unroll
  for (int k = 0; k < NBK; ++k) {
    const int b = (W - 1) + 3 * k;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      v[k][h] = *reinterpret_cast<const double2*>(A + (int64_t)(LB * kLowI[b] + rr + 8 * h) * ld + LB * kLowJ[b] + c);
  }
#pragma unroll
  for (int k = 0; k < NBK; ++k) {
    const int b = (W - 1) + 3 * k;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double* sp = S + sb_off(kLowI[b], kLowJ[b]) + (rr + 8 * h) * TS17 + c;
      sp[0] = v[k][h].x;
      sp[1] = v[k][h].y;
    }
  }
}

// zeros into the blocks above the diagonal of A and of the inverse (the mirror images of the list above, (0, 0)'s aside): stores
// only, issued in the window of step 5 (the early windows carry the big deferred updates)
template <int W>
__device__ __forceinline__ void leaf_zero_upper(double* __restrict__ A, int64_t ld, double* __restrict__ inv, int lane) {
  const int rr = lane >> 3, c = 2 * (lane & 7);
#pragma unroll
  for (int b = W - 1; b < 35; b += 3) {
    if (kLowI[b] == kLowJ[b]) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = LB * kLowJ[b] + rr + 8 * h, cc = LB * kLowI[b] + c;  // block (J, I): above the diagonal
      *reinterpret_cast<double2*>(A + (int64_t)r * ld + cc) = double2{0.0, 0.0};
      *reinterpret_cast<double2*>(inv + r * NB + cc) = double2{0.0, 0.0};
    }
  }
}

// Per step p, every wave passes the same three barriers:
//   [A] column p is in LDS        wave 0: factor + invert diagonal block p
//                                 waves 1..3 (the WINDOW of step p): DEFERRED part of update p-1, block column p-1 of L out to
//                                 global memory, row p-1 of the 128-order inverse finished, row p's sums formed
//   [B] T[p] ready                all: scale the blocks below the diagonal
//   [C] column p scaled           waves 1..3: PRIORITY part of update p (column p+1 -> LDS)
template <int W>
__device__ __forceinline__ void leaf_factor(double* __restrict__ S, double* __restrict__ A, int64_t ld,
                                            double* __restrict__ inv, int t, int g, int q) {
  constexpr int NS = leaf_nslots(W);
  d4 blk[NS];
  d4 xb[8][8], acc[8];  // only this wave's block columns of the inverse are ever touched (constant indices: scalarised)
  const int tt = t - 64;
  leaf_load_blocks<W>(S, A, ld, t & 63);
#define GPX_LEAF_STEP(P_)                                                     \
  do {                                                                        \
    if (P_ > 0) {                                                             \
      leaf_update_deferred<W, (P_ > 0 ? P_ - 1 : 0), NS>(S, blk, g, q);       \
      leaf_store_column<(P_ > 0 ? P_ - 1 : 0)>(S, A, ld, tt);                 \
      if (P_ == 5) leaf_zero_upper<W>(A, ld, inv, t & 63);                    \
      inv_finish_row<W, (P_ > 0 ? P_ - 1 : 0)>(S, inv, xb, acc, g, q);        \
      inv_row_sums<W, P_>(S, xb, acc, g, q);                                  \
    }                                                                         \
    __syncthreads(); /* [B] */                                                \
    if (P_ == 0) { /* the whole block is in LDS now: this wave's trailing blocks into registers */ \
      _Pragma("unroll") for (int s = 0; s < NS; ++s) {                        \
        const LeafBlk bk = leaf_blk(3 * s + (W - 1));                         \
        _Pragma("unroll") for (int v = 0; v < 4; ++v)                         \
          blk[s][v] = S[sb_off(bk.I, bk.K) + (g + 4 * v) * TS17 + q];         \
      }                                                                       \
    }                                                                         \
    leaf_scale(S, P_, W, g, q);                                               \
    __syncthreads(); /* [C] */                                                \
    if (P_ < 7) leaf_update_priority<W, (P_ < 7 ? P_ : 6), NS>(S, blk, g, q);  \
    __syncthreads(); /* [A] of the next step */                               \
  } while (0)
  GPX_LEAF_STEP(0);
  GPX_LEAF_STEP(1);
  GPX_LEAF_STEP(2);
  GPX_LEAF_STEP(3);
  GPX_LEAF_STEP(4);
  GPX_LEAF_STEP(5);
  GPX_LEAF_STEP(6);
  GPX_LEAF_STEP(7);
#undef GPX_LEAF_STEP
  inv_finish_row<W, 7>(S, inv, xb, acc, g, q);
}

// wave 0: the diagonal blocks
__device__ __forceinline__ void leaf_panel_wave(double* __restrict__ S, double* __restrict__ A, int64_t ld, int g, int q,
                                                int lane, int64_t base_index, int64_t n_valid, int* __restrict__ info,
                                                double piv_min, int skip, int fast, double* __restrict__ CB,
                                                long long* __restrict__ stamps) {
  for (int p = 0; p < 8; ++p) {
    double* const Dp = S + sb_off(p, p);
    double* const Ap = A + (int64_t)(LB * p) * ld + LB * p;
    if (stamps && lane == 0) stamps[2 + 3 * p] = (long long)__builtin_amdgcn_s_memtime();
    if (!fast || !leaf_diag_fast(Dp, Ap, ld, q, lane, piv_min, CB))
      leaf_diag(Dp, Ap, ld, LB * p, q, lane, base_index, n_valid, info, piv_min, skip, CB);
    if (stamps && lane == 0) stamps[3 + 3 * p] = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();  // [B]
    leaf_scale(S, p, 0, g, q);
    __syncthreads();  // [C]
    if (stamps && lane == 0) stamps[4 + 3 * p] = (long long)__builtin_amdgcn_s_memtime();
    __syncthreads();  // [A] of the next step
  }
}

// one 128 x 128 leaf: every thread of a 256-thread workgroup calls it; returns without a trailing barrier (wave 0 is done with the
// factor, waves 1..3 with the inverse's last row and their stores)
__device__ __forceinline__ void leaf_body(double* __restrict__ S, double* __restrict__ CB, double* __restrict__ A, int64_t ld,
                                          double* __restrict__ inv, int64_t base_index, int64_t n_valid, int* __restrict__ info,
                                          double piv_min, int skip, int fast, long long* __restrict__ stamps) {
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int g = lane >> 4, q = lane & 15;
  if (stamps && t == 0) { stamps[0] = (long long)__builtin_amdgcn_s_memtime(); stamps[28] = (long long)wall_clock64(); }
  if (wave == 0) {
    // round 5: wave 0 fetches ONLY block (0, 0) and starts factoring it; the other 35 blocks arrive underneath
    double2 v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) v[h] = *reinterpret_cast<const double2*>(A + (int64_t)((lane >> 3) + 8 * h) * ld + 2 * (lane & 7));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double* sp = S + sb_off(0, 0) + ((lane >> 3) + 8 * h) * TS17 + 2 * (lane & 7);
      sp[0] = v[h].x;
      sp[1] = v[h].y;
    }
    if (stamps && t == 0) stamps[1] = (long long)__builtin_amdgcn_s_memtime();
    leaf_panel_wave(S, A, ld, g, q, lane, base_index, n_valid, info, piv_min, skip, fast, CB, stamps);
    if (stamps && t == 0) stamps[27] = (long long)__builtin_amdgcn_s_memtime();
    return;
  }
  switch (wave) {
    case 1: leaf_factor<1>(S, A, ld, inv, t, g, q); break;
    case 2: leaf_factor<2>(S, A, ld, inv, t, g, q); break;
    default: leaf_factor<3>(S, A, ld, inv, t, g, q); break;
  }
  if (stamps && t == 64) stamps[29] = (long long)wall_clock64();
}

__global__ __launch_bounds__(256, 2) void leaf_kernel(double* __restrict__ A, int64_t ld, double* __restrict__ inv,
                                                   int64_t base_index, int64_t n_valid, int* __restrict__ info,
                                                   double piv_min, int skip, int hiprio, int fast,
                                                   long long* __restrict__ stamps) {
  // stamps (debug, gpx_dbg_leaf_stamps; nullptr in the product path): s_memtime of wave 0 at 0 = start, 1 = block (0, 0)
  // loaded, 2 + 3p / 3 + 3p / 4 + 3p = diagonal step p begins / its 16 x 16 factor + inverse done / column p scaled,
  // 27 = wave 0 leaves; 28 / 29 = the 100 MHz wall clock at start / at the end of the LAST wave (written by wave 1)
  __shared__ double S[36 * BSZ];
  __shared__ __attribute__((aligned(16))) double CB[LEAF_CB];
  if (hiprio) __builtin_amdgcn_s_setprio(3);   // gpx_chain_prio: beside resident GEMM waves the CU serves this chain first
  leaf_body(S, CB, A, ld, inv, base_index, n_valid, info, piv_min, skip, fast, stamps);
}

// ---- leaf multiplies: X <- X * inv^T (right) and B <- inv * B (left), in place -----------------------------------
// The triangular solves against a 128x128 diagonal block are products with its inverse.  As 128x128x128 GEMM tiles they
// are bound by the serial MFMA chain and eight k-steps of load latency (21 us however few rows there are, 1024 of them in
// a C4 factorisation).  Here a workgroup owns a strip of ST rows (right) / ST columns (left) over the WHOLE 128-wide
// leaf -- so the in-place update is race-free by construction --, and each wave multiplies the block pair (w, 7-w) of the
// inverse's block rows, whose k-extents 16(w+1) + 16(8-w) always add up to 144: the zero upper blocks are never
// multiplied (36/64 of the MFMAs).
// Round 3: the 36 fragments of the inverse a wave needs go from global memory (L2: every strip reads the same 73 KB) straight
// into REGISTERS, all in flight at once, instead of through a 73 KB LDS image -- one round trip less, and the kernel's LDS
// shrinks to the strip (33 / 66 KB): like the leaf it now co-schedules beside resident GEMM workgroups instead of waiting for a
// whole CU (scripts/dispatch_check2.hip).
template <int ST, int W>
__device__ __forceinline__ void leaf_mul_right_wave(double* __restrict__ Xs, double* __restrict__ Xg, int64_t ldx,
                                                    const double* __restrict__ inv, const double2 (&xv)[ST / 4], int t,
                                                    int g, int q) {
  constexpr int CA = W, CB = 7 - W, SA_ = 4 * (CA + 1), SB_ = 4 * (CB + 1), RB = ST / 16, NX = ST / 4;
  double fa[SA_], fb[SB_];  // B operands: inv[16 cb + q][4 s + g]
  {
    const double* ia = inv + (16 * CA + q) * NB + g;
    const double* ib = inv + (16 * CB + q) * NB + g;
#pragma unroll
    for (int s = 0; s < SA_; ++s) fa[s] = ia[4 * s];
#pragma unroll
    for (int s = 0; s < SB_; ++s) fb[s] = ib[4 * s];
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = t + 256 * i, r = idx >> 6, c = 2 * (idx & 63);
    Xs[r * LS + c] = xv[i].x;
    Xs[r * LS + c + 1] = xv[i].y;
  }
  __syncthreads();  // every read of the strip is done: the stores below cannot race with another workgroup's rows
  d4 accA[RB], accB[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) accA[rb] = accB[rb] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* xs = Xs + q * LS + g;  // X[16rb + q][4s + g]
  // both column blocks share the strip's A fragments while s is inside the shorter extent
#pragma unroll
  for (int s = 0; s < SA_; ++s) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const double a = xs[(16 * rb) * LS + 4 * s];
      accA[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fa[s], accA[rb], 0, 0, 0);
      accB[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fb[s], accB[rb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int s = SA_; s < SB_; ++s) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
      accB[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[(16 * rb) * LS + 4 * s], fb[s], accB[rb], 0, 0, 0);
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double* row = Xg + (int64_t)(16 * rb + g + 4 * v) * ldx + q;
      row[16 * CA] = accA[rb][v];
      row[16 * CB] = accB[rb][v];
    }
}

// X (m x 128, ld ldx) <- X * inv^T.  grid = m / ST workgroups.
template <int ST>
__global__ __launch_bounds__(256, 2) void leaf_mul_right_kernel(double* __restrict__ X, int64_t ldx,
                                                                const double* __restrict__ inv, int hiprio) {
  extern __shared__ double lsm[];
  if (hiprio) __builtin_amdgcn_s_setprio(3);
  double* Xs = lsm;  // [ST][129]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, q = lane & 15;
  double* Xg = X + (int64_t)blockIdx.x * ST * ldx;
  constexpr int NX = ST / 4;  // 16-byte loads per thread for the strip
  double2 xv[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int idx = t + 256 * i, r = idx >> 6, c = 2 * (idx & 63);
    xv[i] = *reinterpret_cast<const double2*>(Xg + (int64_t)r * ldx + c);
  }
  switch (wave) {
    case 0: leaf_mul_right_wave<ST, 0>(Xs, Xg, ldx, inv, xv, t, g, q); break;
    case 1: leaf_mul_right_wave<ST, 1>(Xs, Xg, ldx, inv, xv, t, g, q); break;
    case 2: leaf_mul_right_wave<ST, 2>(Xs, Xg, ldx, inv, xv, t, g, q); break;
    default: leaf_mul_right_wave<ST, 3>(Xs, Xg, ldx, inv, xv, t, g, q); break;
  }
}

template <int ST, int W>
__device__ __forceinline__ void leaf_mul_left_wave(double* __restrict__ Bs, double* __restrict__ Bg, int64_t ldb,
                                                   const double* __restrict__ inv, const double2 (&bv)[ST / 4], int t, int g,
                                                   int q) {
  constexpr int RA = W, RBK = 7 - W, SA_ = 4 * (RA + 1), SB_ = 4 * (RBK + 1), CB = ST / 16, NBV = ST / 4, CPR = ST / 2,
                SBS = ST + 1;
  double fa[SA_], fb[SB_];  // A operands: inv[16 rb + q][4 s + g]
  {
    const double* ia = inv + (16 * RA + q) * NB + g;
    const double* ib = inv + (16 * RBK + q) * NB + g;
#pragma unroll
    for (int s = 0; s < SA_; ++s) fa[s] = ia[4 * s];
#pragma unroll
    for (int s = 0; s < SB_; ++s) fb[s] = ib[4 * s];
  }
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int idx = t + 256 * i, r = idx / CPR, c = 2 * (idx % CPR);
    Bs[r * SBS + c] = bv[i].x;
    Bs[r * SBS + c + 1] = bv[i].y;
  }
  __syncthreads();
  d4 accA[CB], accB[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) accA[cb] = accB[cb] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* bs = Bs + g * SBS + q;  // B operand: strip[4s + g][16cb + q]
#pragma unroll
  for (int s = 0; s < SA_; ++s) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      const double b = bs[(4 * s) * SBS + 16 * cb];
      accA[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[s], b, accA[cb], 0, 0, 0);
      accB[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[s], b, accB[cb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int s = SA_; s < SB_; ++s) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
      accB[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[s], bs[(4 * s) * SBS + 16 * cb], accB[cb], 0, 0, 0);
  }
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      Bg[(int64_t)(16 * RA + g + 4 * v) * ldb + 16 * cb + q] = accA[cb][v];
      Bg[(int64_t)(16 * RBK + g + 4 * v) * ldb + 16 * cb + q] = accB[cb][v];
    }
}

// B (128 x m, ld ldb) <- inv * B.  grid = m / ST workgroups (ST columns each).
template <int ST>
__global__ __launch_bounds__(256, 2) void leaf_mul_left_kernel(double* __restrict__ B, int64_t ldb,
                                                               const double* __restrict__ inv, int hiprio) {
  extern __shared__ double lsm[];
  if (hiprio) __builtin_amdgcn_s_setprio(3);
  double* Bs = lsm;  // [128][ST + 1]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, q = lane & 15;
  double* Bg = B + (int64_t)blockIdx.x * ST;
  constexpr int NBV = 128 * ST / 2 / 256;  // 16-byte loads per thread for the strip (= ST / 4)
  constexpr int CPR = ST / 2;              // 16-byte chunks per strip row
  double2 bv[NBV];
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int idx = t + 256 * i, r = idx / CPR, c = 2 * (idx % CPR);
    bv[i] = *reinterpret_cast<const double2*>(Bg + (int64_t)r * ldb + c);
  }
  switch (wave) {
    case 0: leaf_mul_left_wave<ST, 0>(Bs, Bg, ldb, inv, bv, t, g, q); break;
    case 1: leaf_mul_left_wave<ST, 1>(Bs, Bg, ldb, inv, bv, t, g, q); break;
    case 2: leaf_mul_left_wave<ST, 2>(Bs, Bg, ldb, inv, bv, t, g, q); break;
    default: leaf_mul_left_wave<ST, 3>(Bs, Bg, ldb, inv, bv, t, g, q); break;
  }
}

// ---- TRSV pieces (potrs) ---------------------------------------------------------------------------
// The sweeps follow the same recursion as the factorisation: solve the first half, subtract the off-diagonal block
// times that solution from the second half as ONE bandwidth-bound GEMV over the whole block, solve the second half
// (transposed sweep: the other way round).  Leaves multiply by the inverted 128x128 diagonal block.

// ---- fused sweeps over a diagonal block of up to TRSV_BLOCK rows: the latency-bound bottom of the recursion ----------
// One workgroup solves the whole block -- leaf inverses and the coupling between the leaves -- instead of one launch per
// leaf and per coupling product (7 launches per 512 rows and sweep): potrs was ~45 us per 128 rows, all of it launch
// and load latency, and up to a third of GP.train at N <= 8192.
constexpr int TRSV_BLOCK = 512;

// dots of R consecutive 128-wide rows (row stride `stride`; lane: columns 2*lane, 2*lane+1) with a vector held two entries
// per lane; all R loads are issued before the first use (the sweeps are latency-bound: memory-level parallelism is all
// that counts), results in every lane
template <int R>
__device__ __forceinline__ void wave_rowdots(const double* __restrict__ base, int64_t stride, double2 v, int lane,
                                             double (&out)[R]) {
  double2 a[R];
#pragma unroll
  for (int i = 0; i < R; ++i) a[i] = reinterpret_cast<const double2*>(base + (int64_t)i * stride)[lane];
#pragma unroll
  for (int i = 0; i < R; ++i) out[i] = fma(a[i].x, v.x, a[i].y * v.y);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int i = 0; i < R; ++i) out[i] += __shfl_xor(out[i], off, 64);
}

// y <- L^-1 y for the n x n diagonal block at L (n = 128 * nleaf <= TRSV_BLOCK), leaf inverses at inv
__global__ __launch_bounds__(256) void trsv_block_fwd_kernel(const double* __restrict__ L, int64_t ld,
                                                             const double* __restrict__ inv, double* __restrict__ y,
                                                             int nleaf) {
  __shared__ double ys[TRSV_BLOCK];
  __shared__ double tmp[NB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = NB * nleaf;
  for (int i = t; i < n; i += 256) ys[i] = y[i];
  __syncthreads();
  for (int k = 0; k < nleaf; ++k) {
    const double* ik = inv + (int64_t)k * NB * NB;
    const double2 v = reinterpret_cast<const double2*>(ys + NB * k)[lane];
    // y_k = inv_k y_k: wave w takes rows 32w .. 32w+31, all 32 loads in flight at once (one memory round trip)
    {
      const int r = wave * 32;
      double s[32];
      wave_rowdots<32>(ik + r * NB, NB, v, lane, s);
      if (lane < 32) {
        double mine = s[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) mine = (lane == i) ? s[i] : mine;
        tmp[r + lane] = mine;
      }
    }
    __syncthreads();
    if (t < NB) ys[NB * k + t] = tmp[t];
    __syncthreads();
    // rows of the later leaves: y[r] -= L[r][128k .. 128k+127] . y_k, 32 rows per wave and trip (a whole leaf per trip)
    const double2 w = reinterpret_cast<const double2*>(ys + NB * k)[lane];
    for (int r = NB * (k + 1) + 32 * wave; r < n; r += NB) {
      double s[32];
      wave_rowdots<32>(L + (int64_t)r * ld + NB * k, ld, w, lane, s);
      if (lane < 32) {
        double mine = s[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) mine = (lane == i) ? s[i] : mine;
        ys[r + lane] -= mine;
      }
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += 256) y[i] = ys[i];
}

// z <- L^-T z for the same block: leaves in descending order; thread per column, rows walked with coalesced loads
__global__ __launch_bounds__(256) void trsv_block_bwd_kernel(const double* __restrict__ L, int64_t ld,
                                                             const double* __restrict__ inv, double* __restrict__ z,
                                                             int nleaf) {
  __shared__ double zs[TRSV_BLOCK];
  __shared__ double half[2][NB];
  const int t = threadIdx.x;
  const int n = NB * nleaf;
  for (int i = t; i < n; i += 256) zs[i] = z[i];
  __syncthreads();
  for (int k = nleaf - 1; k >= 0; --k) {
    // z_k = inv_k^T z_k: column c = t & 127, the two thread halves take rows 0..63 / 64..127
    {
      const double* ik = inv + (int64_t)k * NB * NB;
      const int c = t & 127, r0 = (t >> 7) * 64;
      double a[64];  // all 64 loads of this thread in flight at once
#pragma unroll
      for (int r = 0; r < 64; ++r) a[r] = ik[(r0 + r) * NB + c];
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int r = 0; r < 64; r += 2) {
        s0 = fma(a[r], zs[NB * k + r0 + r], s0);
        s1 = fma(a[r + 1], zs[NB * k + r0 + r + 1], s1);
      }
      half[t >> 7][c] = s0 + s1;
    }
    __syncthreads();
    if (t < NB) zs[NB * k + t] = half[0][t] + half[1][t];
    __syncthreads();
    // columns of the earlier leaves: z[c] -= sum_r L[128k + r][c] z_k[r]
    for (int c = t; c < NB * k; c += 256) {
      const double* lc = L + (int64_t)(NB * k) * ld + c;
      double s0 = 0.0, s1 = 0.0;
#pragma unroll 1
      for (int rb = 0; rb < NB; rb += 64) {
        double a[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) a[r] = lc[(int64_t)(rb + r) * ld];
#pragma unroll
        for (int r = 0; r < 64; r += 2) {
          s0 = fma(a[r], zs[NB * k + rb + r], s0);
          s1 = fma(a[r + 1], zs[NB * k + rb + r + 1], s1);
        }
      }
      zs[c] -= s0 + s1;
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += 256) z[i] = zs[i];
}

// y[r] -= sum_c A[r][c] x[c]   (rows x cols block, cols a multiple of 128): one wave per row, 8 rows per workgroup pass
__global__ __launch_bounds__(256) void gemv_sub_kernel(const double* __restrict__ A, int64_t ld, int64_t rows,
                                                       int64_t cols, const double* __restrict__ x,
                                                       double* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const double2* ar = reinterpret_cast<const double2*>(A + r * ld);
    const double2* xr = reinterpret_cast<const double2*>(x);
    double s0 = 0.0, s1 = 0.0;
    for (int64_t c = lane; c < cols / 2; c += 64) {
      const double2 a = ar[c], b = xr[c];
      s0 = fma(a.x, b.x, s0);
      s1 = fma(a.y, b.y, s1);
    }
    double s = s0 + s1;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) y[r] -= s;
  }
}

__global__ __launch_bounds__(256) void logdet_kernel(const double* __restrict__ L, int64_t ld, int64_t n,
                                                     double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += log(L[i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = 2.0 * red[0];
}


// ---- potrs through explicit block inverses ---------------------------------------------------------------------------
// The sweeps are a chain of N / IB dependent steps; with one workgroup streaming each 512-row diagonal block the chain
// cost 40-60 us per step (round 1: 10.5 ms at N = 32768, 5 % of the HBM roof).  Here the diagonal blocks of order IB (up
// to 1024) are INVERTED once per factorisation -- recursively from the 128 x 128 leaf inverses, as batched MFMA products
// over all blocks at once -- so that a diagonal solve becomes one chip-wide matrix-vector product with the inverse (row-
// wise for the forward sweep, with the stored transpose for the backward sweep): ~4 us per step instead of ~50.

// binv[b][t*128 + r][t*128 + c] = invd[(b * (IB/128) + t)][r][c], zero elsewhere
__global__ __launch_bounds__(256) void binv_init_kernel(const double* __restrict__ invd, double* __restrict__ binv,
                                                        int64_t ib, int64_t n) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one double2 of one IB-wide row of one block
  const int64_t per_row = ib / 2, row = idx / per_row, c = 2 * (idx % per_row);
  const int64_t b = row / ib, r = row % ib;
  if (b * ib + r >= (n + ib - 1) / ib * ib) return;
  double2 v = double2{0.0, 0.0};
  const int64_t g = b * ib + r;  // global row
  if (g < n && (c >> 7) == (r >> 7)) v = *reinterpret_cast<const double2*>(invd + (g >> 7) * NB * NB + (r & 127) * NB + (c & 127));
  *reinterpret_cast<double2*>(binv + row * ib + c) = v;
}

// batched transpose of the IB x IB blocks
__global__ __launch_bounds__(256) void binv_transpose_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                             int64_t ib) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* ib_in = in + (int64_t)blockIdx.z * ib * ib;
  double* ib_out = out + (int64_t)blockIdx.z * ib * ib;
  const int64_t i0 = (int64_t)blockIdx.y * 32, j0 = (int64_t)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = ib_in[(i0 + r) * ib + j0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) ib_out[(j0 + r) * ib + i0 + tx] = tile[tx][r];
}

// x[r] = sum_c M[r][c] y[c] over c in [c_lo(r), c_hi(r)): one wave per row; lower != 0: M is lower triangular (columns up
// to the end of r's 128-block), else upper triangular (columns from the start of r's 128-block)
__global__ __launch_bounds__(256) void binv_gemv_kernel(const double* __restrict__ M, int64_t ld, int64_t sz,
                                                        const double* __restrict__ y, double* __restrict__ x, int lower) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)blockIdx.x * 4 + wave;
  if (r >= sz) return;
  const int64_t c0 = lower ? 0 : (r & ~(int64_t)127), c1 = lower ? ((r | 127) + 1) : sz;
  const double2* mr = reinterpret_cast<const double2*>(M + r * ld);
  const double2* yr = reinterpret_cast<const double2*>(y);
  double s0 = 0.0, s1 = 0.0;
  for (int64_t c = c0 / 2 + lane; c < c1 / 2; c += 64) {
    const double2 a = mr[c], b = yr[c];
    s0 = fma(a.x, b.x, s0);
    s1 = fma(a.y, b.y, s1);
  }
  double s = s0 + s1;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) x[r] = s;
}

inline int64_t split(int64_t n) { return (n / NB / 2) * NB; }  // n multiple of 128, n > 128

}  // namespace

// the 64-row / 64-column strips need more than the default 64 KiB of dynamic LDS
template <class K>
static int leaf_mul_allow_lds(K kernel, size_t bytes) {
  GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

// X (m x 128) <- X * inv^T in place; m a multiple of 128
static int launch_leaf_mul_right(gpx_ctx* ctx, double* X, int64_t ldx, const double* inv, int64_t m) {
  GPX_ARG(m % NB == 0 && (ldx % 2) == 0, "leaf multiply: rows must be a multiple of 128 and ld even");
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * (double)m * NB * NB, 0.0);
  // 32-row strips while they fill the chip once, 64-row strips (half the copies of the inverse) beyond that
  if (m <= 32 * (int64_t)ctx->cus) {
    constexpr int ST = 32;
    const size_t sh = (size_t)(ST * LS) * sizeof(double);
    static bool once = false;
    if (!once) { GPX_TRY(leaf_mul_allow_lds(leaf_mul_right_kernel<ST>, sh)); once = true; }
    hipLaunchKernelGGL(leaf_mul_right_kernel<ST>, dim3((unsigned)(m / ST)), dim3(256), sh, ctx->stream, X, ldx, inv, gpx_chain_prio(ctx));
  } else {
    constexpr int ST = 64;
    const size_t sh = (size_t)(ST * LS) * sizeof(double);
    static bool once = false;
    if (!once) { GPX_TRY(leaf_mul_allow_lds(leaf_mul_right_kernel<ST>, sh)); once = true; }
    hipLaunchKernelGGL(leaf_mul_right_kernel<ST>, dim3((unsigned)(m / ST)), dim3(256), sh, ctx->stream, X, ldx, inv, gpx_chain_prio(ctx));
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

// B (128 x m) <- inv * B in place; m a multiple of 128
static int launch_leaf_mul_left(gpx_ctx* ctx, double* B, int64_t ldb, const double* inv, int64_t m) {
  GPX_ARG(m % NB == 0 && (ldb % 2) == 0, "leaf multiply: columns must be a multiple of 128 and ld even");
  ProfScope ps(ctx, GPX_PROF_GEMM, 2.0 * (double)m * NB * NB, 0.0);
  if (m <= 32 * (int64_t)ctx->cus) {
    constexpr int ST = 32;
    const size_t sh = (size_t)(NB * (ST + 1)) * sizeof(double);
    static bool once = false;
    if (!once) { GPX_TRY(leaf_mul_allow_lds(leaf_mul_left_kernel<ST>, sh)); once = true; }
    hipLaunchKernelGGL(leaf_mul_left_kernel<ST>, dim3((unsigned)(m / ST)), dim3(256), sh, ctx->stream, B, ldb, inv, gpx_chain_prio(ctx));
  } else {
    constexpr int ST = 64;
    const size_t sh = (size_t)(NB * (ST + 1)) * sizeof(double);
    static bool once = false;
    if (!once) { GPX_TRY(leaf_mul_allow_lds(leaf_mul_left_kernel<ST>, sh)); once = true; }
    hipLaunchKernelGGL(leaf_mul_left_kernel<ST>, dim3((unsigned)(m / ST)), dim3(256), sh, ctx->stream, B, ldb, inv, gpx_chain_prio(ctx));
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

int launch_leaf(gpx_ctx* ctx, double* A, int64_t ld, double* inv, int64_t base_index, int64_t n_valid) {
  ProfScope ps(ctx, GPX_PROF_LEAF, 2.0 * NB * NB * NB / 3.0, 0.0);
  const int fast = 1;   // round 5's diagonal step (gpx_dbg_leaf_stamps can still time the general one, which is also the bad-pivot fall-back)
  hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), 0, ctx->stream, A, ld, inv, base_index, n_valid, ctx->d_info,
                     ctx->piv_min, ctx->piv_skip, gpx_chain_prio(ctx), fast, (long long*)nullptr);
  GPX_HIP(hipGetLastError());
  return 0;
}

// debug (gpx_debug.h): one leaf launch on the leading 128 x 128 block of K with the phase stamps of leaf_kernel copied out
extern "C" int gpx_dbg_leaf_stamps(gpx_ctx* ctx, gpx_mat* K, int fast, int64_t* out30) {
  GPX_ARG(ctx && K && out30 && K->prows >= NB && K->pcols >= NB, "leaf stamps: bad arguments");
  long long* d = nullptr;
  double* inv = nullptr;
  GPX_HIP(hipMalloc((void**)&d, 30 * sizeof(long long)));
  GPX_HIP(hipMalloc((void**)&inv, NB * NB * sizeof(double)));
  GPX_HIP(hipMemsetAsync(d, 0, 30 * sizeof(long long), ctx->stream));
  hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(256), 0, ctx->stream, K->p, K->ld, inv, (int64_t)0, (int64_t)NB, ctx->d_info,
                     ctx->piv_min, ctx->piv_skip, 0, fast, d);
  GPX_HIP(hipGetLastError());
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  GPX_HIP(hipMemcpy(out30, d, 30 * sizeof(long long), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  (void)hipFree(inv);
  return 0;
}

// X (m x n) <- X * L^-T, L lower n x n at (L, ldl) whose leaf inverses are invd[(row/128)] blocks
int chol_trsm_right(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* X, int64_t ldx,
                    int64_t m, int64_t n) {
  if (m == 0 || n == 0) return 0;
  if (n == NB) return launch_leaf_mul_right(ctx, X, ldx, invd, m);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(chol_trsm_right(ctx, L, ldl, invd, X, ldx, m, n1));
  // X2 -= X1 * L21^T
  GPX_TRY(launch_gemm(ctx, X, ldx, L + n1 * ldl, ldl, X + n1, ldx, m, n2, n1, true, true, false));
  return chol_trsm_right(ctx, L + n1 * ldl + n1, ldl, invd + (n1 / NB) * NB * NB, X + n1, ldx, m, n2);
}

// X (m x n) <- X * L^-1  (right side, lower, NOT transposed): [X1 X2][L11 0; L21 L22] = [A1 A2]
//   X2 = A2 L22^-1;  A1 -= X2 L21;  X1 = A1 L11^-1.   Leaf: X <- X * inv (in place, one 128-wide tile per workgroup)
int chol_trsm_right_n(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* X, int64_t ldx,
                      int64_t m, int64_t n) {
  if (m == 0 || n == 0) return 0;
  if (n == NB) return launch_gemm(ctx, X, ldx, invd, NB, X, ldx, m, NB, NB, false, false, false);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(chol_trsm_right_n(ctx, L + n1 * ldl + n1, ldl, invd + (n1 / NB) * NB * NB, X + n1, ldx, m, n2));
  GPX_TRY(launch_gemm(ctx, X + n1, ldx, L + n1 * ldl, ldl, X, ldx, m, n1, n2, false, true, false));
  return chol_trsm_right_n(ctx, L, ldl, invd, X, ldx, m, n1);
}

// B (n x m) <- L^-1 B
int chol_trsm_left(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* B, int64_t ldb,
                   int64_t n, int64_t m) {
  if (m == 0 || n == 0) return 0;
  if (n == NB) return launch_leaf_mul_left(ctx, B, ldb, invd, m);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(chol_trsm_left(ctx, L, ldl, invd, B, ldb, n1, m));
  // B2 -= L21 * W1
  GPX_TRY(launch_gemm(ctx, L + n1 * ldl, ldl, B, ldb, B + n1 * ldb, ldb, n2, m, n1, false, true, false));
  return chol_trsm_left(ctx, L + n1 * ldl + n1, ldl, invd + (n1 / NB) * NB * NB, B + n1 * ldb, ldb, n2, m);
}

// Right-looking factorisation with 128-wide panels: leaf, one multiply of everything below it by the leaf's inverse, one
// rank-128 lower update of the whole trailing block -- three launches per 128 columns.  Its updates have K = 128 (the C
// read-modify-write dominates), so it only pays where the recursion below would spend its time in launches anyway.
static int potrf_right_looking(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base,
                               int64_t n_valid) {
  for (int64_t j = 0; j < n; j += NB) {
    double* inv_j = invd + (j / NB) * NB * NB;
    GPX_TRY(launch_leaf(ctx, A + j * ld + j, ld, inv_j, base + j, n_valid));
    const int64_t m = n - j - NB;
    if (m > 0) {
      double* X = A + (j + NB) * ld + j;
      GPX_TRY(launch_leaf_mul_right(ctx, X, ld, inv_j, m));
      GPX_TRY(launch_gemm(ctx, X, ld, X, ld, X + NB, ld, m, m, NB, true, true, true));
    }
  }
  return 0;
}

// diagonal blocks up to this order are factored right-looking (leaf, strip multiply, rank-128 update per 128 columns); the panel
// width of the blocked look-ahead factorisation above it.  Re-swept in round 5 (profiles/r05_potrf_variants.txt): right-looking
// limit 2048 / 1024: +1.6 / +3.5 ms at N = 32768; panel 2048 / 8192: +6.7 / +6.3 ms.
constexpr int64_t POTRF_RL_MAX = 4096;
constexpr int64_t POTRF_PANEL = 4096;

static int64_t env_i64(const char* name, int64_t dflt) {
  const char* e = getenv(name);
  return e ? atoll(e) : dflt;
}

static int potrf_rec(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base, int64_t n_valid);
static int binv_build_range(gpx_ctx* ctx, const double* Ld, int64_t ld, const double* invd, double* binv, int64_t ib, int64_t n,
                            double* tmp, int64_t lo = 0, int64_t hi = INT64_MAX);

// X (m x n) <- X L^-T against a diagonal block whose ib-order sub-blocks have explicit inverses (binv, ib x ib each): per
// sub-block X_b <- X_b Binv_b^T as ONE GEMM (triangular operand: half the k range per column tile; through T, the product
// cannot be formed in place), between them X[:, later] -= X[:, done] L[later, done]^T with K >= ib.  The leaf-level recursion
// of chol_trsm_right spends its time in K = 128..512 updates (12-60 TF/s); with 4096-wide panels those were half of the
// panel solves' 58 ms at N = 32768.
// FEW rows (m <= 1024: one batch of design points against the kept factor, gpx_refit_rows): the products of the lower levels are
// m x ib x ib -- 128 64-tiles under a serial k range of 1024, 55-60 us each on half the chip, 2 ms of the 3.6 ms of a
// 512 x 15872 solve whose flops are 1.6 ms.  Those run as slices of the k range (launch_gemm_ksplit_small): >= 512 workgroups.
static inline bool trsm_few_rows(int64_t m) { return m <= 1024 && m % 64 == 0; }
static inline int64_t slices_for(int64_t m, int64_t n, int64_t k) {
  if (n % 64 != 0) return 1;
  const int64_t tiles = (m / 64) * (n / 64);
  int64_t parts = 1;
  while (tiles * parts < 512 && k % (2 * parts * 16) == 0 && k / (2 * parts) >= 256) parts *= 2;
  return parts;
}

static int trsm_right_binv_rec(gpx_ctx* ctx, const double* Ld, int64_t ldl, const double* binv, int64_t ib, double* X,
                               int64_t ldx, int64_t m, int64_t n, int64_t b0, int64_t b1, double* T, int64_t tcap = 0) {
  const bool few = tcap >= 4 * m * ib && trsm_few_rows(m);   // tcap: doubles T holds when it is more than m ib
  auto off = [&](int64_t b) { return b * ib < n ? b * ib : n; };
  if (b1 - b0 == 1) {
    // (round 4 ran tall pieces through an in-place strip kernel -- one workgroup per 128-row strip walking the column tiles from
    // the last to the first: m / 128 workgroups, 41 TF/s on the useful flops.  With the longest-first tile order of the
    // triangular-operand GEMM (gemm_f64.hip tile_of, round 5) the 2-D launch reaches 55 TF/s at m = 24576 and the copy back costs
    // less than the difference: potrf 184.2 -> 181.8 ms at N = 32768; the strip kernel is gone.)
    const int64_t o = off(b0), sz = off(b0 + 1) - o;
    // (few rows: the k range in slices, dense against the stored inverse, whose upper part is zero; the sum goes straight into X)
    const int64_t parts = few ? slices_for(m, sz, sz) : 1;
    if (parts > 1) return launch_gemm_ksplit_small(ctx, X + o, ldx, binv + b0 * ib * ib, ib, X + o, ldx, m, sz, sz, true, parts, T);
    GPX_TRY(launch_gemm_tri(ctx, X + o, ldx, binv + b0 * ib * ib, ib, T, ib, m, sz, sz, true, false, false, 2));
    return gpx_copy2d(ctx, T, ib, X + o, ldx, m, sz);
  }
  const int64_t mid = (b0 + b1) / 2, o0 = off(b0), om = off(mid), o1 = off(b1);
  GPX_TRY(trsm_right_binv_rec(ctx, Ld, ldl, binv, ib, X, ldx, m, n, b0, mid, T, tcap));
  const int64_t parts = few ? slices_for(m, o1 - om, om - o0) : 1;
  // (the upper levels: 128-tiles -- 4 tile rows of them cannot fill the chip either -- over slices of >= 1024 of the k range)
  int64_t p128 = 1;
  if (few && parts == 1 && m % 128 == 0 && (o1 - om) % 128 == 0)
    while ((m / 128) * ((o1 - om) / 128) * p128 < 512 && (om - o0) % (2 * p128 * 16) == 0 && (om - o0) / (2 * p128) >= 1024) p128 *= 2;
  if (parts > 1 && m * (o1 - om) * parts <= tcap)
    GPX_TRY(launch_gemm_ksplit_small(ctx, X + o0, ldx, Ld + om * ldl + o0, ldl, X + om, ldx, m, o1 - om, om - o0, false, parts, T));
  else if (p128 > 1 && m * (o1 - om) * p128 <= tcap)
    GPX_TRY(launch_gemm_ksplit(ctx, X + o0, ldx, Ld + om * ldl + o0, ldl, X + om, ldx, m, o1 - om, om - o0, false, p128, T));
  else
    GPX_TRY(launch_gemm(ctx, X + o0, ldx, Ld + om * ldl + o0, ldl, X + om, ldx, m, o1 - om, om - o0, true, true, false));
  return trsm_right_binv_rec(ctx, Ld, ldl, binv, ib, X, ldx, m, n, mid, b1, T, tcap);
}

// X (m x n) <- X L^-1 (NOT transposed) with the same block inverses: block columns from the last to the first, X_b <- X_b Binv_b
// -- round 5: as a triangular-operand product with the inverse itself (k x n lower, not transposed: the k range of a column tile
// starts at its diagonal; `binv` = the lower inverses, `binvT` their transposes for callers that only hold those) --,
// between them X[:, earlier] -= X[:, done] L[done, earlier] with K >= ib.  The leaf-level recursion of chol_trsm_right_n spends
// its time in K = 128..512 products.
static int trsm_right_n_binv_rec(gpx_ctx* ctx, const double* Ld, int64_t ldl, const double* binvT, int64_t ib, double* X,
                                 int64_t ldx, int64_t m, int64_t n, int64_t b0, int64_t b1, double* T, const double* binv = nullptr) {
  auto off = [&](int64_t b) { return b * ib < n ? b * ib : n; };
  if (b1 - b0 == 1) {
    const int64_t o = off(b0), sz = off(b0 + 1) - o;
    if (binv)
      GPX_TRY(launch_gemm_tri(ctx, X + o, ldx, binv + b0 * ib * ib, ib, T, ib, m, sz, sz, false, false, false, 4));
    else
      GPX_TRY(launch_gemm(ctx, X + o, ldx, binvT + b0 * ib * ib, ib, T, ib, m, sz, sz, true, false, false));
    return gpx_copy2d(ctx, T, ib, X + o, ldx, m, sz);
  }
  const int64_t mid = (b0 + b1) / 2, o0 = off(b0), om = off(mid), o1 = off(b1);
  GPX_TRY(trsm_right_n_binv_rec(ctx, Ld, ldl, binvT, ib, X, ldx, m, n, mid, b1, T, binv));
  GPX_TRY(launch_gemm(ctx, X + om, ldx, Ld + om * ldl + o0, ldl, X + o0, ldx, m, om - o0, o1 - om, false, true, false));
  return trsm_right_n_binv_rec(ctx, Ld, ldl, binvT, ib, X, ldx, m, n, b0, mid, T, binv);
}

// Both right solves against the TRAILING factor L[r0:, r0:] of a complete factor with block inverses (r0 a multiple of the
// inverse order): X <- X L22^-T (transposed != 0) or X <- X L22^-1.  T >= m * ib doubles of scratch.  Used by the slab form of the
// log-marginal gradient (hyper.hip), whose two solves are 2 (N - r0)^2 m flops.
int chol_trsm_right_trailing(gpx_ctx* ctx, gpx_mat* Lm, int64_t r0, double* X, int64_t ldx, int64_t m, int transposed, double* T) {
  GPX_ARG(Lm && Lm->factored && X && T, "trsm trailing: NULL argument / not factored");
  GPX_TRY(chol_binv_ensure(ctx, Lm));
  const int64_t ib = Lm->binv_ib, n2 = Lm->prows - r0;
  GPX_ARG(r0 >= 0 && r0 % ib == 0 && n2 > 0, "trsm trailing: the offset must be a multiple of the block-inverse order");
  const int64_t nblk_all = (Lm->prows + ib - 1) / ib, nb2 = (n2 + ib - 1) / ib;
  const double* Ld = Lm->p + r0 * (Lm->ld + 1);
  const double* binv = Lm->binv + (r0 / ib) * ib * ib;
  const double* binvT = Lm->binv + nblk_all * ib * ib + (r0 / ib) * ib * ib;
  if (transposed) return trsm_right_binv_rec(ctx, Ld, Lm->ld, binv, ib, X, ldx, m, n2, 0, nb2, T);
  return trsm_right_n_binv_rec(ctx, Ld, Lm->ld, binvT, ib, X, ldx, m, n2, 0, nb2, T, binv);
}

// X (m x ncols) <- X L11^-T against the LEADING ncols x ncols block of a complete factor, through its block inverses (every
// product K >= the inverse order instead of the leaf recursion's K = 128..512): the strip solve of gpx_refit_rows, whose few
// rows (one batch of design points) make short-K products latency-bound.  ncols a multiple of 128; T holds tcap >= m * ib
// doubles (from 4 m ib on the small products of a few-row solve run as slices of their k range, with 2 m ncols the large ones too).
int chol_trsm_right_leading(gpx_ctx* ctx, gpx_mat* Lm, int64_t ncols, double* X, int64_t ldx, int64_t m, double* T, int64_t tcap) {
  GPX_ARG(Lm && Lm->factored && X && T && ncols > 0 && ncols <= Lm->prows && ncols % NB == 0, "trsm leading: bad arguments");
  GPX_TRY(chol_binv_ensure(ctx, Lm));
  const int64_t ib = Lm->binv_ib;
  GPX_ARG(tcap >= m * ib, "trsm leading: scratch too small");
  return trsm_right_binv_rec(ctx, Lm->p, Lm->ld, Lm->binv, ib, X, ldx, m, ncols, 0, (ncols + ib - 1) / ib, T, tcap);
}

// X (m x ncols) <- X L11^-1 (NOT transposed) against the LEADING ncols x ncols block of a complete factor through its block
// inverses: rows of L^-1 are X = E L11^-1 (the row form of the sharded log-marginal gradient, hyper.hip gpx_lml_grad_rows).
// ncols a multiple of 128 (a partial last block uses the leading part of its inverse: the inverse of a leading principal block of
// a triangular matrix is the leading block of its inverse); T >= m * ib doubles.
int chol_trsm_right_n_leading(gpx_ctx* ctx, gpx_mat* Lm, int64_t ncols, double* X, int64_t ldx, int64_t m, double* T) {
  GPX_ARG(Lm && Lm->factored && X && T && ncols > 0 && ncols <= Lm->prows && ncols % NB == 0, "trsm leading (n): bad arguments");
  GPX_TRY(chol_binv_ensure(ctx, Lm));
  const int64_t ib = Lm->binv_ib, nblk_all = (Lm->prows + ib - 1) / ib;
  return trsm_right_n_binv_rec(ctx, Lm->p, Lm->ld, Lm->binv + nblk_all * ib * ib, ib, X, ldx, m, ncols, 0, (ncols + ib - 1) / ib, T,
                               Lm->binv);
}

// Blocked right-looking factorisation with panels of width B (4096) and ONE PANEL OF LOOK-AHEAD for large matrices.
// Per panel k: solve the rows below the diagonal block, then the trailing update A22 -= P P^T with K = B.  The diagonal
// block of panel k+1 -- a chain of ~100 latency-bound kernels (128-wide leaves, strip multiplies, rank-128 updates: 1.9 ms
// per 4096 block on an otherwise idle chip) -- only needs block row k+1 of the panel and the diagonal block (k+1, k+1) of that
// update.  CRITICAL PATH FIRST (round 3): solve THOSE rows, update THAT block ("top"), start the chain on the high-priority
// side stream -- and only then solve the rest of the panel, update the rest of block column k+1 and the bulk, all underneath
// the chain, whose kernels fit beside resident GEMM workgroups (leaf 78 KB LDS / 184 VGPRs, strip multiplies 33-66 KB:
// scripts/dispatch_check2.hip) and get a slot whenever one retires.
// History of what this replaced, all measured and removed (profiles/r02_potrf_lookahead.txt, r02_potrf_phases.txt,
// r04_potrf_two_panel_lookahead.txt, r06_potrf_variants.txt): the chain behind the whole column update (8.7 + 2.5 ms
// exposed at N = 32768); a CU-masked chunk of the update beside the chain (a kernel on a masked stream runs at exactly its CU
// share; 190-195 ms); all large kernels masked (226-233 ms); two panels of look-ahead (the bulk update stretches by what the thin
// "top" launches take beside it); the update in slices of its k range; the rest solve as one product with a 4096-order inverse.
// Main stream only (the streams are the context's).  GPX_POTRF_TIMING=1 (debug): per-panel phase spans on stderr.
static int potrf_blocked(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base, int64_t n_valid,
                         int64_t B) {
  hipStream_t M = ctx->stream, S = ctx->streams[1];
  const bool la = M == ctx->streams[0];
  static const int64_t timing = env_i64("GPX_POTRF_TIMING", 0);
  struct Span { const char* what; int64_t panel; hipEvent_t a, b; };
  std::vector<Span> spans;
  auto mark = [&](hipStream_t st) -> hipEvent_t {
    hipEvent_t e = nullptr;
    if (timing && base == 0 && hipEventCreate(&e) == hipSuccess) (void)hipEventRecord(e, st);
    return e;
  };
  if (la && ctx->la_events.empty()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t ev;
      GPX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      ctx->la_events.push_back(ev);
    }
  }
  // explicit inverses of the ib-order diagonal sub-blocks (gpx_potrf provides the storage): built right behind every
  // diagonal block -- part of the chain -- and used by the panel solves; gpx_potrs / posterior reuse them afterwards
  const int64_t ib = ctx->pw_ib;
  const bool bi = ctx->pw_binv != nullptr && ib > 0 && base % ib == 0 && B % ib == 0;
  auto binv_at = [&](int64_t row) { return ctx->pw_binv + ((base + row) / ib) * ib * ib; };
  auto solve = [&](int64_t j0, int64_t w, double* X, int64_t m) -> int {   // X (m x w) <- X D^-T, D the diagonal block at j0
    if (bi) return trsm_right_binv_rec(ctx, A + j0 * (ld + 1), ld, binv_at(j0), ib, X, ld, m, w, 0, (w + ib - 1) / ib, ctx->pw_tmp_T);
    return chol_trsm_right(ctx, A + j0 * (ld + 1), ld, invd + (j0 / NB) * NB * NB, X, ld, m, w);
  };
  GPX_TRY(potrf_rec(ctx, A, ld, n < B ? n : B, invd, base, n_valid));
  if (bi) GPX_TRY(binv_build_range(ctx, A, ld, invd, binv_at(0), ib, n < B ? n : B, ctx->pw_tmp_build));
  for (int64_t j0 = 0; j0 < n; j0 += B) {
    const int64_t w = (n - j0) < B ? (n - j0) : B, below = n - j0 - w;
    if (below == 0) break;                     // the last diagonal block: the main stream has already waited for its chain
    double* P = A + (j0 + w) * ld + j0;        // below x w panel
    double* C = A + (j0 + w) * (ld + 1);       // trailing block, below x below
    const int64_t w2 = below < B ? below : B, rest = below - w2;
    double* invn = invd + ((j0 + w) / NB) * NB * NB;
    if (!la) {   // (a caller on another stream of the context: no side stream to put the chain on)
      GPX_TRY(solve(j0, w, P, below));
      GPX_TRY(launch_gemm(ctx, P, ld, P, ld, C, ld, below, below, w, true, true, true));
      GPX_TRY(potrf_rec(ctx, C, ld, w2, invn, base + j0 + w, n_valid));
      if (bi) GPX_TRY(binv_build_range(ctx, C, ld, invn, binv_at(j0 + w), ib, w2, ctx->pw_tmp_build));
      continue;
    }
    hipEvent_t ev_col = ctx->la_events[0], ev_diag = ctx->la_events[1];
    hipEvent_t t0 = mark(M);
    GPX_TRY(solve(j0, w, P, w2));
    GPX_TRY(launch_gemm(ctx, P, ld, P, ld, C, ld, w2, w2, w, true, true, true));
    spans.push_back({"top", j0 / B, t0, mark(M)});
    GPX_HIP(hipEventRecord(ev_col, M));
    ctx->stream = S;
    int r = 0;
    if (hipStreamWaitEvent(S, ev_col, 0) != hipSuccess) r = -2;
    t0 = mark(S);
    if (r == 0) r = potrf_rec(ctx, C, ld, w2, invn, base + j0 + w, n_valid);
    if (r == 0 && bi) r = binv_build_range(ctx, C, ld, invn, binv_at(j0 + w), ib, w2, ctx->pw_tmp_build);
    spans.push_back({"chain", j0 / B, t0, mark(S)});
    if (r == 0 && hipEventRecord(ev_diag, S) != hipSuccess) r = -2;
    ctx->stream = M;
    if (r != 0) {
      if (r == -2) gpx_set_error("potrf: look-ahead stream plumbing failed");
      return r;
    }
    if (rest > 0) {
      double* P2 = P + w2 * ld;
      t0 = mark(M);
      GPX_TRY(solve(j0, w, P2, rest));   // (scratch shared with the top rows' solve: same stream, so in order)
      spans.push_back({"solve", j0 / B, t0, mark(M)});
      t0 = mark(M);
      GPX_TRY(launch_gemm(ctx, P2, ld, P, ld, C + w2 * ld, ld, rest, w2, w, true, true, false));
      spans.push_back({"column", j0 / B, t0, mark(M)});
      t0 = mark(M);
      GPX_TRY(launch_gemm(ctx, P2, ld, P2, ld, C + w2 * (ld + 1), ld, rest, rest, w, true, true, true));
      spans.push_back({"bulk", j0 / B, t0, mark(M)});
    }
    GPX_HIP(hipStreamWaitEvent(M, ev_diag, 0));
  }
  if (bi && base == 0) ctx->pw_done = 1;
  if (timing && base == 0 && !spans.empty()) {
    (void)hipDeviceSynchronize();
    hipEvent_t first = spans[0].a;
    for (const Span& sp : spans) {
      float off = 0.f, len = 0.f;
      if (sp.a && sp.b && hipEventElapsedTime(&off, first, sp.a) == hipSuccess && hipEventElapsedTime(&len, sp.a, sp.b) == hipSuccess)
        fprintf(stderr, "potrf-timing panel %lld %-8s at %8.3f ms  len %7.3f ms\n", (long long)sp.panel, sp.what, off, len);
    }
    for (const Span& sp : spans) {
      if (sp.a) (void)hipEventDestroy(sp.a);
      if (sp.b) (void)hipEventDestroy(sp.b);
    }
  }
  return 0;
}

int64_t chol_potrf_panel_width(int64_t n) { return n >= 2 * POTRF_PANEL ? POTRF_PANEL : 0; }

static int potrf_rec(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base, int64_t n_valid) {
  if (n == NB) return launch_leaf(ctx, A, ld, invd, base, n_valid);
  // (a one-launch cooperative form of the diagonal block -- one chain workgroup + helper workgroups handing work over through flags
  // in global memory -- was built in round 5 and measured no faster than this launch chain: profiles/r05_potrf_coop_ab.txt; removed
  // in round 6)
  if (n <= POTRF_RL_MAX) return potrf_right_looking(ctx, A, ld, n, invd, base, n_valid);
  if (n >= 2 * POTRF_PANEL) return potrf_blocked(ctx, A, ld, n, invd, base, n_valid, POTRF_PANEL);
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(potrf_rec(ctx, A, ld, n1, invd, base, n_valid));
  double* A21 = A + n1 * ld;
  GPX_TRY(chol_trsm_right(ctx, A, ld, invd, A21, ld, n2, n1));
  GPX_TRY(launch_gemm(ctx, A21, ld, A21, ld, A21 + n1, ld, n2, n2, n1, true, true, true));
  return potrf_rec(ctx, A21 + n1, ld, n2, invd + (n1 / NB) * NB * NB, base + n1, n_valid);
}

int chol_potrf(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t n_valid) {
  GPX_ARG(n > 0 && n % NB == 0, "potrf: padded order must be a positive multiple of 128");
  GPX_HIP(hipMemsetAsync(ctx->d_info, 0, 2 * sizeof(int), ctx->stream));
  return potrf_rec(ctx, A, ld, n, invd, 0, n_valid);
}

int chol_potrf_nozero(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base, int64_t n_valid) {
  GPX_ARG(n > 0 && n % NB == 0, "potrf: padded order must be a positive multiple of 128");
  return potrf_rec(ctx, A, ld, n, invd, base, n_valid);
}

static int trsv_fwd_rec(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* y, int64_t n) {
  if (n <= TRSV_BLOCK) {
    hipLaunchKernelGGL(trsv_block_fwd_kernel, dim3(1), dim3(256), 0, ctx->stream, L, ld, invd, y, (int)(n / NB));
    return 0;
  }
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(trsv_fwd_rec(ctx, L, ld, invd, y, n1));
  int64_t wg = (n2 + 3) / 4;
  if (wg > 4096) wg = 4096;
  hipLaunchKernelGGL(gemv_sub_kernel, dim3((unsigned)wg), dim3(256), 0, ctx->stream, L + n1 * ld, ld, n2, n1, y, y + n1);
  return trsv_fwd_rec(ctx, L + n1 * ld + n1, ld, invd + (n1 / NB) * NB * NB, y + n1, n2);
}

// scratch: tmp (>= n doubles) and part (>= colreduce_partial_elems(n, n) doubles) for the transposed GEMV
static int trsv_bwd_rec(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* z, int64_t n,
                        double* tmp, double* part) {
  if (n <= TRSV_BLOCK) {
    hipLaunchKernelGGL(trsv_block_bwd_kernel, dim3(1), dim3(256), 0, ctx->stream, L, ld, invd, z, (int)(n / NB));
    return 0;
  }
  const int64_t n1 = split(n), n2 = n - n1;
  GPX_TRY(trsv_bwd_rec(ctx, L + n1 * ld + n1, ld, invd + (n1 / NB) * NB * NB, z + n1, n2, tmp, part));
  // z1 -= L21^T z2 : deterministic column reduction of the (n2 x n1) block against z2, subtracted in its final pass
  GPX_TRY(launch_colreduce(ctx, L + n1 * ld, ld, n2, n1, z + n1, z, part, 1));
  return trsv_bwd_rec(ctx, L, ld, invd, z, n1, tmp, part);
}


// ---- potrs through block inverses: host side -----------------------------------------------------------------------------
static int64_t potrs_block(int64_t n) {
  // 1024 minimises build + one solve.  (Order 4096 -- the whole diagonal block of a look-ahead panel, its panel solve then ONE
  // long-K triangular product, potrs 1.75 ms instead of 2.5 -- was measured too: the factorisation loses 13 ms at N = 32768
  // and 3 ms at N = 8192, because a triangular product on few rows cannot use its shorter k ranges: 4096^3 gains 1.2x.  Round 6:
  // the 4096-order inverse for the rows BELOW block row k+1 only, built off the chain: profiles/r06_potrf_variants.txt.)
  int64_t ib = 1024;
  if (ib > n) ib = n;
  return ib;
}

// inverse of the sz x sz diagonal sub-block starting at row/column r0 of every batched block, from the two halves:
//   [A 0; C B]^-1 = [A^-1 0; -B^-1 C A^-1  B^-1]
// Sizes <= lo are taken as already inverted; sizes > hi are only descended into, not combined: the factorisation's chain
// (few CUs) builds up to hi = 1024, the main stream (whole chip) completes the upper levels with lo = 1024.
static int binv_build_rec(gpx_ctx* ctx, const double* L, int64_t ld, int64_t sl, double* binv, int64_t ib, int64_t r0,
                          int64_t sz, double* tmp, int64_t batch, int64_t lo = 0, int64_t hi = INT64_MAX) {
  if (sz <= NB || sz <= lo) return 0;
  const int64_t s1 = split(sz), s2 = sz - s1;
  GPX_TRY(binv_build_rec(ctx, L, ld, sl, binv, ib, r0, s1, tmp, batch, lo, hi));
  GPX_TRY(binv_build_rec(ctx, L, ld, sl, binv, ib, r0 + s1, s2, tmp, batch, lo, hi));
  if (sz > hi) return 0;
  const double* C = L + (r0 + s1) * ld + r0;
  const double* Ai = binv + r0 * ib + r0;
  const double* Bi = binv + (r0 + s1) * ib + r0 + s1;
  double* R = binv + (r0 + s1) * ib + r0;  // zero so far
  // T = C A^-1 ; R = 0 - B^-1 T
  if (batch == 1 && s1 >= 1024) {  // one large block: the tuned kernel, both inverses as triangular operands (A^-1: k range
    // of a column tile starts at its diagonal, mode 4; B^-1: ends at it, mode 1) -- n^3 / 3 for the whole recursion
    GPX_TRY(launch_gemm_tri(ctx, C, ld, Ai, ib, tmp, s1, s2, s1, s1, false, false, false, 4));
    return launch_gemm_tri(ctx, Bi, ib, tmp, s1, R, ib, s2, s1, s2, false, true, false, 1);
  }
  GPX_TRY(launch_gemm_batched(ctx, C, ld, sl, Ai, ib, ib * ib, tmp, s1, ib * ib, s2, s1, s1, false, false, batch));
  return launch_gemm_batched(ctx, Bi, ib, ib * ib, tmp, s1, ib * ib, R, ib, ib * ib, s2, s1, s2, false, true, batch);
}

// ONE block whose order is 128 times a power of two (the 512-order diagonal blocks of the distributed factorisation, a small
// factor's single block): level by level, every combine of a level in one batched launch -- 2 log2(w / 128) launches instead of
// 2 (w / 128 - 1) (512: 4 instead of 6), same products.  They sit on the diagonal chain of the distributed factorisation.
static int binv_build_levels(gpx_ctx* ctx, const double* L, int64_t ld, double* binv, int64_t ib, int64_t w, double* tmp,
                             int64_t lo, int64_t hi) {
  for (int64_t sz = 2 * NB; sz <= w; sz *= 2) {
    if (sz <= lo) continue;
    if (sz > hi) break;
    const int64_t s1 = sz / 2, nbat = w / sz, sl = sz * (ld + 1), sb = sz * (ib + 1);
    // T = C A^-1 ; R = 0 - B^-1 T   (binv_build_rec)
    GPX_TRY(launch_gemm_batched(ctx, L + s1 * ld, ld, sl, binv, ib, sb, tmp, s1, s1 * s1, s1, s1, s1, false, false, nbat));
    GPX_TRY(launch_gemm_batched(ctx, binv + s1 * (ib + 1), ib, sb, tmp, s1, s1 * s1, binv + s1 * ib, ib, sb, s1, s1, s1, false, true,
                                nbat));
  }
  return 0;
}

// explicit inverses of the ib-order diagonal blocks covering n rows (a multiple of 128; the last block may be shorter) that
// start at Ld on the diagonal; invd / binv point at that position too; tmp >= ceil(n / ib) * ib * ib doubles; (lo, hi): see
// binv_build_rec
static int binv_build_range(gpx_ctx* ctx, const double* Ld, int64_t ld, const double* invd, double* binv, int64_t ib, int64_t n,
                            double* tmp, int64_t lo, int64_t hi) {
  const int64_t nblk = (n + ib - 1) / ib, nfull = n / ib, tail = n - nfull * ib;
  const int64_t rows = nblk * ib;
  if (lo == 0)
    hipLaunchKernelGGL(binv_init_kernel, dim3((unsigned)((rows * (ib / 2) + 255) / 256)), dim3(256), 0, ctx->stream, invd, binv,
                       ib, n);
  if (nfull == 1 && tail == 0 && ib <= 1024 && ((ib / NB) & (ib / NB - 1)) == 0) {
    GPX_TRY(binv_build_levels(ctx, Ld, ld, binv, ib, ib, tmp, lo, hi));
    GPX_HIP(hipGetLastError());
    return 0;
  }
  if (nfull > 0) GPX_TRY(binv_build_rec(ctx, Ld, ld, ib * (ld + 1), binv, ib, 0, ib, tmp, nfull, lo, hi));
  if (tail > 0)
    GPX_TRY(binv_build_rec(ctx, Ld + nfull * ib * (ld + 1), ld, 0, binv + nfull * ib * ib, ib, 0, tail, tmp, 1, lo, hi));
  GPX_HIP(hipGetLastError());
  return 0;
}

// inv (w x w, row stride w, zero above the diagonal) = D^-1 for ONE factored diagonal block D (w x w, row stride ldd, leaf
// inverses at invd): the panel solve of the 2-D distributed factorisation multiplies with it instead of walking the
// leaf-level recursion (4 strip kernels + 3 short-K products per 512 columns: 56 ms per C4 factorisation on one rank).
// tmp >= w * w doubles.  Asynchronous on the selected stream.
int chol_block_inverse(gpx_ctx* ctx, const double* D, int64_t ldd, const double* invd, double* inv, int64_t w, double* tmp) {
  GPX_ARG(w > 0 && w % NB == 0, "block inverse: order must be a positive multiple of 128");
  return binv_build_range(ctx, D, ldd, invd, inv, w, w, tmp);
}

// Linv (n x n, row stride n, zero above the diagonal) = L^-1 for the whole factor, by the same halving recursion: n^3/2
// flops instead of the n^3 of a triangular solve against a dense identity.  tmp >= (n/2)^2 doubles.
int chol_trtri(gpx_ctx* ctx, const gpx_mat* Lm, double* Linv, double* tmp) {
  const int64_t n = Lm->prows;
  hipLaunchKernelGGL(binv_init_kernel, dim3((unsigned)((n * (n / 2) + 255) / 256)), dim3(256), 0, ctx->stream, Lm->aux, Linv, n, n);
  GPX_HIP(hipGetLastError());
  // n = 128 * 2^q: the levels up to order 1024 for ALL diagonal blocks at once, one batched pair of launches per level (6
  // launches instead of 2 (n / 128 - n / 1024): 112 at n = 8192, 1.5 ms of the mutual-information design's 6 ms inverse);
  // the recursion then only combines from 1024 up, with the triangular-operand products.
  if (n >= 2048 && ((n / NB) & (n / NB - 1)) == 0) {
    GPX_TRY(binv_build_levels(ctx, Lm->p, Lm->ld, Linv, n, n, tmp, 0, 1024));
    return binv_build_rec(ctx, Lm->p, Lm->ld, 0, Linv, n, 0, n, tmp, 1, 1024);
  }
  return binv_build_rec(ctx, Lm->p, Lm->ld, 0, Linv, n, 0, n, tmp, 1);
}

int64_t chol_binv_order(int64_t n) { return potrs_block(n); }
int64_t chol_binv_elems(int64_t n) {  // lower inverses + their transposes
  const int64_t ib = potrs_block(n);
  return 2 * ((n + ib - 1) / ib) * ib * ib;
}

// the transposed copies the backward sweep of potrs multiplies with; marks the cache valid
int chol_binv_finish(gpx_ctx* ctx, gpx_mat* Lm, int64_t ib) {
  const int64_t nblk = (Lm->prows + ib - 1) / ib, elems = nblk * ib * ib;
  dim3 gt((unsigned)(ib / 32), (unsigned)(ib / 32), (unsigned)nblk);
  hipLaunchKernelGGL(binv_transpose_kernel, gt, dim3(256), 0, ctx->stream, Lm->binv, Lm->binv + elems, ib);
  GPX_HIP(hipGetLastError());
  Lm->binv_ib = ib;
  return 0;
}

// out (n x n, row stride n) = in^T; n a multiple of 32
int chol_block_transpose(gpx_ctx* ctx, const double* in, double* out, int64_t n) {
  GPX_ARG(in && out && n > 0 && n % 32 == 0, "block transpose: order must be a multiple of 32");
  hipLaunchKernelGGL(binv_transpose_kernel, dim3((unsigned)(n / 32), (unsigned)(n / 32), 1), dim3(256), 0, ctx->stream, in, out, n);
  GPX_HIP(hipGetLastError());
  return 0;
}

// x = M y for a triangular sz x sz matrix (row stride ld): lower != 0: M lower triangular, else upper; x != y.  One wave per row.
int chol_tri_gemv(gpx_ctx* ctx, const double* M, int64_t ld, int64_t sz, const double* y, double* x, int lower) {
  GPX_ARG(M && y && x && x != y && sz > 0 && sz % NB == 0 && ld % 2 == 0, "triangular gemv: bad arguments");
  hipLaunchKernelGGL(binv_gemv_kernel, dim3((unsigned)((sz + 3) / 4)), dim3(256), 0, ctx->stream, M, ld, sz, y, x, lower);
  GPX_HIP(hipGetLastError());
  return 0;
}

int chol_binv_ensure(gpx_ctx* ctx, gpx_mat* Lm) {
  const int64_t n = Lm->prows, ib = potrs_block(n);
  if (Lm->binv && Lm->binv_ib == ib) return 0;
  const int64_t nblk = (n + ib - 1) / ib;
  const int64_t bytes = chol_binv_elems(n) * 8;
  if (Lm->binv && Lm->binv_bytes != bytes) {
    gpx_dev_release(ctx, Lm->binv, Lm->binv_bytes);
    Lm->binv = nullptr;
  }
  if (!Lm->binv) {
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, bytes, &p));
    Lm->binv = (double*)p;
    Lm->binv_bytes = bytes;
  }
  void* pt;
  GPX_TRY(gpx_dev_alloc(ctx, nblk * ib * ib * 8, &pt));  // products T, one slab per batched block
  int r = binv_build_range(ctx, Lm->p, Lm->ld, Lm->aux, Lm->binv, ib, n, (double*)pt);
  if (r == 0) r = chol_binv_finish(ctx, Lm, ib);
  (void)hipStreamSynchronize(ctx->stream);  // the product slabs go back to the pool
  gpx_dev_release(ctx, pt, nblk * ib * ib * 8);
  if (r != 0) Lm->binv_ib = 0;
  return r;
}

struct PotrsPlan {
  const double* L;
  int64_t ld, n, ib, nblk;  // ib = order of the diagonal blocks this sweep works with
  int64_t sib;              // order of the STORED inverses (a multiple of ib; row stride of the storage)
  const double* binv;       // [ceil(n/sib)][sib][sib]
  const double* binvT;
  double* part;             // colreduce partials
};
static inline int64_t blk_off(const PotrsPlan& P, int64_t b) { return b * P.ib < P.n ? b * P.ib : P.n; }
// inverse of diagonal block b: the diagonal ib x ib sub-block of the stored sib-order inverse that contains it (the inverse
// of a block-triangular matrix has the inverses of the diagonal blocks on its diagonal)
static inline const double* blk_inv(const PotrsPlan& P, const double* base, int64_t b) {
  const int64_t row = b * P.ib, sb = row / P.sib, within = row - sb * P.sib;
  return base + sb * P.sib * P.sib + within * P.sib + within;
}

// solve L x = rhs for the diagonal blocks [b0, b1): x (separate vector) receives the solution, rhs is consumed
static int potrs_fwd(gpx_ctx* ctx, const PotrsPlan& P, int64_t b0, int64_t b1, double* rhs, double* x) {
  if (b1 - b0 == 1) {
    const int64_t o = blk_off(P, b0), sz = blk_off(P, b0 + 1) - o;
    hipLaunchKernelGGL(binv_gemv_kernel, dim3((unsigned)((sz + 3) / 4)), dim3(256), 0, ctx->stream, blk_inv(P, P.binv, b0),
                       P.sib, sz, rhs + o, x + o, 1);
    return 0;
  }
  const int64_t mid = (b0 + b1) / 2;
  GPX_TRY(potrs_fwd(ctx, P, b0, mid, rhs, x));
  const int64_t r0 = blk_off(P, mid), r1 = blk_off(P, b1), c0 = blk_off(P, b0);
  int64_t wg = (r1 - r0 + 3) / 4;
  if (wg > 4096) wg = 4096;
  hipLaunchKernelGGL(gemv_sub_kernel, dim3((unsigned)wg), dim3(256), 0, ctx->stream, P.L + r0 * P.ld + c0, P.ld, r1 - r0, r0 - c0,
                     x + c0, rhs + r0);
  return potrs_fwd(ctx, P, mid, b1, rhs, x);
}

// solve L^T z = rhs
static int potrs_bwd(gpx_ctx* ctx, const PotrsPlan& P, int64_t b0, int64_t b1, double* rhs, double* z) {
  if (b1 - b0 == 1) {
    const int64_t o = blk_off(P, b0), sz = blk_off(P, b0 + 1) - o;
    hipLaunchKernelGGL(binv_gemv_kernel, dim3((unsigned)((sz + 3) / 4)), dim3(256), 0, ctx->stream, blk_inv(P, P.binvT, b0),
                       P.sib, sz, rhs + o, z + o, 0);
    return 0;
  }
  const int64_t mid = (b0 + b1) / 2;
  GPX_TRY(potrs_bwd(ctx, P, mid, b1, rhs, z));
  const int64_t r0 = blk_off(P, mid), r1 = blk_off(P, b1), c0 = blk_off(P, b0);
  // rhs[c0 : r0] -= L[r0 : r1, c0 : r0]^T z[r0 : r1]  (deterministic column reduction)
  GPX_TRY(launch_colreduce(ctx, P.L + r0 * P.ld + c0, P.ld, r1 - r0, r0 - c0, z + r0, rhs + c0, P.part, 1));
  return potrs_bwd(ctx, P, b0, mid, rhs, z);
}

// W = L^-1 B, out of place: W_b = Binv_b B_b (lower-triangular operand: half the k range per row tile), then the block
// rows below take B -= L[below, b] W_b.  Every update has K >= the inverse order (1024): none of the K = 128..512 products
// of the leaf-level recursion (22-60 TF/s at C4), and the diagonal solves run as chip-filling GEMMs.
static int trsm_left_oop_rec(gpx_ctx* ctx, const PotrsPlan& P, int64_t b0, int64_t b1, double* B, int64_t ldb, double* W,
                             int64_t ldw, int64_t m) {
  if (b1 - b0 == 1) {
    const int64_t o = blk_off(P, b0), sz = blk_off(P, b0 + 1) - o;
    return launch_gemm_tri(ctx, blk_inv(P, P.binv, b0), P.sib, B + o * ldb, ldb, W + o * ldw, ldw, sz, m, sz, false, false,
                           false, 1);
  }
  const int64_t mid = (b0 + b1) / 2;
  GPX_TRY(trsm_left_oop_rec(ctx, P, b0, mid, B, ldb, W, ldw, m));
  const int64_t r0 = blk_off(P, mid), r1 = blk_off(P, b1), c0 = blk_off(P, b0);
  GPX_TRY(launch_gemm(ctx, P.L + r0 * P.ld + c0, P.ld, W + c0 * ldw, ldw, B + r0 * ldb, ldb, r1 - r0, m, r0 - c0, false, true,
                      false));
  return trsm_left_oop_rec(ctx, P, mid, b1, B, ldb, W, ldw, m);
}

int chol_trsm_left_oop(gpx_ctx* ctx, gpx_mat* Lm, double* B, int64_t ldb, double* W, int64_t ldw, int64_t m) {
  GPX_ARG(Lm && Lm->factored && Lm->aux && B && W && B != W, "trsm: matrix has not been factored / bad buffers");
  if (m == 0) return 0;
  GPX_TRY(chol_binv_ensure(ctx, Lm));
  PotrsPlan P;
  P.L = Lm->p;
  P.ld = Lm->ld;
  P.n = Lm->prows;
  P.sib = Lm->binv_ib;
  P.ib = P.sib < 1024 ? P.sib : 1024;  // the diagonal solves are products with a triangular operand: order 1024 keeps the
  P.nblk = (P.n + P.ib - 1) / P.ib;    // extra flops (those of the inverse's sub-diagonal part) at 3 % of the solve
  P.binv = Lm->binv;
  P.binvT = nullptr;
  P.part = nullptr;
  return trsm_left_oop_rec(ctx, P, 0, P.nblk, B, ldb, W, ldw, m);
}

int chol_trsm_left_group(gpx_ctx* ctx, const double* Lg, int64_t ld, const double* invd, int64_t w, int64_t below, int64_t ib,
                         double* B, int64_t ldb, int64_t m, double* inv, double* tmp, double* W, int64_t ldw) {
  GPX_ARG(Lg && invd && B && inv && tmp && W && W != B && w > 0 && w % NB == 0 && ib >= NB && ib % NB == 0, "trsm group: bad arguments");
  if (m == 0) return 0;
  GPX_TRY(binv_build_range(ctx, Lg, ld, invd, inv, ib, w, tmp));
  PotrsPlan P;
  P.L = Lg;
  P.ld = ld;
  P.n = w;
  P.ib = P.sib = ib;
  P.nblk = (w + ib - 1) / ib;
  P.binv = inv;
  P.binvT = nullptr;
  P.part = nullptr;
  GPX_TRY(trsm_left_oop_rec(ctx, P, 0, P.nblk, B, ldb, W, ldw, m));
  if (below > 0) GPX_TRY(launch_gemm(ctx, Lg + w * ld, ld, W, ldw, B + w * ldb, ldb, below, m, w, false, true, false));
  return gpx_copy2d(ctx, W, ldw, B, ldb, w, m);   // the solved rows belong into B (the variances sum over all of its rows)
}

int64_t chol_potrs_scratch_bytes(int64_t n) { return n * 8 + colreduce_partial_elems(n, n) * 8 + 64; }

int chol_potrs(gpx_ctx* ctx, gpx_mat* Lm, double* v, double* scratch) {
  GPX_ARG(Lm && Lm->factored && Lm->aux && v && scratch, "potrs: matrix has not been factored / NULL vector");
  GPX_ARG(Lm->ld % 2 == 0, "potrs: leading dimension must be even");
  GPX_TRY(chol_binv_ensure(ctx, Lm));
  PotrsPlan P;
  P.L = Lm->p;
  P.ld = Lm->ld;
  P.n = Lm->prows;
  P.ib = P.sib = Lm->binv_ib;
  P.nblk = (P.n + P.ib - 1) / P.ib;
  P.binv = Lm->binv;
  P.binvT = Lm->binv + P.nblk * P.ib * P.ib;
  P.part = scratch + P.n;
  double* x = scratch;
  ProfScope ps(ctx, GPX_PROF_TRSV, 2.0 * (double)P.n * P.n, 8.0 * (double)P.n * P.n);
  GPX_TRY(potrs_fwd(ctx, P, 0, P.nblk, v, x));   // v consumed, w = L^-1 y in x
  GPX_TRY(potrs_bwd(ctx, P, 0, P.nblk, x, v));   // x consumed, alpha in v
  GPX_HIP(hipGetLastError());
  return 0;
}

int64_t chol_trsv_scratch_bytes(int64_t n) { return n * 8 + colreduce_partial_elems(n, n) * 8 + 64; }

// asynchronous on the selected stream; scratch (chol_trsv_scratch_bytes) is only used by the transposed sweep
int chol_trsv_with_scratch(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* y, int64_t n,
                           bool transposed, double* scratch) {
  GPX_ARG(n % NB == 0 && n > 0, "trsv: padded length must be a positive multiple of 128");
  GPX_ARG(ld % 2 == 0, "trsv: leading dimension must be even");
  ProfScope ps(ctx, GPX_PROF_TRSV, (double)n * n, 4.0 * (double)n * n);
  int r = transposed ? trsv_bwd_rec(ctx, L, ld, invd, y, n, scratch, scratch + n) : trsv_fwd_rec(ctx, L, ld, invd, y, n);
  if (r != 0) return r;
  GPX_HIP(hipGetLastError());
  return 0;
}

int chol_trsv(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* y, int64_t n, bool transposed) {
  if (!transposed) return chol_trsv_with_scratch(ctx, L, ld, invd, y, n, false, nullptr);
  void* p;
  const int64_t b = chol_trsv_scratch_bytes(n);
  GPX_TRY(gpx_dev_alloc(ctx, b, &p));
  int r = chol_trsv_with_scratch(ctx, L, ld, invd, y, n, true, (double*)p);
  (void)hipStreamSynchronize(ctx->stream);
  gpx_dev_release(ctx, p, b);
  return r;
}

int launch_gemv_sub(gpx_ctx* ctx, const double* A, int64_t ld, int64_t rows, int64_t cols, const double* x, double* y) {
  if (rows <= 0 || cols <= 0) return 0;
  GPX_ARG(cols % 2 == 0 && ld % 2 == 0, "gemv_sub: even column count and leading dimension");
  int64_t wg = (rows + 3) / 4;
  if (wg > 4096) wg = 4096;
  ProfScope ps(ctx, GPX_PROF_TRSV, 2.0 * (double)rows * cols, 8.0 * (double)rows * cols);
  hipLaunchKernelGGL(gemv_sub_kernel, dim3((unsigned)wg), dim3(256), 0, ctx->stream, A, ld, rows, cols, x, y);
  GPX_HIP(hipGetLastError());
  return 0;
}

int launch_logdet(gpx_ctx* ctx, const double* L, int64_t ld, int64_t n, double* d_out) {
  ProfScope ps(ctx, GPX_PROF_REDUCE, 0.0, 8.0 * (double)n);
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, ctx->stream, L, ld, n, d_out);
  GPX_HIP(hipGetLastError());
  return 0;
}
