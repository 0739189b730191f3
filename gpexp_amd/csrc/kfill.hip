// kfill: dense covariance assembly  K[i][j] = k(A_i, B_j) (+ nugget on the diagonal)  -- gfx950.
//
// Replaces the reference's N-iteration Python row loop (gp_kernel_utilities.py:56-60, one
// Kernel.evaluate + np.tile per row) and the column loops that build kernelvals
// (gp.py:132-135, 246-249; experimentalDesign.py:829-831).
//
// HBM-bound by intent: algorithmic bytes = 8*rows*cols written + 8*(rows+cols)*d read.  A pure store kernel with this
// tile pattern reaches 5.6 TB/s on MI355X (scripts/store_peak.hip; hipMemset 6.3 TB/s).  What it takes to get near
// that in fp64 (no hardware exp/sqrt):
//   * 64x64 tiles.  The whole exponent argument comes out of the fp64 MFMA pipe: the staged operands are augmented,
//     A' = [a*scale, |a|^2, 1], B' = [-2 b*scale, 1, |b|^2], so A'.B' = |a-b|^2 (Mehler: [c2 a, pa, 1].[-b, 1, pb] =
//     pa + pb - cross).  The VALU only evaluates the kernel function.  Diagonal entries keep an exact zero distance.
//   * staging is ONE global round trip per tile (4 threads per point, norms reduced with DPP shuffles): under a
//     saturated store queue every dependent load costs microseconds of workgroup lifetime, i.e. occupancy.
//   * symmetric fill: only tiles on/below the diagonal are COMPUTED; each is also written transposed through a
//     padded LDS image (coalesced mirror stores) -- half the exp/sqrt work for the same 8*N^2 bytes.
//   * interior tiles (no padding, no diagonal) take a branch-free path selected by a scalar branch.
//   * placement independence (KParams in gpx_internal.h): coordinates are centred on the bounding-box midpoint before
//     scaling, and fills whose centred, scaled domain is still wide take the EXACT variant, which forms
//     (a_k - b_k) * scale_k from the raw coordinates on the VALU -- the reference's own order of operations
//     (kernels.py:121-122) -- instead of the expanded product.
//   * exp: argument clamped at -750, shifter-trick rounding + one-constant reduction to |r| <= ln2/512 + 256-entry
//     2^(j/256) table in LDS + degree-4 polynomial (truncation 3.8e-17) + v_ldexp; sqrt: v_rsq_f64 (2^-23) + one coupled
//     Newton step + residual correction.  Both stay within 2 ulp (tests: 1e-13 against the oracle / the reference).
#include "gpx_internal.h"
#include <math.h>
#include <stdlib.h>

namespace {

constexpr int TM = 64;  // tile rows
constexpr int TN = 64;  // tile cols
constexpr int TP = TN + 1;  // padded stride of the transpose image
struct KCyclic { int nbt, Pr, pr, Pc, pc; };  // nbt = block size in tiles; 0 = plain (identity) tile map
#ifndef WAVES_PER_EU
#define WAVES_PER_EU 5
#endif
#ifndef WAVES_PER_EU_RECT
#define WAVES_PER_EU_RECT 6
#endif
#ifndef RECT_NT
#define RECT_NT 4   // column tiles per workgroup of kfill_rectn_kernel (measured 2 / 3 / 4: 1.76 / 1.63 / 1.61 ms, one call)
#endif

// exp(x) = 2^n * 2^(j/256) * exp(r):  m = rint(x * 256/ln2), n = m >> 8, j = m & 255, r = x - m*ln2/256, |r| <= ln2/512, so
// a degree-4 polynomial is exact to 3.8e-17; tab[j] = 2^(j/256) sits in LDS (2 KB).  m comes from the add-and-subtract-
// 1.5*2^52 trick: the rounded integer is also the low dword of the shifted sum, so there is no v_rndne / v_cvt.  The
// reduction uses ONE constant: the fma forms m*C exactly, so the only error is m * (C - ln2/256) <= |x| * 8e-17, below the
// rounding of the argument itself.  The integer trick holds for |x| < 2^31 * ln2/256 = 5.8e6, so the argument is clamped
// at -750 first (exp(-750) already underflows to 0; a point pair 1e5 length scales apart must give 0, not garbage from a
// wrapped exponent).  10 fp64 ops + 4 integer ops.  (Rounds 1-2a used 64 entries and degree 5: one fma more per element --
// the rectangular Matern fill is VALU-bound, profiles/r02_kfill_valu.txt.)
constexpr int EXP_TAB = 256;
__device__ __forceinline__ double vmax1(double a, double b);
__device__ __forceinline__ double vmax1_neg(double a, double b);
template <bool NEG = false>  // NEG: exp(-xin), the sign folded into the clamp's source modifier
__device__ __forceinline__ double fast_exp(double xin, const double* __restrict__ tab) {
  const double SHIFT = 6755399441055744.0;                    // 1.5 * 2^52
  const double x = NEG ? vmax1_neg(xin, -750.0) : vmax1(xin, -750.0);
  const double sh = fma(x, 369.3299304675746, SHIFT);          // 256 / ln 2
  const int mi = __double2loint(sh);
  const double m = sh - SHIFT;
  const double r = fma(m, -0.0027076061740622863, x);         // ln2 / 256
  const double tj = tab[mi & (EXP_TAB - 1)];
  double p = 4.1666666666666664354e-02;                       // 1/24
  p = fma(p, r, 1.6666666666666665741e-01);                   // 1/6
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(tj * p, mi >> 8);
}

// max(a, b) as ONE v_max_f64: fmax() compiles to a canonicalising v_max(a, a) in front of the real one (a comes out of the
// MFMA, the compiler cannot know it is no signalling NaN) -- 2 of the 26 VALU instructions per element of a Matern fill
__device__ __forceinline__ double vmax1(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}
__device__ __forceinline__ double vmax1_neg(double a, double b) {  // max(-a, b)
  double r;
  asm("v_max_f64 %0, -%1, %2" : "=v"(r) : "v"(a), "s"(b));
  return r;
}

// smallest squared scaled distance the Matern kinds evaluate: coincident points (exactly 0) and the expanded form's negative
// round-off are clamped to it instead of a compare + two selects; sqrt(1e-300) = 1e-150 leaves every kernel value
// bit-identical to the r = 0 result (1 + 1e-150 == 1, exp(-1e-150) == 1, 1e-300 * sig/3 vanishes).
constexpr double DIST2_MIN = 1e-300;

__device__ __forceinline__ double fast_sqrt(double x) {  // DIST2_MIN <= x (clamped by the caller)
  // v_rsq_f64 is good to 2^-23; one coupled (Goldschmidt) step squares that to ~2^-45 and the residual correction
  // g += (x - g*g) * h is a further Newton step for the root itself: < 1 ulp.
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  const double d = fma(-g, g, x);  // the correction below is second order: h at 2^-23 is plenty
  return fma(d, h, g);
}

// accin = the exponent argument straight from the MFMA (squared scaled distance, or the Mehler exponent); sig3 = sig/3.
// The expanded form's round-off can leave a squared distance at -1e-16: harmless under exp (SE), clamped for the root.
template <int KIND>
__device__ __forceinline__ double kvalue(double accin, double sig, double sig3, const double* __restrict__ tab) {
  if (KIND == GPX_K_SE) {
    return sig * fast_exp(-0.5 * accin, tab);
  } else if (KIND == GPX_K_MATERN32) {
    const double t = fast_sqrt(vmax1(accin, DIST2_MIN));
    return fma(t, sig, sig) * fast_exp<true>(t, tab);
  } else if (KIND == GPX_K_MATERN52) {
    const double acc = vmax1(accin, DIST2_MIN);
    const double t = fast_sqrt(acc);
    return fma(acc, sig3, fma(t, sig, sig)) * fast_exp<true>(t, tab);
  } else {  // Mehler
    return sig * fast_exp<true>(accin, tab);
  }
}

// 2^(j/256), j = 0..255 (correctly rounded; generated with 60-digit decimal arithmetic)
__device__ const double kExp2Tab[EXP_TAB] = {
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175, 1.0108892860517005, 1.0136300849514894,
    1.016378314910953, 1.019133996077738, 1.0218971486541166, 1.0246677928971357, 1.0274459491187637,
    1.030231637686041, 1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264, 1.0556451783605572,
    1.0585073227945128, 1.061377227289262, 1.0642549128844645, 1.0671404006768237, 1.0700337118202419,
    1.0729348675259756, 1.075843889062791, 1.0787607977571199, 1.0816856149932152, 1.0846183622133092,
    1.0875590609177697, 1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175, 1.1143867425958924,
    1.1174081515673693, 1.1204377524096067, 1.12347556733302, 1.1265216186082418, 1.129575928566288,
    1.1326385195987192, 1.1357094141578055, 1.1387886347566916, 1.1418762039695616, 1.1449721444318042,
    1.148076478840179, 1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373, 1.1763969916502812,
    1.1795865274628758, 1.182784710984341, 1.1859915656609938, 1.189207115002721, 1.1924313825831512,
    1.1956643920398273, 1.1989061670743806, 1.202156731452703, 1.2054161090051239, 1.2086843236265816,
    1.2119613992768012, 1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165, 1.241857812073484,
    1.245224830175258, 1.2486009771892048, 1.2519862778663162, 1.255380757024691, 1.2587844395497165,
    1.2621973503942507, 1.2656195145788063, 1.2690509571917332, 1.2724917033894028, 1.275941778396392,
    1.2794012075056693, 1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773, 1.3109612115247644,
    1.3145155879493546, 1.318079601266064, 1.3216532776031575, 1.3252366431597413, 1.3288297242059544,
    1.3324325470831615, 1.3360451382041458, 1.339667524053303, 1.3432997311868353, 1.3469417862329458,
    1.3505937158920345, 1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238, 1.383909881963832,
    1.387662042298529, 1.3914243757719262, 1.3951969099662003, 1.3989796725383112, 1.4027726912202048,
    1.4065759938190154, 1.4103896082172707, 1.4142135623730951, 1.4180478843204152, 1.4218926021691656,
    1.4257477441054942, 1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438, 1.460917794180647,
    1.4648787441464057, 1.4688504333369818, 1.4728328908693675, 1.4768261459394993, 1.4808302278224719,
    1.4848451658727524, 1.488870989524397, 1.4929077282912648, 1.4969554117672355, 1.5010140696264256,
    1.5050837316234065, 1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568, 1.5422108254079407,
    1.5463921831410214, 1.550584877685, 1.5547889397770887, 1.559004400237837, 1.5632312899713576, 1.567469639965553,
    1.5717194812923414, 1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771, 1.6104903319492543,
    1.6148568142048607, 1.6192351351948637, 1.6236253270173289, 1.6280274218573478, 1.632441451987275,
    1.6368674497669644, 1.6413054476440063, 1.645755478153965, 1.6502175739206177, 1.6546917676561943,
    1.6591780921616162, 1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326, 1.7001063537185235,
    1.7047158096580513, 1.709337763100463, 1.713972247929926, 1.718619298122478, 1.723278947746274,
    1.7279512309618377, 1.732636182022311, 1.7373338352737062, 1.7420442251551564, 1.746767386199169,
    1.7515033530318782, 1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401, 1.7947090750031072,
    1.7995750249405351, 1.804454167806624, 1.809346539371032, 1.8142521755003989, 1.8191711121586085,
    1.8241033854070534, 1.8290490314048973, 1.8340080864093424, 1.8389805867758937, 1.843966568958626,
    1.8489660695104508, 1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392, 1.8945759815869656,
    1.8997126981765553, 1.9048633418176741, 1.9100279502703899, 1.9152065613971474, 1.9203992131630474,
    1.925605943636125, 1.930826790987627, 1.9360617934922943, 1.9413109895286405, 1.9465744175792332,
    1.9518521162309783, 1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402};

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int MAXK4 = (GPX_MAXD + 2 + 3) / 4;  // most k slots a staging thread can own

// Stage one 64-point tile into LDS as the augmented MFMA operand (see the header): 4 threads per point, thread c takes
// coordinates c, c+4, ...; every global load of the tile is issued before the first use, the point's own exponent term
// is reduced over the 4 threads with shuffles and lands in slot d (A side) / d+1 (B side), the other slot holds 1.
template <int KIND, bool SIDE_A, int K4>
__device__ __forceinline__ void stage_points(const KParams& kp, int d, int dpad, int sl, const double* __restrict__ X,
                                             int64_t n, int64_t g0, double* P) {
  const int t = threadIdx.x, p = t >> 2, c = t & 3;
  const int64_t g = g0 + p;
  const bool inb = g < n;
  const double* xp = X + g * d;
  double x[K4];
#pragma unroll
  for (int i = 0; i < K4; ++i) {
    const int k = c + 4 * i;
    x[i] = (inb && k < d) ? xp[k] : 0.0;
  }
  double part = 0.0;
#pragma unroll
  for (int i = 0; i < K4; ++i) {
    const int k = c + 4 * i;
    if (k < d) {
      if (KIND == GPX_K_MEHLER) {
        part = fma(kp.c1[k] * x[i], x[i], part);
        x[i] = SIDE_A ? kp.c2[k] * x[i] : -x[i];
      } else {
        const double w = (x[i] - kp.center[k]) * kp.scale[k];
        part = fma(w, w, part);
        x[i] = SIDE_A ? w : -2.0 * w;
      }
    }
  }
  part += __shfl_xor(part, 1);
  part += __shfl_xor(part, 2);
  const double own = inb ? part : 0.0, one = 1.0;
#pragma unroll
  for (int i = 0; i < K4; ++i) {
    const int k = c + 4 * i;
    if (k < dpad) {
      double v = x[i];
      if (k == d) v = SIDE_A ? own : one;
      if (k == d + 1) v = SIDE_A ? one : own;
      P[p * sl + k] = v;
    }
  }
}

// EXACT variant: raw coordinates of a 64-point tile -> LDS ([point][k], odd stride), zeros beyond the set
__device__ __forceinline__ void stage_raw(int d, int sl, const double* __restrict__ X, int64_t n, int64_t g0, double* P) {
  for (int idx = threadIdx.x; idx < TM * d; idx += 256) {
    const int p = idx / d, k = idx - p * d;
    const int64_t g = g0 + p;
    P[p * sl + k] = g < n ? X[g * d + k] : 0.0;
  }
}

// acc[mi][ni][v] = sum_k ((a[r][k] - b[c][k]) * scale_k)^2 for the lane's rows r = rbase + 16 mi + 4 v and columns c0 + ni:
// the coordinate difference is taken FIRST, as the reference does (kernels.py:121-122, 87), so nearby points far from the
// origin or in a wide domain keep their full relative accuracy
__device__ __forceinline__ void exact_dist(const KParams& kp, int d, int sl, const double* __restrict__ As,
                                           const double* __restrict__ Bs, int rbase, int c0, d4 (&acc)[2][2]) {
  const double* ap = As + rbase * sl;
  const double* bp = Bs + c0 * sl;
  for (int k = 0; k < d; ++k) {
    const double sc = kp.scale[k];
    const double b0 = bp[k], b1 = bp[sl + k];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const double a = ap[(mi * 16 + 4 * v) * sl + k];
        const double e0 = (a - b0) * sc, e1 = (a - b1) * sc;
        acc[mi][0][v] = fma(e0, e0, acc[mi][0][v]);
        acc[mi][1][v] = fma(e1, e1, acc[mi][1][v]);
      }
  }
}

// The 16 kernel values of one lane (rows r(mi, v) = wm*32 + mi*16 + g + 4v, columns c0, c0+1) from the MFMA
// accumulators, stored straight to the tile.  INTERIOR = the tile lies fully inside both point sets and does not touch
// the diagonal: no padding, nugget or exact-zero handling.
template <int KIND, bool INTERIOR>
__device__ __forceinline__ void finish_tile(const d4 (&acc)[2][2], double2 (&val)[2][4], double sig,
                                            const double* __restrict__ tab, int64_t i0, int64_t j0, int64_t na,
                                            int64_t nb, int symmetric, const double* __restrict__ nugget,
                                            int64_t nugget_len, double nugget_scalar, char* otile, unsigned ooff,
                                            int64_t ld, int rbase, int c0, bool diag_tile, int64_t row_shift) {
  const double sig3 = sig * (1.0 / 3.0);
  const int64_t gj0 = j0 + c0, gj1 = gj0 + 1;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double s0 = acc[mi][0][v], s1 = acc[mi][1][v];  // squared scaled distance; kvalue deals with negative round-off
      const int64_t gi = i0 + rbase + mi * 16 + 4 * v;
      const int64_t gd = gi + row_shift;  // this row's diagonal column
      if (!INTERIOR && symmetric && KIND != GPX_K_MEHLER) {  // exact zero distance on the diagonal, as the reference has it
        if (gd == gj0) s0 = 0.0;
        if (gd == gj1) s1 = 0.0;
      }
      double v0 = kvalue<KIND>(s0, sig, sig3, tab), v1 = kvalue<KIND>(s1, sig, sig3, tab);
      if (!INTERIOR) {  // edge tiles (padding) and tiles crossing the diagonal (nugget)
        const bool rin = gi < na;
        if (!(rin && gj0 < nb)) v0 = (symmetric && gd == gj0) ? 1.0 : 0.0;
        if (!(rin && gj1 < nb)) v1 = (symmetric && gd == gj1) ? 1.0 : 0.0;
        if (symmetric && rin && nugget_len > 0) {
          const double nz = nugget_len == 1 ? nugget_scalar : nugget[gd];
          if (gd == gj0) v0 += nz;
          if (gd == gj1) v1 += nz;
        }
      }
      val[mi][v].x = v0;
      val[mi][v].y = v1;
      // address = (wave-uniform row-group base, SGPRs) + (per-lane 32-bit byte offset): no 64-bit VALU address math
      double* const dst = reinterpret_cast<double*>(otile + (int64_t)(mi * 16 + 4 * v) * ld * 8 + ooff);
      if (!INTERIOR && diag_tile) {
        // the augmented dot product adds |a|^2 and |b|^2 in operand order, so V[r][c] and V[c][r] can differ in the
        // last bit: the diagonal tile of the mirrored fill stores its lower half here and mirrors it like any other
        if (gd >= gj0) dst[0] = v0;
        if (gd >= gj1) dst[1] = v1;
      } else {
        *reinterpret_cast<double2*>(dst) = val[mi][v];
      }
    }
}

// 64x64 tile per 256-thread workgroup; wave (wm, wn) owns a 32x32 quadrant = 2x2 fp64 MFMA 16x16 blocks.
// MFMA column q of block ni is mapped to tile column 2q+ni, so a lane's two blocks are ADJACENT columns: 16-byte
// stores, 256-byte row segments per 16-lane group.
// SYM: 1-D grid over the tiles on/below the diagonal, mirror-written; otherwise 2-D grid over all tiles.
// K4 = augmented K (d + 2) rounded up to the MFMA step, in units of 4: compile-time so staging holds K4 registers a side.
// EXACT: distances from raw coordinate differences on the VALU (K4 unused), see exact_dist.
template <int KIND, bool SYM, int K4, bool EXACT>
__global__ __launch_bounds__(256, SYM ? WAVES_PER_EU : WAVES_PER_EU_RECT) void kfill_kernel(KParams kp, const double* __restrict__ A, int64_t na,
                                                    const double* __restrict__ B, int64_t nb, int symmetric,
                                                    const double* __restrict__ nugget, int64_t nugget_len,
                                                    double nugget_scalar, double* __restrict__ out, int64_t ld,
                                                    int64_t row_shift, KCyclic cyc) {
  extern __shared__ double sm[];
  const int d = kp.d;
  constexpr int dpad = 4 * K4;  // coordinates + the two augmentation slots, padded to the MFMA K step
  const int sl = EXACT ? (d | 1) : dpad + 1;  // odd stride
  double* As = sm;
  double* Bs = As + TM * sl;
  double* tab = Bs + TN * sl;  // 2^(j/256) table
  double* Tr = tab + EXP_TAB;  // [32][TP] transpose image (SYM only)
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
  // i0, j0: first point of the tile's row / column set; oi0, oj0: where the tile lands in `out`.  They differ only for the
  // 2-D block-cyclic local matrices of the multi-GPU path (cyc.nbt > 0): local block (ti / nbt, tj / nbt) holds the global
  // block (.. * Pr + pr, .. * Pc + pc); tiles strictly above the global diagonal are not needed there.
  const int64_t oi0 = (int64_t)ti * TM, oj0 = (int64_t)tj * TN;
  int64_t i0 = oi0, j0 = oj0;
  if (!SYM && cyc.nbt > 0) {
    i0 = ((int64_t)(ti / cyc.nbt) * cyc.Pr + cyc.pr) * cyc.nbt * TM + (int64_t)(ti % cyc.nbt) * TM;
    j0 = ((int64_t)(tj / cyc.nbt) * cyc.Pc + cyc.pc) * cyc.nbt * TN + (int64_t)(tj % cyc.nbt) * TN;
    if (i0 + TM <= j0) return;
  }
  if (EXACT) {
    stage_raw(d, sl, A, na, i0, As);
    stage_raw(d, sl, B, nb, j0, Bs);
  } else {
    stage_points<KIND, true, K4>(kp, d, dpad, sl, A, na, i0, As);
    stage_points<KIND, false, K4>(kp, d, dpad, sl, B, nb, j0, Bs);
  }
  tab[threadIdx.x] = kExp2Tab[threadIdx.x];  // EXP_TAB == blockDim.x
  __syncthreads();

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, q = lane & 15;

  d4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
  const int c0 = wn * 32 + 2 * q;
  if (EXACT) {
    exact_dist(kp, d, sl, As, Bs, wm * 32 + g, c0, acc);
  } else {
    const double* ap = As + (wm * 32 + q) * sl + g;           // A'[row = mi*16 + q][k = 4s + g]
    const double* bp = Bs + (wn * 32 + 2 * q) * sl + g;       // B'[col = 2q + ni][k = 4s + g]
#pragma unroll
    for (int ks = 0; ks < dpad; ks += 4) {
      const double a0 = ap[ks], a1 = ap[16 * sl + ks];
      const double b0 = bp[ks], b1 = bp[sl + ks];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // row_shift: the row set starts `row_shift` entries into the column set (row-band refill): diagonal at gj == gi + shift
  const int64_t is0 = i0 + row_shift;
  const bool interior = (i0 + TM <= na) && (j0 + TN <= nb) && (!symmetric || (is0 + TM <= j0) || (j0 + TN <= is0));
  char* const otile = reinterpret_cast<char*>(out + oi0 * ld + oj0);
  const unsigned ooff = (unsigned)(((wm * 32 + g) * ld + c0) * 8);
  double2 val[2][4];
  if (__builtin_amdgcn_readfirstlane((int)interior))
    finish_tile<KIND, true>(acc, val, kp.sig, tab, i0, j0, na, nb, symmetric, nugget, nugget_len, nugget_scalar, otile,
                            ooff, ld, wm * 32 + g, c0, false, row_shift);
  else
    finish_tile<KIND, false>(acc, val, kp.sig, tab, i0, j0, na, nb, symmetric, nugget, nugget_len, nugget_scalar, otile,
                             ooff, ld, wm * 32 + g, c0, SYM && ti == tj, row_shift);

  if (SYM) {
    // mirror: out[j0 + c][i0 + r] = V[r][c]; the two row halves (wm = 0 / 1) go through the padded image in turn
    // (the diagonal tile mirrors its strictly-lower part only)
    const int tx = t & 31, ty = t >> 5;
    char* const mbase = reinterpret_cast<char*>(out + j0 * ld + i0);
    const unsigned moff = (unsigned)((ty * ld + tx) * 8);
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
      if (ti != tj) {
#pragma unroll
        for (int qq = 0; qq < 8; ++qq)
          *reinterpret_cast<double*>(mbase + (int64_t)(8 * qq) * ld * 8 + moff + 256 * half) = Tr[tx * TP + ty + 8 * qq];
      } else {
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
          const int c = ty + 8 * qq;
          if (32 * half + tx > c) out[(j0 + c) * ld + i0 + 32 * half + tx] = Tr[tx * TP + c];
        }
      }
    }
  }
}

// Rectangular fills of the Matern kinds: the same tile code, NT = RECT_NT neighbouring column tiles per workgroup.  Those fills
// are VALU-bound (sqrt + exp for every element, profiles/r02_kfill_valu.txt), so every instruction outside the kernel function
// counts: the row points are staged once and the workgroup's fixed costs (launch, table, barrier) are paid once per NT * 64
// columns: 1.94 ms per 8.6 GB with one tile, 1.76 with two, 1.61 with four.  The store-bound kinds (SE, Mehler: 1.50-1.58 ms)
// lose occupancy to it and keep the one-tile kernel.
template <int KIND, int K4, bool EXACT>
__global__ __launch_bounds__(256, WAVES_PER_EU_RECT) void kfill_rectn_kernel(KParams kp, const double* __restrict__ A, int64_t na,
                                                    const double* __restrict__ B, int64_t nb, int symmetric,
                                                    const double* __restrict__ nugget, int64_t nugget_len,
                                                    double nugget_scalar, double* __restrict__ out, int64_t ld,
                                                    int64_t row_shift, KCyclic cyc, int tiles_n) {
  extern __shared__ double sm[];
  constexpr bool SYM = false;
  constexpr int NT = RECT_NT;
  const int d = kp.d;
  constexpr int dpad = 4 * K4;  // coordinates + the two augmentation slots, padded to the MFMA K step
  const int sl = EXACT ? (d | 1) : dpad + 1;  // odd stride
  double* As = sm;
  double* Bs = As + TM * sl;            // NT images
  double* tab = Bs + NT * TN * sl;      // 2^(j/256) table
  double* Tr = tab + EXP_TAB;           // [32][TP] transpose image (SYM only)
  int ti, tj0;
  if (SYM) {
    const int w = blockIdx.x;
    ti = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= w) ++ti;
    while (ti * (ti + 1) / 2 > w) --ti;
    tj0 = w - ti * (ti + 1) / 2;
  } else {
    ti = blockIdx.y;
    tj0 = NT * blockIdx.x;
  }
  // i0, j0: first point of the tile's row / column set; oi0, oj0: where the tile lands in `out`.  They differ only for the
  // 2-D block-cyclic local matrices of the multi-GPU path (cyc.nbt > 0): local block (ti / nbt, tj / nbt) holds the global
  // block (.. * Pr + pr, .. * Pc + pc); tiles strictly above the global diagonal are not needed there.
  const int64_t oi0 = (int64_t)ti * TM;
  int64_t i0 = oi0;
  if (!SYM && cyc.nbt > 0) i0 = ((int64_t)(ti / cyc.nbt) * cyc.Pr + cyc.pr) * cyc.nbt * TM + (int64_t)(ti % cyc.nbt) * TM;
  int64_t oj0s[NT], j0s[NT];
  bool live[NT];
  bool any = false;
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    const int tj = tj0 + it;
    oj0s[it] = (int64_t)tj * TN;
    j0s[it] = oj0s[it];
    live[it] = SYM || tj < tiles_n;
    if (!SYM && cyc.nbt > 0) {
      j0s[it] = ((int64_t)(tj / cyc.nbt) * cyc.Pc + cyc.pc) * cyc.nbt * TN + (int64_t)(tj % cyc.nbt) * TN;
      if (i0 + TM <= j0s[it]) live[it] = false;
    }
    any = any || live[it];
  }
  if (!any) return;
  if (EXACT) {
    stage_raw(d, sl, A, na, i0, As);
#pragma unroll
    for (int it = 0; it < NT; ++it)
      if (live[it]) stage_raw(d, sl, B, nb, j0s[it], Bs + it * TN * sl);
  } else {
    stage_points<KIND, true, K4>(kp, d, dpad, sl, A, na, i0, As);
#pragma unroll
    for (int it = 0; it < NT; ++it)
      if (live[it]) stage_points<KIND, false, K4>(kp, d, dpad, sl, B, nb, j0s[it], Bs + it * TN * sl);
  }
  tab[threadIdx.x] = kExp2Tab[threadIdx.x];  // EXP_TAB == blockDim.x
  __syncthreads();

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, q = lane & 15;
  const int c0 = wn * 32 + 2 * q;
  // row_shift: the row set starts `row_shift` entries into the column set (row-band refill): diagonal at gj == gi + shift
  const int64_t is0 = i0 + row_shift;

#pragma unroll
  for (int it = 0; it < NT; ++it) {
    if (!live[it]) continue;
    const int64_t j0 = j0s[it], oj0 = oj0s[it];
    const double* Bt = Bs + it * TN * sl;
    d4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (d4){0.0, 0.0, 0.0, 0.0};
    if (EXACT) {
      exact_dist(kp, d, sl, As, Bt, wm * 32 + g, c0, acc);
    } else {
      const double* ap = As + (wm * 32 + q) * sl + g;           // A'[row = mi*16 + q][k = 4s + g]
      const double* bp = Bt + (wn * 32 + 2 * q) * sl + g;       // B'[col = 2q + ni][k = 4s + g]
#pragma unroll
      for (int ks = 0; ks < dpad; ks += 4) {
        const double a0 = ap[ks], a1 = ap[16 * sl + ks];
        const double b0 = bp[ks], b1 = bp[sl + ks];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
      }
    }

    const bool interior = (i0 + TM <= na) && (j0 + TN <= nb) && (!symmetric || (is0 + TM <= j0) || (j0 + TN <= is0));
    char* const otile = reinterpret_cast<char*>(out + oi0 * ld + oj0);
    const unsigned ooff = (unsigned)(((wm * 32 + g) * ld + c0) * 8);
    double2 val[2][4];
    if (__builtin_amdgcn_readfirstlane((int)interior))
      finish_tile<KIND, true>(acc, val, kp.sig, tab, i0, j0, na, nb, symmetric, nugget, nugget_len, nugget_scalar, otile,
                              ooff, ld, wm * 32 + g, c0, false, row_shift);
    else
      finish_tile<KIND, false>(acc, val, kp.sig, tab, i0, j0, na, nb, symmetric, nugget, nugget_len, nugget_scalar, otile,
                               ooff, ld, wm * 32 + g, c0, SYM && ti == tj0, row_shift);

    if (SYM) {
      // mirror: out[j0 + c][i0 + r] = V[r][c]; the two row halves (wm = 0 / 1) go through the padded image in turn
      // (the diagonal tile mirrors its strictly-lower part only)
      const int tx = t & 31, ty = t >> 5;
      char* const mbase = reinterpret_cast<char*>(out + j0 * ld + i0);
      const unsigned moff = (unsigned)((ty * ld + tx) * 8);
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
        if (ti != tj0) {
#pragma unroll
          for (int qq = 0; qq < 8; ++qq)
            *reinterpret_cast<double*>(mbase + (int64_t)(8 * qq) * ld * 8 + moff + 256 * half) = Tr[tx * TP + ty + 8 * qq];
        } else {
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            const int c = ty + 8 * qq;
            if (32 * half + tx > c) out[(j0 + c) * ld + i0 + 32 * half + tx] = Tr[tx * TP + c];
          }
        }
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
  out[j] = kvalue<KIND>(acc, kp.sig, kp.sig * (1.0 / 3.0), kExp2Tab);
}

template <int KIND, bool SYM, int K4, bool EXACT = false>
int launch_k4(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb, int symmetric,
              const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out, int64_t prows,
              int64_t pcols, int64_t ld, int64_t row_shift, KCyclic cyc = KCyclic{0, 1, 0, 1, 0}) {
  dim3 grid;
  if (SYM) {
    const int64_t t = prows / TM;
    grid = dim3((unsigned)(t * (t + 1) / 2));
  } else {
    grid = dim3((unsigned)(pcols / TN), (unsigned)(prows / TM));
  }
  static_assert(EXP_TAB == 256, "the exp table is loaded by the 256 threads of the workgroup, one entry each");
  const size_t img = (size_t)TM * (EXACT ? (kp.d | 1) : 4 * K4 + 1);
  if (!SYM && (KIND == GPX_K_MATERN32 || KIND == GPX_K_MATERN52)) {
    grid.x = (grid.x + RECT_NT - 1) / RECT_NT;  // RECT_NT column tiles per workgroup
    const size_t shn = ((1 + RECT_NT) * img + EXP_TAB) * sizeof(double);
    if (shn > 64 * 1024) {  // d >= 19: 97 KB of operand images; say so explicitly rather than rely on the runtime's default
      static bool raised = false;
      if (!raised) {
        GPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfill_rectn_kernel<KIND, K4, EXACT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        raised = true;
      }
    }
    hipLaunchKernelGGL((kfill_rectn_kernel<KIND, K4, EXACT>), grid, dim3(256), shn, ctx->stream,
                       kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar, out, ld, row_shift, cyc,
                       (int)(pcols / TN));
    GPX_HIP(hipGetLastError());
    return 0;
  }
  const size_t sh = (2 * img + EXP_TAB + (SYM ? 32 * TP : 0)) * sizeof(double);
  hipLaunchKernelGGL((kfill_kernel<KIND, SYM, K4, EXACT>), grid, dim3(256), sh, ctx->stream, kp, A, na, B, nb, symmetric,
                     d_nugget, nugget_len, nugget_scalar, out, ld, row_shift, cyc);
  GPX_HIP(hipGetLastError());
  return 0;
}

template <int KIND, bool SYM>
int launch_kind(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb,
                int symmetric, const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out,
                int64_t prows, int64_t pcols, int64_t ld, int64_t row_shift = 0,
                KCyclic cyc = KCyclic{0, 1, 0, 1, 0}) {
#define GPX_K4(N_)                                                                                                      \
  launch_k4<KIND, SYM, N_>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len, nugget_scalar, out, prows, pcols, ld, \
                           row_shift, cyc)
  if (KIND != GPX_K_MEHLER && kp.exact)
    return launch_k4<KIND, SYM, 1, KIND != GPX_K_MEHLER>(ctx, kp, A, na, B, nb, symmetric, d_nugget, nugget_len,
                                                         nugget_scalar, out, prows, pcols, ld, row_shift, cyc);
  const int k4 = (kp.d + 2 + 3) / 4;
  if (k4 <= 2) return GPX_K4(2);   // d <= 6
  if (k4 <= 3) return GPX_K4(3);   // d <= 10
  if (k4 <= 5) return GPX_K4(5);   // d <= 18
  return GPX_K4(MAXK4);
#undef GPX_K4
}

}  // namespace

int launch_kfill(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb,
                 int symmetric, const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out,
                 int64_t prows, int64_t pcols, int64_t ld) {
  GPX_ARG(prows % TM == 0 && pcols % TN == 0, "kfill: padded shape must be a multiple of 64");
  GPX_ARG(prows / TM <= 65535, "kfill: too many row tiles");
  // the mirrored (half-compute) path needs a square output whose row and column point sets coincide
  const bool sym = symmetric && prows == pcols && A == B && na == nb;
  ProfScope ps(ctx, symmetric ? GPX_PROF_KFILL : GPX_PROF_KCROSS, 0.0,
               8.0 * (double)prows * (double)pcols + 8.0 * (double)(na + nb) * kp.d);
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

// rows [row0, prows_total) of the symmetric covariance of X (all columns; diagonal, nugget and identity padding included):
// what a refit needs when only the trailing points changed.  `out` points at row row0 of the padded matrix.
int launch_kfill_rows(gpx_ctx* ctx, const KParams& kp, const double* X, int64_t n, int64_t row0, const double* d_nugget,
                      int64_t nugget_len, double nugget_scalar, double* out, int64_t prows_band, int64_t pcols,
                      int64_t ld) {
  GPX_ARG(row0 % TM == 0 && prows_band % TM == 0 && pcols % TN == 0, "kfill_rows: band must be tile aligned");
  GPX_ARG(prows_band / TM <= 65535, "kfill: too many row tiles");
  int64_t na = n - row0;
  if (na < 0) na = 0;
  ProfScope ps(ctx, GPX_PROF_KFILL, 0.0, 8.0 * (double)prows_band * (double)pcols + 8.0 * (double)(na + n) * kp.d);
  const double* A = X + row0 * kp.d;
#define GPX_KIND(K_) \
  launch_kind<K_, false>(ctx, kp, A, na, X, n, 1, d_nugget, nugget_len, nugget_scalar, out, prows_band, pcols, ld, row0)
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

// Local part of the symmetric covariance of X on rank (pr, pc) of a Pr x Pc process grid, 2-D block-cyclic with block size
// nb: local block (bi, bj) of `out` (prows x pcols, multiples of nb except for the matrix's last, shorter block) is the
// global block (bi * Pr + pr, bj * Pc + pc).  Diagonal, nugget and identity padding follow the global indices; blocks
// strictly above the global diagonal are skipped.
int launch_kfill_cyclic(gpx_ctx* ctx, const KParams& kp, const double* X, int64_t n, const double* d_nugget,
                        int64_t nugget_len, double nugget_scalar, double* out, int64_t prows, int64_t pcols, int64_t ld,
                        int64_t nb, int Pr, int pr, int Pc, int pc) {
  GPX_ARG(nb % TM == 0 && prows % TM == 0 && pcols % TN == 0, "kfill_cyclic: block size and local shape must be multiples of 64");
  GPX_ARG(prows / TM <= 65535, "kfill: too many row tiles");
  if (prows == 0 || pcols == 0) return 0;
  ProfScope ps(ctx, GPX_PROF_KFILL, 0.0, 8.0 * (double)prows * (double)pcols + 8.0 * (double)n * kp.d);
  const KCyclic cyc{(int)(nb / TM), Pr, pr, Pc, pc};
#define GPX_KIND(K_) \
  launch_kind<K_, false>(ctx, kp, X, n, X, n, 1, d_nugget, nugget_len, nugget_scalar, out, prows, pcols, ld, 0, cyc)
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
