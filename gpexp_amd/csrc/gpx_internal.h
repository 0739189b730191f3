// Internal declarations shared by the HIP translation units of libgpx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <unordered_set>
#include "../../include/gpx.h"
#include "../../include/gpx_dist.h"
#include "../../include/gpx_debug.h"

#define GPX_TILE 128          // padding / GEMM tile / Cholesky leaf size
#define GPX_MAXD GPX_MAX_DIM
#define GPX_NSTREAMS 6
#define GPX_SEG_MAX 8         // panels one trailing update of the 2-D distributed factorisation can apply at once
#ifndef GPX_G_SKEW
#define GPX_G_SKEW 0          // doubles added to the row stride of the packed panel pieces of the 2-D distributed loop (see gpx_g_ld)
#endif
#define GPX_MAX_PR 4          // process rows of the 2-D grid the segmented update supports (grids up to 4 x Pc)

// ---- error plumbing ---------------------------------------------------------------------------
void gpx_set_error(const char* fmt, ...);
#define GPX_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      gpx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return -2;                                                                        \
    }                                                                                   \
  } while (0)
#define GPX_ARG(cond, msg)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      gpx_set_error("%s:%d bad argument: %s", __FILE__, __LINE__, msg); \
      return -1;                                             \
    }                                                        \
  } while (0)
#define GPX_TRY(call)        \
  do {                       \
    int r_ = (call);         \
    if (r_ != 0) return r_;  \
  } while (0)

static inline int64_t gpx_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
// Leading dimension for a padded row of `cols` doubles: one extra 128-byte line per row when the row stride would
// be a large power of two (N = 32768 -> 256 KiB).  Measured NEUTRAL on MI355X (GEMM 66.98 vs 66.97 TF/s: the L2 /
// channel address hashing already spreads such strides); kept because it decouples ld from the padded width.
static inline int64_t gpx_skew_ld(int64_t cols) { return (cols >= 1024 && cols % 256 == 0) ? cols + 16 : cols; }

// Row stride of the rows region of a packed panel piece (dist.hip): nb + GPX_G_SKEW doubles.  At nb = 512 a stride of nb is
// exactly 4 KiB, the textbook channel-conflict stride; measured on MI355X it makes NO difference (66.0 TF/s with skew 0, 65.9
// with 16: scripts/probe_update_multi.py -- like gpx_skew_ld, the address hashing already spreads it), so the skew is 0 and the
// pieces travel without 3 % of padding.  The stride stays a separate quantity (gpexp_amd/dist.py Grid2D.gld mirrors it).
static inline int64_t gpx_g_ld(int64_t nb) { return nb + GPX_G_SKEW; }

// ---- covariance-function parameters, passed to kernels by value ---------------------------------
// Coordinates are pre-multiplied by `scale` when staged into LDS so that
//   SE       : k = sig * exp(-0.5 * |a-b|^2)            scale_k = 1/cl_k          (kernels.py:121-122)
//   Matern32 : t = |a-b|, k = sig*(1+t)*exp(-t)          scale   = sqrt(3)/rho     (kernels.py:87-89)
//   Matern52 : t = |a-b|, k = sig*(1+t+t^2/3)*exp(-t)    scale   = sqrt(5)/rho     (not in reference)
//   Mehler   : k = sig * exp(-sum_k c1_k (a^2+b^2) - c2_k a b), sig = prod (1-t^2)^-1/2,
//              c1_k = t^2/(2(1-t^2)), c2_k = t/(1-t^2), scale = 1               (kernels.py:282-285)
//
// Stationary kernels are translation invariant, and the tiled assembly forms |a-b|^2 from the EXPANDED product
// |a|^2 + |b|^2 - 2 a.b on the MFMA pipe, whose absolute error grows like eps * (|a|^2 + |b|^2) * scale^2 -- the reference
// subtracts coordinates first (kernels.py:121-122, 87-89).  Two measures keep the assembly at the reference's accuracy
// for any input placement (gpx_kparams_sets):
//   center  the midpoint of the point sets' bounding box is subtracted from every coordinate before scaling, so the
//           expanded form only ever sees offsets of the size of the domain, not of its distance from the origin;
//   exact   when even the centred, scaled domain is wide (half-width / length scale large: S = sum_k scale_k^2 *
//           (hwA_k^2 + hwB_k^2) above GPX_EXACT_S / sensitivity), the fill forms (a_k - b_k) * scale_k directly from the
//           raw coordinates on the VALU instead -- the reference's own operation order, exact where Sterbenz applies.
// Mehler is not stationary: center = 0, exact = 0 (the reference expands its exponent the same way, kernels.py:282-285).
struct KParams {
  int kind;
  int d;
  double sig;
  double scale[GPX_MAXD];
  double c1[GPX_MAXD];
  double c2[GPX_MAXD];
  double center[GPX_MAXD];
  int exact;
};
int gpx_make_kparams(int kind, int d, const double* hyp, int nhyp, KParams* out);

// ---- objects --------------------------------------------------------------------------------------
struct gpx_mat {
  double* p;
  int64_t rows, cols;    // logical
  int64_t prows, pcols;  // padded shape
  int64_t ld;            // leading dimension >= pcols (skewed off powers of two, see gpx_skew_ld)
  int64_t bytes;
  double* aux;        // after gpx_potrf: inverses of the 128x128 diagonal blocks, (prows/128) x 128 x 128
  int64_t aux_bytes;
  int factored;
  // bounding box of a point set (cols <= GPX_MAXD), computed on the host at upload (gpx_mat_from_host) or on first use
  int bbox_ok;
  double lo[GPX_MAXD], hi[GPX_MAXD];
  // explicit inverses of the IB x IB diagonal blocks of the factor (and their transposes), built on the first gpx_potrs
  // after a factorisation (chol_potrs): ceil(prows / IB) blocks of IB x IB each, twice
  double* binv;
  int64_t binv_bytes;
  int64_t binv_ib;
  // block-cyclic LOCAL matrix of the 2-D distributed factorisation: explicit inverses of the diagonal blocks this rank owns
  // (slot = local block row), kept from the panel solves for the distributed substitution (gpx_dist2_trsv_diag):
  // per slot [inverse | its transpose], dinv_nb x dinv_nb doubles each; dinv_ok[slot] = the slot holds the inverse of the
  // CURRENT factor's block
  double* dinv;
  int64_t dinv_bytes;
  int64_t dinv_nb;
  std::vector<unsigned char> dinv_ok;
};

struct ProfRec {
  hipEvent_t a, b;
  int cls;
  hipStream_t stream;
  bool open;
};

struct gpx_ctx {
  int device;
  hipStream_t stream;        // currently selected stream (all launches go here)
  hipStream_t streams[GPX_NSTREAMS];  // 0 = main, 1 = panel (high priority), 2 = communication (high priority),
                             // 4 = evaluation (low priority, unmasked): the streamed IVAR solve beside the factorisation,
                             // 3 = background, 5 = bulk: CU-masked (leave 4 CUs per XCD to the other streams) when the runtime allows
  std::vector<hipEvent_t> sync_events;  // gpx_event_record / gpx_event_wait ids
  std::unordered_set<const gpx_mat*> live_mats;  // every matrix this context handed out and has not freed (gpx_program_run checks its rows against it)
  // work buffers of a blocked factorisation in flight (set by gpx_potrf around chol_potrf, NULL otherwise): storage of the
  // explicit block inverses being built, order of those blocks, scratch for their build and for the panel solves
  double* pw_binv;
  int64_t pw_ib;
  double* pw_tmp_build;
  double* pw_tmp_T;
  int pw_done;   // set by the factorisation when it has built every block inverse into pw_binv
  std::vector<hipEvent_t> la_events;    // look-ahead factorisation (chol.hip): block row k+1 + next diagonal block ready / chain done
  hipEvent_t ev_side = nullptr;         // fork / join of gpx_refit_rows' copy of the kept rows on the low-priority stream
  int cus;
  // cached device allocations (exact-size reuse)
  std::multimap<int64_t, void*> pool;
  int64_t pool_bytes;
  // deferred release (gpx_mat_free): a freed matrix's buffers go back to the pool AT ONCE, tagged with a fence -- one event per
  // stream of the context, recorded at the time of the free; whoever takes such a block out of the pool waits for the fence
  // first (normally long complete).  Replaces a device-wide synchronisation per free.
  struct Fence {
    std::vector<hipEvent_t> evs;
  };
  std::map<void*, std::shared_ptr<Fence>> pending;
  std::vector<hipEvent_t> fence_free;
  // GPX_ALLOC_GUARD=1 (debug; there is no GPU address sanitizer on this pool): every pooled allocation gets a 4 KiB band of
  // 0xA5 on either side, checked when it goes back to the pool; violations are counted and reported on stderr
  int guard;
  int64_t guard_violations;
  // GPX_CHAOS=seed (debug): every profiled launch site holds its stream back by a random 0.1-3 ms with probability 1/4, so
  // that a missing dependency between the context's streams changes the results instead of hiding behind lucky timing
  uint64_t chaos;
  // scalars
  int* d_info;      // [0] first failing pivot (1-based), 0 = ok; [1] pivots dropped in skip mode
  double piv_min;   // pivot policy of the leaf factorisation (gpx_potrf_policy): pivots <= piv_min are bad ...
  int piv_skip;     // ... and reported (0) or dropped (1)
  double* d_scal;   // small scalar workspace (>= 64 doubles)
  double* trsv_scratch;      // grown on demand, kept until gpx_destroy (lets gpx_potrs_dev stay asynchronous)
  int64_t trsv_scratch_bytes;
  double* d2_scratch;        // 2-D distributed panel solve: explicit inverse of the current diagonal block + build scratch
  int64_t d2_scratch_bytes;  // (2 nb^2 doubles; used on the PANEL stream only, in step order)
  long long* dbg_stamps;     // gpx_dbg_stamp / gpx_dbg_spin_until: wall-clock stamps taken on a stream (GPX_DBG_STAMPS slots)
  const double* d2_inv_src;  // the packed diagonal block whose inverse d2_scratch holds (gpx_dist2_panel_inv), or NULL
  int64_t d2_inv_nb;
  double* ev_scratch;        // streamed evaluation (gpx_dist_ivar_group_at): block inverses of the group + their build scratch +
  int64_t ev_scratch_bytes;  // the solved block rows W (w x m); used on ONE stream only, in group order
  // multi-GPU (RCCL communicator, opaque here; see dist.hip)
  void* comm;
  int rank, world;
  // process grid of the 2-D block-cyclic path (gpx_comm_grid): rank = pr * Pc + pc; sub-communicators from ncclCommSplit:
  // grp[0] = world (== comm), grp[1] = my process row (Pc ranks, rank pc), grp[2] = my process column (Pr ranks, rank pr)
  void* grp[3];
  int grp_size[3], grp_rank[3];
  int Pr, Pc;
  // profiling
  int prof_on;
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_free;
  int64_t prof_launches[GPX_PROF_NCLASS];
  double prof_ms[GPX_PROF_NCLASS];
  double prof_flops[GPX_PROF_NCLASS];
  double prof_bytes[GPX_PROF_NCLASS];
};

// centre + exact-path decision for fills between the given point sets (any may be NULL); see KParams
int gpx_kparams_sets(gpx_ctx* ctx, KParams* kp, const gpx_mat* A, const gpx_mat* B = nullptr, const gpx_mat* C = nullptr);
int gpx_dev_alloc(gpx_ctx* ctx, int64_t bytes, void** out);
void gpx_dev_release(gpx_ctx* ctx, void* p, int64_t bytes);
int gpx_mat_new(gpx_ctx* ctx, int64_t rows, int64_t cols, int pad, gpx_mat** out);

// RAII-ish profiling bracket: records events around launches of one class when profiling is on
struct ProfScope {
  gpx_ctx* ctx;
  int idx;
  ProfScope(gpx_ctx* c, int cls, double flops, double bytes);
  ~ProfScope();
};
int gpx_prof_flush(gpx_ctx* ctx);

// ---- kernel launchers (all asynchronous on ctx->stream) ------------------------------------------
// kfill.hip
int launch_kfill(gpx_ctx* ctx, const KParams& kp, const double* A, int64_t na, const double* B, int64_t nb,
                 int symmetric, const double* d_nugget, int64_t nugget_len, double nugget_scalar,
                 double* out, int64_t prows, int64_t pcols, int64_t ld);
// symmetric sub-block K[off:, off:off+pcols] of the N x N covariance (rows/cols start at point index `off`)
int launch_kfill_offset(gpx_ctx* ctx, const KParams& kp, const double* X, int64_t n, int64_t row_off, int64_t col_off,
                        const double* d_nugget, int64_t nugget_len, double nugget_scalar, double* out, int64_t prows,
                        int64_t pcols, int64_t ld);
int launch_kdiag(gpx_ctx* ctx, const KParams& kp, const double* Z, int64_t m, double* out);
int launch_kfill_cyclic(gpx_ctx* ctx, const KParams& kp, const double* X, int64_t n, const double* d_nugget,
                        int64_t nugget_len, double nugget_scalar, double* out, int64_t prows, int64_t pcols, int64_t ld,
                        int64_t nb, int Pr, int pr, int Pc, int pc);
int gpx_copy2d(gpx_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols);
int gpx_copy2d_lower(gpx_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t n);
int launch_kfill_rows(gpx_ctx* ctx, const KParams& kp, const double* X, int64_t n, int64_t row0, const double* d_nugget,
                      int64_t nugget_len, double nugget_scalar, double* out, int64_t prows_band, int64_t pcols,
                      int64_t ld);

// gemm_f64.hip:  C[m x n] = (accumulate ? C - A*op(B) : A*op(B)),  m,n multiples of 128, k multiple of 16
// Kernels of the latency-bound chains (the diagonal block's factorisation and inversion, the panel stream of the distributed
// loop) run BESIDE chip-filling updates and share their CUs with resident GEMM waves; launched on the context's chain stream
// (streams[1]) they raise their waves' issue priority (s_setprio 3) so that the CU's arbiter serves them first.
static inline int gpx_chain_prio(const gpx_ctx* ctx) { return ctx->stream == ctx->streams[1] ? 1 : 0; }
int launch_gemm(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                int64_t m, int64_t n, int64_t k, bool bt, bool accumulate, bool lower);

int launch_gemm_tri(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                    int64_t m, int64_t n, int64_t k, bool bt, bool accumulate, bool lower, int tri);
int launch_gemm_ksplit(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                       int64_t m, int64_t n, int64_t k, bool lower, int64_t parts, double* P, bool assign = false);
int launch_gemm_ksplit_small(gpx_ctx* ctx, const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                             int64_t m, int64_t n, int64_t k, bool assign, int64_t parts, double* P);
int launch_gemm_batched(gpx_ctx* ctx, const double* A, int64_t lda, int64_t sa, const double* B, int64_t ldb, int64_t sb,
                        double* C, int64_t ldc, int64_t sc, int64_t m, int64_t n, int64_t k, bool bt, bool accumulate,
                        int64_t batch);

int launch_dist2_update(gpx_ctx* ctx, double* C, int64_t ldc, int64_t lr0, int64_t m, int64_t lc0, int64_t n, int64_t nb, int Pr,
                        int Pc, int pr, int pc, int64_t piece_stride, int64_t dsz, int nseg, const double* const* g,
                        const int64_t* ks, int below_diag);

// chol.hip
int launch_leaf(gpx_ctx* ctx, double* A, int64_t ld, double* inv, int64_t base_index, int64_t n_valid);
int chol_potrf(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t n_valid);
// same without resetting the pivot flag; `base` = global index of A's first row (for the reported pivot)
int chol_potrf_nozero(gpx_ctx* ctx, double* A, int64_t ld, int64_t n, double* invd, int64_t base, int64_t n_valid);
// X <- X * L^-T (right, lower, transposed): X is m x n (ld ldx), L n x n lower with leaf inverses invd
int chol_trsm_right(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* X, int64_t ldx,
                    int64_t m, int64_t n);
// X <- X * L^-1 (right, lower, not transposed)
int chol_trsm_right_n(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* X, int64_t ldx,
                      int64_t m, int64_t n);
// B <- L^-1 B (left, lower): B is n x m
int chol_trsm_left(gpx_ctx* ctx, const double* L, int64_t ldl, const double* invd, double* B, int64_t ldb,
                   int64_t n, int64_t m);
int chol_trsv(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* y, int64_t n, bool transposed);
// bytes of scratch chol_trsv needs for the transposed sweep of order n
int64_t chol_trsv_scratch_bytes(int64_t n);
int chol_trsv_with_scratch(gpx_ctx* ctx, const double* L, int64_t ld, const double* invd, double* y, int64_t n,
                           bool transposed, double* scratch);
int launch_logdet(gpx_ctx* ctx, const double* L, int64_t ld, int64_t n, double* d_out);
// alpha = K^-1 y through explicit inverses of the diagonal blocks (built on first use, kept in L): v (padded n doubles) is
// overwritten by the solution; scratch >= chol_potrs_scratch_bytes(n).  Asynchronous on the selected stream.
int chol_trtri(gpx_ctx* ctx, const gpx_mat* L, double* Linv, double* tmp);  // Linv (n x n, ld n) = L^-1; tmp >= (n/2)^2
int chol_binv_ensure(gpx_ctx* ctx, gpx_mat* L);
int chol_trsm_right_trailing(gpx_ctx* ctx, gpx_mat* L, int64_t r0, double* X, int64_t ldx, int64_t m, int transposed, double* T);
int chol_trsm_right_leading(gpx_ctx* ctx, gpx_mat* L, int64_t ncols, double* X, int64_t ldx, int64_t m, double* T, int64_t tcap);
int chol_trsm_right_n_leading(gpx_ctx* ctx, gpx_mat* L, int64_t ncols, double* X, int64_t ldx, int64_t m, double* T);
int chol_block_inverse(gpx_ctx* ctx, const double* D, int64_t ldd, const double* invd, double* inv, int64_t w, double* tmp);
int64_t chol_binv_order(int64_t n);
int64_t chol_binv_elems(int64_t n);
int chol_binv_finish(gpx_ctx* ctx, gpx_mat* L, int64_t ib);  // explicit inverses of L's diagonal blocks (order <= 1024), cached in L
// W (n x m, separate buffer) = L^-1 B through the block inverses: B is consumed (its lower block rows are updated in place)
int chol_trsm_left_oop(gpx_ctx* ctx, gpx_mat* L, double* B, int64_t ldb, double* W, int64_t ldw, int64_t m);
// one GROUP step of a right-looking left solve: Lg = the group's w x w diagonal triangle (row stride ld, `below` more rows
// underneath it), invd = the leaf inverses of its first row block, B = the group's w rows of the right-hand sides (m columns):
//   B[0:w] <- Lg^-1 B[0:w];  B[w : w+below] -= Lg[w:, 0:w] B[0:w]
// through explicit ib-order inverses built here (inv: ceil(w/ib) ib^2 doubles, tmp: the same, W: w x ldw doubles of scratch)
int chol_trsm_left_group(gpx_ctx* ctx, const double* Lg, int64_t ld, const double* invd, int64_t w, int64_t below, int64_t ib,
                         double* B, int64_t ldb, int64_t m, double* inv, double* tmp, double* W, int64_t ldw);
// one right-looking panel step of that solve with the rows [r0, r1) of the factor only (a finished look-ahead panel): W[r0:r1]
// from B[r0:r1] through the block inverses at `binv` (order ib, row stride ib), then B[r1:] -= L[r1:, r0:r1] W[r0:r1]
int64_t chol_potrf_panel_width(int64_t n);  // width of the look-ahead panels chol_potrf uses for order n (0: not blocked)
int64_t chol_potrs_scratch_bytes(int64_t n);
int chol_potrs(gpx_ctx* ctx, gpx_mat* L, double* v, double* scratch);
// y[r] -= sum_c A[r][c] x[c] over a rows x cols block (cols a multiple of 2, ld even)
int launch_gemv_sub(gpx_ctx* ctx, const double* A, int64_t ld, int64_t rows, int64_t cols, const double* x, double* y);
int chol_block_transpose(gpx_ctx* ctx, const double* in, double* out, int64_t n);
int chol_tri_gemv(gpx_ctx* ctx, const double* M, int64_t ld, int64_t sz, const double* y, double* x, int lower);

// fitc.hip: what grad.hip needs of a FITC model (the struct is private to fitc.hip)
struct gpx_fitc;
int64_t fitc_n(const gpx_fitc* f);
int64_t fitc_np(const gpx_fitc* f);
int64_t fitc_nup(const gpx_fitc* f);
int fitc_solve_beta_t(gpx_ctx* ctx, const gpx_fitc* f, const double* B, int64_t mcp, double* Bt, double* U);

// reduce.hip
// out[j] = sum_i B[i][j] * v[i]   (v == nullptr: sum_i B[i][j]^2), i < rows, j < pcols; deterministic
int launch_colreduce(gpx_ctx* ctx, const double* B, int64_t ld, int64_t rows, int64_t pcols, const double* v,
                     double* out, double* d_partial, int subtract = 0);
int64_t colreduce_partial_elems(int64_t rows, int64_t pcols);
int launch_rowreduce(gpx_ctx* ctx, const double* B, int64_t ld, int64_t rows, int64_t cols, const double* v,
                     double* out, int weighted_squares = 0);
int launch_sum(gpx_ctx* ctx, const double* x, int64_t n, double* d_out);

// design.hip
extern "C" int gpx_potri_impl(gpx_ctx* ctx, const gpx_mat* L, gpx_mat** outP, int full);
int launch_transpose(gpx_ctx* ctx, const double* in, int64_t rows, int64_t cols, int64_t ldi, double* out, int64_t ldo);
