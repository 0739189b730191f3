/* gpx_dist.h -- the MULTI-GPU and SCHEDULER primitives of libgpx_hip.so, split off the drop-in ABI (include/gpx.h) in round 4.
 *
 * gpx.h is what a maintainer of the reference binds: one entry point per reference call site.  This header is what only
 * gpexp_amd/dist.py drives: streams and events of the look-ahead pipelines, the RCCL communicator and its sub-communicators,
 * the panel primitives of the 1-D and 2-D block-cyclic factorisations, the recorded programs that replay the Python panel loop
 * natively, and the sharded state machines (MI rows, greedy-IVAR candidates, gradient slabs) whose single-GPU drivers live in
 * gpx.h.  Role replaced: the reference's only parallel backend, the fork + mp.Queue helper parallelizeMcForLoop
 * (parallel_utilities.py:26-80, used at gp.py:258).  Conventions as in gpx.h; every call is asynchronous on the selected stream
 * unless stated.
 */
#ifndef GPX_DIST_H
#define GPX_DIST_H

#include "gpx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- streams and events of the look-ahead pipelines ------------------------------------------------------ */
/* HIP streams per context: 0 main, 1 panel factorisation (high priority), 2 communication (high priority), 3 background
 * (CU-masked: leaves 4 CUs per XCD to the others), 4 evaluation (low priority), 5 bulk (CU-masked like 3: the aggregated
 * trailing updates of the distributed factorisation).
 * All entry points enqueue on the currently selected one; events order work across them (look-ahead pipeline). */
int gpx_stream_select(gpx_ctx* ctx, int which);
int gpx_event_record(gpx_ctx* ctx, int id);  /* id in [0, 65536): recorded on the selected stream */
int gpx_event_wait(gpx_ctx* ctx, int id);    /* the selected stream waits for the last record of id */

/* ---- device-vector glue and raw access to the padded storage --------------------------------------------- */
/* dst[doff : doff+n] = src[soff : soff+n] (mode 0), += src (mode 1), = 0 (mode 2): device-vector glue, asynchronous */
int gpx_vec_op(gpx_ctx* ctx, gpx_mat* dst, int64_t doff, const gpx_mat* src, int64_t soff, int64_t n, int mode);

/* raw access to the padded storage (element offsets); used by the host-staged communicator in tests */
int gpx_mat_read(gpx_ctx* ctx, const gpx_mat* m, int64_t offset, int64_t count, double* dst);
int gpx_mat_write(gpx_ctx* ctx, gpx_mat* m, int64_t offset, int64_t count, const double* src);


/* ---- multi-pick greedy IVAR as a candidate-sharded state machine (gpx_greedy_ivar of gpx.h drives it on one GPU) ---- */
/* A point set that is one rank's SLICE of a larger set takes the whole set's bounding box (lo / hi: d doubles each), so that the
   centring and the exact-difference decision of the fills it enters are the same on every sharding as on one rank. */
int gpx_points_set_box(gpx_ctx* ctx, gpx_mat* P, const double* lo, const double* hi, int d);

/* gpx_greedy_ivar (gpx.h) as a state machine, sharded by CANDIDATES for the multi-GPU form (every rank: its slice of C and the whole of Z):
 * score -> local first minimum; the owner of the merged winner packs its pivot (gpx_givar_pivot_elems doubles: delta, the
 * point, cov(Z, c_s | design) / sqrt(delta), its column of W_C, its coordinates along the earlier picks), the caller broadcasts
 * it, every rank applies it.  The per-candidate arithmetic does not depend on the sharding. */
typedef struct gpx_givar gpx_givar;
int gpx_givar_begin(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* C, const gpx_mat* Z, double noise, int64_t nsel, gpx_givar** out);
int64_t gpx_givar_pivot_elems(const gpx_givar* st);
int gpx_givar_score(gpx_ctx* ctx, gpx_givar* st, double* best_cost, int64_t* best_idx, double* all_costs);
int gpx_givar_pack(gpx_ctx* ctx, gpx_givar* st, int64_t s, gpx_mat* buf);
int gpx_givar_apply(gpx_ctx* ctx, gpx_givar* st, const gpx_mat* buf);
int gpx_givar_end(gpx_ctx* ctx, gpx_givar* st);

/* ---- the slab form of the log-marginal gradient: the unit the multi-GPU gradient shards by --------------- */
/* The traces of gpx_lml_grad (gpx.h) over ONE ROW SLAB [r0, r1) of K^-1 (multiples of 128), un-scaled: sums[q], q < d: sum T K0 e_q^2; q = d: sum T
 * K0; q = d+1: tr T -- over slab rows a and columns b >= a, off-diagonal entries counted twice.  The slab of the inverse is two
 * triangular solves against the TRAILING factor L[r0:, r0:]; no N x N inverse is formed.  Slabs of a partition of the rows add up
 * to the full traces: grad[k] = sums[k] / (2 hyp[k]) (k < d), grad[d] = sums[d] / (2 hyp[d]), grad[d+1] = sums[d+1] / 2.  This is
 * the unit the multi-GPU gradient shards by (gpexp_amd/dist.py dist_lml_grad; gp.py:444-466). */
int gpx_lml_grad_slab(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                      const double* alpha, int64_t r0, int64_t r1, double* sums);

/* The same traces sharded by ROWS [r0, r1) of L^-1 (round 5): one right solve against the leading r1-order block of the factor
   and one lower SYRK G = X^T X (r1 x r1, K = r1 - r0) instead of the slab form's two solves; work (r1 - r0) r1^2, memory
   r1^2 + 2 (r1 - r0) r1 doubles; the range is cut into nsub sub-slabs of equal work whose products accumulate in one matrix,
   traced once.  The partial sums of a partition of the rows add up to gpx_lml_grad's (the slab ending at the
   padded order adds the alpha alpha^T part).  alpha: host, N doubles; sums: host, d+2 doubles. */
int gpx_lml_grad_rows(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                      const double* alpha, int64_t r0, int64_t r1, int nsub, double* sums);

/* ---- row-sharded greedy MI state (gpx_mi_greedy of gpx.h is the single-GPU form) ------------------------- */
/* Greedy MI with the candidate SCORING sharded by rows of the inverse (multi-GPU; gpexp_amd/dist.py dist_mi_greedy).  One
 * state per rank: rows [lo, hi) of the M x M inverse are kept current and exactly those candidates are scored.  Per pick:
 * gpx_mi_row (the owner of the picked row s stages P[s, :] in rowbuf -- the caller broadcasts it), gpx_mi_score (down-date,
 * ratios, local first-max -> host), gpx_mi_select (the winner merged over the ranks).  lo = 0, hi = M reproduces gpx_mi_greedy. */
typedef struct gpx_mi gpx_mi;
int gpx_mi_begin(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* C, double noise, int64_t nsel,
                 int64_t start, int64_t lo, int64_t hi, gpx_mi** out);
int gpx_mi_row(gpx_ctx* ctx, gpx_mi* st, int64_t cur, int64_t s, gpx_mat* rowbuf);
int gpx_mi_score(gpx_ctx* ctx, gpx_mi* st, int64_t cur, const gpx_mat* rowbuf, double* best_val, int64_t* best_idx);
int gpx_mi_select(gpx_ctx* ctx, gpx_mi* st, int64_t slot, int64_t idx);
int gpx_mi_end(gpx_ctx* ctx, gpx_mi* st);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------------
 * Replaces the reference's only parallel backend, the fork + mp.Queue row-sharding helper
 * (parallel_utilities.py:26-80; used at gp.py:258).  The covariance matrix is distributed by block columns
 * (width nb, owner = block index mod world); see gpexp_amd/dist.py for the panel loop that drives these. */
/* rank 0 creates the 128-byte id, the launcher's rendezvous distributes it, every rank calls gpx_comm_init */
int gpx_comm_unique_id(void* out128);
int gpx_comm_init(gpx_ctx* ctx, int rank, int world, const void* id128);
int gpx_comm_destroy(gpx_ctx* ctx);
/* broadcast the first `count` doubles of buf from root (asynchronous on the context's stream) */
int gpx_comm_bcast(gpx_ctx* ctx, gpx_mat* buf, int64_t count, int root);
/* out[world*n] = concatenation in rank order of every rank's in[n] (blocking; scalars such as IVAR partial sums) */
int gpx_comm_allgather_host(gpx_ctx* ctx, const double* in, int64_t n, double* out);
/* assemble only the block columns owned by `rank` (rows on/below the diagonal block) */
int gpx_dist_kfill(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X,
                   const double* nugget, int64_t nugget_len, gpx_mat* K, int64_t nb, int rank, int world);
/* doubles in the packed panel buffer: padded_rows*nb panel + the inverted 128x128 diagonal leaves */
int64_t gpx_dist_panel_elems(int64_t padded_rows, int64_t nb);
/* reset / read the accumulated pivot flag of a panel-wise factorisation (gpx_dist_info synchronises the device) */
int gpx_dist_begin(gpx_ctx* ctx);
int gpx_dist_info(gpx_ctx* ctx, int* info);
/* owner of block column k: pack the panel into P and factor it there (asynchronous on the selected stream) */
int gpx_dist_panel_factor(gpx_ctx* ctx, gpx_mat* K, int64_t k, int64_t nb, gpx_mat* P);
/* every rank, once P has arrived: store the panel and its leaf inverses in the local matrix */
int gpx_dist_panel_store(gpx_ctx* ctx, gpx_mat* K, int64_t k, int64_t nb, const gpx_mat* P);
/* every rank: apply panel k to the owned block columns j0 <= j < j1 (look-ahead updates column k+1 first) */
int gpx_dist_panel_update(gpx_ctx* ctx, gpx_mat* K, int64_t k, int64_t nb, const gpx_mat* P, int64_t j0, int64_t j1,
                          int rank, int world);
/* mark K as a complete factor (every rank now holds all of L) */
/* streamed evaluation (multi-GPU): step k of a right-looking left solve of B = K(X, Z_local) (padded N x m) against block
 * column k of the factor, valid as soon as gpx_dist_panel_store(k) has run; asynchronous on the selected stream */
int gpx_dist_ivar_step(gpx_ctx* ctx, const gpx_mat* K, int64_t k, int64_t nb, gpx_mat* B);
/* the same for the panels k0 .. k1 at once (the update below the group runs with K = (k1 - k0 + 1) nb) */
int gpx_dist_ivar_group(gpx_ctx* ctx, const gpx_mat* K, int64_t k0, int64_t k1, int64_t nb, gpx_mat* B);
/* the same against a WINDOW of the factor: K (padded N rows x window columns) holds the group's block columns from column c0
 * on -- no rank keeps an N x N copy of the factor, each panel is consumed as it arrives (L stays distributed) */
int gpx_dist_ivar_group_at(gpx_ctx* ctx, const gpx_mat* K, int64_t k0, int64_t k1, int64_t nb, gpx_mat* B, int64_t c0);
/* the group's step of the forward substitution L w = y on the vector v (padded N doubles), against the same window: every rank
 * ends with the complete w, the distributed substitution keeps only its backward sweep */
int gpx_dist_fwd_group_at(gpx_ctx* ctx, const gpx_mat* K, int64_t k0, int64_t k1, int64_t nb, gpx_mat* v, int64_t c0);
int gpx_dist_finish(gpx_ctx* ctx, gpx_mat* K);

/* ---- multi-GPU, 2-D block-cyclic (north_star) ---------------------------------------------------------
 * Process grid Pr x Pc, rank (pr, pc) = (rank / Pc, rank % Pc).  Global block (I, J) of the padded matrix (block size nb,
 * a multiple of 128; the last block may be shorter) lives on rank (I % Pr, J % Pc) at local block (I / Pr, J / Pc) of that
 * rank's LOCAL matrix: every rank allocates only its share.  gpexp_amd/dist.py drives the panel loop (diagonal block
 * broadcast down the process column on an ncclCommSplit sub-communicator, panel pieces to every rank over all xGMI links,
 * look-ahead, streamed evaluation) on these primitives; every call is asynchronous on the selected stream unless stated. */
/* sub-communicators: group 0 = world, 1 = the rank's process row, 2 = its process column (ncclCommSplit) */
int gpx_comm_grid(gpx_ctx* ctx, int Pr, int Pc);
int gpx_comm_bcast_grp(gpx_ctx* ctx, gpx_mat* buf, int64_t offset, int64_t count, int root, int grp);
/* out-of-place: the root sends sbuf[soff ..], every member receives into rbuf[roff ..] (ncclBroadcast with two pointers) */
int gpx_comm_bcast_grp2(gpx_ctx* ctx, const gpx_mat* sbuf, int64_t soff, gpx_mat* rbuf, int64_t roff, int64_t count, int root, int grp);
/* in-place sum inside a group, result on group rank `root` (ncclReduce) / over all ranks, result everywhere (ncclAllReduce) */
int gpx_comm_reduce_grp(gpx_ctx* ctx, gpx_mat* buf, int64_t offset, int64_t count, int root, int grp);
int gpx_comm_allreduce(gpx_ctx* ctx, gpx_mat* buf, int64_t offset, int64_t count);
int gpx_comm_allreduce_host(gpx_ctx* ctx, double* inout, int64_t n);   /* <= 64 host scalars, blocking */
/* all-link broadcast: npieces regions of buf (same layout on every rank), region i owned by world rank roots[i], reach
 * every rank as a scatter + all-gather over grouped ncclSend / ncclRecv -- 2/(W-1) of the bytes per xGMI link instead of
 * all of them over one (ring / tree ncclBroadcast).  Every rank calls it with identical arguments. */
int gpx_comm_panel_bcast(gpx_ctx* ctx, gpx_mat* buf, const int64_t* offsets, const int64_t* counts, const int* roots,
                         int npieces);
/* doubles of a packed diagonal block: nb x nb factor + nb/128 inverted 128 x 128 leaves */
int64_t gpx_dist2_diag_elems(int64_t nb);
/* row stride (doubles) of the packed panel rows (nb + a build-time skew, 0 by default) */
int64_t gpx_dist2_row_stride(int64_t nb);
/* local part of K(X) + nugget on rank (pr, pc)   (gp_kernel_utilities.py:34-68, communication-free) */
int gpx_dist2_kfill(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X, const double* nugget,
                    int64_t nugget_len, gpx_mat* A, int64_t nb, int Pr, int Pc, int pr, int pc);
/* diagonal owner: factor the w x w block at local (lr, lc) into the D region of the panel buffer G (offset doff) */
int gpx_dist2_diag_factor(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                          int64_t nb, int64_t base, int64_t n_valid);
/* the same in three parts, which the panel loop issues separately so that only the factorisation itself sits on the chain across
 * ranks: _stage copies the block into the D region BEFORE the block row it still waits for has arrived, _diag_update applies that
 * last update (S[soff]: h x w packed rows, times their transpose) to the staged copy, _factor_staged factors it in place, and
 * _store -- issued behind the event that releases the broadcast -- copies factor and leaf inverses into the local matrix (dslot
 * >= 0: and keeps the explicit inverse gpx_dist2_panel_inv built for this block, for gpx_dist2_trsv_diag) */
int gpx_dist2_diag_stage(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff, int64_t nb);
int gpx_dist2_diag_update(gpx_ctx* ctx, gpx_mat* G, int64_t doff, int64_t h, const gpx_mat* S, int64_t soff, int64_t w, int64_t nb);
int gpx_dist2_diag_factor_staged(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                                 int64_t nb, int64_t base, int64_t n_valid);
int gpx_dist2_diag_store(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, const gpx_mat* G, int64_t doff, int64_t nb,
                         int64_t dslot);
/* size the context's scratches of the distributed loops up front (panel-solve inverse: 2 nb^2 doubles; streamed evaluation:
 * 2 agg nb^2 + agg nb mcols doubles) so that the enqueue path never reallocates mid-step; called by the runners' constructors */
int gpx_dist2_reserve(gpx_ctx* ctx, int64_t nb, int64_t agg, int64_t mcols);
/* holders of block column k, as soon as the diagonal block has arrived in G (doff): its explicit inverse, kept for the panel
 * solves of this step (gpx_dist2_panel_trsm then multiplies with it instead of building it on the panel chain; the block row the
 * next diagonal needs takes the same one-product path instead of the leaf recursion) */
int gpx_dist2_panel_inv(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t nb, int64_t w);
/* holders of block column k: local rows [lr0, lr0+m) of the column <- X L_kk^-T, packed into G at roff */
int gpx_dist2_panel_trsm(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                         int64_t roff, int64_t nb);
/* the same; on the owner of the diagonal block (dslot = its local block row, -1 elsewhere) the explicit inverse the solve
 * builds is KEPT in the local matrix, and gpx_dist2_trsv_diag uses it (one small GEMV instead of a block sweep) */
int gpx_dist2_panel_trsm_keep(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                              int64_t roff, int64_t nb, int64_t dslot);
/* the same with the inverse gpx_dist2_panel_inv prepared for this step's diagonal block (one triangular-operand product, also
 * for a single block row) */
int gpx_dist2_panel_trsm_inv(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                             int64_t roff, int64_t nb, int64_t dslot, int copy_back);
/* copy_back == 0 above: the solved rows go into the packed buffer only; this copies them into the local matrix afterwards (behind
 * the event that releases the panel broadcast: off the chain across ranks) */
int gpx_dist2_panel_copyback(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, const gpx_mat* G, int64_t roff,
                             int64_t nb);
/* The finished factor RE-STREAMED from the block-cyclic local matrix (a rank that keeps no replica; dist2_restream_enqueue): the
   inverses of gpx_dist2_panel_copyback / gpx_dist2_diag_store -- rows of a finished panel, and the factored diagonal block with its
   leaf inverses, from the local matrix into the packed buffer the panel broadcast sends from. */
int gpx_dist2_panel_pack(gpx_ctx* ctx, const gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t roff,
                         int64_t nb);
int gpx_dist2_diag_pack(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff, int64_t nb);

/* A[lr0:lr0+m, lc0:lc0+n] -= G[aoff] (m x w) * G[boff] (n x w)^T : trailing update of one local block column */
int gpx_dist2_update(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc0, int64_t n, const gpx_mat* G,
                     int64_t aoff, int64_t boff, int64_t w, int64_t nb);
/* trailing update by SEVERAL panels at once over the whole local trailing matrix (one launch, K = nseg * nb): panel ks[s]
 * sits in the packed buffer G[s] (Pr pieces of piece_stride doubles); only local blocks on / below (below_diag != 0:
 * strictly below) the global diagonal are touched.  nseg <= 8, Pr <= 4. */
int gpx_dist2_update_multi(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc0, int64_t n, int64_t nb, int Pr, int Pc,
                           int pr, int pc, int64_t piece_stride, int nseg, const gpx_mat* const* G, const int64_t* ks,
                           int below_diag);
/* single-rank replay of the distributed loop: stage what a collective would have delivered out of a complete factor L
 * resident on this GPU (inverse of the two unpack calls below; same bytes, device to device) */
int gpx_dist2_pack_rows(gpx_ctx* ctx, const gpx_mat* L, int64_t first_block, int64_t stride, int64_t col0, gpx_mat* G,
                        int64_t roff, int64_t m, int64_t w, int64_t nb);
int gpx_dist2_pack_diag(gpx_ctx* ctx, const gpx_mat* L, int64_t r0, int64_t w, int64_t nb, gpx_mat* G, int64_t doff);
/* replicated factor for the evaluation phase: piece rows / diagonal block of panel k into the full-size matrix L */
int gpx_dist2_unpack_rows(gpx_ctx* ctx, const gpx_mat* G, int64_t roff, int64_t m, int64_t w, int64_t nb, gpx_mat* L,
                          int64_t first_block, int64_t stride, int64_t col0);
int gpx_dist2_unpack_diag(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t w, int64_t nb, gpx_mat* L, int64_t r0);
/* ... at rows r0, columns c0 of a window of block columns (gpx_dist_ivar_group_at) */
int gpx_dist2_unpack_diag_at(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t w, int64_t nb, gpx_mat* L, int64_t r0,
                             int64_t c0);
/* distributed forward / back substitution on the block-cyclic factor: diagonal-block solve, block GEMV, log-det partial */
int gpx_dist2_trsv_diag(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* v, int64_t voff,
                        int transposed);
int gpx_dist2_gemv(gpx_ctx* ctx, const gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, const gpx_mat* x,
                   int64_t xoff, gpx_mat* acc, int64_t aoff, int transposed);
int gpx_dist2_logdet_acc(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, int64_t n_valid, gpx_mat* acc);

/* Recorded programs: the Python panel loop (gpexp_amd/dist.py) runs once against a recorder; its primitives and collectives
 * become rows of 16 int64 [opcode, handle0, handle1, handle2, a0 .. a11] (variable-length lists in `extra`, referenced by
 * offset) and are replayed natively on every step -- same call sequence, no interpreter in the issue path.  *host_ms
 * (nullable) = host time spent issuing.  Opcodes: */
enum {
  GPX_OP_STREAM = 1, GPX_OP_RECORD, GPX_OP_WAIT, GPX_OP_BEGIN, GPX_OP_DIAG_FACTOR, GPX_OP_PANEL_TRSM, GPX_OP_UPDATE,
  GPX_OP_UPDATE_MULTI, GPX_OP_UNPACK_ROWS, GPX_OP_UNPACK_DIAG, GPX_OP_PACK_ROWS, GPX_OP_PACK_DIAG, GPX_OP_BCAST_GRP,
  GPX_OP_REDUCE_GRP, GPX_OP_ALLREDUCE, GPX_OP_PANEL_BCAST, GPX_OP_IVAR_STEP, GPX_OP_TRSV_DIAG, GPX_OP_GEMV, GPX_OP_LOGDET_ACC,
  GPX_OP_VEC_OP, GPX_OP_SPIN, GPX_OP_COPY, GPX_OP_IVAR_GROUP, GPX_OP_FWD_GROUP, GPX_OP_PANEL_INV, GPX_OP_BCAST_GRP2, GPX_OP_PANEL_COPYBACK,
  GPX_OP_DIAG_STAGE, GPX_OP_DIAG_UPDATE, GPX_OP_DIAG_FACTOR_STAGED, GPX_OP_DIAG_STORE, GPX_OP_PANEL_PACK, GPX_OP_DIAG_PACK
};
int gpx_program_run(gpx_ctx* ctx, const int64_t* ops, int64_t nops, const int64_t* extra, int64_t nextra, double* host_ms);
/* The same program as a hipGraph: captured once (after it has run once the ordinary way; every stream it uses must fork from
 * and join back into stream 0 -- the panel loop emits those rows), then one hipGraphLaunch per step.  Collectives inside a
 * capture are not validated on this project's hardware: see gpexp_amd/dist.py for when the host side captures. */
typedef struct gpx_graph gpx_graph;
int gpx_program_capture(gpx_ctx* ctx, const int64_t* ops, int64_t nops, const int64_t* extra, int64_t nextra, gpx_graph** out);
int gpx_graph_launch(gpx_ctx* ctx, gpx_graph* g, double* host_ms, int64_t* nodes);
int gpx_graph_free(gpx_ctx* ctx, gpx_graph* g);

/* ---- column sums of squares of a solved cross matrix (streamed evaluation) ------------------------------- */
/* out[j] = sum over the first `rows` rows of B[i][j]^2 (host out[B->cols]): variance reduction of a solved cross matrix */
int gpx_col_sumsq(gpx_ctx* ctx, const gpx_mat* B, int64_t rows, double* out);

#ifdef __cplusplus
}
#endif
#endif /* GPX_DIST_H */
