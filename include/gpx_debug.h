/* gpx_debug.h -- TEST HOOKS of libgpx_hip.so (used by tests/ and scripts/ only; NOT part of the drop-in C ABI of gpx.h).
 *
 * The same kernels the entry points of gpx.h launch, reachable one at a time; the debug modes of the allocator; host-side
 * replays of schedules that only a multi-GPU run would otherwise exercise.  Split out of gpx.h in round 3 so that the public
 * header declares only what a maintainer of the reference would bind. */
#ifndef GPX_DEBUG_H
#define GPX_DEBUG_H
#include "gpx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* C (m x n) = beta*C + alpha*A*op(B); bt != 0: B is (n x k) used transposed; alpha,beta in {(-1,1),(1,0)};
 * lower != 0: only tiles on/below the diagonal are touched */
int gpx_dbg_gemm(gpx_ctx* ctx, const gpx_mat* A, const gpx_mat* B, gpx_mat* C, int bt, int accumulate, int lower);
/* triangular-operand GEMM modes: tri = 1 (A lower triangular, k == m), 2 (B lower-triangular n x k used transposed, bt),
 * 3 (lower C = U U^T with A = B = U upper triangular, bt); the structurally zero part of every tile's k range is skipped */
int gpx_dbg_gemm_tri(gpx_ctx* ctx, const gpx_mat* A, const gpx_mat* B, gpx_mat* C, int bt, int accumulate, int tri);
/* the same product as `parts` slices of the k range whose partial products are summed in slice order (B (n x k) used
 * transposed): mode 0 = C -= A B^T on 128-tiles, 1 = the same on/below the diagonal tiles only (square C), 2 = C -= A B^T on
 * 64-tiles (the few-row products of gpx_refit_rows), 3 = C = A B^T on 64-tiles (C may alias A) */
int gpx_dbg_gemm_ksplit(gpx_ctx* ctx, const gpx_mat* A, const gpx_mat* B, gpx_mat* C, int mode, int parts);
/* GPX_CHAOS=<seed> in the environment at gpx_create (debug): every launch site holds its stream back by a random 0.1-3 ms with
 * probability 1/4; results must not change (a dependency between the context's streams that is only met by lucky timing would). */
/* queues a kernel that spins for ~ms milliseconds (<= 500) on the selected stream: lets a test hold one stream back so that a
 * missing cross-stream dependency shows every time instead of once in a dozen runs */
int gpx_dbg_spin(gpx_ctx* ctx, int ms);
/* GPX_ALLOC_GUARD=1 in the environment at gpx_create (debug; this pool has no GPU address sanitizer): every pooled device
 * allocation carries a 4 KiB band of 0xA5 on either side, checked when the block returns to the pool; =2 also fills every
 * block with NaNs when it is handed out (a read of memory nobody wrote then shows in the results).  Returns the number of
 * blocks found overwritten so far (each also reported on stderr), or -1 when the mode is off. */
int64_t gpx_dbg_guard_violations(gpx_ctx* ctx);
/* guard mode only: overruns a scratch block by 16 bytes on purpose; 1 if the check caught it, 0 if not, < 0 on error */
int gpx_dbg_guard_selftest(gpx_ctx* ctx);
/* host logic of gpx_comm_panel_bcast: the ncclSend / ncclRecv schedule of rank `me` in a W-rank communicator, rows of 6 int64
 * (phase 1|2, is_send, piece, offset within the piece, length, peer) in issue order; *nops = rows needed (at most max_ops are
 * written).  `small` = the direct-send threshold in doubles.  No device, no RCCL: tests replay it for all ranks. */
int gpx_dbg_panel_bcast_plan(int W, int me, int64_t small_elems, int npieces, const int64_t* counts, const int* roots,
                             int64_t* ops, int64_t max_ops, int64_t* nops);
/* host logic of the deterministic column reduction: number of row chunks (= partial sums per column) a rows x pcols launch
 * uses, and the scratch bound callers allocate; the bound is monotone in both arguments (one buffer serves every sub-block
 * a sweep reduces).  No device work. */
int gpx_dbg_colreduce_plan(int64_t rows, int64_t pcols, int64_t* nchunk, int64_t* bound_elems);
/* what an assembly between X and Z (NULL: X with itself) would do: *exact = 1 when distances are formed from raw
 * coordinate differences on the VALU (wide domain relative to the length scale) instead of the centred expanded MFMA
 * product; center[d] = the origin subtracted before scaling (bounding-box midpoint; 0 for Mehler) */
int gpx_dbg_kfill_plan(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X, const gpx_mat* Z,
                       int* exact, double* center);

int gpx_dbg_spin_us(gpx_ctx* ctx, int64_t us);   /* gpx_dbg_spin in microseconds (<= 500 000) */
/* a wall-clock stamp taken on the selected stream (slot 0..1023), and a spin on the selected stream that ends `us` microseconds
 * after that stamp (no launch when us <= 0; at once when the moment has passed; never longer than 0.5 s): the paced replay releases
 * a foreign delivery at "arrival of the previous panel + what its producer took" (scripts/replay_comm.py) */
int gpx_dbg_stamp(gpx_ctx* ctx, int slot);
int gpx_dbg_spin_until(gpx_ctx* ctx, int slot, int64_t us);
/* ms between the last records of two pipeline events (gpx_event_record ids; GPX_EVENT_TIMING=1 in the environment makes them
 * carry time stamps): the per-step timeline of the distributed loop's strands without a profiler in the way
 * (scripts/dist_timeline.py).  Returns 1 when an event of the pair was never recorded / has not completed. */
int gpx_dbg_event_elapsed(gpx_ctx* ctx, int id0, int id1, double* ms);

/* One launch of the 128 x 128 Cholesky leaf on the leading block of K (in place) with the kernel's phase time stamps
   (s_memtime of its first wave: 0 start, 1 loaded, 2+3p / 3+3p / 4+3p per diagonal step p, 26 written back, 27 inverse
   done; shader clocks; 28 / 29: the constant 100 MHz clock at start / end).  fast = 1: round 5's diagonal step, 0: the general one throughout.  scripts/probe_leaf.py. */
int gpx_dbg_leaf_stamps(gpx_ctx* ctx, gpx_mat* K, int fast, int64_t* out30);

#ifdef __cplusplus
}
#endif
#endif /* GPX_DEBUG_H */
