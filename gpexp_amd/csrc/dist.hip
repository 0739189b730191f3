// Multi-GPU pieces: the RCCL communicator (dlopen'ed, so single-GPU use never touches librccl) and the device primitives of
// the distributed Cholesky.  One process per GPU; gpexp_amd/dist.py drives the panel loops and owns the rendezvous (the
// 128-byte ncclUniqueId travels through a file keyed by the launcher's port, see FileRendezvous).
//
//   2-D block-cyclic (the default; second half of this file, "gpx_dist2_*"): process grid Pr x Pc, global block (I, J) on rank
//   (I % Pr, J % Pc); ncclCommSplit row / column sub-communicators; diagonal block broadcast down the process column, every
//   piece of a panel to every rank over all xGMI links (grouped ncclSend / ncclRecv scatter + all-gather), trailing updates by
//   groups of panels in one segmented launch; every rank keeps the finished panels for the evaluation phase -- a window of
//   block columns that the streamed evaluation consumes as they arrive (gpx_dist_ivar_group_at), or a replicated copy of L.
//   The recorded-program executor (gpx_program_run) is at the end.
//
//   1-D block columns (round 1, GPX_DIST_LAYOUT=1d; first half, "gpx_dist_*"): every rank holds a full-size matrix, block
//   column j is owned by rank j % world; the owner packs + factors the panel and ncclBroadcast's it, every rank stores it and
//   updates the block columns it owns.
#include "gpx_internal.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// ---- minimal RCCL surface (ABI-compatible with rccl.h / nccl.h) -------------------------------------------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt64 = 4, ncclFloat64 = 8 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
  void* h;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Reduce)(const void*, void*, size_t, int, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*);
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)(void);
  ncclResult_t (*GroupEnd)(void);
  const char* (*GetErrorString)(ncclResult_t);
};
static Rccl g_rccl = {nullptr};

static int rccl_load() {
  if (g_rccl.h) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    gpx_set_error("cannot load librccl.so: %s", dlerror());
    return -3;
  }
#define GPX_SYM(field, name)                                        \
  do {                                                              \
    *(void**)(&g_rccl.field) = dlsym(h, name);                      \
    if (!g_rccl.field) {                                            \
      gpx_set_error("librccl.so lacks symbol %s", name);            \
      return -3;                                                    \
    }                                                               \
  } while (0)
  GPX_SYM(GetUniqueId, "ncclGetUniqueId");
  GPX_SYM(CommInitRank, "ncclCommInitRank");
  GPX_SYM(CommDestroy, "ncclCommDestroy");
  GPX_SYM(Broadcast, "ncclBroadcast");
  GPX_SYM(AllReduce, "ncclAllReduce");
  GPX_SYM(AllGather, "ncclAllGather");
  GPX_SYM(GetErrorString, "ncclGetErrorString");
#undef GPX_SYM
  // what only the 2-D layout needs is optional: a build of librccl without it still serves the 1-D layout (GPX_DIST_LAYOUT=1d),
  // and the entry points that need a missing symbol say so (need_sym below)
#define GPX_OPT(field, name) *(void**)(&g_rccl.field) = dlsym(h, name)
  GPX_OPT(Reduce, "ncclReduce");
  GPX_OPT(CommSplit, "ncclCommSplit");
  GPX_OPT(Send, "ncclSend");
  GPX_OPT(Recv, "ncclRecv");
  GPX_OPT(GroupStart, "ncclGroupStart");
  GPX_OPT(GroupEnd, "ncclGroupEnd");
#undef GPX_OPT
  g_rccl.h = h;
  return 0;
}

#define GPX_NEED_SYM(field, name)                                                                       \
  do {                                                                                                  \
    if (!g_rccl.field) {                                                                                \
      gpx_set_error("librccl.so lacks %s: the 2-D block-cyclic layout is unavailable (GPX_DIST_LAYOUT=1d)", name); \
      return -3;                                                                                        \
    }                                                                                                   \
  } while (0)

#define GPX_NCCL(call)                                                                     \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != 0) {                                                                         \
      gpx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
      return -3;                                                                           \
    }                                                                                      \
  } while (0)

namespace {

// dst[r][c] = src[r][c] for an (rows x cols) block with independent leading dimensions.  A workgroup moves COPY_RPB rows of a
// 512-column stripe (all loads first): with one 4 KB row per workgroup the panel copies of the distributed loop were
// dispatch-bound at ~190 GB/s (profiles/r03_dist_replay_trace_2x4.txt, round-3 first trace: 20 ms of copies per step).
constexpr int COPY_RPB = 8;
// lower_row0 >= 0: only the part on / below the diagonal of a square block whose first row is lower_row0 rows above this launch's
// (to the next 512-column boundary): column groups entirely above the diagonal return at once
__global__ __launch_bounds__(256) void copy2d_kernel(const double* __restrict__ src, int64_t lds_,
                                                     double* __restrict__ dst, int64_t ldd, int64_t rows,
                                                     int64_t cols, int64_t lower_row0) {
  const int64_t c2 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * COPY_RPB;
  if (c2 >= cols) return;
  if (lower_row0 >= 0 && (int64_t)blockIdx.x * 512 > lower_row0 + r0 + COPY_RPB - 1) return;
  double2 v[COPY_RPB];
#pragma unroll
  for (int i = 0; i < COPY_RPB; ++i)
    if (r0 + i < rows) v[i] = *reinterpret_cast<const double2*>(src + (r0 + i) * lds_ + c2);
#pragma unroll
  for (int i = 0; i < COPY_RPB; ++i)
    if (r0 + i < rows) *reinterpret_cast<double2*>(dst + (r0 + i) * ldd + c2) = v[i];
}

}  // namespace

static int copy2d_impl(gpx_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols,
                       bool lower) {
  if (rows <= 0 || cols <= 0) return 0;
  // grid.y is limited to 65535 row groups per launch
  for (int64_t r0 = 0; r0 < rows; r0 += 65535 * (int64_t)COPY_RPB) {
    const int64_t rr = rows - r0 < 65535 * (int64_t)COPY_RPB ? rows - r0 : 65535 * (int64_t)COPY_RPB;
    dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)((rr + COPY_RPB - 1) / COPY_RPB));
    hipLaunchKernelGGL(copy2d_kernel, grid, dim3(256), 0, ctx->stream, src + r0 * lds_, lds_, dst + r0 * ldd, ldd, rr,
                       cols, lower ? r0 : (int64_t)-1);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_copy2d(gpx_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
  return copy2d_impl(ctx, src, lds_, dst, ldd, rows, cols, false);
}

// the lower triangle of an n x n block (to 512-column granularity above the diagonal): what a factor's consumers read
int gpx_copy2d_lower(gpx_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t n) {
  return copy2d_impl(ctx, src, lds_, dst, ldd, n, n, true);
}



extern "C" {

int gpx_comm_unique_id(void* out128) {
  GPX_ARG(out128 != nullptr, "out is NULL");
  GPX_TRY(rccl_load());
  ncclUniqueId id;
  GPX_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
  return 0;
}

int gpx_comm_init(gpx_ctx* ctx, int rank, int world, const void* id128) {
  GPX_ARG(ctx && id128 && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
  GPX_ARG(ctx->comm == nullptr, "communicator already initialised");
  GPX_TRY(rccl_load());
  GPX_HIP(hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  GPX_NCCL(g_rccl.CommInitRank(&c, world, id, rank));
  ctx->comm = c;
  ctx->rank = rank;
  ctx->world = world;
  ctx->grp[0] = c;
  ctx->grp_size[0] = world;
  ctx->grp_rank[0] = rank;
  return 0;
}

int gpx_comm_destroy(gpx_ctx* ctx) {
  if (!ctx || !ctx->comm) return 0;
  (void)hipDeviceSynchronize();
  for (int g = 1; g < 3; ++g) {
    if (ctx->grp[g]) g_rccl.CommDestroy((ncclComm_t)ctx->grp[g]);
    ctx->grp[g] = nullptr;
  }
  g_rccl.CommDestroy((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
  ctx->grp[0] = nullptr;
  return 0;
}

// ---- process grid + sub-communicators (Pr x Pc grid, ncclCommSplit) ---------------------------------
// rank = pr * Pc + pc.  Group 1 = the rank's process ROW (ranks (pr, *), Pc of them, group rank pc); group 2 = its
// process COLUMN (ranks (*, pc), Pr of them, group rank pr).
int gpx_comm_grid(gpx_ctx* ctx, int Pr, int Pc) {
  GPX_ARG(ctx && ctx->comm, "communicator not initialised");
  GPX_ARG(Pr >= 1 && Pc >= 1 && Pr * Pc == ctx->world, "grid does not match the communicator size");
  GPX_ARG(ctx->grp[1] == nullptr && ctx->grp[2] == nullptr, "process grid already set");
  GPX_NEED_SYM(CommSplit, "ncclCommSplit");
  GPX_NEED_SYM(Send, "ncclSend");
  GPX_NEED_SYM(Recv, "ncclRecv");
  GPX_NEED_SYM(GroupStart, "ncclGroupStart");
  GPX_NEED_SYM(GroupEnd, "ncclGroupEnd");
  GPX_NEED_SYM(Reduce, "ncclReduce");
  const int pr = ctx->rank / Pc, pc = ctx->rank % Pc;
  ncclComm_t row = nullptr, col = nullptr;
  GPX_NCCL(g_rccl.CommSplit((ncclComm_t)ctx->comm, /*color*/ pr, /*key*/ pc, &row, nullptr));
  {
    const ncclResult_t r2 = g_rccl.CommSplit((ncclComm_t)ctx->comm, /*color*/ Pr + pc, /*key*/ pr, &col, nullptr);
    if (r2 != 0) {  // no half-built grid: the row communicator goes back
      if (row) g_rccl.CommDestroy(row);
      gpx_set_error("%s:%d ncclCommSplit (process column) -> %s", __FILE__, __LINE__, g_rccl.GetErrorString(r2));
      return -3;
    }
  }
  ctx->grp[1] = row;
  ctx->grp_size[1] = Pc;
  ctx->grp_rank[1] = pc;
  ctx->grp[2] = col;
  ctx->grp_size[2] = Pr;
  ctx->grp_rank[2] = pr;
  ctx->Pr = Pr;
  ctx->Pc = Pc;
  return 0;
}

static int need_group(gpx_ctx* ctx, int grp) {
  GPX_ARG(ctx && grp >= 0 && grp < 3 && ctx->grp[grp], "communicator group not initialised (gpx_comm_init / gpx_comm_grid)");
  return 0;
}

// broadcast buf[offset .. offset+count) inside a group from the group rank `root`; asynchronous on the selected stream
int gpx_comm_bcast_grp(gpx_ctx* ctx, gpx_mat* buf, int64_t offset, int64_t count, int root, int grp) {
  GPX_ARG(buf != nullptr, "buffer is NULL");
  GPX_TRY(need_group(ctx, grp));
  GPX_ARG(offset >= 0 && count >= 0 && (offset + count) * 8 <= buf->bytes, "broadcast range exceeds the buffer");
  GPX_ARG(root >= 0 && root < ctx->grp_size[grp], "root outside the group");
  if (count == 0 || ctx->grp_size[grp] == 1) return 0;
  ProfScope ps(ctx, GPX_PROF_COMM, 0.0, 8.0 * (double)count);
  buf->bbox_ok = 0;  // a cached bounding box (point sets) does not survive a device-side write
  double* p = buf->p + offset;
  GPX_NCCL(g_rccl.Broadcast(p, p, (size_t)count, ncclFloat64, root, (ncclComm_t)ctx->grp[grp], ctx->stream));
  return 0;
}

// out-of-place form: the root sends sbuf[soff ..], EVERY member (the root too) receives into rbuf[roff ..].  Round 4: the block
// row the next diagonal needs travels into a buffer of its own, so that the panel broadcast -- which lands the same bytes on the
// panel buffer a moment later -- never has to wait for the kernel that is reading them (that wait sat on the chain across ranks).
int gpx_comm_bcast_grp2(gpx_ctx* ctx, const gpx_mat* sbuf, int64_t soff, gpx_mat* rbuf, int64_t roff, int64_t count, int root, int grp) {
  GPX_ARG(sbuf && rbuf, "buffer is NULL");
  GPX_TRY(need_group(ctx, grp));
  GPX_ARG(soff >= 0 && roff >= 0 && count >= 0 && (soff + count) * 8 <= sbuf->bytes && (roff + count) * 8 <= rbuf->bytes,
          "broadcast range exceeds a buffer");
  GPX_ARG(root >= 0 && root < ctx->grp_size[grp], "root outside the group");
  if (count == 0) return 0;
  rbuf->bbox_ok = 0;
  if (ctx->grp_size[grp] == 1) {   // alone in the group: the "receive" is a local copy
    GPX_HIP(hipMemcpyAsync(rbuf->p + roff, sbuf->p + soff, (size_t)count * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  ProfScope ps(ctx, GPX_PROF_COMM, 0.0, 8.0 * (double)count);
  GPX_NCCL(g_rccl.Broadcast(sbuf->p + soff, rbuf->p + roff, (size_t)count, ncclFloat64, root, (ncclComm_t)ctx->grp[grp], ctx->stream));
  return 0;
}

// in-place sum of buf[offset .. offset+count) over a group, result on the group rank `root` (others keep their input)
int gpx_comm_reduce_grp(gpx_ctx* ctx, gpx_mat* buf, int64_t offset, int64_t count, int root, int grp) {
  GPX_ARG(buf != nullptr, "buffer is NULL");
  GPX_TRY(need_group(ctx, grp));
  GPX_ARG(offset >= 0 && count >= 0 && (offset + count) * 8 <= buf->bytes, "reduce range exceeds the buffer");
  GPX_ARG(root >= 0 && root < ctx->grp_size[grp], "root outside the group");
  if (count == 0 || ctx->grp_size[grp] == 1) return 0;
  ProfScope ps(ctx, GPX_PROF_COMM, 0.0, 8.0 * (double)count);
  buf->bbox_ok = 0;  // a cached bounding box (point sets) does not survive a device-side write
  double* p = buf->p + offset;
  GPX_NCCL(g_rccl.Reduce(p, p, (size_t)count, ncclFloat64, ncclSum, root, (ncclComm_t)ctx->grp[grp], ctx->stream));
  return 0;
}

// in-place sum over ALL ranks of buf[offset .. offset+count), result everywhere (ncclAllReduce; the replicated alpha)
int gpx_comm_allreduce(gpx_ctx* ctx, gpx_mat* buf, int64_t offset, int64_t count) {
  GPX_ARG(buf != nullptr, "buffer is NULL");
  GPX_TRY(need_group(ctx, 0));
  GPX_ARG(offset >= 0 && count >= 0 && (offset + count) * 8 <= buf->bytes, "allreduce range exceeds the buffer");
  if (count == 0 || ctx->world == 1) return 0;
  ProfScope ps(ctx, GPX_PROF_COMM, 0.0, 8.0 * (double)count);
  buf->bbox_ok = 0;  // a cached bounding box (point sets) does not survive a device-side write
  double* p = buf->p + offset;
  GPX_NCCL(g_rccl.AllReduce(p, p, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

// host scalars summed over all ranks (logdet, y^T alpha partials); blocking
int gpx_comm_allreduce_host(gpx_ctx* ctx, double* inout, int64_t n) {
  GPX_ARG(ctx && inout && ctx->comm && n > 0 && n <= 64, "bad allreduce arguments (at most 64 scalars)");
  double* d = ctx->d_scal;
  GPX_HIP(hipMemcpyAsync(d, inout, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
  if (ctx->world > 1)
    GPX_NCCL(g_rccl.AllReduce(d, d, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  GPX_HIP(hipMemcpyAsync(inout, d, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// ---- all-link panel broadcast ---------------------------------------------------------------------------------------
// npieces regions of `buf` (same layout on every rank), region i owned by world rank roots[i], are delivered to every
// rank.  xGMI is a full point-to-point mesh (7 links x ~153 GB/s per GPU): a ring / tree broadcast moves the whole
// region over one link at a time, so instead every region is cut into world-1 chunks,
//   phase 1 (scatter)     the root sends chunk q to its q-th peer          -- root egress spread over all its links
//   phase 2 (all-gather)  every peer forwards its chunk to the other peers  -- every link of the mesh carries 1/(W-1)
// as two grouped sets of ncclSend / ncclRecv: 2/(W-1) of the region's bytes per link instead of all of them.  Regions
// below 2^16 doubles go from the root to every peer directly (one phase, latency-bound anyway).
// Asynchronous on the selected stream; every rank must call it with identical arguments.
static int64_t sag_min_elems() { return (int64_t)1 << 16; }

}  // extern "C"

// The schedule itself, separated from RCCL so that a host test can replay it for every rank of a communicator (the multi-rank
// RCCL path has never run on hardware; tests/test_host_cpu.py checks that the sends and receives of all ranks pair up in
// order with equal lengths and that every rank ends up with every piece).  op(phase, is_send, piece, off, len, peer): phase 1
// or 2, element range [off, off+len) of the piece.  Returns whether any piece takes the two-phase route.
template <class Op>
static bool panel_bcast_schedule(int W, int me, int64_t small, int npieces, const int64_t* counts, const int* roots, int phase,
                                 Op op) {
  // chunk q of a piece: [q*c, min((q+1)*c, count)), c even (16-byte aligned sends); peer index of rank r w.r.t. the root:
  // r < root ? r : r-1
  auto chunk = [](int64_t count, int W_, int q, int64_t* off, int64_t* len) {
    int64_t c = (count + (W_ - 2)) / (W_ - 1);
    c += c & 1;
    int64_t a = (int64_t)q * c, b = a + c;
    if (a > count) a = count;
    if (b > count) b = count;
    *off = a;
    *len = b - a;
  };
  bool any = false;
  for (int i = 0; i < npieces; ++i) any = any || (counts[i] >= small && W > 2);
  if (phase == 1) {
    for (int i = 0; i < npieces; ++i) {
      const int root = roots[i];
      if (counts[i] == 0) continue;
      const bool direct = counts[i] < small || W == 2;
      if (me == root) {
        for (int r = 0; r < W; ++r) {
          if (r == root) continue;
          if (direct) {
            op(1, true, i, (int64_t)0, counts[i], r);
          } else {
            int64_t off, len;
            chunk(counts[i], W, r < root ? r : r - 1, &off, &len);
            if (len > 0) op(1, true, i, off, len, r);
          }
        }
      } else if (direct) {
        op(1, false, i, (int64_t)0, counts[i], root);
      } else {
        int64_t off, len;
        chunk(counts[i], W, me < root ? me : me - 1, &off, &len);
        if (len > 0) op(1, false, i, off, len, root);
      }
    }
    return any;
  }
  for (int i = 0; i < npieces; ++i) {
    const int root = roots[i];
    if (counts[i] < small || W == 2 || me == root) continue;
    int64_t myoff, mylen;
    chunk(counts[i], W, me < root ? me : me - 1, &myoff, &mylen);
    for (int r = 0; r < W; ++r) {
      if (r == root || r == me) continue;
      int64_t off, len;
      chunk(counts[i], W, r < root ? r : r - 1, &off, &len);
      if (mylen > 0) op(2, true, i, myoff, mylen, r);
      if (len > 0) op(2, false, i, off, len, r);
    }
  }
  return any;
}

extern "C" {

int gpx_comm_panel_bcast(gpx_ctx* ctx, gpx_mat* buf, const int64_t* offsets, const int64_t* counts, const int* roots,
                         int npieces) {
  GPX_ARG(buf && offsets && counts && roots && npieces >= 0, "NULL argument");
  GPX_TRY(need_group(ctx, 0));
  const int W = ctx->world, me = ctx->rank;
  double total = 0.0;
  for (int i = 0; i < npieces; ++i) {
    GPX_ARG(offsets[i] >= 0 && counts[i] >= 0 && (offsets[i] + counts[i]) * 8 <= buf->bytes, "piece exceeds the buffer");
    GPX_ARG(roots[i] >= 0 && roots[i] < W, "piece root outside the communicator");
    total += 8.0 * (double)counts[i];
  }
  buf->bbox_ok = 0;
  if (W == 1 || npieces == 0) return 0;
  ProfScope ps(ctx, GPX_PROF_COMM, 0.0, total);
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  hipStream_t st = ctx->stream;
  int rc = 0;
  auto op = [&](int, bool is_send, int piece, int64_t off, int64_t len, int peer) {
    if (rc != 0) return;
    double* p = buf->p + offsets[piece] + off;
    ncclResult_t r = is_send ? g_rccl.Send(p, (size_t)len, ncclFloat64, peer, comm, st)
                             : g_rccl.Recv(p, (size_t)len, ncclFloat64, peer, comm, st);
    if (r != 0) {  // ncclSuccess
      gpx_set_error("RCCL %s failed: %s", is_send ? "ncclSend" : "ncclRecv", g_rccl.GetErrorString(r));
      rc = -3;
    }
  };
  GPX_NCCL(g_rccl.GroupStart());
  const bool two_phase = panel_bcast_schedule(W, me, sag_min_elems(), npieces, counts, roots, 1, op);
  GPX_NCCL(g_rccl.GroupEnd());
  if (rc != 0) return rc;
  if (!two_phase) return 0;
  GPX_NCCL(g_rccl.GroupStart());
  panel_bcast_schedule(W, me, sag_min_elems(), npieces, counts, roots, 2, op);
  GPX_NCCL(g_rccl.GroupEnd());
  return rc;
}

// test hook (host logic only): the schedule of rank `me` of a W-rank communicator as rows of 6 int64
// (phase, is_send, piece, off, len, peer) in issue order; *nops = number of rows (may exceed max_ops: then only max_ops are written)
int gpx_dbg_panel_bcast_plan(int W, int me, int64_t small, int npieces, const int64_t* counts, const int* roots, int64_t* ops,
                             int64_t max_ops, int64_t* nops) {
  if (W < 1 || me < 0 || me >= W || npieces < 0 || !counts || !roots || !nops) return -1;
  int64_t n = 0;
  auto op = [&](int phase, bool is_send, int piece, int64_t off, int64_t len, int peer) {
    if (ops && n < max_ops) {
      int64_t* r = ops + 6 * n;
      r[0] = phase; r[1] = is_send ? 1 : 0; r[2] = piece; r[3] = off; r[4] = len; r[5] = peer;
    }
    ++n;
  };
  if (W > 1) {
    if (panel_bcast_schedule(W, me, small, npieces, counts, roots, 1, op)) panel_bcast_schedule(W, me, small, npieces, counts, roots, 2, op);
  }
  *nops = n;
  return 0;
}

// broadcast `count` doubles of a device matrix (its first `count` elements) from `root`; asynchronous on the stream
int gpx_comm_bcast(gpx_ctx* ctx, gpx_mat* buf, int64_t count, int root) {
  GPX_ARG(ctx && buf && ctx->comm, "communicator not initialised");
  GPX_ARG(count >= 0 && count * 8 <= buf->bytes, "broadcast count exceeds the buffer");
  buf->bbox_ok = 0;
  ProfScope ps(ctx, GPX_PROF_COMM, 0.0, 8.0 * (double)count);
  GPX_NCCL(g_rccl.Broadcast(buf->p, buf->p, (size_t)count, ncclFloat64, root, (ncclComm_t)ctx->comm, ctx->stream));
  return 0;
}

// host vector of n doubles: gathered from every rank in rank order into out[world*n] (blocking)
int gpx_comm_allgather_host(gpx_ctx* ctx, const double* in, int64_t n, double* out) {
  GPX_ARG(ctx && in && out && ctx->comm && n > 0, "bad allgather arguments");
  void *ps_, *pr;
  GPX_TRY(gpx_dev_alloc(ctx, n * 8, &ps_));
  int r = gpx_dev_alloc(ctx, n * 8 * ctx->world, &pr);
  if (r != 0) {
    gpx_dev_release(ctx, ps_, n * 8);
    return r;
  }
  do {
    if (hipMemcpyAsync(ps_, in, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { r = -2; break; }
    if (g_rccl.AllGather(ps_, pr, (size_t)n, ncclFloat64, (ncclComm_t)ctx->comm, ctx->stream) != 0) { r = -3; break; }
    if (hipMemcpyAsync(out, pr, (size_t)n * 8 * ctx->world, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { r = -2; break; }
  } while (0);
  (void)hipStreamSynchronize(ctx->stream);
  gpx_dev_release(ctx, ps_, n * 8);
  gpx_dev_release(ctx, pr, n * 8 * ctx->world);
  if (r != 0) gpx_set_error("allgather failed (%d)", r);
  return r;
}

// ---- distributed covariance assembly: only block columns owned by `rank` are written ---------------------------
int gpx_dist_kfill(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X, const double* nugget,
                   int64_t nugget_len, gpx_mat* K, int64_t nb, int rank, int world) {
  GPX_ARG(ctx && X && K, "NULL argument");
  GPX_ARG(nb > 0 && nb % GPX_TILE == 0 && world >= 1 && rank >= 0 && rank < world, "bad block-cyclic parameters");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_ARG(X->cols == d && X->pcols == d, "X must be an unpadded (N x d) point set");
  GPX_ARG(K->rows == X->rows && K->cols == X->rows && K->prows == K->pcols, "K must be the padded N x N matrix");
  GPX_ARG(nugget_len == 0 || nugget_len == 1 || nugget_len == X->rows, "nugget_len must be 0, 1 or N");
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X));
  const int64_t n = X->rows, np = K->prows;
  double* d_nug = nullptr;
  int64_t nug_bytes = 0;
  double nscal = nugget_len == 1 ? nugget[0] : 0.0;
  if (nugget_len > 1) {
    nug_bytes = nugget_len * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, nug_bytes, &p));
    d_nug = (double*)p;
    GPX_HIP(hipMemcpyAsync(d_nug, nugget, (size_t)nug_bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  int r = 0;
  const int64_t nblk = (np + nb - 1) / nb;
  for (int64_t j = rank; j < nblk && r == 0; j += world) {
    const int64_t c0 = j * nb;
    const int64_t cw = (np - c0) < nb ? (np - c0) : nb;
    // rows >= c0 only (lower block column): K[c0:, c0:c0+cw]
    r = launch_kfill_offset(ctx, kp, X->p, n, c0, c0, d_nug, nugget_len, nscal, K->p + c0 * K->ld + c0, np - c0, cw,
                            K->ld);
  }
  if (d_nug) {
    (void)hipStreamSynchronize(ctx->stream);
    gpx_dev_release(ctx, d_nug, nug_bytes);
  }
  K->binv_ib = 0;  // block inverses (chol_potrs) belong to the previous contents
  K->factored = 0;
  return r;
}

// elements of the packed panel buffer for step k: rows x nb panel + (nb/128) inverted 128x128 leaves
int64_t gpx_dist_panel_elems(int64_t np, int64_t nb) { return np * nb + (nb / GPX_TILE) * GPX_TILE * GPX_TILE; }

// owner side of step k: pack the panel into P, factor it there, append the leaf inverses.  Asynchronous on the
// selected stream; non-positive pivots accumulate in the context's flag (gpx_dist_info) -- reset it with
// gpx_dist_begin before the first panel.
int gpx_dist_begin(gpx_ctx* ctx) {
  GPX_ARG(ctx != nullptr, "ctx is NULL");
  GPX_HIP(hipMemsetAsync(ctx->d_info, 0, 2 * sizeof(int), ctx->stream));
  return 0;
}

int gpx_dist_info(gpx_ctx* ctx, int* info) {
  GPX_ARG(ctx && info, "NULL argument");
  GPX_HIP(hipSetDevice(ctx->device));
  GPX_HIP(hipDeviceSynchronize());
  GPX_HIP(hipMemcpy(info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}

int gpx_dist_panel_factor(gpx_ctx* ctx, gpx_mat* K, int64_t k, int64_t nb, gpx_mat* P) {
  GPX_ARG(ctx && K && P, "NULL argument");
  const int64_t np = K->prows, r0 = k * nb;
  GPX_ARG(nb % GPX_TILE == 0 && r0 < np, "bad panel index");
  const int64_t w = (np - r0) < nb ? (np - r0) : nb, rows = np - r0;
  GPX_ARG(P->bytes >= gpx_dist_panel_elems(np, nb) * 8, "panel buffer too small");
  double* pb = P->p;
  double* pinv = P->p + rows * nb;  // leaf inverses of this panel
  GPX_TRY(gpx_copy2d(ctx, K->p + r0 * K->ld + r0, K->ld, pb, nb, rows, w));
  GPX_TRY(chol_potrf_nozero(ctx, pb, nb, w, pinv, r0, K->rows));
  if (rows > w) GPX_TRY(chol_trsm_right(ctx, pb, nb, pinv, pb + w * nb, nb, rows - w, w));
  return 0;
}

// every rank, after P has arrived: keep the panel and its leaf inverses in the local matrix
int gpx_dist_panel_store(gpx_ctx* ctx, gpx_mat* K, int64_t k, int64_t nb, const gpx_mat* P) {
  GPX_ARG(ctx && K && P, "NULL argument");
  const int64_t np = K->prows, r0 = k * nb;
  GPX_ARG(nb % GPX_TILE == 0 && r0 < np, "bad panel index");
  const int64_t w = (np - r0) < nb ? (np - r0) : nb, rows = np - r0;
  if (!K->aux) {
    K->aux_bytes = K->prows * GPX_TILE * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, K->aux_bytes, &p));
    K->aux = (double*)p;
  }
  GPX_TRY(gpx_copy2d(ctx, P->p, nb, K->p + r0 * K->ld + r0, K->ld, rows, w));
  GPX_HIP(hipMemcpyAsync(K->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, P->p + rows * nb,
                         (size_t)((w / GPX_TILE) * GPX_TILE * GPX_TILE * 8), hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

// apply panel k to the owned block columns j in [j0, j1):  C[j*nb:, j] -= P[j..] P[j]^T.
// With one rank and j1 == number of blocks the whole trailing range is one lower-triangular SYRK launch.
int gpx_dist_panel_update(gpx_ctx* ctx, gpx_mat* K, int64_t k, int64_t nb, const gpx_mat* P, int64_t j0, int64_t j1,
                          int rank, int world) {
  GPX_ARG(ctx && K && P, "NULL argument");
  const int64_t np = K->prows, r0 = k * nb;
  GPX_ARG(nb % GPX_TILE == 0 && r0 < np && world >= 1, "bad panel index");
  const int64_t w = (np - r0) < nb ? (np - r0) : nb;
  const int64_t nblk = (np + nb - 1) / nb;
  if (j0 <= k) j0 = k + 1;
  if (j1 > nblk) j1 = nblk;
  const double* pb = P->p;
  if (world == 1 && j1 == nblk && j1 - j0 > 1) {
    const int64_t c0 = j0 * nb;
    return launch_gemm(ctx, pb + (c0 - r0) * nb, nb, pb + (c0 - r0) * nb, nb, K->p + c0 * K->ld + c0, K->ld, np - c0,
                       np - c0, w, true, true, true);
  }
  for (int64_t j = j0; j < j1; ++j) {
    if (j % world != rank) continue;
    const int64_t c0 = j * nb;
    const int64_t cw = (np - c0) < nb ? (np - c0) : nb;
    GPX_TRY(launch_gemm(ctx, pb + (c0 - r0) * nb, nb, pb + (c0 - r0) * nb, nb, K->p + c0 * K->ld + c0, K->ld, np - c0,
                        cw, w, true, true, false));
  }
  return 0;
}

// Streamed evaluation: step k of a right-looking LEFT triangular solve of B (np x mcp, the cross matrix K(X, Z_local))
// against the factor, using only block column k of L -- i.e. exactly the panel that has just arrived and been stored:
//   B_k <- L_kk^-1 B_k;   B[(k+1)nb:, :] -= L[(k+1)nb:, k] B_k.
// Issued on the background stream right after panel_store(k), it fills the time the rank would otherwise idle waiting
// for the next panel (at 8 GPUs the panel chain is broadcast-bound and the trailing update per rank is short).
int gpx_dist_ivar_step(gpx_ctx* ctx, const gpx_mat* K, int64_t k, int64_t nb, gpx_mat* B) {
  GPX_ARG(ctx && K && B && K->aux, "NULL argument / no stored panel yet");
  const int64_t np = K->prows, r0 = k * nb;
  GPX_ARG(nb % GPX_TILE == 0 && r0 < np && B->prows == np, "bad panel index / B does not match the factor");
  const int64_t w = (np - r0) < nb ? (np - r0) : nb, below = np - r0 - w, mcp = B->pcols;
  double* Bk = B->p + r0 * B->ld;
  GPX_TRY(chol_trsm_left(ctx, K->p + r0 * K->ld + r0, K->ld, K->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, Bk, B->ld, w,
                         mcp));
  if (below > 0)
    GPX_TRY(launch_gemm(ctx, K->p + (r0 + w) * K->ld + r0, K->ld, Bk, B->ld, B->p + (r0 + w) * B->ld, B->ld, below, mcp, w,
                        false, true, false));
  return 0;
}

// Streamed evaluation, GROUP form: the panels k0 .. k1 of the factor are stored.  With r0 = k0 nb, r1 = min((k1 + 1) nb, np):
//   B[r0:r1] <- L[r0:r1, r0:r1]^-1 B[r0:r1];   B[r1:, :] -= L[r1:, r0:r1] B[r0:r1]
// i.e. (k1 - k0 + 1) steps of gpx_dist_ivar_step at once: the update below the group runs with K = r1 - r0 (2048 at nb = 512,
// four panels) instead of one K = nb product per panel -- the same aggregation as the factorisation's trailing updates.
int gpx_dist_ivar_group(gpx_ctx* ctx, const gpx_mat* K, int64_t k0, int64_t k1, int64_t nb, gpx_mat* B) {
  return gpx_dist_ivar_group_at(ctx, K, k0, k1, nb, B, k0 * nb);
}

// The same against a WINDOW of the factor: K holds only the block columns of the group, starting at its column c0 (rows: all
// of them, leaf inverses per row block as usual).  With c0 = (k0 % window) * nb the streamed evaluation needs no N x N copy of
// the factor on any rank -- each panel is consumed when it arrives and its column slot is reused `window` panels later (SURVEY
// 8e (1): "keep L distributed" above the size where a replica is cheap).
int gpx_dist_ivar_group_at(gpx_ctx* ctx, const gpx_mat* K, int64_t k0, int64_t k1, int64_t nb, gpx_mat* B, int64_t c0) {
  GPX_ARG(ctx && K && B && K->aux, "NULL argument / no stored panel yet");
  const int64_t np = K->prows, r0 = k0 * nb;
  GPX_ARG(nb % GPX_TILE == 0 && k0 >= 0 && k1 >= k0 && r0 < np && B->prows == np, "bad panel range / B does not match the factor");
  const int64_t r1 = (k1 + 1) * nb < np ? (k1 + 1) * nb : np, w = r1 - r0, below = np - r1, mcp = B->pcols;
  GPX_ARG(c0 >= 0 && c0 % GPX_TILE == 0 && c0 + w <= K->pcols, "the group's columns fall outside the stored window");
  double* Bk = B->p + r0 * B->ld;
  // The group's triangle through EXPLICIT inverses of its nb-order diagonal blocks (built here from the leaf inverses, batched:
  // ~0.1 ms per group): tri-GEMMs + K >= nb updates instead of the leaf-level recursion (128-row strip kernels and K = 128..1024
  // products: 15-20 ms of the evaluation stream per C4 step on a rank of 8, against 4 ms of flops).  Order of the inverses: nb.
  const int64_t ib = nb;
  if (ib >= GPX_TILE && w > ib) {
    const int64_t nblk = (w + ib - 1) / ib;
    const int64_t need = (2 * nblk * ib * ib + w * mcp) * 8;
    if (ctx->ev_scratch_bytes < need) {
      GPX_HIP(hipDeviceSynchronize());
      if (ctx->ev_scratch) (void)hipFree(ctx->ev_scratch);
      ctx->ev_scratch = nullptr;
      ctx->ev_scratch_bytes = 0;
      GPX_HIP(hipMalloc((void**)&ctx->ev_scratch, (size_t)need));
      ctx->ev_scratch_bytes = need;
    }
    double* inv = ctx->ev_scratch;
    double* tmp = inv + nblk * ib * ib;
    double* W = tmp + nblk * ib * ib;
    return chol_trsm_left_group(ctx, K->p + r0 * K->ld + c0, K->ld, K->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, w, below, ib, Bk,
                                B->ld, mcp, inv, tmp, W, mcp);
  }
  GPX_TRY(chol_trsm_left(ctx, K->p + r0 * K->ld + c0, K->ld, K->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, Bk, B->ld, w, mcp));
  if (below > 0)
    GPX_TRY(launch_gemm(ctx, K->p + r1 * K->ld + c0, K->ld, Bk, B->ld, B->p + r1 * B->ld, B->ld, below, mcp, w, false, true, false));
  return 0;
}

// Forward substitution riding along the streamed evaluation: v (padded N doubles, y at the start) takes the group's step of
// L w = y against the same window of the factor --  v[r0:r1] <- L[r0:r1, r0:r1]^-1 v[r0:r1];  v[r1:] -= L[r1:, r0:r1] v[r0:r1].
// Every rank receives every panel, so every rank ends with the complete w = L^-1 y and the distributed substitution needs no
// forward sweep (N / nb block steps with a reduce and a broadcast each); memory-bound: one more read of the window's group.
int gpx_dist_fwd_group_at(gpx_ctx* ctx, const gpx_mat* K, int64_t k0, int64_t k1, int64_t nb, gpx_mat* v, int64_t c0) {
  GPX_ARG(ctx && K && v && K->aux, "NULL argument / no stored panel yet");
  const int64_t np = K->prows, r0 = k0 * nb;
  GPX_ARG(nb % GPX_TILE == 0 && k0 >= 0 && k1 >= k0 && r0 < np && v->bytes >= np * 8, "bad panel range / vector shorter than the factor");
  const int64_t r1 = (k1 + 1) * nb < np ? (k1 + 1) * nb : np, w = r1 - r0, below = np - r1;
  GPX_ARG(c0 >= 0 && c0 % GPX_TILE == 0 && c0 + w <= K->pcols, "the group's columns fall outside the stored window");
  v->bbox_ok = 0;
  GPX_TRY(chol_trsv_with_scratch(ctx, K->p + r0 * K->ld + c0, K->ld, K->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, v->p + r0, w,
                                 false, nullptr));
  if (below > 0) GPX_TRY(launch_gemv_sub(ctx, K->p + r1 * K->ld + c0, K->ld, below, w, v->p + r0, v->p + r1));
  return 0;
}

// =====================================================================================================================
// 2-D block-cyclic distributed Cholesky (north_star).  Process grid Pr x Pc, rank (pr, pc) = (rank / Pc,
// rank % Pc); global block (I, J) of the padded matrix (block size nb, the last block may be shorter) lives on rank
// (I % Pr, J % Pc) at local block (I / Pr, J / Pc) of that rank's LOCAL matrix -- each rank allocates only its
// ~N^2 / (Pr Pc) share.  gpexp_amd/dist.py drives the panel loop; the pieces below are its device primitives.  A panel
// step k moves data through ONE packed buffer G per rank with Pr "pieces", piece p = what process row p contributes:
//     [ D: nb x nb factored diagonal block (row stride nb) | nb/128 inverted 128x128 leaves | rows: m_p x nb, row stride
//       gpx_dist2_row_stride(nb) ]
// where rows = the blocks L_Ik, I > k, I % Pr == p, in ascending I (D is only meaningful in piece k % Pr).
// =====================================================================================================================
namespace {

// dst[((r / nb) * stride + first) * nb + r % nb][c] = src[r][c]: packed piece rows -> rows of the replicated factor
__global__ __launch_bounds__(256) void copy_cyclic_rows_kernel(const double* __restrict__ src, int64_t lds_,
                                                               double* __restrict__ dst, int64_t ldd, int64_t rows,
                                                               int64_t cols, int64_t nb, int64_t first, int64_t stride) {
  const int64_t c2 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * COPY_RPB;   // COPY_RPB divides nb: the rows of a group share their block
  if (c2 >= cols || r0 >= rows) return;
  const int64_t g0 = ((r0 / nb) * stride + first) * nb + r0 % nb;
  double2 v[COPY_RPB];
#pragma unroll
  for (int i = 0; i < COPY_RPB; ++i)
    if (r0 + i < rows) v[i] = *reinterpret_cast<const double2*>(src + (r0 + i) * lds_ + c2);
#pragma unroll
  for (int i = 0; i < COPY_RPB; ++i)
    if (r0 + i < rows) *reinterpret_cast<double2*>(dst + (g0 + i) * ldd + c2) = v[i];
}

// src[((r / nb) * stride + first) * nb + r % nb][c] -> dst[r][c]: the inverse of copy_cyclic_rows_kernel (rows of a
// full-size factor -> packed piece rows; the single-rank REPLAY of the distributed loop stages "received" pieces with it)
__global__ __launch_bounds__(256) void gather_cyclic_rows_kernel(const double* __restrict__ src, int64_t lds_,
                                                                 double* __restrict__ dst, int64_t ldd, int64_t rows,
                                                                 int64_t cols, int64_t nb, int64_t first, int64_t stride) {
  const int64_t c2 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r0 = (int64_t)blockIdx.y * COPY_RPB;
  if (c2 >= cols || r0 >= rows) return;
  const int64_t g0 = ((r0 / nb) * stride + first) * nb + r0 % nb;
  double2 v[COPY_RPB];
#pragma unroll
  for (int i = 0; i < COPY_RPB; ++i)
    if (r0 + i < rows) v[i] = *reinterpret_cast<const double2*>(src + (g0 + i) * lds_ + c2);
#pragma unroll
  for (int i = 0; i < COPY_RPB; ++i)
    if (r0 + i < rows) *reinterpret_cast<double2*>(dst + (r0 + i) * ldd + c2) = v[i];
}

// acc[c] -= sum_r A[r][c] x[r] over an m x w block (row stride ld), m a multiple of 4: a workgroup owns 128 columns (64
// column pairs x 4 row groups), every row group walks its quarter of the rows with coalesced 16-byte loads, the four
// partial sums meet in LDS in a fixed order -- deterministic, no scratch, no host synchronisation (the column reduction
// this replaces allocated its partial sums from the pool and synchronised the stream on every block of the back
// substitution: 64 host round trips per solve at C4)
__global__ __launch_bounds__(256) void gemv_t_sub_kernel(const double* __restrict__ A, int64_t ld, int64_t m, int64_t w,
                                                         const double* __restrict__ x, double* __restrict__ acc) {
  __shared__ double2 part[4][64];
  const int cp = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 128 + 2 * cp;
  const int64_t rows_per = (m + 3) / 4, r0 = rg * rows_per, r1 = (r0 + rows_per) < m ? (r0 + rows_per) : m;
  double s0 = 0.0, s1 = 0.0;
  if (c < w) {
    const double* ap = A + r0 * ld + c;
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
      double2 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2*>(ap + (int64_t)i * ld);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double xv = x[r + i];
        s0 = fma(v[i].x, xv, s0);
        s1 = fma(v[i].y, xv, s1);
      }
      ap += 8 * ld;
    }
    for (; r < r1; ++r) {
      const double2 v = *reinterpret_cast<const double2*>(ap);
      const double xv = x[r];
      s0 = fma(v.x, xv, s0);
      s1 = fma(v.y, xv, s1);
      ap += ld;
    }
  }
  part[rg][cp] = double2{s0, s1};
  __syncthreads();
  if (rg == 0 && c < w) {
    const double t0 = (part[0][cp].x + part[1][cp].x) + (part[2][cp].x + part[3][cp].x);
    const double t1 = (part[0][cp].y + part[1][cp].y) + (part[2][cp].y + part[3][cp].y);
    double2 a = *reinterpret_cast<double2*>(acc + c);
    a.x -= t0;
    a.y -= t1;
    *reinterpret_cast<double2*>(acc + c) = a;
  }
}

// out[0] += 2 * sum_i log(L[i][i]) over a w x w diagonal block (deterministic: one workgroup, fixed tree)
__global__ __launch_bounds__(256) void logdet_acc_kernel(const double* __restrict__ L, int64_t ld, int64_t w,
                                                         int64_t n_valid, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < w && i < n_valid; i += 256) s += log(L[i * ld + i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] += 2.0 * red[0];
}

}  // namespace

static int check_local(const gpx_mat* A, int64_t lr, int64_t m, int64_t lc, int64_t w) {
  GPX_ARG(A != nullptr, "local matrix is NULL");
  GPX_ARG(lr >= 0 && lc >= 0 && m >= 0 && w >= 0 && lr + m <= A->prows && lc + w <= A->pcols, "block outside the local matrix");
  GPX_ARG(lr % GPX_TILE == 0 && lc % GPX_TILE == 0 && m % GPX_TILE == 0 && w % GPX_TILE == 0, "blocks must be 128-aligned");
  return 0;
}

int64_t gpx_dist2_diag_elems(int64_t nb) { return nb * nb + (nb / GPX_TILE) * GPX_TILE * GPX_TILE; }
// row stride (doubles) of the rows region of a packed piece (see gpx_g_ld); the host side checks its constant against it
int64_t gpx_dist2_row_stride(int64_t nb) { return gpx_g_ld(nb); }

// assemble the local part of K(X) + nugget on rank (pr, pc): A is the local (rows_local x cols_local) matrix
int gpx_dist2_kfill(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X, const double* nugget,
                    int64_t nugget_len, gpx_mat* A, int64_t nb, int Pr, int Pc, int pr, int pc) {
  GPX_ARG(ctx && X && A, "NULL argument");
  GPX_ARG(nb > 0 && nb % GPX_TILE == 0 && Pr >= 1 && Pc >= 1 && pr >= 0 && pr < Pr && pc >= 0 && pc < Pc, "bad grid");
  KParams kp;
  GPX_TRY(gpx_make_kparams(kind, d, hyp, nhyp, &kp));
  GPX_ARG(X->cols == d && X->pcols == d, "X must be an unpadded (N x d) point set");
  GPX_ARG(nugget_len == 0 || nugget_len == 1 || nugget_len == X->rows, "nugget_len must be 0, 1 or N");
  GPX_TRY(gpx_kparams_sets(ctx, &kp, X));
  const int64_t n = X->rows;
  double* d_nug = nullptr;
  int64_t nug_bytes = 0;
  const double nscal = nugget_len == 1 ? nugget[0] : 0.0;
  if (nugget_len > 1) {
    nug_bytes = nugget_len * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, nug_bytes, &p));
    d_nug = (double*)p;
    GPX_HIP(hipMemcpyAsync(d_nug, nugget, (size_t)nug_bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  int r = 0;
  if (A->rows > 0 && A->cols > 0)
    r = launch_kfill_cyclic(ctx, kp, X->p, n, d_nug, nugget_len, nscal, A->p, gpx_round_up(A->rows, GPX_TILE),
                            gpx_round_up(A->cols, GPX_TILE), A->ld, nb, Pr, pr, Pc, pc);
  if (d_nug) {
    (void)hipStreamSynchronize(ctx->stream);
    gpx_dev_release(ctx, d_nug, nug_bytes);
  }
  A->binv_ib = 0;  // block inverses (chol_potrs) belong to the previous contents
  A->factored = 0;
  return r;
}

// diagonal owner of step k: copy the w x w block at local (lr, lc) into the D region of G (offset doff), factor it there
// (leaf inverses behind it), keep the inverses in A->aux (by local row) and write L_kk back into A.  `base` = global index
// of the block's first row (pivot report), n_valid = number of real points.  Asynchronous on the selected stream.
// = gpx_dist2_diag_stage + gpx_dist2_diag_factor_staged + gpx_dist2_diag_store below, which the panel loop issues separately
// (round 4) so that only the factorisation itself sits between the arrival of the block row the diagonal waits for and the
// broadcast of the factor: the copy INTO the packed buffer runs before that arrival (the block's last update is then applied
// to the copy, gpx_dist2_diag_update), the copies OUT of it behind the event that releases the broadcast -- three launches less
// on the chain across ranks, whose per-step latency, summed, is the factorisation time of a multi-rank run.
static int d2_scratch_ensure(gpx_ctx* ctx, int64_t nb);
static int64_t d2_slices(int64_t m, int64_t n, int64_t k, int64_t nb);
static int diag_region_check(const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, const gpx_mat* G, int64_t doff, int64_t nb) {
  GPX_ARG(G, "NULL argument");
  GPX_TRY(check_local(A, lr, w, lc, w));
  GPX_ARG(w <= nb && nb % GPX_TILE == 0 && doff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes, "D region outside G");
  return 0;
}

int gpx_dist2_diag_stage(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff, int64_t nb) {
  GPX_ARG(ctx, "NULL argument");
  GPX_TRY(diag_region_check(A, lr, lc, w, G, doff, nb));
  return gpx_copy2d(ctx, A->p + lr * A->ld + lc, A->ld, G->p + doff, nb, w, w);
}

// staged block (h x h at doff of G, row stride nb) -= S[soff](h x w, packed row stride) * the same^T: the LAST update of a
// diagonal block (by the block row that came ahead of its panel), applied to the staged copy
int gpx_dist2_diag_update(gpx_ctx* ctx, gpx_mat* G, int64_t doff, int64_t h, const gpx_mat* S, int64_t soff, int64_t w, int64_t nb) {
  GPX_ARG(ctx && G && S, "NULL argument");
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(nb > 0 && nb % GPX_TILE == 0 && h > 0 && h <= nb && h % GPX_TILE == 0 && w > 0 && w <= nb && w % 16 == 0 && doff >= 0 &&
              soff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes && (soff + h * gld) * 8 <= S->bytes, "operand outside its buffer");
  const int64_t parts = d2_slices(h, h, w, nb);
  if (parts > 1) {
    GPX_TRY(d2_scratch_ensure(ctx, nb));
    return launch_gemm_ksplit_small(ctx, S->p + soff, gld, S->p + soff, gld, G->p + doff, nb, h, h, w, false, parts,
                                    ctx->d2_scratch + 2 * nb * nb);
  }
  return launch_gemm(ctx, S->p + soff, gld, S->p + soff, gld, G->p + doff, nb, h, h, w, true, true, false);
}

int gpx_dist2_diag_factor_staged(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                                 int64_t nb, int64_t base, int64_t n_valid) {
  GPX_ARG(ctx, "NULL argument");
  GPX_TRY(diag_region_check(A, lr, lc, w, G, doff, nb));
  if (!A->aux) {   // (the store below needs it; allocated here, ahead of the chain)
    A->aux_bytes = A->prows * GPX_TILE * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, A->aux_bytes, &p));
    A->aux = (double*)p;
  }
  if (A->dinv && A->dinv_nb == nb && (size_t)(lr / nb) < A->dinv_ok.size()) A->dinv_ok[(size_t)(lr / nb)] = 0;  // a new factor
  double* D = G->p + doff;
  return chol_potrf_nozero(ctx, D, nb, w, D + nb * nb, base, n_valid);
}

static int keep_inverse(gpx_ctx* ctx, gpx_mat* A, int64_t dslot, int64_t nb, const double* inv) {
  const int64_t slots = (A->prows + nb - 1) / nb;
  GPX_ARG(dslot < slots, "diagonal slot outside the local matrix");
  if (!A->dinv || A->dinv_nb != nb) {
    if (A->dinv) gpx_dev_release(ctx, A->dinv, A->dinv_bytes);
    A->dinv = nullptr;
    void* pd;
    GPX_TRY(gpx_dev_alloc(ctx, slots * 2 * nb * nb * 8, &pd));
    A->dinv = (double*)pd;
    A->dinv_bytes = slots * 2 * nb * nb * 8;
    A->dinv_nb = nb;
    A->dinv_ok.assign((size_t)slots, 0);
  }
  double* slot = A->dinv + dslot * 2 * nb * nb;   // [inverse | its transpose: the transposed sweep reads rows too]
  GPX_HIP(hipMemcpyAsync(slot, inv, (size_t)(nb * nb * 8), hipMemcpyDeviceToDevice, ctx->stream));
  GPX_TRY(chol_block_transpose(ctx, slot, slot + nb * nb, nb));
  A->dinv_ok[(size_t)dslot] = 1;
  return 0;
}

// the factored block and its leaf inverses from the packed buffer into the local matrix (what the distributed substitution
// reads); dslot >= 0: and the explicit inverse gpx_dist2_panel_inv built for THIS block (still in the context's scratch: same
// stream, before the next gpx_dist2_panel_inv) into the local matrix's slot for the substitution's diagonal solves
int gpx_dist2_diag_store(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, const gpx_mat* G, int64_t doff, int64_t nb,
                         int64_t dslot) {
  GPX_ARG(ctx, "NULL argument");
  GPX_TRY(diag_region_check(A, lr, lc, w, G, doff, nb));
  GPX_ARG(A->aux != nullptr, "diag_store: the block was not factored through gpx_dist2_diag_factor_staged");
  const double* D = G->p + doff;
  GPX_TRY(gpx_copy2d(ctx, D, nb, A->p + lr * A->ld + lc, A->ld, w, w));
  GPX_HIP(hipMemcpyAsync(A->aux + (lr / GPX_TILE) * GPX_TILE * GPX_TILE, D + nb * nb, (size_t)((w / GPX_TILE) * GPX_TILE * GPX_TILE * 8),
                         hipMemcpyDeviceToDevice, ctx->stream));
  if (dslot >= 0 && w == nb && w > GPX_TILE) {
    GPX_ARG(ctx->d2_scratch && ctx->d2_inv_src == D && ctx->d2_inv_nb == nb, "diag_store: no inverse was prepared for this diagonal block");
    GPX_TRY(keep_inverse(ctx, A, dslot, nb, ctx->d2_scratch));
  }
  return 0;
}

int gpx_dist2_diag_factor(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                          int64_t nb, int64_t base, int64_t n_valid) {
  GPX_TRY(gpx_dist2_diag_stage(ctx, A, lr, lc, w, G, doff, nb));
  GPX_TRY(gpx_dist2_diag_factor_staged(ctx, A, lr, lc, w, G, doff, nb, base, n_valid));
  return gpx_dist2_diag_store(ctx, A, lr, lc, w, G, doff, nb, -1);
}

// holders of block column k (ranks with pc == k % Pc): X = A[lr0 : lr0+m, lc : lc+w] <- X L_kk^-T (L_kk and its leaf
// inverses from the D region at doff), packed (row stride nb) into G at roff AND kept in A.  Asynchronous.
// Tall pieces (m >= GPX_DIST2_INV_MIN rows, default 2 nb) multiply with the EXPLICIT inverse of L_kk -- built here from the
// leaf inverses by four batched MFMA products (every holder builds its own copy: nothing more to broadcast, and it is off the
// diagonal chain, which solves its one early block through the leaves) -- as one triangular-operand GEMM straight into the
// packed buffer; the leaf-level recursion (4 strip kernels + 3 short-K products per 512 columns) cost 56 ms per C4
// factorisation on one rank against 190 ms for everything else (profiles/r03_dist_w1_before.txt).  The inverse lives in a
// context-owned scratch: calls must be issued in order on ONE stream (the PANEL stream of the panel loop).
int gpx_dist2_panel_trsm(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                         int64_t roff, int64_t nb) {
  return gpx_dist2_panel_trsm_keep(ctx, A, lr0, m, lc, w, G, doff, roff, nb, -1);
}

// [0, nb^2): the explicit inverse of the step's diagonal block; [nb^2, 2 nb^2): its build scratch; [2 nb^2, 6 nb^2): partial
// products of the chain's two small launches when they run as slices of their k range (d2_slices)
static int d2_scratch_ensure(gpx_ctx* ctx, int64_t nb) {
  const int64_t need = 6 * nb * nb * 8;
  if (ctx->d2_scratch_bytes < need) {
    GPX_HIP(hipDeviceSynchronize());
    if (ctx->d2_scratch) (void)hipFree(ctx->d2_scratch);
    ctx->d2_scratch = nullptr;
    ctx->d2_scratch_bytes = 0;
    ctx->d2_inv_src = nullptr;
    GPX_HIP(hipMalloc((void**)&ctx->d2_scratch, (size_t)need));
    ctx->d2_scratch_bytes = need;
  }
  return 0;
}

// Size the context's two distributed-loop scratches UP FRONT: the panel solve's inverse (2 nb^2 doubles) and the
// streamed evaluation's group scratch (2 ceil(w / nb) nb^2 + w mcols doubles, w = agg nb).  Both used to grow on first use from
// inside the enqueue path -- hipDeviceSynchronize + hipFree + hipMalloc in the middle of a step that may have collectives in
// flight (bench.py's preflight runs nb = 256 and the real runner 512), or inside a stream capture.  The runner's constructor
// calls this; the hot path then finds them large enough.  Blocking.
// The two products that sit on the chain ACROSS ranks with one block of rows -- the solve of block row k+1 against the
// prepared inverse and the last update of the staged diagonal block: nb x nb x nb, 256 64-tiles under a serial k range of nb,
// 55-60 us each -- as slices of the k range (launch_gemm_ksplit_small, as in gpx_refit_rows: 23 + 6 us).
static int64_t d2_slices(int64_t m, int64_t n, int64_t k, int64_t nb) {
  if (m % 64 != 0 || n % 64 != 0 || m > nb || n > nb) return 1;
  int64_t parts = 1;
  while (parts < 4 && (m / 64) * (n / 64) * parts < 1024 && k % (2 * parts * 16) == 0 && k / (2 * parts) >= 256) parts *= 2;
  return parts;
}

int gpx_dist2_reserve(gpx_ctx* ctx, int64_t nb, int64_t agg, int64_t mcols) {
  GPX_ARG(ctx && nb > 0 && nb % GPX_TILE == 0 && agg >= 1 && mcols >= 0, "bad arguments");
  GPX_TRY(d2_scratch_ensure(ctx, nb));
  const int64_t w = agg * nb, mcp = gpx_round_up(mcols > 0 ? mcols : 1, GPX_TILE);
  const int64_t need = (2 * agg * nb * nb + w * mcp) * 8;
  if (mcols > 0 && ctx->ev_scratch_bytes < need) {
    GPX_HIP(hipDeviceSynchronize());
    if (ctx->ev_scratch) (void)hipFree(ctx->ev_scratch);
    ctx->ev_scratch = nullptr;
    ctx->ev_scratch_bytes = 0;
    GPX_HIP(hipMalloc((void**)&ctx->ev_scratch, (size_t)need));
    ctx->ev_scratch_bytes = need;
  }
  return 0;
}

// Round 4: the explicit inverse of the diagonal block of panel k, built AHEAD of the panel solve.  It depends on the factored
// block alone (in G at doff: available when the column broadcast lands, typically a millisecond before the column has its last
// update), but gpx_dist2_panel_trsm used to build it inside the solve: eight small dependent kernels on the panel chain --
// whose per-step latency, summed over the steps, IS the factorisation time of a multi-rank run (scripts/dist_replay.py
// --paced-grid) -- and the block row the next diagonal needs went through the leaf recursion, eight more.  The panel loop now
// calls this right behind the broadcast of the diagonal block, before it waits for the column; both solves of the step are then
// one triangular-operand product each.  PANEL stream, step order: the scratch holds one inverse at a time.
int gpx_dist2_panel_inv(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t nb, int64_t w) {
  GPX_ARG(ctx && G && nb > 0 && nb % GPX_TILE == 0 && doff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes, "bad arguments");
  if (w != nb || w <= GPX_TILE) return 0;   // ragged last block / single leaf: the solve walks the leaf recursion
  GPX_TRY(d2_scratch_ensure(ctx, nb));
  const double* D = G->p + doff;
  double* inv = ctx->d2_scratch;
  ctx->d2_inv_src = nullptr;
  GPX_TRY(chol_block_inverse(ctx, D, nb, D + nb * nb, inv, w, inv + nb * nb));
  ctx->d2_inv_src = D;
  ctx->d2_inv_nb = nb;
  return 0;
}

// ... and, on the OWNER of the diagonal block (dslot = its local block row, -1 elsewhere), the explicit inverse is kept in the
// local matrix for the distributed substitution: its diagonal solves then are one small GEMV instead of a 512-row sweep
// through one workgroup (45 us), 2 N / nb of them in a chain.
static int panel_trsm_impl(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                           int64_t roff, int64_t nb, int64_t dslot, int use_prepared);

int gpx_dist2_panel_trsm_keep(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                              int64_t roff, int64_t nb, int64_t dslot) {
  return panel_trsm_impl(ctx, A, lr0, m, lc, w, G, doff, roff, nb, dslot, 0);
}

// the same with the inverse gpx_dist2_panel_inv built for THIS step's diagonal block (the caller vouches for that: the panel
// loop calls the two back to back on the PANEL stream); falls back to the leaf recursion for a ragged / single-leaf block
int gpx_dist2_panel_trsm_inv(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                             int64_t roff, int64_t nb, int64_t dslot, int copy_back) {
  return panel_trsm_impl(ctx, A, lr0, m, lc, w, G, doff, roff, nb, dslot, copy_back ? 1 : 2);
}

// the solved rows from the packed buffer back into the local matrix (the block-cyclic factor the substitution sweeps read):
// the second half of gpx_dist2_panel_trsm_inv(copy_back = 0), issued BEHIND the event that releases the panel broadcast -- the
// broadcast reads the packed buffer only, the copy has no business on the chain across ranks
int gpx_dist2_panel_copyback(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, const gpx_mat* G, int64_t roff,
                             int64_t nb) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc, w));
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(w <= nb && roff >= 0 && (roff + m * gld) * 8 <= G->bytes, "region outside G");
  if (m == 0) return 0;
  return gpx_copy2d(ctx, G->p + roff, gld, A->p + lr0 * A->ld + lc, A->ld, m, w);
}

// Round 5, the factor RE-STREAMED (gpexp_amd/dist.py dist2_restream_enqueue: evaluation against a rank that kept no replica of
// the factor -- the block-cyclic local matrix is all there is): the inverse of gpx_dist2_panel_copyback / gpx_dist2_diag_store.
// Rows [lr0, lr0 + m) of local block column lc (a piece of finished panel k) from the local matrix into the packed buffer ...
int gpx_dist2_panel_pack(gpx_ctx* ctx, const gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t roff,
                         int64_t nb) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc, w));
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(w <= nb && roff >= 0 && (roff + m * gld) * 8 <= G->bytes, "region outside G");
  if (m == 0) return 0;
  return gpx_copy2d(ctx, A->p + lr0 * A->ld + lc, A->ld, G->p + roff, gld, m, w);
}

// ... and the factored diagonal block with its leaf inverses into the D region (owner of the block)
int gpx_dist2_diag_pack(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff, int64_t nb) {
  GPX_ARG(ctx, "NULL argument");
  GPX_TRY(diag_region_check(A, lr, lc, w, G, doff, nb));
  GPX_ARG(A->aux != nullptr, "diag_pack: the local matrix holds no factored diagonal block");
  double* D = G->p + doff;
  GPX_TRY(gpx_copy2d(ctx, A->p + lr * A->ld + lc, A->ld, D, nb, w, w));
  GPX_HIP(hipMemcpyAsync(D + nb * nb, A->aux + (lr / GPX_TILE) * GPX_TILE * GPX_TILE, (size_t)((w / GPX_TILE) * GPX_TILE * GPX_TILE * 8),
                         hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

static int panel_trsm_impl(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                           int64_t roff, int64_t nb, int64_t dslot, int use_prepared) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc, w));
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(w <= nb && doff >= 0 && roff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes &&
              (roff + m * gld) * 8 <= G->bytes, "region outside G");
  if (m == 0) return 0;
  const double* D = G->p + doff;
  double* X = A->p + lr0 * A->ld + lc;
  const int64_t inv_min = 2;   // rows from which the explicit inverse is used, in units of nb
  const bool prepared = use_prepared != 0 && w == nb && w > GPX_TILE;   // use_prepared == 2: the caller copies back later
  GPX_ARG(!prepared || (ctx->d2_inv_src == D && ctx->d2_inv_nb == nb), "panel solve: no inverse was prepared for this diagonal block");
  if (prepared || (inv_min > 0 && m >= inv_min * nb && w == nb && w > GPX_TILE)) {
    GPX_TRY(d2_scratch_ensure(ctx, nb));
    double* inv = ctx->d2_scratch;
    if (!prepared) {
      ctx->d2_inv_src = nullptr;
      GPX_TRY(chol_block_inverse(ctx, D, nb, D + nb * nb, inv, w, inv + nb * nb));
      ctx->d2_inv_src = D;
      ctx->d2_inv_nb = nb;
    }
    // (use_prepared == 2: gpx_dist2_diag_store keeps the inverse, behind the broadcast's event -- not on the chain)
    if (dslot >= 0 && use_prepared != 2) GPX_TRY(keep_inverse(ctx, A, dslot, nb, inv));
    const int64_t parts = prepared ? d2_slices(m, w, w, nb) : 1;   // (one block of rows: dense against the inverse's zero upper part)
    if (parts > 1)
      GPX_TRY(launch_gemm_ksplit_small(ctx, X, A->ld, inv, w, G->p + roff, gld, m, w, w, true, parts, inv + 2 * nb * nb));
    else
      GPX_TRY(launch_gemm_tri(ctx, X, A->ld, inv, w, G->p + roff, gld, m, w, w, true, false, false, 2));
    if (prepared && use_prepared == 2) return 0;     // gpx_dist2_panel_copyback follows behind the broadcast's event
    return gpx_copy2d(ctx, G->p + roff, gld, X, A->ld, m, w);
  }
  GPX_TRY(chol_trsm_right(ctx, D, nb, D + nb * nb, X, A->ld, m, w));
  return gpx_copy2d(ctx, X, A->ld, G->p + roff, gld, m, w);
}

// A[lr0 : lr0+m, lc0 : lc0+n] -= G[aoff](m x w, stride nb) * G[boff](n x w, stride nb)^T   (trailing update of one local
// block column by panel k).  Asynchronous.
int gpx_dist2_update(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc0, int64_t n, const gpx_mat* G,
                     int64_t aoff, int64_t boff, int64_t w, int64_t nb) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc0, n));
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(w > 0 && w <= nb && w % 16 == 0 && aoff >= 0 && boff >= 0 && (aoff + m * gld) * 8 <= G->bytes &&
              (boff + n * gld) * 8 <= G->bytes, "operand outside G");
  if (m == 0 || n == 0) return 0;
  return launch_gemm(ctx, G->p + aoff, gld, G->p + boff, gld, A->p + lr0 * A->ld + lc0, A->ld, m, n, w, true, true, false);
}

// A[lr0 : lr0+m, lc0 : lc0+n] -= the contributions of nseg panels at once, for the local blocks on / below (below_diag != 0:
// strictly below) the global diagonal: panel ks[s] sits in the packed buffer G[s] (piece layout above; piece_stride doubles per
// piece).  One launch over the whole local trailing matrix with K = nseg * nb -- see dist2_update_kernel (gemm_f64.hip).
int gpx_dist2_update_multi(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc0, int64_t n, int64_t nb, int Pr, int Pc,
                           int pr, int pc, int64_t piece_stride, int nseg, const gpx_mat* const* G, const int64_t* ks,
                           int below_diag) {
  GPX_ARG(ctx && G && ks, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc0, n));
  GPX_ARG(nb > 0 && nb % GPX_TILE == 0 && Pr >= 1 && Pr <= GPX_MAX_PR && Pc >= 1 && pr >= 0 && pr < Pr && pc >= 0 && pc < Pc &&
              nseg >= 0 && nseg <= GPX_SEG_MAX, "bad grid / segment count");
  GPX_ARG(lr0 % nb == 0 && lc0 % nb == 0, "the updated block must start on a block boundary");
  if (m == 0 || n == 0 || nseg == 0) return 0;
  const int64_t dsz = gpx_dist2_diag_elems(nb);
  GPX_ARG(piece_stride >= dsz, "piece stride smaller than a diagonal region");
  const double* g[GPX_SEG_MAX];
  const int64_t li_first = lr0 / nb, li_last = (lr0 + m - 1) / nb, lj_first = lc0 / nb, lj_last = (lc0 + n - 1) / nb;
  for (int s = 0; s < nseg; ++s) {
    GPX_ARG(G[s] != nullptr && ks[s] >= 0, "NULL panel buffer / negative panel index");
    GPX_ARG(Pr * piece_stride * 8 <= G[s]->bytes, "panel buffer smaller than Pr pieces");
    g[s] = G[s]->p;
    // every operand row must lie inside its piece: rows of local block li are block (li - li0(pr, k)) of piece pr, rows of
    // global block J are block (J / Pr - li0(J % Pr, k)) of piece J % Pr, li0(p, k) = #{I' <= k : I' % Pr == p}
    auto li0 = [&](int64_t p) { return ks[s] + 1 <= p ? (int64_t)0 : (ks[s] - p) / Pr + 1; };
    GPX_ARG(li_first * Pr + pr > ks[s] && lj_first * Pc + pc > ks[s], "update touches blocks that are not behind the panel");
    GPX_ARG(dsz + ((li_last - li0(pr)) * nb + (lr0 + m - li_last * nb)) * gpx_g_ld(nb) <= piece_stride, "A rows beyond the piece");
    for (int64_t lj = lj_first; lj <= lj_last; ++lj) {
      const int64_t J = lj * Pc + pc, p = J % Pr;
      const int64_t cw = (lc0 + n - lj * nb) < nb ? (lc0 + n - lj * nb) : nb;
      GPX_ARG(J / Pr >= li0(p) && dsz + ((J / Pr - li0(p)) * nb + cw) * gpx_g_ld(nb) <= piece_stride, "B rows beyond the piece");
    }
  }
  return launch_dist2_update(ctx, A->p, A->ld, lr0, m, lc0, n, nb, Pr, Pc, pr, pc, piece_stride, dsz, nseg, g, ks, below_diag);
}

// single-rank REPLAY of the distributed loop (gpexp_amd/dist.py ReplayComm): what a collective would have delivered is
// copied -- same byte count, device to device -- out of a complete factor L that is resident on this GPU.
// rows: the inverse of gpx_dist2_unpack_rows; diag: the inverse of gpx_dist2_unpack_diag (block + its leaf inverses)
int gpx_dist2_pack_rows(gpx_ctx* ctx, const gpx_mat* L, int64_t first_block, int64_t stride, int64_t col0, gpx_mat* G,
                        int64_t roff, int64_t m, int64_t w, int64_t nb) {
  GPX_ARG(ctx && G && L, "NULL argument");
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(roff >= 0 && m >= 0 && (roff + m * gld) * 8 <= G->bytes && w % 2 == 0 && col0 + w <= L->pcols, "bad piece");
  if (m == 0) return 0;
  const int64_t last = ((m - 1) / nb * stride + first_block) * nb + (m - 1) % nb;
  GPX_ARG(last < L->prows, "piece rows fall outside the factor");
  for (int64_t r0 = 0; r0 < m; r0 += 65535 / nb * nb) {
    int64_t rr = m - r0;
    if (rr > 65535 / nb * nb) rr = 65535 / nb * nb;
    dim3 grid((unsigned)((w / 2 + 255) / 256), (unsigned)((rr + COPY_RPB - 1) / COPY_RPB));
    hipLaunchKernelGGL(gather_cyclic_rows_kernel, grid, dim3(256), 0, ctx->stream, L->p + col0, L->ld, G->p + roff + r0 * gld, gld,
                       rr, w, nb, first_block + (r0 / nb) * stride, stride);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_dist2_pack_diag(gpx_ctx* ctx, const gpx_mat* L, int64_t r0, int64_t w, int64_t nb, gpx_mat* G, int64_t doff) {
  GPX_ARG(ctx && G && L && L->aux, "NULL argument / factor without leaf inverses");
  GPX_ARG(doff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes && r0 + w <= L->prows && w <= nb, "bad D region");
  GPX_TRY(gpx_copy2d(ctx, L->p + r0 * L->ld + r0, L->ld, G->p + doff, nb, w, w));
  GPX_HIP(hipMemcpyAsync(G->p + doff + nb * nb, L->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE,
                         (size_t)((w / GPX_TILE) * GPX_TILE * GPX_TILE * 8), hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

// replicated factor: the m x w rows of a piece (blocks first, first+stride, ... in ascending order) -> rows of block
// column `col0 / nb` of the full-size matrix L
int gpx_dist2_unpack_rows(gpx_ctx* ctx, const gpx_mat* G, int64_t roff, int64_t m, int64_t w, int64_t nb, gpx_mat* L,
                          int64_t first_block, int64_t stride, int64_t col0) {
  GPX_ARG(ctx && G && L, "NULL argument");
  const int64_t gld = gpx_g_ld(nb);
  GPX_ARG(roff >= 0 && m >= 0 && (roff + m * gld) * 8 <= G->bytes && w % 2 == 0 && col0 + w <= L->pcols, "bad piece");
  if (m == 0) return 0;
  const int64_t last = ((m - 1) / nb * stride + first_block) * nb + (m - 1) % nb;
  GPX_ARG(last < L->prows, "piece rows fall outside the replicated factor");
  for (int64_t r0 = 0; r0 < m; r0 += 65535 / nb * nb) {
    int64_t rr = m - r0;
    if (rr > 65535 / nb * nb) rr = 65535 / nb * nb;
    dim3 grid((unsigned)((w / 2 + 255) / 256), (unsigned)((rr + COPY_RPB - 1) / COPY_RPB));
    hipLaunchKernelGGL(copy_cyclic_rows_kernel, grid, dim3(256), 0, ctx->stream, G->p + roff + r0 * gld, gld,
                       L->p + col0, L->ld, rr, w, nb, first_block + (r0 / nb) * stride, stride);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

// replicated factor: diagonal block k (global offset r0 = k * nb) and its leaf inverses from the D region
int gpx_dist2_unpack_diag(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t w, int64_t nb, gpx_mat* L, int64_t r0) {
  return gpx_dist2_unpack_diag_at(ctx, G, doff, w, nb, L, r0, r0);
}

// ... to rows r0.., columns c0.. of L (c0 != r0: L is a window of block columns, see gpx_dist_ivar_group_at)
int gpx_dist2_unpack_diag_at(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t w, int64_t nb, gpx_mat* L, int64_t r0,
                             int64_t c0) {
  GPX_ARG(ctx && G && L, "NULL argument");
  GPX_ARG(doff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes && r0 + w <= L->prows && w <= nb && c0 >= 0 &&
              c0 + w <= L->pcols, "bad D region");
  if (!L->aux) {
    L->aux_bytes = L->prows * GPX_TILE * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, L->aux_bytes, &p));
    L->aux = (double*)p;
  }
  GPX_TRY(gpx_copy2d(ctx, G->p + doff, nb, L->p + r0 * L->ld + c0, L->ld, w, w));
  GPX_HIP(hipMemcpyAsync(L->aux + (r0 / GPX_TILE) * GPX_TILE * GPX_TILE, G->p + doff + nb * nb,
                         (size_t)((w / GPX_TILE) * GPX_TILE * GPX_TILE * 8), hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

// ---- distributed solves on the block-cyclic factor (north_star: "RCCL broadcast/reduce") -----------------------------
// v[voff : voff+w] <- L_kk^-1 v (transposed: L_kk^-T v) with the diagonal block at local (lr, lc) and its leaf inverses
int gpx_dist2_trsv_diag(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* v, int64_t voff,
                        int transposed) {
  GPX_ARG(ctx && v, "NULL argument");
  GPX_TRY(check_local(A, lr, w, lc, w));
  GPX_ARG(A->aux != nullptr, "local matrix holds no factored diagonal block");
  GPX_ARG(voff >= 0 && voff % 2 == 0 && (voff + w) * 8 <= v->bytes, "vector segment out of range");
  const int64_t need = chol_trsv_scratch_bytes(w);
  if (ctx->trsv_scratch_bytes < need) {
    GPX_HIP(hipDeviceSynchronize());
    if (ctx->trsv_scratch) (void)hipFree(ctx->trsv_scratch);
    ctx->trsv_scratch = nullptr;
    ctx->trsv_scratch_bytes = 0;
    GPX_HIP(hipMalloc((void**)&ctx->trsv_scratch, (size_t)need));
    ctx->trsv_scratch_bytes = need;
  }
  if (A->dinv && w == A->dinv_nb && lr % w == 0 && (size_t)(lr / w) < A->dinv_ok.size() && A->dinv_ok[(size_t)(lr / w)]) {
    // the block's explicit inverse is at hand (kept by the panel solve): v <- Dinv v or Dinv^T v as a row-per-wave GEMV
    // (the transposed form reads the stored transpose), through the scratch vector
    v->bbox_ok = 0;
    const double* slot = A->dinv + (lr / w) * 2 * w * w;
    GPX_TRY(chol_tri_gemv(ctx, transposed ? slot + w * w : slot, w, w, v->p + voff, ctx->trsv_scratch, transposed ? 0 : 1));
    GPX_HIP(hipMemcpyAsync(v->p + voff, ctx->trsv_scratch, (size_t)(w * 8), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  return chol_trsv_with_scratch(ctx, A->p + lr * A->ld + lc, A->ld, A->aux + (lr / GPX_TILE) * GPX_TILE * GPX_TILE,
                                v->p + voff, w, transposed != 0, ctx->trsv_scratch);
}

// acc[aoff : aoff+m] -= A[lr0 : lr0+m, lc : lc+w] x[xoff : xoff+w]          (transposed == 0)
// acc[aoff : aoff+w] -= A[lr0 : lr0+m, lc : lc+w]^T x[xoff : xoff+m]        (transposed != 0; deterministic reduction)
int gpx_dist2_gemv(gpx_ctx* ctx, const gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, const gpx_mat* x,
                   int64_t xoff, gpx_mat* acc, int64_t aoff, int transposed) {
  GPX_ARG(ctx && x && acc, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc, w));
  if (m == 0 || w == 0) return 0;
  const double* Ab = A->p + lr0 * A->ld + lc;
  if (!transposed) {
    GPX_ARG((xoff + w) * 8 <= x->bytes && (aoff + m) * 8 <= acc->bytes && xoff % 2 == 0, "vector segment out of range");
    return launch_gemv_sub(ctx, Ab, A->ld, m, w, x->p + xoff, acc->p + aoff);
  }
  GPX_ARG((xoff + m) * 8 <= x->bytes && (aoff + w) * 8 <= acc->bytes && aoff % 2 == 0 && A->ld % 2 == 0,
          "vector segment out of range / misaligned");
  ProfScope ps(ctx, GPX_PROF_TRSV, 2.0 * (double)m * w, 8.0 * (double)m * w);
  hipLaunchKernelGGL(gemv_t_sub_kernel, dim3((unsigned)((w + 127) / 128)), dim3(256), 0, ctx->stream, Ab, A->ld, m, w,
                     x->p + xoff, acc->p + aoff);
  GPX_HIP(hipGetLastError());
  return 0;
}

// acc += 2 sum log diag of the w x w block at local (lr, lc); acc is a device scalar (>= 1 double), zero it first
int gpx_dist2_logdet_acc(gpx_ctx* ctx, const gpx_mat* A, int64_t lr, int64_t lc, int64_t w, int64_t n_valid, gpx_mat* acc) {
  GPX_ARG(ctx && acc, "NULL argument");
  GPX_TRY(check_local(A, lr, w, lc, w));
  hipLaunchKernelGGL(logdet_acc_kernel, dim3(1), dim3(256), 0, ctx->stream, A->p + lr * A->ld + lc, A->ld, w, n_valid,
                     acc->p);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_dist_finish(gpx_ctx* ctx, gpx_mat* K) {
  GPX_ARG(ctx && K && K->aux, "matrix was not factored by the distributed panel loop");
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  K->binv_ib = 0;  // block inverses (chol_potrs) belong to the previous contents
  K->factored = 1;
  return 0;
}

}  // extern "C"

// =====================================================================================================================
// Recorded programs.  The panel loop of the 2-D factorisation is Python on purpose (gpexp_amd/dist.py: the same loop drives
// the NumPy device double of the CPU tests), but its ~60 C-ABI calls per step cost more host time than an 8-rank step leaves
// (0.45 ms of GPU work per step at C4).  The loop is therefore RUN ONCE against a recorder, which turns every primitive and
// collective into a row of 16 int64 -- [opcode, handle0..2, a0..a11] -- and the rows are replayed here, natively, on every
// step: no Python, no ctypes marshalling, identical call sequence.  Variable-length arguments (piece lists, segment lists) live
// in the `extra` pool and are referenced by offset.  The direct (unrecorded) mode of the Python side executes the very same
// rows one at a time, so there is one dispatch table, not two.
// =====================================================================================================================
#include <chrono>


extern "C" {

int gpx_program_run(gpx_ctx* ctx, const int64_t* ops, int64_t nops, const int64_t* extra, int64_t nextra, double* host_ms) {
  GPX_ARG(ctx && (ops || nops == 0) && nops >= 0 && nextra >= 0, "bad program");
  const auto t0 = std::chrono::steady_clock::now();
  for (int64_t i = 0; i < nops; ++i) {
    const int64_t* o = ops + 16 * i;
    gpx_mat* h0 = reinterpret_cast<gpx_mat*>((uintptr_t)o[1]);
    gpx_mat* h1 = reinterpret_cast<gpx_mat*>((uintptr_t)o[2]);
    gpx_mat* h2 = reinterpret_cast<gpx_mat*>((uintptr_t)o[3]);
    // the rows carry raw handles recorded earlier: a matrix freed since then (or a context closed under a kept Program) must be
    // an error here, not a use-after-free in a kernel
    if ((h0 && !ctx->live_mats.count(h0)) || (h1 && !ctx->live_mats.count(h1)) || (h2 && !ctx->live_mats.count(h2))) {
      gpx_set_error("program op %lld (opcode %lld): a matrix handle of the recorded row is no longer alive", (long long)i, (long long)o[0]);
      return -1;
    }
    const int64_t* a = o + 4;
    int r = 0;
    switch ((int)o[0]) {
      case GPX_OP_STREAM: r = gpx_stream_select(ctx, (int)a[0]); break;
      case GPX_OP_RECORD: r = gpx_event_record(ctx, (int)a[0]); break;
      case GPX_OP_WAIT: r = gpx_event_wait(ctx, (int)a[0]); break;
      case GPX_OP_BEGIN: r = gpx_dist_begin(ctx); break;
      case GPX_OP_DIAG_FACTOR: r = gpx_dist2_diag_factor(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4], a[5], a[6]); break;
      case GPX_OP_PANEL_TRSM:
        r = a[8] ? gpx_dist2_panel_trsm_inv(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], a[5], a[6], a[7] - 1, a[8] == 1)
                 : gpx_dist2_panel_trsm_keep(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], a[5], a[6], a[7] - 1);
        break;
      case GPX_OP_PANEL_COPYBACK: r = gpx_dist2_panel_copyback(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], a[5]); break;
      case GPX_OP_DIAG_STAGE: r = gpx_dist2_diag_stage(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4]); break;
      case GPX_OP_DIAG_UPDATE: r = gpx_dist2_diag_update(ctx, h0, a[0], a[1], h1, a[2], a[3], a[4]); break;
      case GPX_OP_DIAG_FACTOR_STAGED: r = gpx_dist2_diag_factor_staged(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4], a[5], a[6]); break;
      case GPX_OP_DIAG_STORE: r = gpx_dist2_diag_store(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4], a[5] - 1); break;
      case GPX_OP_PANEL_PACK: r = gpx_dist2_panel_pack(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], a[5]); break;
      case GPX_OP_DIAG_PACK: r = gpx_dist2_diag_pack(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4]); break;
      case GPX_OP_UPDATE: r = gpx_dist2_update(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], a[5], a[6], a[7]); break;
      case GPX_OP_UPDATE_MULTI: {
        // a: lr0, m, lc0, n, nb, Pr, Pc, pr, pc, piece_stride, nseg | below_diag << 8, extra offset of [G handles..., ks...]
        const int nseg = (int)(a[10] & 0xff), below = (int)(a[10] >> 8);
        if (nseg < 0 || nseg > GPX_SEG_MAX || a[11] < 0 || a[11] + 2 * nseg > nextra) {
          gpx_set_error("program op %lld: segment list outside the extra pool", (long long)i);
          return -1;
        }
        const gpx_mat* G[GPX_SEG_MAX];
        for (int s = 0; s < nseg; ++s) {
          G[s] = reinterpret_cast<const gpx_mat*>((uintptr_t)extra[a[11] + s]);
          if (!ctx->live_mats.count(G[s])) {
            gpx_set_error("program op %lld: a packed panel buffer of the segment list is no longer alive", (long long)i);
            return -1;
          }
        }
        r = gpx_dist2_update_multi(ctx, h0, a[0], a[1], a[2], a[3], a[4], (int)a[5], (int)a[6], (int)a[7], (int)a[8], a[9], nseg,
                                   G, extra + a[11] + nseg, below);
        break;
      }
      case GPX_OP_UNPACK_ROWS: r = gpx_dist2_unpack_rows(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], a[5], a[6]); break;
      case GPX_OP_UNPACK_DIAG: r = gpx_dist2_unpack_diag_at(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4] ? a[4] - 1 : a[3]); break;
      case GPX_OP_PACK_ROWS: r = gpx_dist2_pack_rows(ctx, h0, a[0], a[1], a[2], h1, a[3], a[4], a[5], a[6]); break;
      case GPX_OP_PACK_DIAG: r = gpx_dist2_pack_diag(ctx, h0, a[0], a[1], a[2], h1, a[3]); break;
      case GPX_OP_BCAST_GRP: r = gpx_comm_bcast_grp(ctx, h0, a[0], a[1], (int)a[2], (int)a[3]); break;
      case GPX_OP_BCAST_GRP2: r = gpx_comm_bcast_grp2(ctx, h0, a[0], h1, a[1], a[2], (int)a[3], (int)a[4]); break;
      case GPX_OP_REDUCE_GRP: r = gpx_comm_reduce_grp(ctx, h0, a[0], a[1], (int)a[2], (int)a[3]); break;
      case GPX_OP_ALLREDUCE: r = gpx_comm_allreduce(ctx, h0, a[0], a[1]); break;
      case GPX_OP_PANEL_BCAST: {
        // a: npieces, extra offset of [offsets..., counts..., roots...]
        const int64_t np_ = a[0];
        if (np_ < 0 || np_ > 64 || a[1] < 0 || a[1] + 3 * np_ > nextra) {
          gpx_set_error("program op %lld: piece list outside the extra pool", (long long)i);
          return -1;
        }
        int roots[64];
        for (int64_t q = 0; q < np_; ++q) roots[q] = (int)extra[a[1] + 2 * np_ + q];
        r = gpx_comm_panel_bcast(ctx, h0, extra + a[1], extra + a[1] + np_, roots, (int)np_);
        break;
      }
      case GPX_OP_IVAR_STEP: r = gpx_dist_ivar_step(ctx, h0, a[0], a[1], h1); break;
      case GPX_OP_IVAR_GROUP: r = gpx_dist_ivar_group_at(ctx, h0, a[0], a[1], a[2], h1, a[3] ? a[3] - 1 : a[0] * a[2]); break;
      case GPX_OP_PANEL_INV: r = gpx_dist2_panel_inv(ctx, h0, a[0], a[1], a[2]); break;
      case GPX_OP_FWD_GROUP: r = gpx_dist_fwd_group_at(ctx, h0, a[0], a[1], a[2], h1, a[3] ? a[3] - 1 : a[0] * a[2]); break;
      case GPX_OP_TRSV_DIAG: r = gpx_dist2_trsv_diag(ctx, h0, a[0], a[1], a[2], h1, a[3], (int)a[4]); break;
      case GPX_OP_GEMV: r = gpx_dist2_gemv(ctx, h0, a[0], a[1], a[2], a[3], h1, a[4], h2, a[5], (int)a[6]); break;
      case GPX_OP_LOGDET_ACC: r = gpx_dist2_logdet_acc(ctx, h0, a[0], a[1], a[2], a[3], h1); break;
      case GPX_OP_VEC_OP: r = gpx_vec_op(ctx, h0, a[0], h1, a[1], a[2], (int)a[3]); break;
      case GPX_OP_SPIN:   // a1 = 0: a0 ms; 1: a0 us; 2: stamp slot a0; 3: spin until a2 us after the stamp in slot a0
        r = a[1] == 3 ? gpx_dbg_spin_until(ctx, (int)a[0], a[2])
            : a[1] == 2 ? gpx_dbg_stamp(ctx, (int)a[0])
            : a[1] ? gpx_dbg_spin_us(ctx, a[0]) : gpx_dbg_spin(ctx, (int)a[0]);
        break;
      case GPX_OP_COPY: {  // h0[a0 : a0+a2] <- h1[a1 : a1+a2], device to device (replay stand-in for a point-to-point transfer)
        if (!h0 || !h1 || a[0] < 0 || a[1] < 0 || a[2] < 0 || (a[0] + a[2]) * 8 > h0->bytes || (a[1] + a[2]) * 8 > h1->bytes) {
          gpx_set_error("program op %lld: copy outside its buffers", (long long)i);
          return -1;
        }
        if (a[2] > 0 && hipMemcpyAsync(h0->p + a[0], h1->p + a[1], (size_t)a[2] * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
          gpx_set_error("program op %lld: hipMemcpyAsync failed", (long long)i);
          return -2;
        }
        break;
      }
      default:
        gpx_set_error("program op %lld: unknown opcode %lld", (long long)i, (long long)o[0]);
        return -1;
    }
    if (r != 0) return r;
  }
  if (host_ms) *host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}


// ---- the recorded program as a hipGraph ------------------------------------------------------------------------------------------
// gpx_program_run still pays one HIP launch per kernel (~3 us per row: 6-10 ms per C4 factorisation).  A program whose streams
// all fork from and join back into the main stream (the panel loop emits the fork / join rows itself) can instead be CAPTURED
// once -- the rows are issued under hipStreamBeginCapture, every kernel, copy and event edge becomes a graph node -- and the
// whole step is then ONE hipGraphLaunch.  Nothing that allocates or synchronises may run during capture: capture a program only
// after it has run once in the ordinary way (scratch buffers, leaf-inverse storage and kernel attributes exist by then).
// RCCL collectives inside a capture are NOT validated on this project's hardware (RCCL never ran with two ranks here): the
// Python side captures only when no rank of a real communicator would have to (world 1 / replay), unless told otherwise.
struct gpx_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  int64_t nodes;
};

int gpx_graph_free(gpx_ctx* ctx, gpx_graph* g) {
  if (!g) return 0;
  (void)ctx;
  (void)hipDeviceSynchronize();
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

int gpx_program_capture(gpx_ctx* ctx, const int64_t* ops, int64_t nops, const int64_t* extra, int64_t nextra, gpx_graph** out) {
  GPX_ARG(ctx && out && (ops || nops == 0), "bad program");
  GPX_ARG(ctx->prof_on == 0, "profiling events cannot be captured: disable the profile first");
  GPX_HIP(hipDeviceSynchronize());
  hipStream_t origin = ctx->streams[0];
  ctx->stream = origin;
  GPX_HIP(hipStreamBeginCapture(origin, hipStreamCaptureModeRelaxed));
  int r = gpx_program_run(ctx, ops, nops, extra, nextra, nullptr);
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(origin, &graph);
  ctx->stream = origin;
  if (r != 0 || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    if (r == 0) gpx_set_error("program capture: hipStreamEndCapture -> %s (do all streams join the main stream?)", hipGetErrorString(e));
    return r != 0 ? r : -2;
  }
  gpx_graph* g = new gpx_graph();
  g->graph = graph;
  g->exec = nullptr;
  size_t n = 0;
  (void)hipGraphGetNodes(graph, nullptr, &n);
  g->nodes = (int64_t)n;
  e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    gpx_set_error("program capture: hipGraphInstantiate -> %s", hipGetErrorString(e));
    gpx_graph_free(ctx, g);
    return -2;
  }
  *out = g;
  return 0;
}

// one step = one launch; *host_ms (nullable) = host time of the launch call; *nodes (nullable) = nodes of the graph
int gpx_graph_launch(gpx_ctx* ctx, gpx_graph* g, double* host_ms, int64_t* nodes) {
  GPX_ARG(ctx && g && g->exec, "no graph");
  const auto t0 = std::chrono::steady_clock::now();
  GPX_HIP(hipGraphLaunch(g->exec, ctx->streams[0]));
  if (host_ms) *host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (nodes) *nodes = g->nodes;
  ctx->stream = ctx->streams[0];
  return 0;
}

}  // extern "C"
