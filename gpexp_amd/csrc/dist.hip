// Multi-GPU pieces: RCCL communicator (dlopen'ed, so single-GPU use never touches librccl) and the panel
// primitives of the 1-D block-cyclic distributed Cholesky.  One process per GPU; the Python side
// (gpexp_amd/dist.py) drives the panel loop and owns the rendezvous (the 128-byte ncclUniqueId travels over
// whatever the launcher provides -- torch.distributed/gloo under torchrun).
//
// Layout: every rank holds a full-size padded matrix.  Block column j (width nb, a multiple of 128) is OWNED
// by rank j % world: only the owner assembles and updates it.  At step k the owner packs its panel
// (rows >= k*nb of block column k) into a contiguous buffer, factors it there (diagonal block: recursive
// potrf; rows below: TRSM against it), appends the inverted 128x128 diagonal leaves, and broadcasts the
// buffer.  Every rank then (a) stores the panel and the leaf inverses into its own matrix -- so that at the
// end each rank holds the complete factor, which lets posterior/IVAR evaluation shard the evaluation points
// with no further exchange of L -- and (b) applies  C_j -= P_j.. P_j^T  to each owned block column j > k.
#include "gpx_internal.h"
#include <dlfcn.h>
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
  GPX_SYM(Reduce, "ncclReduce");
  GPX_SYM(CommSplit, "ncclCommSplit");
  GPX_SYM(Send, "ncclSend");
  GPX_SYM(Recv, "ncclRecv");
  GPX_SYM(GroupStart, "ncclGroupStart");
  GPX_SYM(GroupEnd, "ncclGroupEnd");
  GPX_SYM(GetErrorString, "ncclGetErrorString");
#undef GPX_SYM
  g_rccl.h = h;
  return 0;
}

#define GPX_NCCL(call)                                                                     \
  do {                                                                                     \
    ncclResult_t r_ = (call);                                                              \
    if (r_ != 0) {                                                                         \
      gpx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_)); \
      return -3;                                                                           \
    }                                                                                      \
  } while (0)

namespace {

// dst[r][c] = src[r][c] for an (rows x cols) block with independent leading dimensions
__global__ __launch_bounds__(256) void copy2d_kernel(const double* __restrict__ src, int64_t lds_,
                                                     double* __restrict__ dst, int64_t ldd, int64_t rows,
                                                     int64_t cols) {
  const int64_t c2 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r = blockIdx.y;
  if (c2 >= cols || r >= rows) return;
  *reinterpret_cast<double2*>(dst + r * ldd + c2) = *reinterpret_cast<const double2*>(src + r * lds_ + c2);
}

}  // namespace

int gpx_copy2d(gpx_ctx* ctx, const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  // grid.y is limited to 65535 rows per launch
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t rr = rows - r0 < 65535 ? rows - r0 : 65535;
    dim3 grid((unsigned)((cols / 2 + 255) / 256), (unsigned)rr);
    hipLaunchKernelGGL(copy2d_kernel, grid, dim3(256), 0, ctx->stream, src + r0 * lds_, lds_, dst + r0 * ldd, ldd, rr,
                       cols);
  }
  GPX_HIP(hipGetLastError());
  return 0;
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
  const int pr = ctx->rank / Pc, pc = ctx->rank % Pc;
  ncclComm_t row = nullptr, col = nullptr;
  GPX_NCCL(g_rccl.CommSplit((ncclComm_t)ctx->comm, /*color*/ pr, /*key*/ pc, &row, nullptr));
  GPX_NCCL(g_rccl.CommSplit((ncclComm_t)ctx->comm, /*color*/ Pr + pc, /*key*/ pr, &col, nullptr));
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
  double* p = buf->p + offset;
  GPX_NCCL(g_rccl.Broadcast(p, p, (size_t)count, ncclFloat64, root, (ncclComm_t)ctx->grp[grp], ctx->stream));
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
// below GPX_SAG_MIN doubles go from the root to every peer directly (one phase, latency-bound anyway).
// Asynchronous on the selected stream; every rank must call it with identical arguments.
static int64_t sag_min_elems() {
  static int64_t v = -1;
  if (v < 0) {
    const char* e = getenv("GPX_SAG_MIN");
    v = e ? atoll(e) : (int64_t)1 << 16;
  }
  return v;
}

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

// =====================================================================================================================
// 2-D block-cyclic distributed Cholesky (north_star).  Process grid Pr x Pc, rank (pr, pc) = (rank / Pc,
// rank % Pc); global block (I, J) of the padded matrix (block size nb, the last block may be shorter) lives on rank
// (I % Pr, J % Pc) at local block (I / Pr, J / Pc) of that rank's LOCAL matrix -- each rank allocates only its
// ~N^2 / (Pr Pc) share.  gpexp_amd/dist.py drives the panel loop; the pieces below are its device primitives.  A panel
// step k moves data through ONE packed buffer G per rank with Pr "pieces", piece p = what process row p contributes:
//     [ D: nb x nb factored diagonal block (row stride nb) | nb/128 inverted 128x128 leaves | rows: m_p x nb ]
// where rows = the blocks L_Ik, I > k, I % Pr == p, in ascending I (D is only meaningful in piece k % Pr).
// =====================================================================================================================
namespace {

// dst[((r / nb) * stride + first) * nb + r % nb][c] = src[r][c]: packed piece rows -> rows of the replicated factor
__global__ __launch_bounds__(256) void copy_cyclic_rows_kernel(const double* __restrict__ src, int64_t lds_,
                                                               double* __restrict__ dst, int64_t ldd, int64_t rows,
                                                               int64_t cols, int64_t nb, int64_t first, int64_t stride) {
  const int64_t c2 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  const int64_t r = blockIdx.y;
  if (c2 >= cols || r >= rows) return;
  const int64_t gr = ((r / nb) * stride + first) * nb + r % nb;
  *reinterpret_cast<double2*>(dst + gr * ldd + c2) = *reinterpret_cast<const double2*>(src + r * lds_ + c2);
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
int gpx_dist2_diag_factor(gpx_ctx* ctx, gpx_mat* A, int64_t lr, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                          int64_t nb, int64_t base, int64_t n_valid) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr, w, lc, w));
  GPX_ARG(w <= nb && nb % GPX_TILE == 0 && doff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes, "D region outside G");
  if (!A->aux) {
    A->aux_bytes = A->prows * GPX_TILE * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, A->aux_bytes, &p));
    A->aux = (double*)p;
  }
  double* D = G->p + doff;
  double* Dinv = D + nb * nb;
  double* Ablk = A->p + lr * A->ld + lc;
  GPX_TRY(gpx_copy2d(ctx, Ablk, A->ld, D, nb, w, w));
  GPX_TRY(chol_potrf_nozero(ctx, D, nb, w, Dinv, base, n_valid));
  GPX_TRY(gpx_copy2d(ctx, D, nb, Ablk, A->ld, w, w));
  GPX_HIP(hipMemcpyAsync(A->aux + (lr / GPX_TILE) * GPX_TILE * GPX_TILE, Dinv, (size_t)((w / GPX_TILE) * GPX_TILE * GPX_TILE * 8),
                         hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

// holders of block column k (ranks with pc == k % Pc): X = A[lr0 : lr0+m, lc : lc+w] <- X L_kk^-T in place (L_kk and its
// leaf inverses from the D region at doff), then packed (row stride nb) into G at roff.  Asynchronous.
int gpx_dist2_panel_trsm(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc, int64_t w, gpx_mat* G, int64_t doff,
                         int64_t roff, int64_t nb) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc, w));
  GPX_ARG(w <= nb && doff >= 0 && roff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes &&
              (roff + m * nb) * 8 <= G->bytes, "region outside G");
  if (m == 0) return 0;
  const double* D = G->p + doff;
  double* X = A->p + lr0 * A->ld + lc;
  GPX_TRY(chol_trsm_right(ctx, D, nb, D + nb * nb, X, A->ld, m, w));
  return gpx_copy2d(ctx, X, A->ld, G->p + roff, nb, m, w);
}

// A[lr0 : lr0+m, lc0 : lc0+n] -= G[aoff](m x w, stride nb) * G[boff](n x w, stride nb)^T   (trailing update of one local
// block column by panel k).  Asynchronous.
int gpx_dist2_update(gpx_ctx* ctx, gpx_mat* A, int64_t lr0, int64_t m, int64_t lc0, int64_t n, const gpx_mat* G,
                     int64_t aoff, int64_t boff, int64_t w, int64_t nb) {
  GPX_ARG(ctx && G, "NULL argument");
  GPX_TRY(check_local(A, lr0, m, lc0, n));
  GPX_ARG(w > 0 && w <= nb && w % 16 == 0 && aoff >= 0 && boff >= 0 && (aoff + m * nb) * 8 <= G->bytes &&
              (boff + n * nb) * 8 <= G->bytes, "operand outside G");
  if (m == 0 || n == 0) return 0;
  return launch_gemm(ctx, G->p + aoff, nb, G->p + boff, nb, A->p + lr0 * A->ld + lc0, A->ld, m, n, w, true, true, false);
}

// replicated factor: the m x w rows of a piece (blocks first, first+stride, ... in ascending order) -> rows of block
// column `col0 / nb` of the full-size matrix L
int gpx_dist2_unpack_rows(gpx_ctx* ctx, const gpx_mat* G, int64_t roff, int64_t m, int64_t w, int64_t nb, gpx_mat* L,
                          int64_t first_block, int64_t stride, int64_t col0) {
  GPX_ARG(ctx && G && L, "NULL argument");
  GPX_ARG(roff >= 0 && m >= 0 && (roff + m * nb) * 8 <= G->bytes && w % 2 == 0 && col0 + w <= L->pcols, "bad piece");
  if (m == 0) return 0;
  const int64_t last = ((m - 1) / nb * stride + first_block) * nb + (m - 1) % nb;
  GPX_ARG(last < L->prows, "piece rows fall outside the replicated factor");
  for (int64_t r0 = 0; r0 < m; r0 += 65535 / nb * nb) {
    int64_t rr = m - r0;
    if (rr > 65535 / nb * nb) rr = 65535 / nb * nb;
    dim3 grid((unsigned)((w / 2 + 255) / 256), (unsigned)rr);
    hipLaunchKernelGGL(copy_cyclic_rows_kernel, grid, dim3(256), 0, ctx->stream, G->p + roff + r0 * nb, nb,
                       L->p + col0, L->ld, rr, w, nb, first_block + (r0 / nb) * stride, stride);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

// replicated factor: diagonal block k (global offset r0 = k * nb) and its leaf inverses from the D region
int gpx_dist2_unpack_diag(gpx_ctx* ctx, const gpx_mat* G, int64_t doff, int64_t w, int64_t nb, gpx_mat* L, int64_t r0) {
  GPX_ARG(ctx && G && L, "NULL argument");
  GPX_ARG(doff >= 0 && (doff + gpx_dist2_diag_elems(nb)) * 8 <= G->bytes && r0 + w <= L->prows && w <= nb, "bad D region");
  if (!L->aux) {
    L->aux_bytes = L->prows * GPX_TILE * 8;
    void* p;
    GPX_TRY(gpx_dev_alloc(ctx, L->aux_bytes, &p));
    L->aux = (double*)p;
  }
  GPX_TRY(gpx_copy2d(ctx, G->p + doff, nb, L->p + r0 * L->ld + r0, L->ld, w, w));
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
  GPX_ARG((xoff + m) * 8 <= x->bytes && (aoff + w) * 8 <= acc->bytes, "vector segment out of range");
  const int64_t pe = colreduce_partial_elems(m, w);
  void* pp;
  GPX_TRY(gpx_dev_alloc(ctx, pe * 8 + 8, &pp));
  int r = launch_colreduce(ctx, Ab, A->ld, m, w, x->p + xoff, acc->p + aoff, (double*)pp, 1);
  (void)hipStreamSynchronize(ctx->stream);  // the partials go back to the pool
  gpx_dev_release(ctx, pp, pe * 8 + 8);
  return r;
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
