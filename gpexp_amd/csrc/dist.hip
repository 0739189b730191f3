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
  return 0;
}

int gpx_comm_destroy(gpx_ctx* ctx) {
  if (!ctx || !ctx->comm) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  g_rccl.CommDestroy((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
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
  GPX_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
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

int gpx_dist_finish(gpx_ctx* ctx, gpx_mat* K) {
  GPX_ARG(ctx && K && K->aux, "matrix was not factored by the distributed panel loop");
  GPX_HIP(hipStreamSynchronize(ctx->stream));
  K->factored = 1;
  return 0;
}

}  // extern "C"
