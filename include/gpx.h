/* gpx.h -- C ABI of libgpx_hip.so: the MI355X (gfx950) GP-inference hot path behind GPEXP's API.
 *
 * The reference (goroda/GPEXP) is pure Python and has NO existing FFI/plugin layer;
 * its boundary is the Python class API.  Each entry point below replaces the NumPy/LAPACK work
 * of the reference call site cited next to it (file:line relative to the reference root) and is
 * bound from Python by ctypes in gpexp_amd/_lib.py (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - All matrices/vectors are IEEE fp64.  Host buffers are C-contiguous row-major (NumPy default),
 *    caller-owned, and must outlive the call only.
 *  - gpx_mat is a library-owned dense device matrix (row-major, leading dimension >= cols, storage
 *    padded to a multiple of 128 in both dimensions; the padding is kept as an identity / zero
 *    extension so that factorisations and solves never see an edge tile).  Free with gpx_mat_free.
 *  - Return value: 0 = ok; >0 = 1-based index of the first non-positive Cholesky pivot
 *    (the reference never fails here because numpy.linalg.pinv silently truncates, gp.py:181);
 *    <0 = argument / HIP / RCCL error, text via gpx_last_error().
 *  - Calls are blocking unless stated; one gpx_ctx per process (= per GPU); not thread-safe by
 *    contract (the reference's callers are single-threaded).
 *  - Placement independence: the stationary kernels subtract coordinates before anything else in the reference
 *    (kernels.py:121-122, 87-89).  The assembly centres every point set on its bounding-box midpoint (computed on the
 *    host at gpx_mat_from_host) and switches to raw coordinate differences when the centred domain is still wide
 *    relative to the length scales, so results do not depend on where the inputs sit (gpx_dbg_kfill_plan reports it).
 *  - Covariance kernels are passed flat as (kind, d, hyp[nhyp]):
 *      GPX_K_SE        hyp = {cl_0..cl_{d-1}, signalSize}          kernels.py:100-123
 *      GPX_K_MATERN32  hyp = {rho, signalSize}                      kernels.py:72-91
 *      GPX_K_MATERN52  hyp = {rho, signalSize}                      (absent from the reference)
 *      GPX_K_MEHLER    hyp = {t_0..t_{d-1}}                         kernels.py:183-228, 250-293
 */
#ifndef GPX_H
#define GPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_ABI_VERSION 2
#define GPX_MAX_DIM 32

typedef struct gpx_ctx gpx_ctx;
typedef struct gpx_mat gpx_mat;

enum gpx_kernel_kind { GPX_K_SE = 0, GPX_K_MATERN32 = 1, GPX_K_MATERN52 = 2, GPX_K_MEHLER = 3 };

/* names of the timed kernel classes reported by gpx_profile_get */
enum gpx_prof_class {
  GPX_PROF_KFILL = 0,   /* symmetric covariance assembly K(X,X) (HBM-bound: 8*rows*cols bytes written) */
  GPX_PROF_GEMM = 1,    /* fp64 MFMA GEMM/SYRK/TRSM-update tiles (MFMA-bound: 2*m*n*k flops) */
  GPX_PROF_LEAF = 2,    /* 128x128 diagonal potf2 + trtri */
  GPX_PROF_TRSV = 3,    /* potrs sweeps */
  GPX_PROF_REDUCE = 4,  /* column reductions / logdet */
  GPX_PROF_GREEDY = 5,  /* greedy-design scoring kernels */
  GPX_PROF_COMM = 6,    /* RCCL collectives */
  GPX_PROF_KCROSS = 7,  /* rectangular cross-covariance assembly K(X,Z): every element computed (VALU: sqrt + exp) */
  GPX_PROF_NCLASS = 8
};

/* ---- lifecycle ------------------------------------------------------------------------------ */
int gpx_abi_version(void);
const char* gpx_last_error(void);
/* device = HIP ordinal (LOCAL_RANK in the one-process-per-GPU launch).  Fails (<0) when no GPU. */
int gpx_create(int device, gpx_ctx** out);
int gpx_destroy(gpx_ctx* ctx);
int gpx_sync(gpx_ctx* ctx);   /* device-wide: every stream of the context */
/* release cached workspace back to HIP */
int gpx_trim(gpx_ctx* ctx);
/* device facts for reports: name[<=256], CU count, HBM bytes, clock MHz */
int gpx_device_info(gpx_ctx* ctx, char* name, int name_len, int* cus, int64_t* hbm_bytes, int* clock_mhz);

/* ---- device matrices ------------------------------------------------------------------------- */
/* upload a (rows x cols) host array; pad != 0 pads storage to multiples of 128 (zero filled) */
int gpx_mat_from_host(gpx_ctx* ctx, const double* src, int64_t rows, int64_t cols, int pad, gpx_mat** out);
int gpx_mat_alloc(gpx_ctx* ctx, int64_t rows, int64_t cols, int pad, gpx_mat** out);
int gpx_mat_free(gpx_ctx* ctx, gpx_mat* m);
/* device-to-device duplicate of a matrix INCLUDING its factor state (leaf inverses): the class API under a multi-process
 * launch hands every GP object its own copy of the replicated factor the distributed runner assembled (gp.py:181's
 * precisionMatrix role); blocking on the selected stream */
int gpx_mat_clone(gpx_ctx* ctx, const gpx_mat* src, gpx_mat** out);
int gpx_mat_shape(const gpx_mat* m, int64_t* rows, int64_t* cols, int64_t* ld);
/* tri: 0 = as stored, 1 = lower triangle (strict upper written as 0), 2 = lower mirrored to upper.
 * Backs the lazy GP.covarianceMatrix / GP.precisionMatrix attributes (gp.py:178-181). */
int gpx_mat_to_host(gpx_ctx* ctx, const gpx_mat* m, double* dst, int tri);

/* ---- L0/L1: covariance assembly -------------------------------------------------------------- */
/* K[i][j] = k(X_i, X_j) + nugget_i*delta_ij  (Z == NULL; N x N)      gp_kernel_utilities.py:34-68
 * K[i][j] = k(X_i, Z_j)                      (Z != NULL; N x M)      gp.py:132-135, 246-249;
 *                                                                    experimentalDesign.py:829-831
 * X, Z: device point sets (rows = points, cols = d).  nugget: host, nugget_len in {0,1,N}. */
int gpx_kfill(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
              const gpx_mat* X, const gpx_mat* Z, const double* nugget, int64_t nugget_len,
              gpx_mat** outK);
/* same, into an existing matrix of the right shape (no allocation inside timed loops) */
int gpx_kfill_into(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                   const gpx_mat* X, const gpx_mat* Z, const double* nugget, int64_t nugget_len,
                   gpx_mat* K);
/* k(Z_j, Z_j) for every point -> host out[M]                          gp.py:140, 251 */
int gpx_kdiag(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* Z, double* out);

/* Kernel.evaluate semantics (kernels.py:49-65): out[i] = k(A_i, B_i) for equally sized host point sets, or
 * one point against n (na == 1 or nb == 1); out has max(na, nb) entries.  Any other shape pair is an error. */
int gpx_kernel_eval(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                    const double* A, int64_t na, const double* B, int64_t nb, double* out);

/* ---- L2: factorisation and solves (replace numpy.linalg.pinv / slogdet) ---------------------- */
/* in-place lower Cholesky K = L L^T; replaces pinv at gp.py:181, 400.
   INVARIANT of every factored matrix (gpx_potrf, gpx_refit_rows, the distributed fits): only the lower triangle is defined.
   The strict upper part holds whatever the storage held before -- finite leftovers of K after gpx_potrf, possibly NaN / Inf
   pool contents after gpx_refit_rows (which copies the old factor's lower triangle only).  No entry point reads it: solves,
   posterior, gradients, gpx_potri and further refits take the lower triangle; gpx_mat_to_host(tri = 1 / 2) masks / mirrors it.
   tests/test_gpu_refit.py runs them on a refit factor whose pool blocks were NaN-filled. */
int gpx_potrf(gpx_ctx* ctx, gpx_mat* K);
/* Pivot policy of every factorisation that follows: a pivot <= piv_min is bad; skip == 0 reports it (status > 0 from
 * gpx_potrf, the default with piv_min = 0), skip != 0 DROPS the point instead -- L_jj = 1, the rest of column j and row j of
 * the inverse are 0, so solves return 0 in that component, as if the point were not in the set: what numpy.linalg.pinv
 * (gp.py:181, experimentalDesign.py:826) makes of an exactly duplicated point.  gpx_potrf_dropped: pivots dropped by the
 * last factorisation. */
int gpx_potrf_policy(gpx_ctx* ctx, double piv_min, int skip);
int gpx_potrf_dropped(gpx_ctx* ctx, int* count);
/* factor K(X)+nugget when its leading `keep` (multiple of 128) rows/columns equal the matrix Lold factors
 * (the design loop pins earlier points by bounds, experimentalDesign.py:722-724, and only the last batch moves): the
 * leading factor block is copied, rows >= keep are assembled and the factorisation is completed in O(N^2 b).
 * keep == 0 (Lold may be NULL) is a plain assemble + factor.  Status as gpx_potrf; *outL is a new library-owned matrix. */
int gpx_refit_rows(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X,
                   const double* nugget, int64_t nugget_len, const gpx_mat* Lold, int64_t keep, gpx_mat** outL);
/* alpha = K^{-1} y from the factor; y, alpha host (N)                 gp.py:101, 435 */
int gpx_potrs(gpx_ctx* ctx, const gpx_mat* L, const double* y, double* alpha);
/* the same on device vectors (y, alpha: at least padded-N doubles, zero padded), ASYNCHRONOUS on the selected
 * stream -- lets the latency-bound sweeps run on the side stream underneath the evaluation GEMMs */
int gpx_potrs_dev(gpx_ctx* ctx, const gpx_mat* L, const gpx_mat* y, gpx_mat* alpha);
/* log det K = 2 sum log L_ii                                          gp.py:434 (slogdet) */
int gpx_logdet(gpx_ctx* ctx, const gpx_mat* L, double* out);
/* explicit inverse (lower triangle valid) for the lazy precisionMatrix attribute and lml_grad */
int gpx_potri(gpx_ctx* ctx, const gpx_mat* L, gpx_mat** outP);

/* posterior at M points: mean_j = k_j^T alpha (gp.py:137), var_j = k(z_j,z_j) - k_j^T K^{-1} k_j
 * (signed, gp.py:253-256; the caller applies abs for GP.evaluate, gp.py:145).
 * alpha (host, N) may be NULL when mean == NULL; mean / var (host, M) may each be NULL. */
int gpx_posterior(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                  const gpx_mat* L, const gpx_mat* X, const double* alpha,
                  const gpx_mat* Z, double* mean, double* var);
/* full M x M posterior covariance (compvar=2, gp.py:146-152) -> host cov[M*M] */
int gpx_posterior_cov(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                      const gpx_mat* L, const gpx_mat* X, const gpx_mat* Z, double* cov);

/* ---- L3: design-cost evaluators ---------------------------------------------------------------- */
/* IVAR = (1/M) sum_j var_j (signed mean; caller applies abs)          experimentalDesign.py:104-117 */
int gpx_ivar(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
             const gpx_mat* L, const gpx_mat* X, const gpx_mat* Z, double* out);
/* The same cost, keeping W = L^-1 K(X, Z) (N x M, a matrix of the context: release with gpx_mat_free) for the gradient AT THE
 * SAME DESIGN: an optimiser asks for the cost and then its gradient at one point (experimentalDesign.py:471-489), and the forward
 * solve is a third of the gradient's work.  *W = NULL when Z does not fit one evaluation chunk. */
int gpx_ivar_keep(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                  const gpx_mat* L, const gpx_mat* X, const gpx_mat* Z, double* out, gpx_mat** W);
/* The cost once more after gpx_refit_rows kept the leading `keep` rows of the factor: W (from gpx_ivar_keep / an earlier update, for
 * the design whose factor the refit started from) keeps its leading rows, the rows from `keep` on are re-assembled and re-solved in
 * place -- 2 (N - keep) keep M flops instead of N^2 M (the batch loop of experimentalDesign.py:694-751 moves the last batch only). */
int gpx_ivar_update(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* Z, gpx_mat* W, int64_t keep, double* out);
/* GP fit + IVAR in one call -- what costFunctionGP_IVAR.evaluate (experimentalDesign.py:104-117: refit, then
 * evaluateVariance over the MC points) amounts to per optimiser evaluation: K (assembled, gpx_kfill) is factored in place
 * as by gpx_potrf and *out receives what gpx_ivar would return on the finished factor.  With GPX_FIT_IVAR_STREAMED=1 the
 * evaluation solve is streamed underneath the factorisation (panel events of the blocked look-ahead Cholesky, a
 * low-priority stream of its own); on one GPU that measured slower than factor-then-solve, so it is opt-in.
 * Status as gpx_potrf.  Main stream only. */
int gpx_fit_ivar(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, gpx_mat* K, const gpx_mat* X, const gpx_mat* Z,
                 double* out);
/* greedy maximum-posterior-variance selection among M candidates, nugget 0 (experimentalDesign.py:787-845).
 * keep[nkeep] = indices already selected; selects until nsel indices in total; out_idx[nsel] receives
 * keep followed by the new picks; w (host, M) optional weights; first-max tie rule (np.argmax). */
int gpx_greedy_var(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                   const gpx_mat* C, const double* w, const int64_t* keep, int64_t nkeep,
                   int64_t nsel, int64_t* out_idx);
/* one-step-lookahead greedy IVAR among candidates C for a GP already factored on X (L):
 * cost_j = IVAR(X u {c_j}) over MC points Z with noise variance `noise` on the new point;
 * out_cost[M] (host, nullable) all costs, *out_best = first arg-min.  Composition oracle: SURVEY.md 8c. */
int gpx_greedy_ivar_step(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp,
                         const gpx_mat* L, const gpx_mat* X, const gpx_mat* C, const gpx_mat* Z,
                         double noise, double* out_cost, int64_t* out_best);
/* Multi-pick greedy IVAR with RESIDENT state (composition of experimentalDesign.py:79-117 per SURVEY 8c): nsel picks cost one
 * set-up (the work of ONE gpx_greedy_ivar_step) + per pick one pass over W_C = L^-1 K(X, C) and one over cov(Z, C | design) --
 * no refit, no N^2 solve.  Picks equal nsel rounds of gpx_greedy_ivar_step + refit on the winner (a candidate may be picked
 * again, as there).  out_idx[nsel]; out_cost[nsel] (the winner's cost at each pick, optional); all_costs (nsel x M, optional). */
int gpx_greedy_ivar(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* C, const gpx_mat* Z, double noise, int64_t nsel, int64_t* out_idx, double* out_cost,
                    double* all_costs);

/* greedy mutual-information design among M candidates with noise variance `noise`, seeded with `start`
 * (experimentalDesign.py:223-285, 753-785): out_idx[nsel] = start followed by the picks (first-max tie rule);
 * out_ratio[nsel-1] (nullable) = winning ratio var(c|A)/var(c|all\A\c) of every step.  M <= 65535. */
int gpx_mi_greedy(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* C, double noise,
                  int64_t nsel, int64_t start, int64_t* out_idx, double* out_ratio);

/* ---- hyper-parameter gradient -------------------------------------------------------------------- */
/* grad[k] = 1/2 tr((alpha alpha^T - K^-1) dK/d theta_k), theta = {hyp[0..nhyp-1], noise}; the noise entry is
 * the raw 1/2 tr(alpha alpha^T - K^-1) (the caller applies the reference's x 2*noise, gp.py:463-464).
 * Squared-exponential kernel only (gp.py:444-466 + kernels.py:125-144; the reference raises for the others). */
int gpx_lml_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                 const double* alpha, double* grad);

/* the same raw sums over ALL rows at once, for ONE GPU that can hold two more N x N buffers: L^-1 (N^3/3, large products),
 * U = L^-T, lower K^-1 = U U^T with the zero part of every tile's k range skipped (N^3/3), written over L^-1 -- the 2 N^3 / 3
 * flops of the slabs at the rate of large GEMMs, and N^2 less memory than gpx_potri + gpx_lml_grad (gp.py:444-466) */
int gpx_lml_grad_linv(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                      const double* alpha, double* sums);


/* ---- point-location gradients of the posterior variance ----------------------------------------------
 * For the two kernels the reference differentiates: squared exponential (kernels.py:146-181, including its doubled
 * signalSize, :177) and 1-D Mehler (GPX_K_MEHLER with d == 1; kernels.py:295-324); any other kind is an argument error.
 * noise_deriv (host N x d, nullable) = d noise(x_j) / d x_jl of a heteroscedastic noise model (space.noiseFunc.deriv,
 * gp.py:314-317): it enters the derivative of the covariance at coincident training points.  Nothing N x N reaches the host. */
/* grad[a*d + l] = d IVAR / d X[a][l] = (1/M) sum_m d var(z_m) / d X[a][l]
 * (costFunctionGP_IVAR.derivative, experimentalDesign.py:168-179 -> gp.py:282-341).  grad: host, N*d. */
int gpx_ivar_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                  const gpx_mat* Z, const double* noise_deriv, double* grad);
/* ... with the forward solve kept by gpx_ivar_keep for the same L, X, Z (W == NULL: exactly gpx_ivar_grad) */
int gpx_ivar_grad_w(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                    const gpx_mat* Z, const double* noise_deriv, const gpx_mat* W, double* grad);
/* ... for the design points from r0 (a multiple of 128) on only: the batch loop pins the earlier ones by equal bounds
 * (experimentalDesign.py:719-724).  Squared exponential, homoscedastic; grad: (N - r0) x d.  2 (N - r0) N M flops instead of 2 N^2 M. */
int gpx_ivar_grad_rows(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                       const gpx_mat* Z, const gpx_mat* W, int64_t r0, double* grad);
/* out[(j*d + l) * M + m] = d var(z_m) / d X[j][l]  (GP.evaluateVarianceDerivative, gp.py:282-341; host, (N*d) x M).
 * eval_bias (host N, nullable) / dk_bias (host N x d, nullable): the terms of gp.py:318-320 -- noise(x_j) added to
 * k(x_j, z_m) and noise'(x_j) subtracted from -dk(z_m, x_j)/dz for every m -- which the reference applies when the WHOLE
 * evaluation set coincides with training point j; the caller decides (rows of zeros otherwise). */
int gpx_var_grad(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                 const gpx_mat* Z, const double* noise_deriv, const double* eval_bias, const double* dk_bias,
                 double* out);
/* out[m*d + l] = d var(z_m) / d z_m[l]  (GP.evaluateVarianceDerivWRTnewpt, gp.py:261-280; host, M*d) */
int gpx_var_grad_newpt(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* L, const gpx_mat* X,
                       const gpx_mat* Z, double* out);


/* out[rows] = A v for a resident matrix (covTimesV, gp_kernel_utilities.py:107-143: the Nystrom operator application) */
int gpx_matvec(gpx_ctx* ctx, const gpx_mat* A, const double* v, double* out);

/* ---- f4: FITC sparse approximation (gp.py:182-210, 401-426; gp_kernel_utilities.py:70-104) ---------------------
 * Inducing points S (nu x d, a subset of the nodes in the reference: np.random.permutation, gp.py:188).  The model keeps
 * chol(Quu), Kuf, G = diag(K - Q) and chol(Quu + Kuf G^-1 Kfu) on the device; no N x N matrix is formed. */
typedef struct gpx_fitc gpx_fitc;
int gpx_fitc_fit(gpx_ctx* ctx, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X, const gpx_mat* S,
                 double noise, gpx_fitc** out);
int gpx_fitc_free(gpx_ctx* ctx, gpx_fitc* f);
int gpx_fitc_shape(const gpx_fitc* f, int64_t* n, int64_t* nu);
/* coeff = P y with the Woodbury precision (GP.train, gp.py:100-101); quad (nullable) = y^T P y */
int gpx_fitc_solve(gpx_ctx* ctx, const gpx_fitc* f, const double* y, double* coeff, double* quad);
/* log det(Q + G) (loglikeParams, gp.py:434) */
int gpx_fitc_logdet(gpx_ctx* ctx, const gpx_fitc* f, double* out);
/* GP.evaluate / evaluateVariance with the FITC precision (gp.py:132-145, 246-255): mean (nullable; needs coeff) and the
 * SIGNED variance (nullable) at the M points of Z */
int gpx_fitc_posterior(gpx_ctx* ctx, const gpx_fitc* f, const gpx_mat* X, const double* coeff, const gpx_mat* Z,
                       double* mean, double* var);
/* dense Q + G and P (host n x n, each nullable): the covarianceMatrix / precisionMatrix attributes (gp.py:200-206) */
int gpx_fitc_dense(gpx_ctx* ctx, const gpx_fitc* f, double* cov, double* prec);
/* GP.evaluateVarianceDerivative / evaluateVarianceDerivWRTnewpt on a FITC model: the reference computes both from whatever
 * `precisionMatrix` holds (gp.py:275, 322), for FITC the Woodbury precision of gp.py:204-206.  Same outputs and optional
 * arguments as gpx_var_grad / gpx_var_grad_newpt; (kind, d, hyp) = the kernel the model was fitted with. */
int gpx_fitc_var_grad(gpx_ctx* ctx, const gpx_fitc* f, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X,
                      const gpx_mat* Z, const double* noise_deriv, const double* eval_bias, const double* dk_bias, double* out);
int gpx_fitc_var_grad_newpt(gpx_ctx* ctx, const gpx_fitc* f, int kind, int d, const double* hyp, int nhyp, const gpx_mat* X,
                            const gpx_mat* Z, double* out);

/* ---- measurement ------------------------------------------------------------------------------- */
/* when enabled every kernel launch of a class is bracketed by HIP events on the launch stream */
int gpx_profile_enable(gpx_ctx* ctx, int on);
int gpx_profile_reset(gpx_ctx* ctx);
/* sums since reset: launches, elapsed ms (HIP events), algorithmic flops and bytes */
int gpx_profile_get(gpx_ctx* ctx, int prof_class, int64_t* launches, double* ms, double* flops, double* bytes);

/* Test hooks (gpx_dbg_*: kernel-level entry points, allocator guards, schedule replays) are declared in gpx_debug.h; they are
 * exported by the same library but are not part of the drop-in ABI. */

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
