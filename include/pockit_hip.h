/* pockit_hip.h -- C ABI of the MI355X NLP-callback evaluator (libpockit_hip.so).
 *
 * Drop-in boundary: these entry points are what a host binding for pockit's evaluator path
 * binds.  Each eval function replaces one method of the reference's cyipopt ``problem_obj``
 * (class SystemBase in /root/reference/pockit/base/systembase.py):
 *
 *   pk_eval_f     <-  SystemBase.objective(x)                     systembase.py:602-605
 *   pk_eval_grad  <-  SystemBase.gradient(x)                      systembase.py:646-657
 *   pk_eval_g     <-  SystemBase.constraints(x)                   systembase.py:613-623
 *   pk_eval_jac   <-  SystemBase.jacobian(x)                      systembase.py:676-693
 *   pk_eval_hess  <-  SystemBase.hessian(x, lagrange, obj_factor) systembase.py:820-835
 *   pk_get_structure <- jacobianstructure()/hessianstructure()    systembase.py:671-674,811-818
 * and the caller they serve is cyipopt.Problem(...) built in
 * /root/reference/pockit/optimizer/ipopt.py:41-53.
 *
 * Conventions: every function returns 0 on success and a non-zero code on error, with a message
 * retrievable through pk_last_error(); the caller owns all host buffers; the library owns device
 * memory, its stream and the loaded code object; ``x`` is never written (the reference mutates
 * boundary slots in place, phasebase.py:840-847 -- we evaluate as if, without touching x); one
 * context per GPU, calls on one context are serialized by the caller (as IPOPT does).
 * No function falls back to the CPU: without a GPU / code object every eval returns an error.
 *
 * The *_dev variants take device pointers (e.g. torch tensors' data_ptr()) and a hipStream_t
 * (NULL = the context's stream); they enqueue only and do not synchronize.
 */
#ifndef POCKIT_HIP_H
#define POCKIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pk_ctx pk_ctx;

/* Compile-time facts of a generated model (pockit_amd/codegen.py) the runtime needs to launch it. */
typedef struct pk_model_desc {
  int32_t n_phase;
  int32_t n_I;          /* number of integral symbols (length of the device I buffer)            */
  int32_t nred;         /* PK_NRED the code object was compiled with                             */
  int32_t lds_g;        /* LDS doubles per wave for eval_g / eval_jac / eval_hess                */
  int32_t lds_j;
  int32_t lds_h;
  int32_t ne_j;         /* number of boundary/system scalar expressions of eval_jac / eval_hess  */
  int32_t ne_h;
  int32_t prepass_f;    /* 1 if the callback needs the integral pre-pass (pk_int)                 */
  int32_t prepass_grad;
  int32_t prepass_g;
  int32_t prepass_jac;
  int32_t prepass_hess;
  int32_t lds_x;        /* LDS doubles per wave of the fused x-kernel (pk_xall)                   */
  int32_t ne_a;         /* scalar expressions of the auxiliary pass (outer-product Hessian path)  */
  int32_t ne_hc;        /* scalar expressions of the compact Hessian                              */
  int32_t lds_e;        /* LDS doubles per wave of the mesh error estimation kernel (pk_err)      */
  int32_t tab_cap;      /* PK_TAB_CAP the code object was compiled with: entries of a staged pattern table, 64 or 256 */
  int32_t sharded;      /* 1 if the code object was generated with PK_SHARDED (in-launch exchange between GPUs)       */
  int32_t lds_jc;       /* LDS doubles per wave of the compact Jacobian kernel (pk_jacc)                                */
  int32_t ne_jc;        /* scalar expressions of the compact Jacobian                                                   */
  int32_t max_phases;   /* PK_MAX_PHASES the code object was compiled with (0: 8): phase records in its kernel arguments */
  int32_t cycle_subs;   /* workgroups per tile block of pk_cycle for a model evaluated in groups (1 + passes of J + passes of H); 0: 3 / 2 */
  int32_t hess_subs;    /* workgroups per tile block of pk_hess for such a model (passes of H); 0: 1 */
  int32_t hessc_subs;   /* ... of the compact Hessian kernel for such a model (its passes); 0: 1 */
  int32_t jacc_subs;    /* ... of the compact Jacobian kernel for such a model (its passes); 0: 1 */
  int32_t big_global;   /* 1: intervals with more than 64 points stage their rows in the device staging buffer whatever their
                           length (PK_BIG_GLOBAL: the rows of every state do not fit a workgroup's LDS); 0: only beyond 256 */
  int32_t big_rows;     /* rows one such interval stages per sub-slot (0: derived from lds_x / lds_h / lds_jc) */
  int32_t wide;         /* 1: the model has a WIDE phase (its states are evaluated in passes over chunks).  The library then refuses
                           every launch of the kernel pk_xall with error 27: the sequential values role of such a phase has an open defect
                           (round 5); the one-launch cycle and the five callbacks do not use that kernel */
} pk_model_desc;

/* One (model, mesh) instance: sizes plus the table blobs built by pockit_amd/evaluator.py.
 * ``phases``/``tiles``/``kinds``/``items_*`` are arrays of the PkPhase/PkTile/PkKind/PkItem
 * structs of pockit_amd/csrc/pk_abi.h passed as raw bytes. */
typedef struct pk_problem_desc {
  int32_t n, m, n_sys, n_s, l_s;
  int32_t n_phase, n_tiles, n_kinds;
  int64_t nnz_J, nnz_H;
  const void* phases;
  const void* tiles;
  const void* kinds;
  const void* items_jac;
  int32_t n_items_jac;
  const void* items_hess;
  int32_t n_items_hess;
  const int32_t* ib;
  int64_t n_ib;
  const double* db;
  int64_t n_db;
  const int64_t* lb;
  int64_t n_lb;
  int32_t gz_off, n_gz;
  /* outer-product path (objective / system constraints nonlinear in the integrals); all may be empty */
  const void* items_aux;
  int32_t n_items_aux;
  const void* outer;      /* PkOuter[n_outer] */
  int32_t n_outer;
  int32_t n_aux;          /* length of the auxiliary buffer */
  /* compact (coalesced) Hessian layout, optional (nnz_Hc = 0: not available) */
  const void* items_hessc;
  int32_t n_items_hessc;
  int64_t nnz_Hc;
  /* optional COO structure in the reference's order (copied; may be NULL) */
  const int32_t* jac_row;
  const int32_t* jac_col;
  const int32_t* hess_row;
  const int32_t* hess_col;
  /* compact (coalesced) Jacobian layout, optional (nnz_Jc = 0: not available) */
  const void* items_jacc;
  int32_t n_items_jacc;
  int64_t nnz_Jc;
} pk_problem_desc;

int pk_create(pk_ctx** out, int device_id);
void pk_destroy(pk_ctx* ctx);
const char* pk_last_error(pk_ctx* ctx); /* ctx may be NULL: last error of a failed pk_create */
int pk_device_count(void);

int pk_load_model(pk_ctx* ctx, const void* code_object, size_t len, const pk_model_desc* md);
int pk_set_problem(pk_ctx* ctx, const pk_problem_desc* pd);
int pk_get_structure(pk_ctx* ctx, int32_t* jac_row, int32_t* jac_col, int32_t* hess_row, int32_t* hess_col);

/* host-buffer API (what the cyipopt shim calls): H2D, launch, D2H, synchronize */
int pk_eval_f(pk_ctx* ctx, const double* x, double* f);
int pk_eval_grad(pk_ctx* ctx, const double* x, double* grad /* n */);
int pk_eval_g(pk_ctx* ctx, const double* x, double* g /* m */);
int pk_eval_jac(pk_ctx* ctx, const double* x, double* vals /* nnz_J */);
int pk_eval_hess(pk_ctx* ctx, const double* x, const double* lambda, double sigma, double* vals /* nnz_H */);

/* all five outputs of one cycle on the same x (fused x-kernel + Hessian), host buffers */
int pk_eval_cycle(pk_ctx* ctx, const double* x, const double* lambda, double sigma, double* f, double* grad,
                  double* g, double* jac, double* hess);

/* "new x" protocol for host shims.  cyipopt calls objective / gradient / constraints / jacobian separately but on the
 * same iterate, then hessian with fresh multipliers (the five methods ipopt.py:41-53 hands over as ``problem_obj``).
 *   pk_same_x      1 if ``x`` equals the x of the last pk_prepare_x bit for bit (its results are still held)
 *   pk_prepare_x   stages x in pinned memory, uploads it, runs the fused x-kernel (every node evaluated once for f,
 *                  grad f, g, J) and queues the copies of the results into pinned host memory right behind it, in the
 *                  order a solver asks for them -- nothing waits; x and lambda staging is double-buffered
 *   pk_fetch       waits for ONE result (what: 0 f, 1 grad[n], 2 g[m], 3 jac[nnz_J]); out == NULL leaves it where it
 *                  landed (pk_result_location), otherwise it is copied on to ``out``
 *   pk_eval_hess_prepared   Hessian of the Lagrangian on the prepared x (x is not uploaded again)
 *   pk_set_result_targets   where the results of the NEXT prepare / Hessian land: pinned memory of the caller
 *                  (pk_host_alloc; a solver-side array that outlives the call), NULL = the context's own buffers
 *   pk_set_host_mode        prefetch 1 (default): all four x-results are copied out behind the kernel; 0: f and g
 *                  always, grad f and J on first request (a line search's rejected trial points never ask);
 *                  host_direct 1: the kernels store into the pinned host targets themselves (no DMA; A/B switch)
 *   pk_invalidate_x         forget the prepared x (the context's buffers were used by another entry point) */
/* the compact Hessian layout (pk_eval_hessc: one value per distinct position of a node) on the x of the last pk_prepare_x,
 * multipliers given or staged by pk_stage_lambda; vals_pinned = 1: `vals` is pk_host_alloc memory the DMA writes directly.
 * What SystemBase.hessian (systembase.py:820-835) becomes for a solver that was handed the compact structure. */
int pk_eval_hessc_prepared(pk_ctx* ctx, const double* lambda, double sigma, double* vals, int vals_pinned);
int pk_same_x(pk_ctx* ctx, const double* x);
int pk_same_bits(const double* a, const double* b, size_t n);   /* 1 if equal bit for bit (memcmp), for host shims */
int pk_copy_bits(double* dst, const double* src, size_t n);      /* dst = src (non-overlapping), for host shims */
/* k helper threads (0 = none, the default; <= 16) that take slices of pk_same_bits / pk_copy_bits passes of 256 KB and more:
 * rank 0 of the host-landed sharded cycle stages and compares an x that is N times as long as one GPU's.
 * Process-wide; helpers spin for 1 ms after the last pass, otherwise they sleep in 20 us steps. */
int pk_host_threads(int k);
long pk_host_threads_jobs(void); /* slices of passes the helpers have executed so far (diagnostics) */
int pk_host_threads_hot(void);   /* how many of them are spinning right now (diagnostics); -1: the pool gave up -- the caller waited
                                  * more than 1 ms for a helper three times (a host whose CPUs are time slices of fewer cores) */
int pk_prepare_x(pk_ctx* ctx, const double* x);
int pk_fetch(pk_ctx* ctx, int what, double* out);
int pk_eval_hess_prepared(pk_ctx* ctx, const double* lambda /* NULL: staged */, double sigma, double* vals);
/* queue the upload of the next Hessian's multipliers and return (the x check then overlaps the DMA) */
int pk_stage_lambda(pk_ctx* ctx, const double* lambda);
int pk_set_result_targets(pk_ctx* ctx, double* f, double* grad, double* g, double* jac, double* hess);
/* ONE call per callback of a host shim (what a cyipopt binding makes of the five methods it is handed, ipopt.py:41-53):
 *   pk_callback_x     what = 0 objective (systembase.py:602), 1 gradient (:646), 2 constraints (:613), 3 jacobian (:676).
 *                     If `x` is not the prepared iterate it becomes it (pk_prepare_x) and *fresh = 1; its results then land
 *                     in `block`, pinned memory of the caller's laid out like the library's own buffers,
 *                     [J (nnz_J) | grad f (n) | g (m)] (NULL: the context's buffers, pk_host_buffer) -- the pieces of J that
 *                     change with x, grad f and g leave the device in ONE copy.  Then result `what` is waited for; f_out
 *                     receives f for what = 0.  A block must stay allocated until every result of its iterate has been
 *                     fetched or the context's stream has been synchronized (pk_sync, pk_destroy): all results of a new x
 *                     are on their way into it whether or not they are asked for.
 *   pk_callback_hess  SystemBase.hessian (systembase.py:820-835): stages the multipliers, prepares a new x as above, evaluates
 *                     the Hessian of the Lagrangian into `hess` (pinned memory of the caller's or NULL = the context's
 *                     buffer; compact = 1: the compact layout of pk_eval_hessc, `hess` required) and waits for it.
 *   pk_set_jac_constant_runs   runs [start, stop) of the Jacobian values that do not depend on x (the +-1 translation entries
 *                     of phasebase.py:1071-1081 and constant boundary items: 19 % of J at 12k quadrotor nodes): the copy to the
 *                     host skips them from then on; every landing array must have been filled once with
 *   pk_fill_jac_constants      (the context's own buffer is filled by pk_set_jac_constant_runs itself).
 *   pk_set_host_option         A/B switches (default): "spin_wait" (1), "lambda_direct" (1), "chunk_upload" (1), "kernel_upload"
 *                              "separate_x" (0; 1: the five callbacks and the cycle through the stand-alone kernels one after the other,
 *                              never the fused kernel -- the fallback of a code object whose fused kernel fails its self-check),
 *                     (1), "kernel_download" (8: up to that many MiB per copy), "split_copy" (1), "speculative_hess" (1), "mark_wait" (1: the callbacks wait on a word a one-thread
 *                     kernel stores behind the result copies instead of on the stream's state), "hess_direct"
 *                     (1: a Hessian of at most "kernel_download" MiB is stored into its pinned landing array by the kernel itself),
 *                     "xpart_single" (1: the x-results of a new iterate come from ONE launch, pk_cycle without its Hessian role),
 *                     "adaptive_prefetch" (1: grad f and J of a new iterate are copied ahead only
 *                     if they were asked for at the previous one -- a line search's rejected trial points ask for f and g only),
 *                     "small_x_kb" (128: the x threshold of small_direct in KB, an A/B knob),
 *                     "small_direct" (1: an x of <= 128 KB is read by the kernels from its pinned staging buffer, x-results of
 *                     <= 1 MB are stored by the kernel straight into the landing block: no upload / copy launches)
 *                     -- see pk_runtime.cpp. */
int pk_callback_x(pk_ctx* ctx, int what, const double* x, double* block, double* f_out, int* fresh);
int pk_callback_hess(pk_ctx* ctx, const double* x, const double* lambda, double sigma, double* block, double* hess,
                     int compact, int* fresh);
/* all five results of one iterate in ONE call and ONE launch, for a caller that has the multipliers together with x
 * (Evaluator.cycle; SystemBase's five callbacks evaluated at once): same staging and landing as the two callbacks above --
 * `block` = [J | grad f | g] pinned with its constant entries filled in, `hess` = nnz_H pinned values -- returns when
 * everything has landed; the iterate is then the prepared one (pk_callback_x on the same x is served from the block). */
int pk_callback_cycle(pk_ctx* ctx, const double* x, const double* lambda, double sigma, double* block, double* hess,
                      double* f_out);
int pk_set_jac_constant_runs(pk_ctx* ctx, int n_runs, const int64_t* start, const int64_t* stop);
int pk_fill_jac_constants(pk_ctx* ctx, double* jac /* nnz_J */);
int pk_set_host_option(pk_ctx* ctx, const char* name, int value);
/* layout of the Jacobian the host shim serves (pk_fetch(3), pk_callback_x(3), the J part of a landing block): 0 the reference's
 * triplets (default), 1 the compact layout of pk_eval_jacc (a landing block is then [J compact (nnz_Jc) | grad f | g]) */
int pk_set_jacobian_layout(pk_ctx* ctx, int compact);
int pk_result_location(pk_ctx* ctx, int what /* 0..4 */, double** ptr);
int pk_set_host_mode(pk_ctx* ctx, int prefetch, int host_direct);
int pk_invalidate_x(pk_ctx* ctx);
/* Pinned (page-locked) result buffers owned by the context: what = 0 f, 1 grad, 2 g, 3 jac, 4 hess.  The default
 * landing place of the results; reused by the next iterate. */
int pk_host_buffer(pk_ctx* ctx, int what, double** ptr, int64_t* count);
/* Pinned, device-visible host memory that is NOT tied to a context (result arrays handed to a solver may outlive the
 * evaluator): DMA targets at full PCIe rate.  pk_last_error(NULL) holds the message of a failure. */
int pk_host_alloc(size_t bytes, void** out);
int pk_host_free(void* p);

/* Compact Hessian of the Lagrangian: the reference repeats every dynamics entry for
 * each nonzero of the integration matrix (phasebase.py:923-928,1280-1285; 10-20x duplication); here lambda is
 * contracted first (mu = I^T lambda) and entries of a node with equal (row, col) are summed, so one value per
 * distinct position is produced.  Layout: pockit_amd.transcription.SystemPlan.hessc_row/col. */
int pk_eval_hessc(pk_ctx* ctx, const double* x, const double* lambda, double sigma, double* vals /* nnz_Hc */);
/* Compact Jacobian (the other half of SURVEY.md 8(f) rank 1): the reference repeats every derivative entry of a dynamics
 * function for each nonzero of the integration matrix (phasebase.py:885-887,1120-1124); where the entry's column is the
 * same on every node (t_0, t_f, a static parameter) that is K triplets on one (row, column).  pk_jacc contracts such
 * entries with the integration block first -- one value per defect row -- and sums scalar items that meet on one
 * position; entries with a per-node column keep the reference's form.  Layout: SystemPlan.jacc_row/col; scatter-added, the
 * triplets give the matrix SystemBase.jacobian (systembase.py:676-693) assembles to. */
int pk_eval_jacc(pk_ctx* ctx, const double* x, double* vals /* nnz_Jc */);
int pk_eval_jacc_dev(pk_ctx* ctx, const double* d_x, double* d_vals, void* stream);
int pk_eval_hessc_dev(pk_ctx* ctx, const double* d_x, const double* d_lambda, double sigma, double* d_vals,
                      void* stream);

/* Mesh error estimation (reference: phasebase.py:1339-1372
 * _error_estimation_data_continuous, called by check_continuous / refine_continuous, phasebase.py:1374-1437,
 * 1522-1617): every mesh interval is re-collocated with one more point; the kernel (pk_err, one wavefront per
 * group of intervals) interpolates states/controls to the augmented nodes, evaluates the dynamics there and returns both
 * sides of the integral-form collocation equation,  T = T_aug x  and  I = dt (I_aug d/2) f, per phase as
 * [n_x][rows] (rows = sum_j (K_j + 1) for LGR, sum_j K_j for LGL).  The per-interval comparison and the
 * hp-refinement decision are host logic (pockit_amd/refine.py).
 * ``intervals``: one PkErrIv (csrc/pk_abi.h) per mesh interval; ``groups``: (first record, count) pairs -- the run of
 * consecutive intervals of one phase and one K that ONE wavefront handles (K + 1 lanes per interval), padded per phase to a
 * multiple of 4 groups with count 0;
 * ``tables``: the interpolation / translation / integration blocks they index; ``n_out``: doubles per output. */
int pk_set_mesh_error_tables(pk_ctx* ctx, const void* intervals, int32_t n_intervals, const int32_t* groups,
                             int32_t n_groups, const double* tables, int64_t n_tables, int64_t n_out);
int pk_eval_mesh_error(pk_ctx* ctx, const double* x, double* T /* n_out */, double* I /* n_out */);
int pk_eval_mesh_error_dev(pk_ctx* ctx, const double* d_x, double* d_T, double* d_I, void* stream);

/* Device-resident CSR hand-off (the reference hands host triplets to IPOPT,
 * optimizer/ipopt.py:41-53).  The triplet values of J (which = 0) or H (which = 1, lower triangle) are gathered
 * into CSR order on the device, repeated (row, col) entries summed in triplet order: the matrices can feed a
 * GPU KKT solve without crossing PCIe.  ``perm[q]``: triplet index of the q-th entry in (row, col) order;
 * ``seg[p] .. seg[p+1]``: the run of q belonging to CSR entry p (NULL when no entry repeats).  The CSR
 * structure itself (indptr, indices) is host data: pockit_amd/csr.py.  which = 2 maps the COMPACT Hessian values
 * (pk_eval_hessc, one value per distinct entry) onto the same CSR entries: when it is set, pk_eval_hess_csr(_dev) evaluate
 * the compact form and permute instead of writing and re-adding every repeated triplet.  which = 3 does the same for the
 * Jacobian with the compact values of pk_eval_jacc (its few repeated positions are summed by the gather). */
int pk_set_csr_map(pk_ctx* ctx, int which, const int32_t* seg /* n_unique + 1 or NULL */, const int32_t* perm,
                   int64_t n_unique, int64_t n_triplets);
int pk_gather_csr_dev(pk_ctx* ctx, int which, const double* d_triplets, double* d_csr, void* stream);
int pk_eval_jac_csr_dev(pk_ctx* ctx, const double* d_x, double* d_csr, void* stream);
int pk_eval_hess_csr_dev(pk_ctx* ctx, const double* d_x, const double* d_lambda, double sigma, double* d_csr,
                         void* stream);
int pk_eval_jac_csr(pk_ctx* ctx, const double* x, double* vals /* n_unique */);
int pk_eval_hess_csr(pk_ctx* ctx, const double* x, const double* lambda, double sigma, double* vals);

/* device-pointer API: enqueue on ``stream`` (hipStream_t, NULL = context stream), no sync */
int pk_eval_f_dev(pk_ctx* ctx, const double* d_x, double* d_f, void* stream);
int pk_eval_grad_dev(pk_ctx* ctx, const double* d_x, double* d_grad, void* stream);
int pk_eval_g_dev(pk_ctx* ctx, const double* d_x, double* d_g, void* stream);
int pk_eval_jac_dev(pk_ctx* ctx, const double* d_x, double* d_vals, void* stream);
int pk_eval_hess_dev(pk_ctx* ctx, const double* d_x, const double* d_lambda, double sigma, double* d_vals,
                     void* stream);
/* the four x-only outputs of one iterate (what a line search's trial point needs, what the host shim runs on a new x):
 * the fused x-kernel + the one-workgroup reduction, two launches */
int pk_eval_xpart_dev(pk_ctx* ctx, const double* d_x, double* d_f, double* d_grad, double* d_g, double* d_jac, void* stream);
/* one NLP-callback cycle f, grad f, g, J, H on the same x (IPOPT's per-iteration pattern), as ONE launch
 * (pk_cycle: the workgroups of the fused x-kernel -- each node evaluated once for f, grad f, g, J -- and of the
 * Hessian kernel side by side, plus a finalize workgroup fed by the same launch).  One cycle may be in flight
 * per context at a time (the launch owns the context's hand-off slots). */
int pk_eval_cycle_dev(pk_ctx* ctx, const double* d_x, const double* d_lambda, double sigma, double* d_f,
                      double* d_grad, double* d_g, double* d_jac, double* d_hess, void* stream);
/* `count` back-to-back cycles on the same buffers, enqueued by the library (a solver written against the C ABI launches from
 * compiled code; bench.py's timed batches go through this so that no interpreter loop paces the stream).  xchg = 1: every
 * cycle is followed by pk_exchange_sums_dev(d_x, d_xgrad, d_f) -- the two-launch form of a sharded cycle.  No reference
 * counterpart (the reference's callbacks are synchronous NumPy calls, systembase.py:602-835). */
int pk_eval_cycle_dev_repeat(pk_ctx* ctx, const double* d_x, const double* d_lambda, double sigma, double* d_f, double* d_grad,
                             double* d_g, double* d_jac, double* d_hess, void* stream, int count, int xchg, double* d_xgrad);
/* Replay the fused cycle from a cached hipGraph while its pointers, sigma and stream do not change (a solver's
 * steady state); any change re-captures.  Off by default. */
int pk_set_cycle_graph(pk_ctx* ctx, int enable);
/* single_launch = 1 (default): pk_cycle; 0: the two-launch form, pk_xall then pk_hess, which also reduces */
int pk_set_cycle_mode(pk_ctx* ctx, int single_launch);
/* Which layouts pk_eval_cycle_dev[_repeat] writes: 0 = the reference's triplet lists (systembase.py:676-693, 820-835), 1 = the
 * compact layouts of pk_set_problem (nnz_Jc / nnz_Hc values: pk_eval_jacc / pk_eval_hessc).  A compact
 * cycle stays ONE launch: the compact kernels' tile code runs in the Jacobian / Hessian roles of pk_cycle.  Error 69 for a
 * model whose system functions are nonlinear in the integrals or a mesh with an interval of more than 64 points. */
int pk_set_cycle_layout(pk_ctx* ctx, int jac_compact, int hess_compact);
int pk_sync(pk_ctx* ctx, void* stream);
int pk_wait_idle(pk_ctx* ctx, void* stream);   /* the same by polling the stream (returns a few microseconds earlier) */

/* Mesh-interval sharding across GPUs (one context per GPU, each holding its shard of the tiles):
 * ``secondary`` shards skip the boundary-node / system-level work (done once, on the primary);
 * with ``external_prepass`` the callbacks do not run the integral pre-pass themselves: the caller
 * runs pk_eval_integrals_dev, sums ``d_integrals`` (n_I doubles, caller-owned) across shards
 * (RCCL all-reduce) and only then calls the callbacks / pk_eval_f_from_integrals_dev. */
int pk_set_shard(pk_ctx* ctx, int secondary, int external_prepass, double* d_integrals);
int pk_eval_integrals_dev(pk_ctx* ctx, const double* d_x, void* stream);
/* models nonlinear in the integrals (outer-product Hessian blocks, easyderiv.py:323-459) as shards: pk_eval_hess_dev leaves
 * the quadrature-weighted gradient entries of the integrals of THIS shard's nodes in the auxiliary buffer (pk_aux_buffer;
 * entries of other shards' nodes stay zero); the caller sums the buffers over the ranks into a buffer of its own and the
 * primary rank forms the blocks from the sum with pk_eval_outer_dev (into the Hessian values, reference positions). */
int pk_aux_buffer(pk_ctx* ctx, double** d_ptr, int64_t* count);
int pk_eval_outer_dev(pk_ctx* ctx, const double* d_aux_sum, double* d_vals /* nnz_H */, void* stream);
int pk_eval_f_from_integrals_dev(pk_ctx* ctx, const double* d_x, double* d_f, void* stream);

/* Sharded cycles without a collective in the data path.  Every rank leaves its shard's slices of grad f / g / J / H in
 * its own HBM (pk_eval_cycle_dev on its tiles); what couples the shards is the handful of sums over all nodes -- the
 * integrals (-> f) and the gradient entries of t0 / tf / static parameters.  pk_exchange_sums_dev posts this rank's
 * partial vector into every peer's mailbox (peer-mapped fine-grained device memory: pk_device_alloc + pk_ipc_export on
 * the owner, pk_ipc_open on the peers), waits for theirs and adds them in rank order inside ONE one-workgroup launch.
 * pk_set_shared_grad_target redirects a shard's partial sums of the shared gradient slots (used when its gradient
 * output points at another GPU's buffer: the reassembly of the triplets on one GPU by direct peer stores).
 * pk_copy_runs_dev is the pack / unpack pass of the RCCL gather / all-gather forms of the reassembly (A/B). */
int pk_device_alloc(pk_ctx* ctx, size_t bytes, int finegrained, void** out);
int pk_device_free(pk_ctx* ctx, void* p);
int pk_ipc_export(pk_ctx* ctx, void* dptr, void* handle64 /* 64 bytes out */);
int pk_ipc_open(pk_ctx* ctx, const void* handle64, void** out);
int pk_ipc_close(pk_ctx* ctx, void* p);
int pk_set_shared_grad_target(pk_ctx* ctx, double* d_grad_shared);
/* Host-landed sharded cycle (every GPU lands its slices in ONE host array over its own PCIe link): a host
 * region several processes map (shared memory) is page-locked and made device-visible in every process
 * (pk_host_register), the ranks' run-copy kernels (pk_copy_runs_dev) store their owned runs straight into it;
 * pk_copy_dev is an asynchronous copy between any two device-visible addresses. */
int pk_host_register(pk_ctx* ctx, void* p, size_t bytes, void** dev_ptr);
int pk_host_unregister(pk_ctx* ctx, void* p);
int pk_copy_dev(pk_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream);
int pk_set_exchange(pk_ctx* ctx, int world, int rank, const void* d_boxes, const int32_t* d_idx, int n_sh, int stride);
/* pk_set_exchange: every rank's mailbox holds 2 * world * stride words + 16 state words (zeroed here on this rank: the caller
 * puts a barrier between the set-up and the first cycle).  pk_exchange_status: cycles exchanged so far and how many of them
 * timed out waiting for a peer (then this rank's sums read NaN). */
int pk_exchange_status(pk_ctx* ctx, void* stream, int64_t* cycles, int64_t* timed_out);
int pk_exchange_sums_dev(pk_ctx* ctx, const double* d_x, double* d_grad, double* d_f, int epoch /* <= 0: counted on the device */,
                         int write_f, void* stream);
/* 1: the finalize workgroup of pk_eval_cycle_dev's launch exchanges the partial sums itself -- a sharded cycle is ONE
 * launch per GPU; 0 (default): pk_exchange_sums_dev is a launch of its own behind it. */
int pk_set_exchange_inline(pk_ctx* ctx, int enable);
int pk_copy_runs_dev(pk_ctx* ctx, const int64_t* d_table, int n_chunks, const double* d_src, double* d_dst, void* stream);
/* a progress mark: *d_dst = value once everything enqueued before it on the stream has finished (d_dst: device address of an
 * 8-byte aligned word, e.g. inside a segment registered with pk_host_register -- the host-landed sharded cycle lets rank 0
 * poll such words instead of waiting for the other processes to notice that their GPU has finished) */
int pk_store_word_dev(pk_ctx* ctx, void* d_dst, int64_t value, void* stream);

/* HIP-event timing of the individual kernels on the launch stream.
 * kernel ids: 0 pk_int, 1 pk_fin, 2 pk_g, 3 pk_grad, 4 pk_jac, 5 pk_hess, 6 pk_xall, 7 pk_aux, 8 pk_outer,
 * 9 pk_hessc, 10 pk_err, 11 pk_csr, 12 pk_cycle, 13 pk_xchg, 14 pk_runs, 15 pk_jacc.  pk_profile_sampling(n): only every n-th launch of a selected kernel is timed (a timed
 * launch costs ~2-3 us more than a plain one, so timing every launch slows the loop being measured). */
int pk_profile(pk_ctx* ctx, int kernel_mask /* bit k: time kernel k; 0 = off */);
int pk_profile_sampling(pk_ctx* ctx, int period);
/* Developer tracing: with a model generated under POCKIT_AMD_TRACE=1 the waves of pk_cycle / pk_xall store the
 * constant-rate device clock (s_memrealtime) at up to 16 checkpoints of their record.  The first call arms the
 * buffer; later calls copy the [3 n_tiles + 3][16] marks out (records: [tile][values, Jacobian, Hessian wave], then
 * pk_cycle's boundary-J, boundary-H and finalize workgroups) and clear the buffer. */
int pk_trace_read(pk_ctx* ctx, uint64_t* out, int64_t count);
int pk_profile_read(pk_ctx* ctx, int kernel_id, int64_t* launches, double* total_ms);
const char* pk_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* POCKIT_HIP_H */
