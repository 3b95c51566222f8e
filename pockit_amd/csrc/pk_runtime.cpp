// pk_runtime.cpp -- host runtime behind the C ABI of include/pockit_hip.h.
//
// Owns: the HIP context objects of one GPU (stream, loaded code object, kernel handles), the
// device copies of the per-(model, mesh) tables, and device work buffers (x, lambda, outputs,
// integrals, per-tile partial sums).  It launches the kernels of the generated code object
// (pockit_amd/codegen.py + csrc/pk_kernels.hip.h) in the order each NLP callback needs:
//
//   eval_f     pk_int, pk_fin(integrals, f)
//   eval_grad  [pk_int, pk_fin(integrals)]?  pk_grad, pk_fin(gradient slots)
//   eval_g     [pk_int, pk_fin(integrals)]?  pk_g
//   eval_jac   [pk_int, pk_fin(integrals)]?  pk_jac
//   eval_hess  [pk_int, pk_fin(integrals)]?  pk_hess
//   cycle      pk_cycle: ONE launch holding pk_xall's workgroups (f partials, grad f, g, J from one node
//              evaluation), pk_hess's workgroups and a finalize workgroup that receives the partial sums of
//              the same launch through hand-off slots (integrals, f, shared gradient slots);
//              pk_set_cycle_mode(0) selects the older two-launch form pk_xall, pk_hess(+ reductions)
// ("?" = only when a system-level function is nonlinear in the integrals, pk_model_desc.prepass_*).
//
// There is no CPU evaluation path: every entry point fails with an error code when no device /
// code object / problem is present.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <climits>
#include <cstddef>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pockit_hip.h"
#define PK_MAX_PHASES 128     // (= PK_HOST_MAX_PHASES: the host-side PkArgs holds the most a code object may ask for)
#include "pk_abi.h"

namespace {

enum { K_INT = 0, K_FIN, K_G, K_GRAD, K_JAC, K_HESS, K_XALL, K_AUX, K_OUTER, K_HESSC, K_ERR, K_CSR, K_CYCLE, K_XCHG, K_RUNS, K_JACC,
       K_CYCLEC, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"pk_int", "pk_fin", "pk_g", "pk_grad", "pk_jac", "pk_hess", "pk_xall",
                                           "pk_aux", "pk_outer", "pk_hessc", "pk_err", "pk_csr", "pk_cycle", "pk_xchg", "pk_runs",
                                           "pk_jacc", "pk_cyclec"};
enum { F_WRITE_F = 1, F_SECONDARY = 2, F_FIN_INT = 8, F_FIN_GRAD = 16, F_SPLIT = 32, F_XCHG = 64, F_NO_HESS = 128,
       F_COMPACT_H = 256, F_COMPACT_J = 512 };

thread_local std::string g_create_error;

struct EventPair {
  hipEvent_t a, b;
};

}  // namespace

struct pk_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipModule_t module = nullptr;
  hipFunction_t fn[K_COUNT] = {};
  bool have_model = false, have_problem = false;
  int shard_flags = 0;          // OR-ed into PkArgs.flags (bit 1: secondary shard)
  bool external_prepass = false; // sharded mode: the caller all-reduces the integrals itself
  double* ext_I = nullptr;      // caller-owned integral buffer (sharded mode)
  double* gshared = nullptr;    // pk_set_shared_grad_target: where the shared gradient slots go (NULL: the gradient itself)
  // pk_set_exchange: peer-mapped mailboxes of the partial-sum exchange (pk_xchg)
  const unsigned long long* const* xc_box = nullptr;
  const int32_t* xc_idx = nullptr;
  int32_t xc_world = 0, xc_rank = 0, xc_nsh = 0, xc_stride = 0;
  unsigned long long* xc_own = nullptr;   // this rank's own mailbox (host copy of the pointer: state block, pk_exchange_status)
  bool xc_inline = false;       // pk_cycle's finalize workgroup exchanges the partial sums itself (pk_set_exchange_inline)
  bool split_xall = false;      // pk_xall with two waves per tile (values / Jacobian), see pk_set_problem
  int cycle_mode = 1;           // 1: single-launch pk_cycle; 0: pk_xall + pk_hess (pk_set_cycle_mode)
  bool has_big = false;         // the mesh has intervals with more than 64 points (one workgroup each, PK_BIG code objects)
  // staging rows of intervals with more than 256 points (they do not fit the workgroup's LDS rows): slots in device memory
  double *d_big_stage = nullptr, *d_err_stage = nullptr;
  int32_t big_row = 0, big_slot = 0, err_row = 0, err_slot = 0;
  unsigned long long *d_cpart = nullptr, *d_cpart2 = nullptr;   // pk_cycle's hand-off slots (PK_EMPTY between launches)
  size_t cpart_slots = 0;
  unsigned long long status_seen[2] = {0, 0};   // PkArgs.status as of the last check (handoff_check)
  int poll_limit = 0;                           // > 0: poll rounds before a hand-off gives up ("poll_limit" host option; tests)
  unsigned profile_mask = 0;
  unsigned profile_period = 1;  // time every n-th launch of a selected kernel
  unsigned profile_seen[K_COUNT] = {};
  int debug_flags = 0;          // diagnostic kernel switches (POCKIT_AMD_DEBUG_FLAGS), never set in production
  pk_model_desc md{};
  // problem
  int32_t n = 0, m = 0, n_sys = 0, n_s = 0, l_s = 0, n_phase = 0, n_tiles = 0;
  int64_t nnz_J = 0, nnz_H = 0;
  int32_t n_items_jac = 0, n_items_hess = 0, n_items_aux = 0, n_outer = 0, n_aux = 0, gz_off = 0, n_gz = 0;
  int32_t n_items_hessc = 0, n_items_jacc = 0;
  int64_t nnz_Hc = 0, nnz_Jc = 0;
  void* d_items_jacc = nullptr;
  double* d_Jc = nullptr;
  void *d_phases = nullptr, *d_tiles = nullptr, *d_kinds = nullptr, *d_items_jac = nullptr, *d_items_hess = nullptr,
       *d_items_aux = nullptr, *d_outer = nullptr, *d_items_hessc = nullptr;
  double *d_aux = nullptr, *d_Hc = nullptr;
  // cached hipGraph of the fused callback cycle (pk_set_cycle_graph)
  bool use_graph = false;
  hipGraphExec_t cyc_exec = nullptr;
  hipGraphExec_t rep_exec = nullptr;      // pk_eval_cycle_dev_repeat: a batch of rep_count cycles as one graph
  int rep_count = 0;
  struct CycleKey {
    const void *x, *lam, *f, *grad, *g, *jac, *hess;
    double sigma;
    hipStream_t st;
    bool operator==(const CycleKey& o) const {
      return x == o.x && lam == o.lam && f == o.f && grad == o.grad && g == o.g && jac == o.jac && hess == o.hess &&
             sigma == o.sigma && st == o.st;
    }
  } cyc_key{}, rep_key{};
  unsigned long long* d_trace = nullptr;   // developer tracing buffer, [n_tiles][16]
  // triplet -> CSR maps (pk_set_csr_map): [0] Jacobian, [1] Hessian of the Lagrangian (lower triangle)
  struct CsrMap {
    int32_t *d_seg = nullptr, *d_perm = nullptr;
    double* d_vals = nullptr;
    int64_t n_unique = 0, n_triplets = 0;
  } csr[4];      // + [2]: compact Hessian values -> the same CSR entries (a pure permutation: one value per entry)
                 // + [3]: compact Jacobian values -> the CSR entries of J (the few repeated positions summed)
  // mesh error estimation (pk_set_mesh_error_tables)
  void* d_erriv = nullptr;
  int32_t* d_errgrp = nullptr;     // (first record, count) per wavefront of pk_err
  double *d_errdb = nullptr, *d_errT = nullptr, *d_errI = nullptr;
  int32_t n_erriv = 0;
  int64_t n_err_out = 0;
  int32_t* d_ib = nullptr;
  double* d_db = nullptr;
  int64_t* d_lb = nullptr;
  // work buffers
  double* h_Hc = nullptr;      // pinned landing place of the compact Hessian (pk_eval_hessc_prepared), allocated on first use
  double *d_x = nullptr, *d_lam = nullptr, *d_f = nullptr, *d_grad = nullptr, *d_g = nullptr, *d_J = nullptr,
         *d_H = nullptr, *d_I = nullptr, *d_partial = nullptr, *d_partial2 = nullptr;
  std::vector<PkPhase> h_phases;
  std::vector<EventPair> free_events;
  std::vector<int32_t> jac_row, jac_col, hess_row, hess_col;
  // pinned host staging of the host shim: x and lambda are double-buffered (the upload of iterate k + 1 does not wait
  // for anything of iterate k), results land in h_out (f, grad, g, J, H) or in caller-supplied pinned targets
  double *h_xs[2] = {nullptr, nullptr}, *h_lams[2] = {nullptr, nullptr};
  hipEvent_t ev_xs[2] = {nullptr, nullptr}, ev_lams[2] = {nullptr, nullptr};   // upload k of the buffer has left it
  int xbuf = 0, lambuf = 0;
  double* h_x = nullptr;                   // the staging buffer holding the x of the last pk_prepare_x (pk_same_x)
  bool x_valid = false;
  bool lam_staged = false;                 // pk_stage_lambda ran, pk_eval_hess_prepared has not consumed it yet
  double* h_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  double* target[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // pk_set_result_targets (NULL: h_out[k])
  bool target_visible[5] = {true, true, true, true, true};             // the device can store into target[k] itself
  double* landed[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // where result k of the current iterate went
  hipEvent_t ev_out[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool enq[5] = {false, false, false, false, false};                   // copy of result k is enqueued
  bool done[5] = {false, false, false, false, false};                  // ... and known to have landed
  bool stored_direct[5] = {false, false, false, false, false};         // the kernel stored result k into its landing place
  int prefetch = 1;            // 1: every x-only result is copied out right behind the kernel; 0: on first request
  int adaptive_prefetch = 1;   // ... but grad f and J only while the solver keeps asking for them: an iterate whose Jacobian was
                               // never asked for was a rejected trial point of a line search (f and g only), and the copy of
                               // its J (122 us of link time at 12k nodes) stood in the way of the next trial point's upload;
                               // the new x behind such an iterate gets grad f / J on request
  bool cur_J_asked = true;     // grad f or J of the prepared iterate has been asked for
  int host_direct = 0;         // 1: the kernels store into the (pinned, device-visible) host targets themselves
  // Host-shim tuning (pk_set_host_option; defaults = what measured fastest on MI355X, tools/dma_probe.cpp):
  int spin_wait = 1;           // results are awaited by polling (the event's state / f's own pinned word), not hipEventSynchronize
  int lambda_direct = 1;       // the Hessian kernel of the prepared protocol reads the multipliers from the pinned staging
                               // buffer itself (one pass over PCIe inside the kernel) instead of an upload in front of it;
                               // applied up to 2 MB of multipliers (12k nodes: -8 us; at 3.2 MB the chunk-pipelined upload
                               // wins by 16 us: the staging memcpy then overlaps the link)
  int chunk_upload = 1;        // staging of large inputs is pipelined with their upload in a few chunks
  int kernel_upload = 1;       // x (and lambda) go up through a copy kernel on the compute queue instead of the DMA engine: the
                               // kernel behind it then starts without a cross-engine hand-off (~10 us on the path to f)
  int kernel_download = 8;     // results of up to this many MiB per piece come down through a copy kernel instead of the DMA
                               // engine (0: never): no cross-engine hand-off behind the kernel that produced them (~10 us per
                               // copy), but 51 instead of 56 GB/s on the link -- the DMA engine wins from ~5 MB on
  int split_copy = 1;          // grad f | g leave in a copy of their own in front of J (+1 DMA), with an event behind it: the
                               // gradient and constraints callbacks return while J is still on the link, and the bitwise
                               // compares of x they and the Jacobian callback start with are hidden behind that copy
  int small_direct = 1;        // small systems are bound by the number of launches, not by bytes: a kernel reads an x of at most
                               // 128 KB from its pinned staging buffer (no upload launch) and stores x-results of at most 1 MB
                               // straight into their pinned landing places (no copy launches) -- LQR 10x10: 53 -> us per iterate
  int small_x_kb = 128;            // (the x threshold of small_direct, in KB: an A/B knob)
  const double* x_src = nullptr;   // where the kernels read the prepared x: d_x, or (small_direct) the pinned staging buffer
  int xpart_single = 1;        // pk_eval_xpart_dev as ONE launch (pk_cycle without its Hessian role) instead of pk_xall + pk_fin
  bool separate_x = false;     // the five callbacks one after the other through the STAND-ALONE kernels (pk_int + pk_fin, pk_grad, pk_g,
                               // pk_jac, pk_hess), as for a model that needs the integrals first: what pockit_amd.Evaluator.checked falls back
                               // to when a code object's fused kernel fails its self-check (round 5)
  int hess_direct = 1;         // the Hessian kernel stores into the pinned landing place itself when H is small enough for the
                               // copy kernel (kernel_download): no launch behind it, its reads of lambda and its stores share
                               // the link in both directions (12k nodes: 97 -> 93 us; at 83 MB the copy is faster)
  int speculative_hess = 1;    // pk_callback_hess launches on the prepared x BEFORE comparing x with it (the compare then runs
                               // while the GPU works; a different x -- rare -- discards the launch and starts over)
  bool target_pinned[5] = {false, false, false, false, false};   // target[k] is pinned memory by contract (landing blocks)
  // reuse guard of the staging buffers without events: every enqueue takes a sequence number; an idle stream seen by the host
  // (wait_result) retires all numbers issued so far
  uint64_t op_seq = 0, idle_seq = 0;
  // mark_wait: behind the last result copy of a batch a one-word kernel stores a counter into pinned memory (h_out[0][4]) and
  // the waiting callback polls that word instead of the stream's state (the runtime's query answers several microseconds
  // after the word is there; the stream is still asked now and then, so a failed launch does not hang the caller)
  int mark_wait = 1;
  unsigned long long mark_val = 0;     // value of the last mark enqueued
  uint64_t mark_op_seq = 0;            // op_seq when it was enqueued: everything up to it has finished once the mark is seen
  bool mark_pending = false;           // the last thing enqueued for the results is a mark nobody has waited for yet
  uint64_t xs_seq[2] = {0, 0}, lams_seq[2] = {0, 0};
  hipEvent_t ev_early = nullptr;     // behind the grad f | g copy of the current iterate (split_copy)
  bool early_valid = false;
  const double* lam_src = nullptr;   // where the staged multipliers are read from (d_lam, or the pinned staging buffer)
  int ev_of[5] = {0, 1, 2, 3, 4};    // the event that covers result k of the current iterate (one event per batch of copies)
  // Pieces [start, stop) of the Jacobian values that CHANGE with x.  Default: everything.  pk_set_jac_constant_runs takes
  // x-independent runs (the +-1 translation entries of phasebase.py:1071-1081 are 19 % of J at 12k nodes) out of the
  // per-iterate copy: they are put into a landing array once (pk_fill_jac_constants) and never cross PCIe again.
  std::vector<std::pair<int64_t, int64_t>> jruns, jconst;      // (of the layout the shim serves; the other layout's are parked)
  std::vector<std::pair<int64_t, int64_t>> jruns_other, jconst_other;
  bool jac_compact = false;    // the host shim's Jacobian callback serves the compact layout (pk_set_jacobian_layout)
  int cycle_layout = 0;        // what pk_eval_cycle_dev writes: bit 0 compact Jacobian, bit 1 compact Hessian (pk_set_cycle_layout)
  bool target_filled = false;  // the caller's J landing array (target[3]) already holds the constant runs (pk_callback_x blocks)
  bool jac_filled = false;     // ... and so does the landing place of the CURRENT iterate: its copy skips them
  // profiling
  bool profiling = false;
  std::vector<EventPair> pending[K_COUNT];
  int64_t launches[K_COUNT] = {};
  double total_ms[K_COUNT] = {};
  std::string error;
};

namespace {

int fail(pk_ctx* c, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->error = buf; else g_create_error = buf;
  return code;
}

#define PK_HIP(c, call)                                                                                  \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) return fail((c), 100 + (int)e_, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

void drop_cycle_graph(pk_ctx* c) {
  if (c->cyc_exec) { (void)hipGraphExecDestroy(c->cyc_exec); c->cyc_exec = nullptr; }
  if (c->rep_exec) { (void)hipGraphExecDestroy(c->rep_exec); c->rep_exec = nullptr; }
}

template <class T>
void release(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

void free_problem(pk_ctx* c) {
  release(c->d_items_jacc); release(c->d_Jc);
  release(c->d_phases); release(c->d_tiles); release(c->d_kinds); release(c->d_items_jac); release(c->d_items_hess); release(c->d_items_aux); release(c->d_outer); release(c->d_aux); release(c->d_items_hessc); release(c->d_Hc);
  release(c->d_erriv); release(c->d_errgrp); release(c->d_errdb); release(c->d_errT); release(c->d_errI);
  release(c->d_big_stage); release(c->d_err_stage);
  c->n_erriv = 0; c->n_err_out = 0;
  drop_cycle_graph(c);
  release(c->d_trace);
  for (auto& m : c->csr) { release(m.d_seg); release(m.d_perm); release(m.d_vals); m.n_unique = m.n_triplets = 0; }
  release(c->d_ib); release(c->d_db); release(c->d_lb);
  c->d_g = c->d_grad = nullptr;   // (interior pointers of the d_J allocation: one block [J | grad f | g])
  c->jruns.clear(); c->jconst.clear(); c->jruns_other.clear(); c->jconst_other.clear();
  c->jac_compact = false;
  c->lam_src = nullptr;
  if (c->h_Hc) { (void)hipHostFree(c->h_Hc); c->h_Hc = nullptr; }
  release(c->d_x); release(c->d_lam); release(c->d_f); release(c->d_J);
  release(c->d_H); release(c->d_I); release(c->d_partial); release(c->d_partial2);
  release(c->d_cpart); release(c->d_cpart2);
  for (int b = 0; b < 2; ++b) {
    if (c->h_xs[b]) (void)hipHostFree(c->h_xs[b]);
    if (c->h_lams[b]) (void)hipHostFree(c->h_lams[b]);
    if (c->ev_xs[b]) (void)hipEventDestroy(c->ev_xs[b]);
    if (c->ev_lams[b]) (void)hipEventDestroy(c->ev_lams[b]);
    c->h_xs[b] = c->h_lams[b] = nullptr;
    c->ev_xs[b] = c->ev_lams[b] = nullptr;
    c->xs_seq[b] = c->lams_seq[b] = 0;
  }
  if (c->ev_early) { (void)hipEventDestroy(c->ev_early); c->ev_early = nullptr; }
  c->early_valid = false;
  c->h_x = nullptr;
  c->x_valid = false;
  for (int k = 0; k < 5; ++k) {
    if (c->h_out[k] && k != 1 && k != 2) (void)hipHostFree(c->h_out[k]);      // (h_out[1], h_out[2] live inside h_out[3]'s block)
    if (c->ev_out[k]) (void)hipEventDestroy(c->ev_out[k]);
    c->h_out[k] = c->target[k] = c->landed[k] = nullptr;
    c->ev_out[k] = nullptr;
    c->enq[k] = false;
  }
  c->have_problem = false;
}

int upload(pk_ctx* c, void** dst, const void* src, size_t bytes) {
  *dst = nullptr;
  const size_t alloc = bytes ? bytes : 8;
  PK_HIP(c, hipMalloc(dst, alloc));
  if (bytes) PK_HIP(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}

int ready(pk_ctx* c) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!c->have_model) return fail(c, 2, "no model loaded (pk_load_model)");
  if (!c->have_problem) return fail(c, 3, "no problem set (pk_set_problem)");
  return 0;
}

size_t args_bytes(const pk_ctx* c);

PkArgs base_args(pk_ctx* c, const double* d_x, const double* d_lam, double sigma) {
  PkArgs A;      // (only the bytes the code object declares are filled and launched: the head and its phase records)
  std::memset(static_cast<void*>(&A), 0, args_bytes(c));
  A.x = d_x; A.lam = d_lam; A.sigma = sigma;
  A.phase = (const PkPhase*)c->d_phases; A.tile = (const PkTile*)c->d_tiles; A.kind = (const PkKind*)c->d_kinds;
  A.items = nullptr; A.ib = c->d_ib; A.db = c->d_db; A.lb = c->d_lb;
  A.Ibuf = c->ext_I ? c->ext_I : c->d_I; A.partial = c->d_partial; A.partial2 = c->d_partial2;
  A.cpart = c->d_cpart; A.cpart2 = c->d_cpart2; A.o_aux = c->d_aux; A.outer = (const PkOuter*)c->d_outer; A.n_outer = c->n_outer;
  A.n_tiles = c->n_tiles; A.n_items = 0; A.n_phase = c->n_phase; A.n = c->n;
  A.l_s = c->l_s; A.n_s = c->n_s; A.n_sys = c->n_sys; A.m = c->m;
  A.gz_off = c->gz_off; A.n_gz = c->n_gz; A.flags = c->shard_flags | c->debug_flags;
  for (size_t k = 0; k < c->h_phases.size(); ++k) A.ph[k] = c->h_phases[k];
  A.trace = c->d_trace;
  A.o_gshared = c->gshared;
  A.big_stage = c->d_big_stage; A.big_row = c->big_row; A.big_slot = c->big_slot;
  A.status = c->h_out[0] ? reinterpret_cast<unsigned long long*>(c->h_out[0] + 6) : nullptr;
  A.poll_limit = c->poll_limit;
  return A;
}

// bytes of PkArgs the loaded code object declares: the head and as many phase records as it was compiled for
size_t args_bytes(const pk_ctx* c) {
  return offsetof(PkArgs, ph) + sizeof(PkPhase) * (size_t)(c->md.max_phases > 0 ? c->md.max_phases : 8);
}

int launch_raw(pk_ctx* c, int k, void* args, size_t sz, unsigned grid, size_t lds_bytes, hipStream_t st) {
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  EventPair ev{};
  if (grid == 0) return 0;
  // the tile kernels that stage their pattern tables keep one table block per wave in front of the model's staging area
  if (k == K_G || k == K_JAC || k == K_HESS || k == K_XALL || k == K_CYCLE || k == K_CYCLEC || k == K_JACC || k == K_HESSC)
    lds_bytes += sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)(2 * c->md.tab_cap + 2 * PK_WAVE + c->md.tab_cap / 2);
  if ((k == K_CYCLE || k == K_CYCLEC) && c->xc_inline && c->xc_world > 1 && lds_bytes < sizeof(double) * 2 * 512)
    lds_bytes = sizeof(double) * 2 * 512;      // the finalize workgroup's exchange vectors (2 x PK_XC_CAP doubles)
  if (lds_bytes > 160 * 1024) return fail(c, 22, "%s needs %zu bytes of LDS per workgroup (> 160 KiB)", kKernelNames[k], lds_bytes);
  // every `profile_period`-th launch of a selected kernel is timed (the timed launch path costs ~2-3 us of host
  // and command-processor work, so timing all of them would slow the loop being measured)
  const bool timed = c->profiling && ((c->profile_mask >> k) & 1u) && (c->profile_seen[k]++ % c->profile_period == 0);
  if (timed) {
    // Timed launch: hipExtModuleLaunchKernel attaches the events to the dispatch packet itself, so
    // elapsed(a, b) is the kernel's own start->end on this stream (what rocprofv3 reports), without
    // the command-processor gaps a hipEventRecord pair around the launch would add.
    if (!c->free_events.empty()) {
      ev = c->free_events.back();
      c->free_events.pop_back();
    } else {
      PK_HIP(c, hipEventCreate(&ev.a));
      PK_HIP(c, hipEventCreate(&ev.b));
    }
    PK_HIP(c, hipExtModuleLaunchKernel(c->fn[k], grid * PK_BLOCK, 1, 1, PK_BLOCK, 1, 1, lds_bytes, st, nullptr, config,
                                       ev.a, ev.b, 0));
    c->pending[k].push_back(ev);
    return 0;
  }
  PK_HIP(c, hipModuleLaunchKernel(c->fn[k], grid, 1, 1, PK_BLOCK, 1, 1, (unsigned)lds_bytes, st, nullptr, config));
  return 0;
}

int launch(pk_ctx* c, int k, PkArgs& A, unsigned grid, size_t lds_bytes, hipStream_t st) {
  // pk_xall runs the values role of a WIDE phase with its dynamics passes inside the values wave: wrong f / grad / g for
  // some models and GPU memory faults (round 5, an open defect on that kernel's SGPR-spill path).
  // Every route to it -- the two-launch cycle, a profiled context, pk_set_option("xpart_single", 0), the x-part of a
  // sharded context with the in-launch exchange, integrals-first models with intervals of more than 64 points -- ends here.
  if (k == K_XALL && c->md.wide)
    return fail(c, 27, "pk_xall is not offered for a model with a wide phase (open defect of its sequential values role, DESIGN.md "
                       "section 11): use the one-launch cycle / the callbacks of an unprofiled context (the default)");
  return launch_raw(c, k, &A, args_bytes(c), grid, lds_bytes, st);
}

unsigned tile_blocks(const pk_ctx* c) { return (unsigned)((c->n_tiles + PK_WAVES_PER_BLOCK - 1) / PK_WAVES_PER_BLOCK); }
// pk_hess: edge and reduction workgroups + one workgroup per tile block -- per PASS of a block for a model evaluated in groups
// whose code object runs the passes as workgroups of their own (md.hess_subs, codegen.py)
unsigned hess_grid(const pk_ctx* c) { return tile_blocks(c) * (c->md.hess_subs > 0 ? (unsigned)c->md.hess_subs : 1u) + 2u; }

int prepass(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_f, bool write_f, hipStream_t st) {
  PkArgs A = base_args(c, d_x, d_lam, sigma);
  A.o_f = d_f;
  int rc = launch(c, K_INT, A, tile_blocks(c), 0, st);
  if (rc) return rc;
  A.flags |= F_FIN_INT | (write_f ? F_WRITE_F : 0);
  return launch(c, K_FIN, A, 1, 0, st);
}

// pk_xall's launch shape: one wave per tile, or -- split launch -- two waves (of two workgroups) per tile
unsigned xall_blocks(const pk_ctx* c) { return (c->split_xall ? 2u : 1u) * tile_blocks(c) + 1u; }
int xall_flags(const pk_ctx* c) { return c->split_xall ? F_SPLIT : 0; }

// the cycle as ONE launch (pk_cycle): [edge J | edge H | finalize | tile slots: x block(s) + Hessian block per group]
// (d_lam == NULL: the x-part alone -- the Hessian workgroups of the grid leave at once)
// layout: bit 0 -- d_jac receives the COMPACT Jacobian (the Jacobian role of the launch runs pk_jacc's tile code), bit 1 --
// d_hess receives the COMPACT Hessian (the Hessian workgroups run pk_hessc's); -1: what pk_set_cycle_layout chose
int enqueue_single_launch_cycle(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_f,
                                double* d_grad, double* d_g, double* d_jac, double* d_hess, hipStream_t st, int layout = -1) {
  if (layout < 0) layout = c->cycle_layout;
  PkArgs A = base_args(c, d_x, d_lam, sigma);
  if (!d_lam) A.flags |= F_NO_HESS;
  A.o_f = d_f; A.o_grad = d_grad; A.o_g = d_g; A.o_jac = d_jac; A.o_hess = d_hess;
  A.items = (const PkItem*)((layout & 1) ? c->d_items_jacc : c->d_items_jac);
  A.n_items = (layout & 1) ? c->n_items_jacc : c->n_items_jac;
  A.items2 = (const PkItem*)((layout & 2) ? c->d_items_hessc : c->d_items_hess);
  A.n_items2 = (layout & 2) ? c->n_items_hessc : c->n_items_hess;
  if (layout & 1) A.flags |= F_COMPACT_J;
  if (layout & 2) A.flags |= F_COMPACT_H;
  A.flags |= F_FIN_INT | F_WRITE_F | F_FIN_GRAD | xall_flags(c);
  if (c->xc_inline && c->xc_world > 1) {      // sharded: the sums over the ranks are exchanged inside this launch
    A.flags |= F_XCHG;
    A.xc_box = (unsigned long long* const*)c->xc_box; A.xc_idx = c->xc_idx;
    A.xc_world = c->xc_world; A.xc_rank = c->xc_rank; A.xc_epoch = 0; A.xc_nsh = c->xc_nsh; A.xc_stride = c->xc_stride;
  }      // (xc_epoch = 0: the cycle number is kept in device memory, so these arguments never change -> graph-replayable)
  size_t dbl = PK_WAVES_PER_BLOCK * (size_t)(c->md.lds_x > c->md.lds_h ? c->md.lds_x : c->md.lds_h);
  if (dbl < (size_t)c->md.ne_j) dbl = (size_t)c->md.ne_j;
  if (dbl < (size_t)c->md.ne_h) dbl = (size_t)c->md.ne_h;
  if (layout & 1) {      // (tile_jacc stages in the x-part's rows: lds_x >= lds_jc by construction, codegen.py)
    if (dbl < (size_t)c->md.ne_jc) dbl = (size_t)c->md.ne_jc;
  }
  if (layout & 2) {
    if (dbl < PK_WAVES_PER_BLOCK * (size_t)c->md.lds_g) dbl = PK_WAVES_PER_BLOCK * (size_t)c->md.lds_g;
    if (dbl < (size_t)c->md.ne_hc) dbl = (size_t)c->md.ne_hc;
  }
  // workgroups per tile block: [Jacobian | values | Hessian] (x-part split) or [x-part | Hessian]; a model evaluated in groups:
  // one per pass of the Jacobian / Hessian role beside the values workgroup (md.cycle_subs, codegen.py)
  const unsigned per_group = c->md.cycle_subs > 0 ? (unsigned)c->md.cycle_subs : (c->split_xall ? 3u : 2u);
  const unsigned grid = tile_blocks(c) * per_group + 3u;
  // pk_cycle's kernarg segment: the scalars a tile wave needs first (preloaded into SGPRs), then the PkArgs
  struct CycleArgs {
    const PkTile* tile;
    int32_t n_tiles, flags, grid, pad;
    PkArgs A;
  } K;
  static_assert(offsetof(CycleArgs, A) == PK_CYCLE_ARGS_OFFSET, "layout of pk_cycle's kernel arguments");
  K.tile = A.tile; K.n_tiles = A.n_tiles; K.flags = A.flags; K.grid = (int32_t)grid; K.pad = 0;
  std::memcpy(static_cast<void*>(&K.A), &A, args_bytes(c));
  // (the compact layouts: pk_cyclec, the same launch compiled with their roles -- a kernel of its own so that pk_cycle's
  //  register count stays what the reference layouts need)
  return launch_raw(c, layout ? K_CYCLEC : K_CYCLE, &K, offsetof(CycleArgs, A) + args_bytes(c), grid, sizeof(double) * dbl, st);
}

int enqueue_fused_cycle(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_f, double* d_grad,
                        double* d_g, double* d_jac, double* d_hess, hipStream_t st) {
  int rc;
  if (c->cycle_mode == 1) return enqueue_single_launch_cycle(c, d_x, d_lam, sigma, d_f, d_grad, d_g, d_jac, d_hess, st);
  if (c->cycle_layout) return fail(c, 69, "the two-launch cycle (pk_set_cycle_mode 0) writes the reference layouts only");
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_f = d_f; A.o_grad = d_grad; A.o_g = d_g; A.o_jac = d_jac;
  A.items = (const PkItem*)c->d_items_jac;
  A.n_items = c->n_items_jac;
  size_t lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_x;
  if (lds < sizeof(double) * (size_t)c->md.ne_j) lds = sizeof(double) * (size_t)c->md.ne_j;
  A.flags |= xall_flags(c);
  if ((rc = launch(c, K_XALL, A, xall_blocks(c), lds, st))) return rc;
  // pk_hess's boundary workgroup also performs pk_fin's reductions (f, shared gradient slots)
  PkArgs H = base_args(c, d_x, d_lam, sigma);
  H.o_f = d_f; H.o_grad = d_grad; H.o_hess = d_hess;
  H.items = (const PkItem*)c->d_items_hess;
  H.n_items = c->n_items_hess;
  H.flags |= F_FIN_INT | F_WRITE_F | F_FIN_GRAD | xall_flags(c);
  lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_h;
  if (lds < sizeof(double) * (size_t)c->md.ne_h) lds = sizeof(double) * (size_t)c->md.ne_h;
  return launch(c, K_HESS, H, hess_grid(c), lds, st);
}

hipStream_t pick(pk_ctx* c, void* stream) { return stream ? (hipStream_t)stream : c->stream; }

}  // namespace

// ---- helpers of the host shim (the "new x" protocol further down)
// Copy kernel of the host shim: n doubles between pinned host memory and device memory, 16 bytes per lane.  src and dst
// are congruent modulo 16 bytes (the caller checks), so at most one leading and one trailing double go alone.
__global__ void __launch_bounds__(256) pk_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  size_t head = ((uintptr_t)src >> 3) & 1;
  if (head > n) head = n;
  const size_t pairs = (n - head) >> 1;
  const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src + head);
  double2* __restrict__ d2 = reinterpret_cast<double2*>(dst + head);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) d2[i] = s2[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (head) dst[0] = src[0];
    if (head + 2 * pairs < n) dst[n - 1] = src[n - 1];
  }
}

// One 64-bit word stored behind everything enqueued before it on the stream (a progress mark in pinned host memory that
// the host, or another process mapping the same segment, polls: seen a few microseconds before an event would report).
__global__ void __launch_bounds__(64) pk_store_word_kernel(unsigned long long* dst, unsigned long long value) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *(volatile unsigned long long*)dst = value;
}

namespace {

// dst[0 .. n) = src[0 .. n) on the context's stream: the copy kernel when asked for and possible, else the DMA engine
int copy_async(pk_ctx* c, double* dst, const double* src, size_t n, hipMemcpyKind kind, bool by_kernel) {
  if (!n) return 0;
  if (by_kernel && !(((uintptr_t)dst ^ (uintptr_t)src) & 8)) {
    // (tools/copy_kernel_probe.cpp: device -> host does not care about the grid, 49-50 GB/s from 32 to 2048 workgroups;
    //  host -> device prefers FEW workgroups: 0.77 MB 28 us with 64, 32 us with 1024; 4.8 MB 101 vs 123 us)
    const size_t pairs = n / 2 + 1;
    unsigned grid = (unsigned)((pairs + 255) / 256);
    const unsigned cap = kind == hipMemcpyHostToDevice ? 64u : 512u;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(pk_copy_kernel, dim3(grid), dim3(256), 0, c->stream, src, dst, n);
    PK_HIP(c, hipGetLastError());
    return 0;
  }
  PK_HIP(c, hipMemcpyAsync(dst, src, sizeof(double) * n, kind, c->stream));
  return 0;
}

size_t result_count(const pk_ctx* c, int what) {
  const size_t cnt[5] = {1, (size_t)c->n, (size_t)c->m, (size_t)(c->jac_compact ? c->nnz_Jc : c->nnz_J), (size_t)c->nnz_H};
  return cnt[what];
}

double* device_result(pk_ctx* c, int what) {
  double* src[5] = {c->d_f, c->d_grad, c->d_g, c->jac_compact ? c->d_Jc : c->d_J, c->d_H};
  return src[what];
}

// Queue the copies of the results in `mask` (bit k: result k; 0 f, 1 grad f, 2 g, 3 J, 4 H) of the current iterate that are
// not on their way yet, and -- unless results are awaited by polling the stream, see wait_result -- ONE event behind them.  J, grad f and g are neighbours on the device ([J | grad | g], one
// allocation); where their landing places are neighbours in the same order (the context's own block, or one block of the
// caller's) the pieces are merged: the changing part of J, grad f and g leave in one DMA.  The pieces of J that never
// change (pk_set_jac_constant_runs) are not copied at all.  f needs no copy when the kernel stored it into its pinned
// landing place itself (a DMA of 8 bytes costs as much as one of 100 KB).
int enqueue_mark(pk_ctx* c) {
  if (!c->mark_wait || !c->spin_wait || !c->h_out[0]) return 0;
  unsigned long long* word = reinterpret_cast<unsigned long long*>(c->h_out[0] + 4);
  const unsigned long long value = ++c->mark_val;      // (a copy: the launch takes its arguments by value, now)
  c->mark_op_seq = c->op_seq;
  hipStream_t st = c->stream;
  hipLaunchKernelGGL(pk_store_word_kernel, dim3(1), dim3(64), 0, st, word, value);
  PK_HIP(c, hipGetLastError());
  c->mark_pending = true;
  return 0;
}

// A hand-off that gave up waiting (pk_cycle's finalize workgroup for the partial sums of its own launch; the exchange of the
// sums between the ranks) leaves NaN in f and in the gradient entries shared by all nodes -- indistinguishable, for a solver,
// from a model that evaluates to NaN (which the reference passes on unchecked, examples/_plotting.py:58-63, and so do we).
// The kernels count such events in two status words in pinned memory (PkArgs.status); every waiting entry point compares
// them with what it saw last and turns a change into error 97: the hand-off slots are put back to PK_EMPTY (a publisher that
// arrived after the give-up left its value behind), the staged iterate is dropped, the message says what happened.
int handoff_check(pk_ctx* c) {
  if (!c->h_out[0]) return 0;
  const volatile unsigned long long* st = reinterpret_cast<const volatile unsigned long long*>(c->h_out[0] + 6);
  const unsigned long long a = st[0], b = st[1];
  if (a == c->status_seen[0] && b == c->status_seen[1]) return 0;
  const unsigned long long da = a - c->status_seen[0], db = b - c->status_seen[1];
  // (rare path: the device, not only the context's stream -- pk_sync / pk_wait_idle come here for a caller's stream too, and
  //  cycles still in flight on it must not race with the reset of the hand-off slots below)
  (void)hipDeviceSynchronize();
  if (c->d_cpart && c->cpart_slots) {
    const std::vector<unsigned long long> empty(c->cpart_slots, (unsigned long long)PK_EMPTY);
    (void)hipMemcpy(c->d_cpart, empty.data(), sizeof(unsigned long long) * c->cpart_slots, hipMemcpyHostToDevice);
    (void)hipMemcpy(c->d_cpart2, empty.data(), sizeof(unsigned long long) * c->cpart_slots, hipMemcpyHostToDevice);
  }
  c->status_seen[0] = st[0];
  c->status_seen[1] = st[1];
  c->x_valid = false;
  c->lam_staged = false;
  if (da)
    return fail(c, 97, "pk_cycle: the finalize workgroup gave up waiting for %llu partial sum(s) of its own launch; f and the "
                       "gradient entries shared by all nodes of this iterate are NaN (hand-off slots reset)", da);
  return fail(c, 97, "%llu exchange(s) of the partial sums between the ranks gave up waiting for a peer; f and the shared "
                     "gradient entries of this iterate are NaN on this rank", db);
}

// everything enqueued for the results has finished: the pending mark has been stored, or (no mark) the stream is idle
int wait_results_landed_raw(pk_ctx* c) {
  hipError_t e;
  if (c->mark_pending) {
    const volatile unsigned long long* word = reinterpret_cast<const volatile unsigned long long*>(c->h_out[0] + 4);
    const unsigned long long want = c->mark_val;
    for (long spins = 1; *word < want; ++spins) {
      if ((spins & 0x3FFF) == 0) {          // now and then: has the stream finished (or failed) without storing the mark?
        e = hipStreamQuery(c->stream);
        if (e == hipSuccess) {
          if (*word < want) return fail(c, 65, "the progress mark was not stored by its kernel");
          break;
        }
        if (e != hipErrorNotReady) return fail(c, 100 + (int)e, "waiting for the results: %s", hipGetErrorString(e));
      }
    }
    c->mark_pending = false;
    if (c->mark_op_seq > c->idle_seq) c->idle_seq = c->mark_op_seq;
    return 0;
  }
  const uint64_t seen = c->op_seq;
  while ((e = hipStreamQuery(c->stream)) == hipErrorNotReady) { }
  if (e != hipSuccess) return fail(c, 100 + (int)e, "hipStreamQuery failed: %s", hipGetErrorString(e));
  c->idle_seq = seen;
  return 0;
}

int wait_results_landed(pk_ctx* c) {
  const int rc = wait_results_landed_raw(c);
  return rc ? rc : handoff_check(c);
}

int enqueue_result_copies(pk_ctx* c, unsigned mask) {
  struct Piece { const double* src; double* dst; size_t count; bool pinned; };
  std::vector<Piece> pcs;
  pcs.reserve(8);
  int first = -1;
  auto add = [&](const double* src, double* dst, size_t count, bool pinned) {
    if (!count) return;
    if (!pcs.empty() && pcs.back().src + pcs.back().count == src && pcs.back().dst + pcs.back().count == dst &&
        pcs.back().pinned == pinned) {
      pcs.back().count += count;
      return;
    }
    pcs.push_back(Piece{src, dst, count, pinned});
  };
  // device order of the x-results is [J | grad f | g]: one piece when nothing is left out; split_copy sends grad f | g first
  const int joined[5] = {0, 3, 1, 2, 4}, split[5] = {0, 1, 2, 3, 4};
  const int* order = c->split_copy ? split : joined;
  size_t early_pieces = 0;
  bool early = false;
  for (int o = 0; o < 5; ++o) {
    const int k = order[o];
    if (!((mask >> k) & 1u) || c->enq[k]) continue;
    if (first < 0) first = k;
    if (c->stored_direct[k]) continue;
    const bool pinned = !c->target[k] || c->target_pinned[k];
    if (k == 3 && c->jac_filled) {
      const double* dj = device_result(c, 3);
      for (const auto& r : c->jruns) add(dj + r.first, c->landed[3] + r.first, (size_t)(r.second - r.first), pinned);
    } else {
      add(device_result(c, k), c->landed[k], result_count(c, k), pinned);
    }
    if (c->split_copy && (k == 1 || k == 2)) { early = true; early_pieces = pcs.size(); }
  }
  if (first < 0) return 0;
  // an event behind grad f | g only when something (J) follows them in this batch: otherwise the stream's state tells
  const bool want_early = early && c->spin_wait && ((mask >> 3) & 1u) && !c->enq[3] && !c->stored_direct[3] && pcs.size() > early_pieces;
  int rc;
  bool any_dma = false;
  for (size_t i = 0; i < pcs.size(); ++i) {
    const bool by_kernel = pcs[i].pinned && sizeof(double) * pcs[i].count <= ((size_t)c->kernel_download << 20);
    any_dma |= !by_kernel || (((uintptr_t)pcs[i].dst ^ (uintptr_t)pcs[i].src) & 8) != 0;
    if ((rc = copy_async(c, pcs[i].dst, pcs[i].src, pcs[i].count, hipMemcpyDeviceToHost, by_kernel))) return rc;
    if (want_early && i + 1 == early_pieces) {
      PK_HIP(c, hipEventRecord(c->ev_early, c->stream));
      c->early_valid = true;
    }
  }
  ++c->op_seq;
  // (a mark kernel behind a DMA would wait for the hand-off between the two engines, ~10 us: large copies keep the stream poll)
  c->mark_pending = false;
  if (!any_dma && (rc = enqueue_mark(c))) return rc;
  if (!c->spin_wait) PK_HIP(c, hipEventRecord(c->ev_out[first], c->stream));      // (see wait_result)
  for (int k = 0; k < 5; ++k)
    if (((mask >> k) & 1u) && !c->enq[k]) { c->enq[k] = true; c->ev_of[k] = first; }
  return 0;
}

// Wait for result k of the current iterate.  Measured on MI355X (tools/dma_probe.cpp, profiles/r03_b_dma_probe.txt): a
// hipEventRecord behind a copy plus hipEventSynchronize (or polling hipEventQuery) returns ~8 us after polling
// hipStreamQuery alone does, and a word the kernel itself stores into pinned memory is seen ~4 us before its event.  So
// (spin_wait, the default) no event is recorded for the results at all: f, which the finalize kernel stores into its
// pinned landing place, is awaited on its own word (PK_EMPTY until the system-scope store lands), every copied result by
// polling the stream -- the result copies are the last thing an iterate enqueues, and an idle stream means every result
// enqueued so far has landed.
int wait_result_raw(pk_ctx* c, int k) {
  if (c->done[k]) return 0;
  if (k == 0 && c->stored_direct[0] && c->spin_wait) {
    const volatile unsigned long long* word = (const volatile unsigned long long*)c->landed[0];
    for (long spins = 1; *word == (unsigned long long)PK_EMPTY; ++spins) {
      if ((spins & 0x3FFF) == 0) {          // now and then: has the stream finished (or failed) without storing f?
        const uint64_t seen = c->op_seq;
        const hipError_t e = hipStreamQuery(c->stream);
        if (e == hipSuccess) {
          c->idle_seq = seen;
          if (*word == (unsigned long long)PK_EMPTY) {
            // (an f computed from a hand-off that gave up carries the sentinel's own NaN payload: say what happened)
            const int hc = handoff_check(c);
            return hc ? hc : fail(c, 65, "the objective was not stored by its kernel");
          }
          break;
        }
        if (e != hipErrorNotReady) return fail(c, 100 + (int)e, "waiting for f: %s", hipGetErrorString(e));
      }
    }
    c->done[0] = true;
    return 0;
  }
  if (c->spin_wait) {
    hipError_t e;
    if ((k == 1 || k == 2) && c->early_valid) {       // grad f | g went ahead of J with an event of their own (split_copy)
      while ((e = hipEventQuery(c->ev_early)) == hipErrorNotReady) { }
      if (e != hipSuccess) return fail(c, 100 + (int)e, "hipEventQuery failed: %s", hipGetErrorString(e));
      c->done[1] = c->done[2] = true;
      return 0;
    }
    int rc = wait_results_landed_raw(c);
    if (rc) return rc;
    for (int j = 0; j < 5; ++j)
      if (c->enq[j]) c->done[j] = true;
    return 0;
  }
  PK_HIP(c, hipEventSynchronize(c->ev_out[c->ev_of[k]]));
  c->done[k] = true;
  return 0;
}

int wait_result(pk_ctx* c, int k) {
  const int rc = wait_result_raw(c, k);
  return rc ? rc : handoff_check(c);
}

// stage `count` doubles in the next staging buffer of a double-buffered pair and queue their upload (dst == nullptr: stage
// only -- the consumer kernel reads the pinned buffer itself).  Large inputs are staged and uploaded in a few chunks so that
// the host's memcpy of chunk i + 1 runs while chunk i is on the link (4.8 MB: 202 -> 157 us; every extra DMA costs ~10 us,
// so small inputs go in one piece).
int stage_upload(pk_ctx* c, double* const bufs[2], hipEvent_t const evs[2], uint64_t seqs[2], int& cur, const double* src,
                 double* dst, size_t count, double** staged) {
  cur ^= 1;
  // the buffer was last read two iterates ago -- by an upload whose completion an idle stream seen since then implies
  // (polling waits), or whose event says so (event waits)
  if (seqs[cur] > c->idle_seq) {
    if (c->spin_wait || !dst) {          // (no upload, no event: the kernels that read the buffer in place are awaited on the stream)
      const uint64_t seen = c->op_seq;
      PK_HIP(c, hipStreamSynchronize(c->stream));
      c->idle_seq = seen;
    } else {
      PK_HIP(c, hipEventSynchronize(evs[cur]));
    }
  }
  const size_t bytes = sizeof(double) * count;
  const int chunks = (dst && c->chunk_upload && bytes >= ((size_t)2 << 20)) ? (c->kernel_upload ? 4 : 3) : 1;
  const size_t step = ((count + chunks - 1) / chunks + 7) & ~(size_t)7;
  int rc;
  for (size_t lo = 0; lo < count; lo += step) {
    const size_t len = count - lo < step ? count - lo : step;
    (void)pk_copy_bits(bufs[cur] + lo, src + lo, len);      // (memcpy; with the helper threads of pk_host_threads from 1 MB on)
    if (dst && (rc = copy_async(c, dst + lo, bufs[cur] + lo, len, hipMemcpyHostToDevice, c->kernel_upload != 0))) return rc;
  }
  seqs[cur] = ++c->op_seq;          // (a consumer kernel reading the buffer itself is enqueued right behind: same number)
  if (dst && !c->spin_wait) PK_HIP(c, hipEventRecord(evs[cur], c->stream));
  if (staged) *staged = bufs[cur];
  return 0;
}

}  // namespace

extern "C" int pk_eval_xpart_dev(pk_ctx* c, const double* d_x, double* d_f, double* d_grad, double* d_g, double* d_jac, void* stream);

namespace {
// Meshes with big intervals (more than 64 points) are served by the fused x-kernel only: a single callback on device
// pointers runs it with the context's own buffers for the outputs nobody asked for.
int eval_one_via_xpart(pk_ctx* c, const double* d_x, int which, double* d_out, void* stream) {
  c->x_valid = false;
  double* o[4] = {c->d_f, c->d_grad, c->d_g, c->d_J};
  o[which] = d_out;
  return pk_eval_xpart_dev(c, d_x, o[0], o[1], o[2], o[3], stream);
}
}  // namespace

extern "C" {

int pk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* pk_kernel_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }

int pk_create(pk_ctx** out, int device_id) {
  if (!out) return fail(nullptr, 1, "pk_create: null output pointer");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, 10, "pk_create: no HIP device available (%s); the evaluator has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, 11, "pk_create: device %d out of range [0,%d)", device_id, ndev);
  pk_ctx* c = new pk_ctx();
  c->device = device_id;
  if (const char* dbg = getenv("POCKIT_AMD_DEBUG_FLAGS")) c->debug_flags = atoi(dbg) & (256 | 512 | 1024 | 2048 | 4096 | 8192 | 16384 | 32768 | 65536 | 131072);
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
    int rc = fail(nullptr, 12, "pk_create: %s", hipGetErrorString(e));
    delete c;
    return rc;
  }
  *out = c;
  return 0;
}

void pk_destroy(pk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int k = 0; k < K_COUNT; ++k)
    for (auto& ev : c->pending[k]) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  for (auto& ev : c->free_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  free_problem(c);
  if (c->module) (void)hipModuleUnload(c->module);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* pk_last_error(pk_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int pk_load_model(pk_ctx* c, const void* code_object, size_t len, const pk_model_desc* md) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!code_object || len == 0 || !md) return fail(c, 20, "pk_load_model: empty code object or descriptor");
  PK_HIP(c, hipSetDevice(c->device));
  if (c->module) { (void)hipModuleUnload(c->module); c->module = nullptr; c->have_model = false; }
  PK_HIP(c, hipModuleLoadData(&c->module, code_object));
  for (int k = 0; k < K_COUNT; ++k) PK_HIP(c, hipModuleGetFunction(&c->fn[k], c->module, kKernelNames[k]));
  c->md = *md;
  if (md->tab_cap != 64 && md->tab_cap != 256) return fail(c, 23, "pk_load_model: table capacity %d (64 or 256)", md->tab_cap);
  if (md->cycle_subs < 0 || md->cycle_subs > 4096) return fail(c, 25, "pk_load_model: cycle_subs %d", md->cycle_subs);
  if (md->hess_subs < 0 || md->hess_subs > 4096 || (md->hess_subs > 0) != (md->cycle_subs > 0))
    return fail(c, 25, "pk_load_model: hess_subs %d with cycle_subs %d", md->hess_subs, md->cycle_subs);
  if (md->wide < 0 || md->wide > 1) return fail(c, 25, "pk_load_model: wide %d", md->wide);
  if (md->big_global < 0 || md->big_global > 1 || md->big_rows < 0 || md->big_rows > (1 << 20))
    return fail(c, 25, "pk_load_model: big_global %d / big_rows %d", md->big_global, md->big_rows);
  if (md->hessc_subs < 0 || md->hessc_subs > 4096 || md->jacc_subs < 0 || md->jacc_subs > 4096 ||
      ((md->hessc_subs > 0 || md->jacc_subs > 0) && md->cycle_subs == 0))
    return fail(c, 25, "pk_load_model: hessc_subs %d / jacc_subs %d with cycle_subs %d", md->hessc_subs, md->jacc_subs, md->cycle_subs);
  if (md->max_phases < 0 || md->max_phases > PK_HOST_MAX_PHASES || md->n_phase > (md->max_phases > 0 ? md->max_phases : 8))
    return fail(c, 24, "pk_load_model: %d phases, code object compiled for %d (the library passes at most %d phase records in "
                       "the kernel arguments)", md->n_phase, md->max_phases > 0 ? md->max_phases : 8, PK_HOST_MAX_PHASES);
  // what a launch will ask for (launch_raw): the model's staging rows of the workgroup's waves + their table blocks -- the
  // same bytes pockit_amd.codegen.ModelSource.launch_lds_bytes counts when it chooses the group size
  const size_t lds_max = 160 * 1024;
  const size_t tab = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)(2 * md->tab_cap + 2 * PK_WAVE + md->tab_cap / 2);
  const size_t need[5] = {(size_t)md->lds_g, (size_t)md->lds_j, (size_t)md->lds_h, (size_t)md->lds_x, (size_t)md->lds_jc};
  for (size_t v : need)
    if (v * PK_WAVES_PER_BLOCK * sizeof(double) + tab > lds_max)
      return fail(c, 21, "pk_load_model: model needs %zu bytes of LDS per workgroup (> 160 KiB)",
                  v * PK_WAVES_PER_BLOCK * sizeof(double) + tab);
  c->have_model = true;
  return 0;
}

int pk_set_problem(pk_ctx* c, const pk_problem_desc* pd) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!c->have_model) return fail(c, 2, "pk_set_problem: load a model first");
  if (!pd) return fail(c, 30, "pk_set_problem: null descriptor");
  if (pd->n_phase > (c->md.max_phases > 0 ? c->md.max_phases : 8))
    return fail(c, 32, "pk_set_problem: %d phases, but the code object was generated for at most %d", pd->n_phase,
                c->md.max_phases > 0 ? c->md.max_phases : 8);
  if (pd->n_phase != c->md.n_phase) return fail(c, 31, "pk_set_problem: %d phases but the model was generated for %d", pd->n_phase, c->md.n_phase);
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  free_problem(c);
  c->n = pd->n; c->m = pd->m; c->n_sys = pd->n_sys; c->n_s = pd->n_s; c->l_s = pd->l_s;
  c->n_phase = pd->n_phase; c->n_tiles = pd->n_tiles; c->nnz_J = pd->nnz_J; c->nnz_H = pd->nnz_H;
  c->n_items_jac = pd->n_items_jac; c->n_items_hess = pd->n_items_hess; c->gz_off = pd->gz_off; c->n_gz = pd->n_gz;
  c->n_items_aux = pd->n_items_aux; c->n_outer = pd->n_outer; c->n_aux = pd->n_aux;
  c->cycle_layout = 0;
  c->n_items_hessc = pd->n_items_hessc; c->nnz_Hc = pd->nnz_Hc;
  c->n_items_jacc = pd->n_items_jacc; c->nnz_Jc = pd->nnz_Jc;
  // Small meshes are bound by the serial chain of one wave, not by throughput: let two waves share a tile in
  // pk_xall as long as that still leaves at most two waves per SIMD (POCKIT_AMD_SPLIT=0/1 overrides).
  {
    const char* env = getenv("POCKIT_AMD_SPLIT");
    c->split_xall = env ? atoi(env) != 0 : (pd->n_tiles > 0 && pd->n_tiles <= 1024);
  }
  int rc;
  if ((rc = upload(c, &c->d_phases, pd->phases, sizeof(PkPhase) * (size_t)pd->n_phase))) return rc;
  c->h_phases.assign((const PkPhase*)pd->phases, (const PkPhase*)pd->phases + pd->n_phase);

  {   // intervals with more than 256 points stage their rows in device memory: slot numbers into the tile records
    std::vector<PkTile> tiles((const PkTile*)pd->tiles, (const PkTile*)pd->tiles + pd->n_tiles);
    int32_t n_stage = 0, kmax = 0;
    for (PkTile& t : tiles) {
      t.stage = 0;
      if (t.nj > 0 && t.K > (c->md.big_global ? PK_WAVE : 256)) {
        t.stage = n_stage++;
        if (t.K > kmax) kmax = t.K;
      }
    }
    release(c->d_big_stage);
    c->big_row = c->big_slot = 0;
    if (n_stage) {
      size_t rows = (size_t)(c->md.lds_x > c->md.lds_h ? c->md.lds_x : c->md.lds_h) / PK_WAVE;
      if ((size_t)c->md.lds_jc / PK_WAVE > rows) rows = (size_t)c->md.lds_jc / PK_WAVE;
      if ((size_t)c->md.big_rows > rows) rows = (size_t)c->md.big_rows;      // (big_global: lds_x is sized for ordinary tiles only)
      c->big_row = (kmax + 7) & ~7;
      const size_t slot = rows * (size_t)c->big_row;
      if (slot > (size_t)INT32_MAX) return fail(c, 33, "pk_set_problem: an interval with %d points is too long for the staging buffer", kmax);
      c->big_slot = (int32_t)slot;
      PK_HIP(c, hipMalloc((void**)&c->d_big_stage, sizeof(double) * slot * 4 * (size_t)n_stage));
    }
    if ((rc = upload(c, &c->d_tiles, tiles.data(), sizeof(PkTile) * tiles.size()))) return rc;
  }
  if ((rc = upload(c, &c->d_kinds, pd->kinds, sizeof(PkKind) * (size_t)pd->n_kinds))) return rc;
  if ((rc = upload(c, &c->d_items_jac, pd->items_jac, sizeof(PkItem) * (size_t)pd->n_items_jac))) return rc;
  if ((rc = upload(c, &c->d_items_hess, pd->items_hess, sizeof(PkItem) * (size_t)pd->n_items_hess))) return rc;
  if ((rc = upload(c, &c->d_items_aux, pd->items_aux, sizeof(PkItem) * (size_t)pd->n_items_aux))) return rc;
  if ((rc = upload(c, &c->d_outer, pd->outer, sizeof(PkOuter) * (size_t)pd->n_outer))) return rc;
  if ((rc = upload(c, &c->d_items_hessc, pd->items_hessc, sizeof(PkItem) * (size_t)pd->n_items_hessc))) return rc;
  if ((rc = upload(c, &c->d_items_jacc, pd->items_jacc, sizeof(PkItem) * (size_t)pd->n_items_jacc))) return rc;
  if ((rc = upload(c, (void**)&c->d_ib, pd->ib, sizeof(int32_t) * (size_t)pd->n_ib))) return rc;
  if ((rc = upload(c, (void**)&c->d_db, pd->db, sizeof(double) * (size_t)pd->n_db))) return rc;
  if ((rc = upload(c, (void**)&c->d_lb, pd->lb, sizeof(int64_t) * (size_t)pd->n_lb))) return rc;
  auto dalloc = [&](double** p, size_t count) -> int {
    PK_HIP(c, hipMalloc((void**)p, sizeof(double) * (count ? count : 1)));
    PK_HIP(c, hipMemset(*p, 0, sizeof(double) * (count ? count : 1)));
    return 0;
  };
  // J, grad f and g of the host shim share ONE allocation in this order, on the device and in pinned host memory: the
  // pieces of J that change with x, grad f and g then leave in one DMA (every extra DMA costs ~10 us on this link)
  if ((rc = dalloc(&c->d_x, c->n)) || (rc = dalloc(&c->d_lam, c->m)) || (rc = dalloc(&c->d_f, 1)) ||
      (rc = dalloc(&c->d_J, (size_t)c->nnz_J + (size_t)c->n + (size_t)c->m)) ||
      (rc = dalloc(&c->d_H, (size_t)c->nnz_H)) || (rc = dalloc(&c->d_aux, (size_t)c->n_aux)) || (rc = dalloc(&c->d_Hc, (size_t)c->nnz_Hc)) || (rc = dalloc(&c->d_Jc, (size_t)c->nnz_Jc)) || (rc = dalloc(&c->d_I, c->md.n_I)) ||
      (rc = dalloc(&c->d_partial, (2 * (size_t)c->n_tiles / PK_WAVES_PER_BLOCK + 2) * (size_t)c->md.nred)) ||
      (rc = dalloc(&c->d_partial2, (2 * (size_t)c->n_tiles / PK_WAVES_PER_BLOCK + 2) * (size_t)c->md.nred)))
    return rc;
  c->has_big = false;
  {
    const PkTile* tl = (const PkTile*)pd->tiles;
    for (int32_t t = 0; t < pd->n_tiles; ++t)
      if (tl[t].nj > 0 && tl[t].K > PK_WAVE) {
        c->has_big = true;
        if (t % PK_WAVES_PER_BLOCK || tl[t].nj != 1)
          return fail(c, 33, "pk_set_problem: tile %d: an interval with more than %d points must be alone in the first slot of a tile block", t, PK_WAVE);
        for (int32_t u = 1; u < PK_WAVES_PER_BLOCK && t + u < pd->n_tiles; ++u)
          if (tl[t + u].nj != 0) return fail(c, 33, "pk_set_problem: tile %d shares a block with a big interval", t + u);
      }
  }
  c->d_grad = c->d_J + c->nnz_J;
  c->d_g = c->d_grad + c->n;
  c->jruns.assign(1, std::make_pair((int64_t)0, (int64_t)c->nnz_J));
  c->jconst.clear();
  c->jruns_other.assign(1, std::make_pair((int64_t)0, (int64_t)c->nnz_Jc));
  c->jconst_other.clear();
  {   // hand-off slots of pk_cycle: one per x-kernel workgroup and reduction row, PK_EMPTY between launches
    const size_t slots = (2 * (size_t)c->n_tiles / PK_WAVES_PER_BLOCK + 2) * (size_t)c->md.nred;
    const std::vector<unsigned long long> empty(slots, (unsigned long long)PK_EMPTY);
    if ((rc = upload(c, (void**)&c->d_cpart, empty.data(), sizeof(unsigned long long) * slots))) return rc;
    if ((rc = upload(c, (void**)&c->d_cpart2, empty.data(), sizeof(unsigned long long) * slots))) return rc;
    c->cpart_slots = slots;
  }
  {
    const size_t cnt[5] = {1, (size_t)c->n, (size_t)c->m, (size_t)c->nnz_J, (size_t)c->nnz_H};
    for (int b = 0; b < 2; ++b) {
      PK_HIP(c, hipHostMalloc((void**)&c->h_xs[b], sizeof(double) * (size_t)(c->n ? c->n : 1), hipHostMallocDefault));
      PK_HIP(c, hipHostMalloc((void**)&c->h_lams[b], sizeof(double) * (size_t)(c->m ? c->m : 1), hipHostMallocDefault));
      PK_HIP(c, hipEventCreateWithFlags(&c->ev_xs[b], hipEventDisableTiming));
      PK_HIP(c, hipEventCreateWithFlags(&c->ev_lams[b], hipEventDisableTiming));
    }
    PK_HIP(c, hipHostMalloc((void**)&c->h_out[0], sizeof(double) * 8, hipHostMallocDefault));
    std::memset(c->h_out[0], 0, sizeof(double) * 8);      // ([0] f, [4] the progress mark of mark_wait, [6] [7] PkArgs.status)
    c->status_seen[0] = c->status_seen[1] = 0;
    c->mark_val = 0; c->mark_op_seq = 0; c->mark_pending = false;
    PK_HIP(c, hipHostMalloc((void**)&c->h_out[3], sizeof(double) * (cnt[3] + cnt[1] + cnt[2] + 1), hipHostMallocDefault));
    PK_HIP(c, hipHostMalloc((void**)&c->h_out[4], sizeof(double) * (cnt[4] + 1), hipHostMallocDefault));
    c->h_out[1] = c->h_out[3] + cnt[3];                      // (one block [J | grad f | g], like the device's)
    c->h_out[2] = c->h_out[1] + cnt[1];
    for (int k = 0; k < 5; ++k) PK_HIP(c, hipEventCreateWithFlags(&c->ev_out[k], hipEventDisableTiming));
    PK_HIP(c, hipEventCreateWithFlags(&c->ev_early, hipEventDisableTiming));
    c->xbuf = c->lambuf = 0;
  }
  auto keep = [](std::vector<int32_t>& v, const int32_t* src, int64_t cnt) {
    v.clear();
    if (src) v.assign(src, src + cnt);
  };
  keep(c->jac_row, pd->jac_row, pd->nnz_J); keep(c->jac_col, pd->jac_col, pd->nnz_J);
  keep(c->hess_row, pd->hess_row, pd->nnz_H); keep(c->hess_col, pd->hess_col, pd->nnz_H);
  c->have_problem = true;
  return 0;
}

int pk_get_structure(pk_ctx* c, int32_t* jr, int32_t* jc, int32_t* hr, int32_t* hc) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->jac_row.empty() && c->nnz_J) return fail(c, 40, "pk_get_structure: no structure was supplied to pk_set_problem");
  if (jr) std::memcpy(jr, c->jac_row.data(), sizeof(int32_t) * c->jac_row.size());
  if (jc) std::memcpy(jc, c->jac_col.data(), sizeof(int32_t) * c->jac_col.size());
  if (hr) std::memcpy(hr, c->hess_row.data(), sizeof(int32_t) * c->hess_row.size());
  if (hc) std::memcpy(hc, c->hess_col.data(), sizeof(int32_t) * c->hess_col.size());
  return 0;
}

// ---------------------------------------------------------------- device-pointer API
int pk_eval_f_dev(pk_ctx* c, const double* d_x, double* d_f, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  return prepass(c, d_x, nullptr, 0.0, d_f, true, pick(c, stream));
}

// sharded mode, step 1: this shard's contribution to every integral -> integral buffer
int pk_eval_integrals_dev(pk_ctx* c, const double* d_x, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  return prepass(c, d_x, nullptr, 0.0, c->d_f, false, pick(c, stream));
}

// sharded mode, step 2 (after the caller all-reduced the integral buffer): f = F_o(I, s)
int pk_eval_f_from_integrals_dev(pk_ctx* c, const double* d_x, double* d_f, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_f = d_f;
  A.flags |= F_WRITE_F;
  return launch(c, K_FIN, A, 1, 0, pick(c, stream));
}

int pk_set_shard(pk_ctx* c, int secondary, int external_prepass, double* d_integrals) {
  if (!c) return fail(nullptr, 1, "null context");
  c->shard_flags = secondary ? F_SECONDARY : 0;
  c->external_prepass = external_prepass != 0;
  c->ext_I = d_integrals;
  drop_cycle_graph(c);            // (a captured cycle holds the old flags / integral buffer)
  return 0;
}

int pk_eval_grad_dev(pk_ctx* c, const double* d_x, double* d_grad, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->has_big) return eval_one_via_xpart(c, d_x, 1, d_grad, stream);
  hipStream_t st = pick(c, stream);
  if (c->md.prepass_grad && !c->external_prepass && (rc = prepass(c, d_x, nullptr, 0.0, c->d_f, false, st))) return rc;
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_grad = d_grad;
  if ((rc = launch(c, K_GRAD, A, tile_blocks(c), 0, st))) return rc;
  A.flags |= F_FIN_GRAD;
  return launch(c, K_FIN, A, 1, 0, st);
}

int pk_eval_g_dev(pk_ctx* c, const double* d_x, double* d_g, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->has_big) return eval_one_via_xpart(c, d_x, 2, d_g, stream);
  hipStream_t st = pick(c, stream);
  if (c->md.prepass_g && !c->external_prepass && (rc = prepass(c, d_x, nullptr, 0.0, c->d_f, false, st))) return rc;
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_g = d_g;
  return launch(c, K_G, A, tile_blocks(c) + 1, sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_g, st);
}

int pk_eval_jac_dev(pk_ctx* c, const double* d_x, double* d_vals, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->has_big) return eval_one_via_xpart(c, d_x, 3, d_vals, stream);
  hipStream_t st = pick(c, stream);
  if (c->md.prepass_jac && !c->external_prepass && (rc = prepass(c, d_x, nullptr, 0.0, c->d_f, false, st))) return rc;
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_jac = d_vals;
  A.items = (const PkItem*)c->d_items_jac;
  A.n_items = c->n_items_jac;
  size_t lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_j;
  if (lds < sizeof(double) * (size_t)c->md.ne_j) lds = sizeof(double) * (size_t)c->md.ne_j;
  return launch(c, K_JAC, A, tile_blocks(c) + 1, lds, st);
}

int pk_eval_hess_dev(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_vals, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (!d_lam) return fail(c, 50, "pk_eval_hess: lambda is required");
  hipStream_t st = pick(c, stream);
  if (c->md.prepass_hess && !c->external_prepass && (rc = prepass(c, d_x, d_lam, sigma, c->d_f, false, st))) return rc;
  PkArgs A = base_args(c, d_x, d_lam, sigma);
  A.o_hess = d_vals;
  A.items = (const PkItem*)c->d_items_hess;
  A.n_items = c->n_items_hess;
  size_t lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_h;
  if (lds < sizeof(double) * (size_t)c->md.ne_h) lds = sizeof(double) * (size_t)c->md.ne_h;
  if ((rc = launch(c, K_HESS, A, hess_grid(c), lds, st))) return rc;
  if (c->n_outer > 0) {   // objective / system constraints nonlinear in the integrals: outer-product blocks
    PkArgs X = base_args(c, d_x, d_lam, sigma);
    X.o_hess = d_vals;
    X.items = (const PkItem*)c->d_items_aux;
    X.n_items = c->n_items_aux;
    if ((rc = launch(c, K_AUX, X, tile_blocks(c) + 1, sizeof(double) * (size_t)(c->md.ne_a > 0 ? c->md.ne_a : 1), st))) return rc;
    // a shard stops here: its auxiliary buffer holds the entries of ITS nodes, the caller sums the buffers over the
    // ranks and has the primary rank form the blocks (pk_eval_outer_dev)
    if (c->external_prepass) return 0;
    const unsigned grid = (unsigned)(c->n_outer < 4096 ? c->n_outer : 4096);
    return launch(c, K_OUTER, X, grid, 0, st);
  }
  return 0;
}

// sharded mode, models nonlinear in the integrals: the auxiliary buffer pk_eval_hess_dev fills on a shard (NULL / 0 for
// models without outer-product blocks) ...
int pk_aux_buffer(pk_ctx* c, double** ptr, int64_t* count) {
  int rc = ready(c);
  if (rc) return rc;
  if (ptr) *ptr = c->n_outer > 0 ? c->d_aux : nullptr;
  if (count) *count = c->n_outer > 0 ? (int64_t)c->n_aux : 0;
  return 0;
}

// ... and the outer-product blocks of the Hessian from the buffer summed over the ranks (d_aux_sum: n_aux doubles, device)
int pk_eval_outer_dev(pk_ctx* c, const double* d_aux_sum, double* d_vals, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->n_outer <= 0) return 0;
  if (!d_aux_sum || !d_vals) return fail(c, 50, "pk_eval_outer: null device buffer");
  PkArgs X = base_args(c, nullptr, nullptr, 0.0);
  X.o_hess = d_vals;
  X.o_aux = const_cast<double*>(d_aux_sum);
  const unsigned grid = (unsigned)(c->n_outer < 4096 ? c->n_outer : 4096);
  return launch(c, K_OUTER, X, grid, 0, pick(c, stream));
}

// compact (coalesced) Hessian of the Lagrangian: one value per distinct (row, col) class of a node
int pk_eval_hessc_dev(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_vals, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (!d_lam) return fail(c, 50, "pk_eval_hessc: lambda is required");
  if (c->nnz_Hc <= 0) return fail(c, 51, "pk_eval_hessc: no compact Hessian layout was supplied to pk_set_problem");
  hipStream_t st = pick(c, stream);
  if (c->md.prepass_hess && !c->external_prepass && (rc = prepass(c, d_x, d_lam, sigma, c->d_f, false, st))) return rc;
  PkArgs A = base_args(c, d_x, d_lam, sigma);
  A.o_hess = d_vals;
  A.items = (const PkItem*)c->d_items_hessc;
  A.n_items = c->n_items_hessc;
  size_t lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_g;      // the tile's multiplier rows, [state][row]
  if (lds < sizeof(double) * (size_t)c->md.ne_hc) lds = sizeof(double) * (size_t)c->md.ne_hc;
  return launch(c, K_HESSC, A, tile_blocks(c) * (c->md.hessc_subs > 0 ? (unsigned)c->md.hessc_subs : 1u) + 1, lds, st);
}

// compact (coalesced) Jacobian: dense-column entries of the dynamics contracted with the integration block first
int pk_eval_jacc_dev(pk_ctx* c, const double* d_x, double* d_vals, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->nnz_Jc <= 0) return fail(c, 52, "pk_eval_jacc: no compact Jacobian layout was supplied to pk_set_problem");
  hipStream_t st = pick(c, stream);
  if (c->md.prepass_jac && !c->external_prepass && (rc = prepass(c, d_x, nullptr, 0.0, c->d_f, false, st))) return rc;
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_jac = d_vals;
  A.items = (const PkItem*)c->d_items_jacc;
  A.n_items = c->n_items_jacc;
  size_t lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_jc;
  if (lds < sizeof(double) * (size_t)c->md.ne_jc) lds = sizeof(double) * (size_t)c->md.ne_jc;
  return launch(c, K_JACC, A, tile_blocks(c) * (c->md.jacc_subs > 0 ? (unsigned)c->md.jacc_subs : 1u) + 1, lds, st);
}

int pk_eval_jacc(pk_ctx* c, const double* x, double* vals) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !vals) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  PK_HIP(c, hipMemcpyAsync(c->d_x, x, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
  if ((rc = pk_eval_jacc_dev(c, c->d_x, c->d_Jc, nullptr))) return rc;
  PK_HIP(c, hipMemcpyAsync(vals, c->d_Jc, sizeof(double) * (size_t)c->nnz_Jc, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------------------------------------------------------- device-resident CSR hand-off
int pk_set_csr_map(pk_ctx* c, int which, const int32_t* seg, const int32_t* perm, int64_t n_unique, int64_t n_triplets) {
  int rc = ready(c);
  if (rc) return rc;
  if (which < 0 || which > 3)
    return fail(c, 80, "pk_set_csr_map: which must be 0 (Jacobian), 1 (Hessian), 2 (compact Hessian) or 3 (compact Jacobian)");
  const int64_t expect = which == 0 ? c->nnz_J : which == 1 ? c->nnz_H : which == 2 ? c->nnz_Hc : c->nnz_Jc;
  if (!perm || n_unique <= 0 || n_unique > n_triplets || n_triplets != expect || n_triplets > INT32_MAX)
    return fail(c, 81, "pk_set_csr_map: map does not match the problem (%lld triplets expected)", (long long)expect);
  // validate on the host: the kernel indexes with these
  for (int64_t q = 0; q < n_triplets; ++q)
    if (perm[q] < 0 || perm[q] >= n_triplets) return fail(c, 82, "pk_set_csr_map: perm[%lld] out of range", (long long)q);
  if (seg) {
    if (seg[0] != 0 || seg[n_unique] != n_triplets) return fail(c, 83, "pk_set_csr_map: segment table does not cover the triplets");
    for (int64_t p = 0; p < n_unique; ++p)
      if (seg[p + 1] <= seg[p]) return fail(c, 83, "pk_set_csr_map: empty or decreasing segment %lld", (long long)p);
  } else if (n_unique != n_triplets) {
    return fail(c, 83, "pk_set_csr_map: a segment table is required when entries repeat");
  }
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  auto& m = c->csr[which];
  release(m.d_seg); release(m.d_perm); release(m.d_vals);
  m.n_unique = m.n_triplets = 0;
  if (seg) {
    // repeated entries: the device gets the runs per slice of 256 consecutive CSR entries, transposed and padded to the
    // slice's longest run (see kernel_csr)
    const int64_t nblk = (n_unique + PK_BLOCK - 1) / PK_BLOCK;
    std::vector<int32_t> off((size_t)nblk + 1, 0);
    int64_t total = 0;
    for (int64_t b = 0; b < nblk; ++b) {
      int32_t width = 0;
      for (int64_t p = b * PK_BLOCK; p < n_unique && p < (b + 1) * PK_BLOCK; ++p) width = std::max(width, seg[p + 1] - seg[p]);
      off[(size_t)b] = (int32_t)total;
      total += (int64_t)width * PK_BLOCK;
      if (total > INT32_MAX) return fail(c, 85, "pk_set_csr_map: the padded run table does not fit 32-bit offsets");
    }
    off[(size_t)nblk] = (int32_t)total;
    std::vector<int32_t> sell((size_t)total, -1);
    for (int64_t p = 0; p < n_unique; ++p) {
      const int64_t b = p / PK_BLOCK, t = p % PK_BLOCK;
      for (int32_t k = 0; k < seg[p + 1] - seg[p]; ++k) sell[(size_t)(off[(size_t)b] + (int64_t)k * PK_BLOCK + t)] = perm[seg[p] + k];
    }
    if ((rc = upload(c, (void**)&m.d_seg, off.data(), sizeof(int32_t) * off.size()))) return rc;
    if ((rc = upload(c, (void**)&m.d_perm, sell.data(), sizeof(int32_t) * sell.size()))) return rc;
  } else if ((rc = upload(c, (void**)&m.d_perm, perm, sizeof(int32_t) * (size_t)n_triplets))) {
    return rc;
  }
  PK_HIP(c, hipMalloc((void**)&m.d_vals, sizeof(double) * (size_t)n_unique));
  m.n_unique = n_unique;
  m.n_triplets = n_triplets;
  return 0;
}

int pk_gather_csr_dev(pk_ctx* c, int which, const double* d_triplets, double* d_csr, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (which < 0 || which > 3 || c->csr[which].n_unique == 0) return fail(c, 84, "pk_gather_csr: call pk_set_csr_map first");
  const auto& m = c->csr[which];
  PkArgs A = base_args(c, nullptr, nullptr, 0.0);
  A.csr_in = d_triplets; A.csr_seg = m.d_seg; A.csr_perm = m.d_perm; A.csr_out = d_csr; A.n_csr = (int32_t)m.n_unique;
  unsigned grid = (unsigned)((m.n_unique + PK_BLOCK - 1) / PK_BLOCK);
  if (grid > 4096) grid = 4096;
  return launch(c, K_CSR, A, grid, 0, pick(c, stream));
}

int pk_eval_jac_csr_dev(pk_ctx* c, const double* d_x, double* d_csr, void* stream) {
  if (c) c->x_valid = false;      // (the triplets pass through the context's J buffer)
  if (c && c->have_problem && c->csr[3].n_unique > 0 && c->nnz_Jc > 0) {      // from the compact evaluation, like the Hessian's
    int rc = pk_eval_jacc_dev(c, d_x, c->d_Jc, stream);
    return rc ? rc : pk_gather_csr_dev(c, 3, c->d_Jc, d_csr, stream);
  }
  int rc = pk_eval_jac_dev(c, d_x, c ? c->d_J : nullptr, stream);
  return rc ? rc : pk_gather_csr_dev(c, 0, c->d_J, d_csr, stream);
}

// The CSR values of the Hessian come from the COMPACT evaluation when its map is set (which = 2): pk_hessc writes one value
// per distinct (row, col) -- the multipliers contracted with the integration block first -- and the gather is a pure
// permutation of nnz_Hc values; the route through the reference layout writes every repeated triplet (6.6 per entry at the
// humanoid's size) and adds them up again (40k nodes: 17 + 52 us vs 8 + 6 us).
int pk_eval_hess_csr_dev(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_csr, void* stream) {
  if (c && c->have_problem && c->csr[2].n_unique > 0 && c->nnz_Hc > 0) {
    int rc = pk_eval_hessc_dev(c, d_x, d_lam, sigma, c->d_Hc, stream);
    return rc ? rc : pk_gather_csr_dev(c, 2, c->d_Hc, d_csr, stream);
  }
  int rc = pk_eval_hess_dev(c, d_x, d_lam, sigma, c ? c->d_H : nullptr, stream);
  return rc ? rc : pk_gather_csr_dev(c, 1, c->d_H, d_csr, stream);
}

int pk_eval_jac_csr(pk_ctx* c, const double* x, double* vals) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !vals) return fail(c, 60, "null host buffer");
  const int jm = c->csr[3].n_unique > 0 ? 3 : 0;      // (both maps fill the same CSR entries)
  if (c->csr[jm].n_unique == 0) return fail(c, 84, "pk_eval_jac_csr: call pk_set_csr_map first");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  PK_HIP(c, hipMemcpyAsync(c->d_x, x, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
  if ((rc = pk_eval_jac_csr_dev(c, c->d_x, c->csr[jm].d_vals, nullptr))) return rc;
  PK_HIP(c, hipMemcpyAsync(vals, c->csr[jm].d_vals, sizeof(double) * (size_t)c->csr[jm].n_unique, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

int pk_eval_hess_csr(pk_ctx* c, const double* x, const double* lambda, double sigma, double* vals) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !lambda || !vals) return fail(c, 60, "null host buffer");
  const int hm = c->csr[2].n_unique > 0 ? 2 : 1;      // (both maps fill the same CSR entries)
  if (c->csr[hm].n_unique == 0) return fail(c, 84, "pk_eval_hess_csr: call pk_set_csr_map first");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  PK_HIP(c, hipMemcpyAsync(c->d_x, x, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
  PK_HIP(c, hipMemcpyAsync(c->d_lam, lambda, sizeof(double) * (size_t)c->m, hipMemcpyHostToDevice, c->stream));
  if ((rc = pk_eval_hess_csr_dev(c, c->d_x, c->d_lam, sigma, c->csr[hm].d_vals, nullptr))) return rc;
  PK_HIP(c, hipMemcpyAsync(vals, c->csr[hm].d_vals, sizeof(double) * (size_t)c->csr[hm].n_unique, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------------------------------------------------------- mesh error estimation
int pk_set_mesh_error_tables(pk_ctx* c, const void* intervals, int32_t n_intervals, const int32_t* groups,
                             int32_t n_groups, const double* tables, int64_t n_tables, int64_t n_out) {
  int rc = ready(c);
  if (rc) return rc;
  if (!intervals || n_intervals <= 0 || !groups || n_groups <= 0 || !tables || n_tables <= 0 || n_out <= 0)
    return fail(c, 70, "pk_set_mesh_error_tables: empty tables");
  if (n_groups % PK_WAVES_PER_BLOCK)
    return fail(c, 71, "pk_set_mesh_error_tables: wave groups must be padded to a multiple of %d per phase", PK_WAVES_PER_BLOCK);
  if (((size_t)c->md.lds_e / PK_WAVE) * 264 * sizeof(double) > 160 * 1024)
    return fail(c, 72, "pk_set_mesh_error_tables: model needs more than 160 KiB of LDS per workgroup");
  // host-side validation of everything the kernel indexes with (a faulting kernel can take the node down)
  const PkErrIv* iv = (const PkErrIv*)intervals;
  for (int32_t g = 0; g < n_intervals; ++g) {
    const PkErrIv& r = iv[g];
    if (r.phase < 0 || r.phase >= c->n_phase) return fail(c, 73, "pk_set_mesh_error_tables: record %d: bad phase", g);
    const PkPhase& ph = c->h_phases[r.phase];
    const int na = r.K + 1, ncx = r.K + 1 - ph.scheme, nr = ncx;
    const int64_t tab = (int64_t)na * ncx + (int64_t)na * r.K + (int64_t)nr * ncx + (int64_t)nr * na;
    if (r.K < 1 || r.lm < 0 || r.lm + ncx > ph.state_len || r.lm + r.K > ph.L_m || r.tab_off < 0 ||
        r.tab_off + tab > n_tables || r.tau_off < 0 || r.tau_off + na > n_tables || r.row0 < 0 || r.row0 + nr > r.rows ||
        r.out_off < 0 || r.out_off + (int64_t)ph.n_x * r.rows > n_out)
      return fail(c, 74, "pk_set_mesh_error_tables: record %d is inconsistent with the problem", g);
  }
  for (int32_t g = 0; g < n_groups; ++g) {     // a wave's intervals: in range, one phase, one K, K + 1 lanes each
    const int32_t first = groups[2 * g], cnt = groups[2 * g + 1];
    if (first < 0 || first >= n_intervals)
      return fail(c, 76, "pk_set_mesh_error_tables: wave group %d is out of range", g);
    if (cnt == 1 && iv[first].K + 1 > PK_WAVE) {     // K + 1 > 64: a workgroup of its own (first group of the block, count 1;
      if (g % PK_WAVES_PER_BLOCK)                    //  the block's other groups carry count -1)
        return fail(c, 76, "pk_set_mesh_error_tables: wave group %d: an interval with K + 1 > %d must start a block", g, PK_WAVE);
      for (int32_t u = 1; u < PK_WAVES_PER_BLOCK; ++u)
        if (groups[2 * (g + u) + 1] != -1)
          return fail(c, 76, "pk_set_mesh_error_tables: wave group %d shares its block with a workgroup-wide interval", g + u);
      g += PK_WAVES_PER_BLOCK - 1;
      continue;
    }
    if (cnt < 0 || first + cnt > n_intervals || cnt * (iv[first].K + 1) > PK_WAVE)
      return fail(c, 76, "pk_set_mesh_error_tables: wave group %d is out of range", g);
    for (int32_t j = 1; j < cnt; ++j)
      if (iv[first + j].K != iv[first].K || iv[first + j].phase != iv[first].phase)
        return fail(c, 76, "pk_set_mesh_error_tables: wave group %d mixes phases or orders", g);
  }
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  release(c->d_erriv); release(c->d_errgrp); release(c->d_errdb); release(c->d_errT); release(c->d_errI); release(c->d_err_stage);
  c->n_erriv = 0; c->n_err_out = 0; c->err_row = c->err_slot = 0;
  {   // intervals whose K + 1 augmented nodes do not fit the LDS rows of 264 doubles: slots of a staging buffer
    std::vector<PkErrIv> ivs(iv, iv + n_intervals);
    int32_t n_stage = 0, namax = 0;
    for (PkErrIv& r : ivs) {
      r.stage = 0;
      if (r.K + 1 > 264) {
        r.stage = n_stage++;
        if (r.K + 1 > namax) namax = r.K + 1;
      }
    }
    if (n_stage) {
      c->err_row = (namax + 7) & ~7;
      const size_t slot = ((size_t)c->md.lds_e / PK_WAVE) * (size_t)c->err_row;
      if (slot > (size_t)INT32_MAX) return fail(c, 72, "pk_set_mesh_error_tables: an interval with %d points is too long for the staging buffer", namax - 1);
      c->err_slot = (int32_t)slot;
      PK_HIP(c, hipMalloc((void**)&c->d_err_stage, sizeof(double) * slot * (size_t)n_stage));
    }
    if ((rc = upload(c, &c->d_erriv, ivs.data(), sizeof(PkErrIv) * ivs.size()))) return rc;
  }
  if ((rc = upload(c, (void**)&c->d_errgrp, groups, sizeof(int32_t) * 2 * (size_t)n_groups))) return rc;
  if ((rc = upload(c, (void**)&c->d_errdb, tables, sizeof(double) * (size_t)n_tables))) return rc;
  PK_HIP(c, hipMalloc((void**)&c->d_errT, sizeof(double) * (size_t)n_out));
  PK_HIP(c, hipMalloc((void**)&c->d_errI, sizeof(double) * (size_t)n_out));
  PK_HIP(c, hipMemset(c->d_errT, 0, sizeof(double) * (size_t)n_out));
  PK_HIP(c, hipMemset(c->d_errI, 0, sizeof(double) * (size_t)n_out));
  c->n_erriv = n_groups;
  c->n_err_out = n_out;
  return 0;
}

int pk_eval_mesh_error_dev(pk_ctx* c, const double* d_x, double* d_T, double* d_I, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (c->n_erriv <= 0) return fail(c, 75, "pk_eval_mesh_error: call pk_set_mesh_error_tables first");
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.erriv = (const PkErrIv*)c->d_erriv;
  A.errgrp = c->d_errgrp;
  A.errdb = c->d_errdb;
  A.n_erriv = c->n_erriv;
  A.o_errT = d_T;
  A.o_errI = d_I;
  A.big_stage = c->d_err_stage; A.big_row = c->err_row; A.big_slot = c->err_slot;
  // (lds_e = 64 (2 n_x + n_u) doubles per wave; a workgroup-wide interval stages rows of 264 doubles)
  return launch(c, K_ERR, A, (unsigned)(c->n_erriv / PK_WAVES_PER_BLOCK),
                sizeof(double) * ((size_t)c->md.lds_e / PK_WAVE) * 264, pick(c, stream));
}

int pk_eval_mesh_error(pk_ctx* c, const double* x, double* T, double* I) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !T || !I) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  PK_HIP(c, hipMemcpyAsync(c->d_x, x, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
  if ((rc = pk_eval_mesh_error_dev(c, c->d_x, c->d_errT, c->d_errI, nullptr))) return rc;
  PK_HIP(c, hipMemcpyAsync(T, c->d_errT, sizeof(double) * (size_t)c->n_err_out, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipMemcpyAsync(I, c->d_errI, sizeof(double) * (size_t)c->n_err_out, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

int pk_eval_hessc(pk_ctx* c, const double* x, const double* lambda, double sigma, double* vals) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !lambda || !vals) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  PK_HIP(c, hipMemcpyAsync(c->d_x, x, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
  PK_HIP(c, hipMemcpyAsync(c->d_lam, lambda, sizeof(double) * (size_t)c->m, hipMemcpyHostToDevice, c->stream));
  if ((rc = pk_eval_hessc_dev(c, c->d_x, c->d_lam, sigma, c->d_Hc, nullptr))) return rc;
  PK_HIP(c, hipMemcpyAsync(vals, c->d_Hc, sizeof(double) * (size_t)c->nnz_Hc, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

int pk_exchange_sums_dev(pk_ctx* c, const double* d_x, double* d_grad, double* d_f, int epoch, int write_f, void* stream);

int pk_eval_cycle_dev(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_f, double* d_grad,
                      double* d_g, double* d_jac, double* d_hess, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (!d_lam) return fail(c, 50, "pk_eval_cycle: lambda is required");
  hipStream_t st = pick(c, stream);
  const bool needs_I = c->md.prepass_grad || c->md.prepass_g || c->md.prepass_jac || c->md.prepass_hess || c->separate_x;
  // general path: the five callbacks one after the other.  A shard (pk_set_shard) may take the single launch too: its
  // finalize workgroup then leaves THIS shard's share of the integrals and of the shared gradient slots for the
  // caller's all-reduce, and f is the caller's to recompute (pk_eval_f_from_integrals_dev).
  if (c->cycle_layout && (needs_I || c->has_big || c->cycle_mode != 1))
    return fail(c, 69, "pk_eval_cycle: the compact layouts ride in the single-launch cycle only");
  if (needs_I && c->has_big) {       // (big intervals: the fused x-kernel behind the integral prepass, then H)
    if ((rc = pk_eval_xpart_dev(c, d_x, d_f, d_grad, d_g, d_jac, stream))) return rc;
    return pk_eval_hess_dev(c, d_x, d_lam, sigma, d_hess, stream);
  }
  if (needs_I || ((c->external_prepass || c->shard_flags) && c->cycle_mode != 1)) {
    if (!c->external_prepass && (rc = pk_eval_f_dev(c, d_x, d_f, stream))) return rc;
    if ((rc = pk_eval_grad_dev(c, d_x, d_grad, stream))) return rc;
    if ((rc = pk_eval_g_dev(c, d_x, d_g, stream))) return rc;
    if ((rc = pk_eval_jac_dev(c, d_x, d_jac, stream))) return rc;
    return pk_eval_hess_dev(c, d_x, d_lam, sigma, d_hess, stream);
  }
  // fused path: every x-only output from one evaluation of each node, then H (whose boundary workgroup also
  // performs the reductions).  With pk_set_cycle_graph the two launches are replayed from a cached hipGraph as
  // long as the pointers, sigma and the stream stay the same (an NLP solver's steady state).
  const pk_ctx::CycleKey key{d_x, d_lam, d_f, d_grad, d_g, d_jac, d_hess, sigma, st};
  // (a sharded cycle replays too: its exchange counts the cycles in device memory, and every pk_set_* call that changes
  //  a launch argument -- shard flags, shared-slot target, integral buffer, exchange form -- drops the captured graph)
  const bool graph = c->use_graph && c->profile_mask == 0;
  if (graph && c->cyc_exec && c->cyc_key == key) {
    PK_HIP(c, hipGraphLaunch(c->cyc_exec, st));
    return 0;
  }
  if (graph) {
    drop_cycle_graph(c);
    PK_HIP(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  }
  rc = enqueue_fused_cycle(c, d_x, d_lam, sigma, d_f, d_grad, d_g, d_jac, d_hess, st);
  if (!graph) return rc;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(st, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return fail(c, 3, "pk_eval_cycle: graph capture failed: %s", hipGetErrorString(e));
  e = hipGraphInstantiate(&c->cyc_exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { c->cyc_exec = nullptr; return fail(c, 3, "pk_eval_cycle: graph instantiation failed: %s", hipGetErrorString(e)); }
  c->cyc_key = key;
  PK_HIP(c, hipGraphLaunch(c->cyc_exec, st));
  return 0;
}

// `count` back-to-back cycles on the same buffers, enqueued from here (a solver written against the C ABI launches from
// compiled code; bench.py's timed batches use this so that a Python loop does not pace the stream).  xchg = 1: every cycle
// is followed by pk_exchange_sums_dev(d_x, d_xgrad, d_f) -- the two-launch form of a sharded cycle.
int pk_eval_cycle_dev_repeat(pk_ctx* c, const double* d_x, const double* d_lam, double sigma, double* d_f, double* d_grad,
                             double* d_g, double* d_jac, double* d_hess, void* stream, int count, int xchg, double* d_xgrad) {
  int rc = ready(c);
  if (rc) return rc;
  if (count < 0 || (xchg && !d_xgrad)) return fail(c, 50, "pk_eval_cycle_dev_repeat: bad count, or xchg without the gradient buffer of the exchange");
  // pk_set_cycle_graph(1): the whole batch is ONE hipGraph of `count` kernel nodes, captured once and replayed while
  // pointers, sigma, stream and count stay the same -- the host then pays one graph launch per batch instead of `count`
  // kernel launches.  Sharded cycles with the in-launch exchange included (the cycle number lives in device memory).
  const bool needs_I = c->md.prepass_grad || c->md.prepass_g || c->md.prepass_jac || c->md.prepass_hess || c->separate_x;
  const bool graph = c->use_graph && c->profile_mask == 0 && !xchg && !needs_I && count > 1 && c->cycle_mode == 1 &&
                     !c->external_prepass && d_lam;
  if (graph) {
    hipStream_t st = pick(c, stream);
    const pk_ctx::CycleKey key{d_x, d_lam, d_f, d_grad, d_g, d_jac, d_hess, sigma, st};
    if (!(c->rep_exec && c->rep_key == key && c->rep_count == count)) {
      if (c->rep_exec) { (void)hipGraphExecDestroy(c->rep_exec); c->rep_exec = nullptr; }
      PK_HIP(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < count && !rc; ++k) rc = enqueue_single_launch_cycle(c, d_x, d_lam, sigma, d_f, d_grad, d_g, d_jac, d_hess, st);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(st, &g);
      if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      if (e != hipSuccess) return fail(c, 3, "pk_eval_cycle_dev_repeat: graph capture failed: %s", hipGetErrorString(e));
      e = hipGraphInstantiate(&c->rep_exec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (e != hipSuccess) { c->rep_exec = nullptr; return fail(c, 3, "pk_eval_cycle_dev_repeat: graph instantiation failed: %s", hipGetErrorString(e)); }
      c->rep_key = key;
      c->rep_count = count;
    }
    PK_HIP(c, hipGraphLaunch(c->rep_exec, st));
    return 0;
  }
  for (int k = 0; k < count; ++k) {
    rc = pk_eval_cycle_dev(c, d_x, d_lam, sigma, d_f, d_grad, d_g, d_jac, d_hess, stream);
    if (!rc && xchg) rc = pk_exchange_sums_dev(c, d_x, d_xgrad, d_f, 0, 1, stream);
    if (rc) return rc;
  }
  return 0;
}

// The four x-only outputs (f, grad f, g, J) of one iterate on device pointers: the fused x-kernel (every node evaluated
// once, one joint CSE) + the one-workgroup reduction -- what a line search's trial point needs, and what the host shim
// runs on a new x.  A shard (pk_set_shard) leaves ITS share of the integrals in the integral buffer and its partial sums
// in the shared gradient slots; its f is not meaningful (the caller adds the integrals over the shards first).  Models
// whose system functions are nonlinear in the integrals run the callbacks one after the other.
// can the x-part of an iterate come from ONE pk_cycle launch without its Hessian role?
static bool xpart_is_one_launch(const pk_ctx* c) {
  const bool needs_I = c->md.prepass_grad || c->md.prepass_g || c->md.prepass_jac || c->md.prepass_hess || c->separate_x;
  return c->xpart_single && c->cycle_mode == 1 && !needs_I && c->profile_mask == 0 && !(c->xc_inline && c->xc_world > 1);
}

int pk_eval_xpart_dev(pk_ctx* c, const double* d_x, double* d_f, double* d_grad, double* d_g, double* d_jac, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  hipStream_t st = pick(c, stream);
  const bool needs_I = c->md.prepass_grad || c->md.prepass_g || c->md.prepass_jac || c->md.prepass_hess || c->separate_x;
  if (needs_I && !c->has_big) {
    if (!c->external_prepass && (rc = pk_eval_f_dev(c, d_x, d_f, stream))) return rc;
    if ((rc = pk_eval_grad_dev(c, d_x, d_grad, stream))) return rc;
    if ((rc = pk_eval_g_dev(c, d_x, d_g, stream))) return rc;
    return pk_eval_jac_dev(c, d_x, d_jac, stream);
  }
  // (a mesh with intervals of more than 64 points has the fused x-kernel only: the integrals it needs come from the
  //  integral prepass in front of it)
  if (needs_I && !c->external_prepass && (rc = prepass(c, d_x, nullptr, 0.0, d_f, true, st))) return rc;
  // ONE launch instead of pk_xall + pk_fin: pk_cycle's grid without its Hessian role -- the partial sums reach the finalize
  // workgroup inside the launch (host shim at 12k nodes: f is in pinned memory ~5 us earlier, and so is everything behind it)
  // (a shard takes it too, like pk_eval_cycle_dev: its finalize workgroup leaves THIS shard's share of the integrals and of
  //  the shared gradient slots for the caller to add up; never with the in-launch exchange, which belongs to whole cycles)
  if (xpart_is_one_launch(c))
    return enqueue_single_launch_cycle(c, d_x, nullptr, 0.0, d_f, d_grad, d_g, d_jac, nullptr, st, 0);
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_f = d_f; A.o_grad = d_grad; A.o_g = d_g; A.o_jac = d_jac;
  A.items = (const PkItem*)c->d_items_jac;
  A.n_items = c->n_items_jac;
  size_t lds = sizeof(double) * PK_WAVES_PER_BLOCK * (size_t)c->md.lds_x;
  if (lds < sizeof(double) * (size_t)c->md.ne_j) lds = sizeof(double) * (size_t)c->md.ne_j;
  A.flags |= xall_flags(c);
  if ((rc = launch(c, K_XALL, A, xall_blocks(c), lds, st))) return rc;
  // (a shard of a model nonlinear in the integrals: the caller has summed the integrals over the ranks, they stay as they are)
  A.flags |= ((needs_I && c->external_prepass) ? 0 : (F_FIN_INT | F_WRITE_F)) | F_FIN_GRAD;
  return launch(c, K_FIN, A, 1, 0, st);
}

// Which layouts pk_eval_cycle_dev writes into d_jac / d_hess: 0 the reference's triplets (default), 1 the compact layout
// (nnz_Jc / nnz_Hc values, the structures of pk_set_problem).  The compact layouts ride in the SAME single launch: the
// Jacobian role runs pk_jacc's tile code, the Hessian workgroups pk_hessc's -- a compact cycle is one launch too.
int pk_set_cycle_layout(pk_ctx* c, int jac_compact, int hess_compact) {
  int rc = ready(c);
  if (rc) return rc;
  if (jac_compact && c->nnz_Jc <= 0) return fail(c, 52, "pk_set_cycle_layout: no compact Jacobian layout was supplied to pk_set_problem");
  if (hess_compact && c->nnz_Hc <= 0) return fail(c, 51, "pk_set_cycle_layout: no compact Hessian layout was supplied to pk_set_problem");
  const bool needs_I = c->md.prepass_grad || c->md.prepass_g || c->md.prepass_jac || c->md.prepass_hess || c->separate_x;
  if ((jac_compact || hess_compact) && (needs_I || c->has_big))
    return fail(c, 69, "pk_set_cycle_layout: the compact layouts ride in the single-launch cycle, which this model / mesh does not "
                       "use (system functions nonlinear in the integrals, or an interval with more than 64 points)");
  c->cycle_layout = (jac_compact ? 1 : 0) | (hess_compact ? 2 : 0);
  drop_cycle_graph(c);
  if (c->rep_exec) { (void)hipGraphExecDestroy(c->rep_exec); c->rep_exec = nullptr; }
  return 0;
}

// 1 (default): the cycle is ONE launch (pk_cycle); 0: two launches (pk_xall, then pk_hess with the reductions)
int pk_set_cycle_mode(pk_ctx* c, int single_launch) {
  if (!c) return fail(nullptr, 1, "null context");
  c->cycle_mode = single_launch ? 1 : 0;
  drop_cycle_graph(c);
  return 0;
}

int pk_set_cycle_graph(pk_ctx* c, int enable) {
  if (!c) return fail(nullptr, 1, "null context");
  c->use_graph = enable != 0;
  if (!c->use_graph) drop_cycle_graph(c);
  return 0;
}

int pk_sync(pk_ctx* c, void* stream) {
  if (!c) return fail(nullptr, 1, "null context");
  PK_HIP(c, hipStreamSynchronize(pick(c, stream)));
  return handoff_check(c);
}

// the same by polling the stream's state: the host learns ~8 us earlier than through hipStreamSynchronize that a copy has
// landed (tools/dma_probe.cpp); for callers on a latency path (the host-landed sharded cycle)
int pk_wait_idle(pk_ctx* c, void* stream) {
  if (!c) return fail(nullptr, 1, "null context");
  hipStream_t st = pick(c, stream);
  hipError_t e;
  while ((e = hipStreamQuery(st)) == hipErrorNotReady) { }
  if (e != hipSuccess) return fail(c, 100 + (int)e, "hipStreamQuery failed: %s", hipGetErrorString(e));
  return handoff_check(c);
}

// ---------------------------------------------------------------- host-buffer API
#define PK_HOST_EVAL(IN_COPY, CALL, D_OUT, OUT, COUNT)                                                      \
  int rc = ready(c);                                                                                        \
  if (rc) return rc;                                                                                        \
  if (!x || !(OUT)) return fail(c, 60, "null host buffer");                                                 \
  PK_HIP(c, hipSetDevice(c->device));                                                                       \
  c->x_valid = false; /* the context's x and result buffers now hold another evaluation */                  \
  PK_HIP(c, hipMemcpyAsync(c->d_x, x, sizeof(double) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));    \
  IN_COPY;                                                                                                  \
  if ((rc = (CALL))) return rc;                                                                             \
  PK_HIP(c, hipMemcpyAsync((OUT), (D_OUT), sizeof(double) * (size_t)(COUNT), hipMemcpyDeviceToHost, c->stream)); \
  PK_HIP(c, hipStreamSynchronize(c->stream));                                                               \
  return handoff_check(c);

int pk_eval_f(pk_ctx* c, const double* x, double* f) { PK_HOST_EVAL((void)0, pk_eval_f_dev(c, c->d_x, c->d_f, nullptr), c->d_f, f, 1) }

int pk_eval_grad(pk_ctx* c, const double* x, double* grad) {
  PK_HOST_EVAL((void)0, pk_eval_grad_dev(c, c->d_x, c->d_grad, nullptr), c->d_grad, grad, c->n)
}

int pk_eval_g(pk_ctx* c, const double* x, double* g) { PK_HOST_EVAL((void)0, pk_eval_g_dev(c, c->d_x, c->d_g, nullptr), c->d_g, g, c->m) }

int pk_eval_jac(pk_ctx* c, const double* x, double* vals) {
  PK_HOST_EVAL((void)0, pk_eval_jac_dev(c, c->d_x, c->d_J, nullptr), c->d_J, vals, c->nnz_J)
}

int pk_eval_hess(pk_ctx* c, const double* x, const double* lambda, double sigma, double* vals) {
  if (c && !lambda) return fail(c, 50, "pk_eval_hess: lambda is required");
  PK_HOST_EVAL(PK_HIP(c, hipMemcpyAsync(c->d_lam, lambda, sizeof(double) * (size_t)c->m, hipMemcpyHostToDevice, c->stream)),
               pk_eval_hess_dev(c, c->d_x, c->d_lam, sigma, c->d_H, nullptr), c->d_H, vals, c->nnz_H)
}

int pk_eval_cycle(pk_ctx* c, const double* x, const double* lambda, double sigma, double* f, double* grad, double* g,
                  double* jac, double* hess) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !lambda || !f || !grad || !g || !jac || !hess) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  if ((rc = stage_upload(c, c->h_xs, c->ev_xs, c->xs_seq, c->xbuf, x, c->d_x, (size_t)c->n, nullptr))) return rc;
  if ((rc = stage_upload(c, c->h_lams, c->ev_lams, c->lams_seq, c->lambuf, lambda, c->d_lam, (size_t)c->m, nullptr))) return rc;
  if ((rc = pk_eval_cycle_dev(c, c->d_x, c->d_lam, sigma, c->d_f, c->d_grad, c->d_g, c->d_J, c->d_H, nullptr))) return rc;
  PK_HIP(c, hipMemcpyAsync(f, c->d_f, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipMemcpyAsync(grad, c->d_grad, sizeof(double) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipMemcpyAsync(g, c->d_g, sizeof(double) * (size_t)c->m, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipMemcpyAsync(jac, c->d_J, sizeof(double) * (size_t)c->nnz_J, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipMemcpyAsync(hess, c->d_H, sizeof(double) * (size_t)c->nnz_H, hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return handoff_check(c);
}

// ---------------------------------------------------------------- host shim: the "new x" protocol
// IPOPT evaluates f, grad f, g, J separately but on the same iterate (ipopt.py:41-53 hands the five methods of the
// problem object to cyipopt): pk_prepare_x uploads a new x ONCE, runs the fused x-kernel and -- prefetch mode -- queues
// the copy of every result into pinned host memory right behind it, in the order a solver asks for them; pk_fetch then
// only waits for the event of its result.  Nothing in here synchronizes the stream: the staging buffers of x and lambda
// are double-buffered and guarded by events, the results by one event each.
// 1 if x equals the x of the last pk_prepare_x bit for bit (the results held for it are still valid), else 0
int pk_same_x(pk_ctx* c, const double* x) {
  if (!c || !c->have_problem || !x || !c->x_valid || !c->h_x) return 0;
  return pk_same_bits(c->h_x, x, (size_t)c->n);      // (with the helper threads of pk_host_threads, if the caller started any)
}

}  // extern "C"

// ---------------------------------------------------------------- helper threads for the host's passes over x and lambda
// The solver's thread compares x with the prepared iterate in every callback and copies x / lambda into staging memory: one
// pass over n doubles each (14 us per 0.77 MB).  For one GPU they hide behind the transfers; for the host-landed sharded
// cycle at N times the size they are rank 0's serial part.  pk_host_threads(k) starts k helpers that
// take slices of such a pass.  A helper is "hot" (spinning on its mailbox) for 1 ms after the pool was last used and only
// hot helpers are given work -- the caller never waits for a thread to wake up; cold helpers look at an activity counter
// every 20 us (no condition variable, nothing to miss).  The caller always takes a slice itself and finishes alone when
// no helper is hot.  One caller at a time (the solver's thread).
namespace {
struct HostPool {
  struct alignas(128) Box {
    std::atomic<uint64_t> posted{0}, done{0};
    std::atomic<int> hot{0};
    int op = 0;                       // 0 compare, 1 copy
    const char* a = nullptr;
    char* b = nullptr;
    size_t bytes = 0;
    std::atomic<int> differs{0};
  };
  std::vector<std::unique_ptr<Box>> box;
  std::vector<std::thread> th;
  std::atomic<uint64_t> activity{0};
  std::atomic<bool> stop{false};
  std::atomic<long> jobs{0};          // slices executed by helpers (diagnostics)
  int slow_waits = 0;                 // passes in which the caller waited more than 1 ms for a helper (a host whose
  bool given_up = false;              // "CPUs" are time slices of fewer cores): after three of them the helpers are left alone

  static void relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
  }
  void work(Box* bx) {
    using clock = std::chrono::steady_clock;
    uint64_t seen = 0, act = activity.load(std::memory_order_acquire);
    auto hot_until = clock::now();
    for (;;) {
      const uint64_t p = bx->posted.load(std::memory_order_acquire);
      if (p != seen) {
        if (bx->op == 0) { if (std::memcmp(bx->a, bx->b, bx->bytes) != 0) bx->differs.store(1, std::memory_order_relaxed); }
        else std::memcpy(bx->b, bx->a, bx->bytes);
        seen = p;
        jobs.fetch_add(1, std::memory_order_relaxed);
        bx->done.store(p, std::memory_order_release);
        hot_until = clock::now() + std::chrono::milliseconds(1);
        continue;
      }
      if (stop.load(std::memory_order_acquire)) return;
      if (bx->hot.load(std::memory_order_relaxed)) {
        for (int i = 0; i < 64; ++i) relax();
        const uint64_t a2 = activity.load(std::memory_order_acquire);
        if (a2 != act) { act = a2; hot_until = clock::now() + std::chrono::milliseconds(1); }
        else if (clock::now() > hot_until) {
          bx->hot.store(0, std::memory_order_seq_cst);       // (a job posted while this store was on its way is seen by the
          continue;                                          //  next pass of the loop: posted is read first)
        }
      } else {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        const uint64_t a2 = activity.load(std::memory_order_acquire);
        if (a2 != act) {
          act = a2;
          hot_until = clock::now() + std::chrono::milliseconds(1);
          bx->hot.store(1, std::memory_order_seq_cst);
        }
      }
    }
  }
  explicit HostPool(int k) {
    for (int i = 0; i < k; ++i) box.emplace_back(new Box());
    for (int i = 0; i < k; ++i) th.emplace_back([this, i]() { work(box[(size_t)i].get()); });
  }
  ~HostPool() {
    stop.store(true, std::memory_order_release);
    for (auto& t : th) t.join();
  }
  // op over [a, a + bytes) / [b, b + bytes); returns 1 if a compare found a difference.  While no helper is hot the caller
  // works through the pass in 256 KB pieces itself (a cold helper needs up to 20 us to notice the activity); as soon as
  // some are, what is left is cut into one slice each (whole 4 KB pages) and the caller takes the first.
  int run(int op, const char* a, char* b, size_t bytes) {
    activity.fetch_add(1, std::memory_order_release);
    auto one = [op](const char* pa, char* pb, size_t len) -> int {
      if (op == 0) return std::memcmp(pa, pb, len) != 0;
      std::memcpy(pb, pa, len);
      return 0;
    };
    size_t at = 0;
    std::vector<Box*> use;
    use.reserve(box.size());
    while (at < bytes) {
      use.clear();
      if (!given_up)
        for (auto& bx : box)
          if (bx->hot.load(std::memory_order_seq_cst)) use.push_back(bx.get());
      const size_t left = bytes - at;
      if (use.empty() || left <= ((size_t)256 << 10)) {
        const size_t len = left < ((size_t)256 << 10) ? left : ((size_t)256 << 10);
        if (one(a + at, b + at, len)) return 1;
        at += len;
        continue;
      }
      const size_t parts = use.size() + 1;
      const size_t slice = (left / parts + 4095) & ~(size_t)4095;
      size_t pos = at + slice;                         // [at, at + slice) is the caller's
      std::vector<std::pair<Box*, uint64_t>> waits;
      waits.reserve(use.size());
      for (Box* bx : use) {
        if (pos >= bytes) break;
        const size_t len = bytes - pos < slice ? bytes - pos : slice;
        bx->op = op; bx->a = a + pos; bx->b = b + pos; bx->bytes = len;
        bx->differs.store(0, std::memory_order_relaxed);
        const uint64_t id = bx->posted.load(std::memory_order_relaxed) + 1;
        bx->posted.store(id, std::memory_order_release);
        waits.emplace_back(bx, id);
        pos += len;
      }
      int differs = one(a + at, b + at, slice < left ? slice : left);
      if (pos < bytes) differs |= one(a + pos, b + pos, bytes - pos);
      const auto t_wait = std::chrono::steady_clock::now();
      bool slow = false;
      for (auto& w : waits) {                          // (always: a helper must not be left reading the caller's arrays)
        for (unsigned spins = 1; w.first->done.load(std::memory_order_acquire) != w.second; ++spins) {
          relax();
          if (!slow && (spins & 0xFF) == 0 && std::chrono::steady_clock::now() - t_wait > std::chrono::milliseconds(1)) slow = true;
        }
        differs |= w.first->differs.load(std::memory_order_relaxed);
      }
      if (slow && ++slow_waits >= 3) given_up = true;
      return differs;
    }
    return 0;
  }
};
std::unique_ptr<HostPool> g_pool;
constexpr size_t kPoolMinBytes = (size_t)256 << 10;   // shorter passes are done by the caller alone
}  // namespace

extern "C" {

// k helper threads for the host's passes over x / lambda (0: none, the default; at most 16).  Process-wide; call it from
// the thread that calls pk_same_bits / pk_copy_bits / the callbacks, while none of them is running.
int pk_host_threads(int k) {
  if (k < 0 || k > 16) return fail(nullptr, 69, "pk_host_threads: between 0 and 16 helper threads");
  g_pool.reset();
  if (k > 0) g_pool.reset(new HostPool(k));
  return 0;
}

long pk_host_threads_jobs(void) { return g_pool ? g_pool->jobs.load(std::memory_order_relaxed) : 0; }   // slices helpers have taken

// how many helpers are spinning right now (diagnostics)
int pk_host_threads_hot(void) {
  int k = 0;
  if (g_pool) {
    if (g_pool->given_up) return -1;       // (the caller waited a millisecond for a helper three times: not used any more)
    for (auto& bx : g_pool->box) k += bx->hot.load(std::memory_order_relaxed);
  }
  return k;
}

// 1 if the two arrays of n doubles are equal bit for bit (what decides "is this the iterate I already evaluated": one pass
// at memcmp speed, no temporary -- numpy.array_equal builds a boolean array of n elements first)
int pk_same_bits(const double* a, const double* b, size_t n) {
  if (!a || !b) return 0;
  const size_t bytes = sizeof(double) * n;
  // (iterates that differ usually differ at the front: look there before anybody else is asked to help)
  const size_t head = bytes < 4096 ? bytes : 4096;
  if (g_pool) g_pool->activity.fetch_add(1, std::memory_order_release);      // (a callback is running: helpers, get ready)
  if (std::memcmp(a, b, head) != 0) return 0;
  if (g_pool && bytes >= kPoolMinBytes)
    return g_pool->run(0, (const char*)a + head, (char*)const_cast<double*>(b) + head, bytes - head) ? 0 : 1;
  return std::memcmp((const char*)a + head, (const char*)b + head, bytes - head) == 0 ? 1 : 0;
}

// dst[0 .. n) = src[0 .. n) (non-overlapping), with the helper threads of pk_host_threads when the arrays are large
int pk_copy_bits(double* dst, const double* src, size_t n) {
  if (!dst || !src) return fail(nullptr, 60, "null host buffer");
  const size_t bytes = sizeof(double) * n;
  if (g_pool && bytes >= kPoolMinBytes) (void)g_pool->run(1, (const char*)src, (char*)dst, bytes);
  else std::memcpy(dst, src, bytes);
  return 0;
}

// the context's x / result buffers were used for something else (mesh error, one-shot evals, the cycle call)
int pk_invalidate_x(pk_ctx* c) {
  if (!c) return fail(nullptr, 1, "null context");
  c->x_valid = false;
  return 0;
}

// Where the results of the NEXT pk_prepare_x / pk_eval_hess_prepared land: pinned host memory of the caller (from
// pk_host_alloc), NULL = the context's own pinned buffer of that output (pk_host_buffer).
int pk_set_result_targets(pk_ctx* c, double* f, double* grad, double* g, double* jac, double* hess) {
  int rc = ready(c);
  if (rc) return rc;
  double* t[5] = {f, grad, g, jac, hess};
  c->target_filled = false;       // (an arbitrary array of the caller's: the whole Jacobian is copied into it)
  for (int k = 0; k < 5; ++k) {
    c->target[k] = t[k];
    c->target_pinned[k] = false;
    // A kernel may store into a target only if the device can see it (pinned / registered host memory); a pageable
    // target still works as the destination of a copy.
    c->target_visible[k] = true;
    if (t[k]) {
      hipPointerAttribute_t attr;
      std::memset(&attr, 0, sizeof attr);
      const hipError_t e = hipPointerGetAttributes(&attr, t[k]);
      if (e != hipSuccess) (void)hipGetLastError();
      c->target_visible[k] = e == hipSuccess && (attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice ||
                                                 attr.type == hipMemoryTypeManaged);
    }
  }
  return 0;
}

// prefetch = 1 (default): every x-only result is copied to the host right behind the kernel; 0: a result is copied
// when it is first asked for (a request for the gradient also queues the Jacobian -- a solver wants both at an accepted
// point, and neither at a rejected trial point).  host_direct = 1: the kernels store f / grad / g / J (and H) straight
// into the pinned host targets over PCIe, no device-side staging and no DMA (A/B switch).
int pk_set_host_mode(pk_ctx* c, int prefetch, int host_direct) {
  if (!c) return fail(nullptr, 1, "null context");
  c->prefetch = prefetch ? 1 : 0;
  c->host_direct = host_direct ? 1 : 0;
  c->x_valid = false;
  return 0;
}

// Pinned (page-locked, device-visible) host memory for result arrays that outlive a call: process-wide, not tied to a
// context (a host array handed to the solver may outlive the evaluator that filled it).
int pk_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(nullptr, 60, "null host buffer");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault);
  if (e != hipSuccess) return fail(nullptr, 100 + (int)e, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return 0;
}

int pk_host_free(void* p) {
  if (!p) return 0;
  // (a landing block may still be the target of copies nobody asked for -- every result of a new x is on its way into it:
  //  nothing may be in flight when the memory goes; found by the sanitized build, tests/fake_hip)
  (void)hipDeviceSynchronize();
  hipError_t e = hipHostFree(p);
  if (e != hipSuccess) return fail(nullptr, 100 + (int)e, "hipHostFree failed: %s", hipGetErrorString(e));
  return 0;
}

namespace {
bool small_x(const pk_ctx* c) { return c->small_direct && sizeof(double) * (size_t)c->n <= ((size_t)c->small_x_kb << 10); }
bool small_results(const pk_ctx* c) {
  const size_t nj = (size_t)(c->jac_compact ? c->nnz_Jc : c->nnz_J);
  return c->small_direct && sizeof(double) * (nj + (size_t)c->n + (size_t)c->m) <= ((size_t)1 << 20);
}
bool hess_goes_direct(const pk_ctx* c) {
  return c->hess_direct && sizeof(double) * (size_t)c->nnz_H <= ((size_t)c->kernel_download << 20);
}

// multipliers of the next Hessian: staged in pinned memory; uploaded by DMA, or -- lambda_direct -- left there for the
// Hessian kernel to read over PCIe itself (0.77 MB: DMA + kernel 34 us, kernel reading pinned memory 26 us)
int stage_lambda(pk_ctx* c, const double* lambda) {
  double* staged = nullptr;
  const bool direct = c->lambda_direct != 0 && sizeof(double) * (size_t)c->m <= ((size_t)2 << 20);
  int rc = stage_upload(c, c->h_lams, c->ev_lams, c->lams_seq, c->lambuf, lambda, direct ? nullptr : c->d_lam, (size_t)c->m, &staged);
  if (rc) return rc;
  c->lam_src = direct ? staged : c->d_lam;
  c->lam_staged = true;
  return 0;
}

// a landing block of the caller's for the x-results of the next new iterate: [J (nnz_J) | grad f (n) | g (m)]
void take_block(pk_ctx* c, double* block) {
  const int64_t nj = c->jac_compact ? c->nnz_Jc : c->nnz_J;      // (a block follows the layout the Jacobian callback serves)
  c->target[3] = block;
  c->target[1] = block ? block + nj : nullptr;
  c->target[2] = block ? block + nj + c->n : nullptr;
  c->target_visible[1] = c->target_visible[2] = c->target_visible[3] = !c->host_direct;   // (copy targets; see pk_set_result_targets)
  c->target_filled = block != nullptr;      // (the contract of pk_callback_x: blocks were filled by pk_fill_jac_constants)
  c->target_pinned[1] = c->target_pinned[2] = c->target_pinned[3] = block != nullptr;      // (... and are pinned memory)
}
}  // namespace

int pk_prepare_x(pk_ctx* c, const double* x) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  const bool sx = small_x(c), sr = small_results(c);
  if ((rc = stage_upload(c, c->h_xs, c->ev_xs, c->xs_seq, c->xbuf, x, sx ? nullptr : c->d_x, (size_t)c->n, &c->h_x))) return rc;
  // (small x: every kernel of this iterate -- the x-part now, the Hessian later -- reads the staging buffer in place.  The
  //  buffer is written again two stagings from now; every kernel of this iterate has been awaited by then: a callback returns
  //  only when its result has landed, a discarded speculative launch is synchronized, and stage_upload waits for the stream
  //  if nothing since the buffer's staging has been seen idle)
  c->x_src = sx ? c->h_x : c->d_x;
  for (int k = 0; k < 5; ++k) {
    c->landed[k] = c->target[k] ? c->target[k] : c->h_out[k];
    c->enq[k] = c->done[k] = false;
  }
  c->jac_filled = !c->jconst.empty() && (c->target[3] ? c->target_filled : true) && !c->host_direct && !sr;
  c->early_valid = false;
  double* o[4];
  for (int k = 0; k < 4; ++k) {     // (f: stored by the kernel itself whenever its landing place is device-visible)
    c->stored_direct[k] = (c->host_direct || sr || k == 0) && (!c->target[k] || c->target_visible[k]);
    o[k] = c->stored_direct[k] ? c->landed[k] : device_result(c, k);
  }
  if (c->stored_direct[0]) *(volatile unsigned long long*)c->landed[0] = (unsigned long long)PK_EMPTY;     // (see wait_result)
  if (c->jac_compact && !c->has_big && xpart_is_one_launch(c)) {
    // the compact layout of the Jacobian from the SAME launch: its Jacobian role runs pk_jacc's tile code (no reference-layout
    // J is written, no second kernel)
    if ((rc = enqueue_single_launch_cycle(c, c->x_src, nullptr, 0.0, o[0], o[1], o[2], o[3], nullptr, c->stream, 1))) return rc;
  } else {
    if ((rc = pk_eval_xpart_dev(c, c->x_src, o[0], o[1], o[2], c->jac_compact ? c->d_J : o[3], nullptr))) return rc;
    // the compact layout of the Jacobian: its own kernel behind the fused x-kernel (whose reference-layout J stays on the device)
    if (c->jac_compact && (rc = pk_eval_jacc_dev(c, c->x_src, o[3], nullptr))) return rc;
  }
  // f and g are what a line search asks for at every trial point: always on their way; grad f and J in prefetch mode
  const bool ahead = c->prefetch && (!c->adaptive_prefetch || c->cur_J_asked);
  c->cur_J_asked = false;
  if ((rc = enqueue_result_copies(c, ahead ? 0xFu : 0x5u))) return rc;
  c->x_valid = true;
  return 0;
}

// result `what` (0 f, 1 grad, 2 g, 3 jac) of the last pk_prepare_x: waits for its copy.  out == NULL: the result stays
// where it landed (pk_result_location); otherwise it is copied on to `out` (a second host copy).
int pk_fetch(pk_ctx* c, int what, double* out) {
  int rc = ready(c);
  if (rc) return rc;
  if (what < 0 || what > 3) return fail(c, 61, "pk_fetch: what must be 0 (f), 1 (grad), 2 (g) or 3 (jac)");
  if (!c->x_valid) return fail(c, 64, "pk_fetch: no prepared x (pk_prepare_x)");
  if (what == 1 || what == 3) c->cur_J_asked = true;
  if (!c->enq[what] && (rc = enqueue_result_copies(c, (1u << what) | (what == 1 ? 8u : 0u)))) return rc;   // (an accepted point: J follows grad f)
  if ((rc = wait_result(c, what))) return rc;
  if (out && out != c->landed[what]) std::memcpy(out, c->landed[what], sizeof(double) * result_count(c, what));
  return 0;
}

// ONE call per x-callback of a host shim (objective / gradient / constraints / jacobian of the cyipopt protocol,
// ipopt.py:41-53): if `x` is not the prepared iterate it becomes it -- its results landing in `block`, pinned memory of the
// caller's holding [J (nnz_J) | grad f (n) | g (m)] (NULL: the context's own buffers; f always lands in the context's
// pinned word) and *fresh = 1 -- then result `what` is waited for; f_out (may be NULL) receives f when what == 0.
int pk_callback_x(pk_ctx* c, int what, const double* x, double* block, double* f_out, int* fresh) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x) return fail(c, 60, "null host buffer");
  if (what < 0 || what > 3) return fail(c, 61, "pk_callback_x: what must be 0 (f), 1 (grad), 2 (g) or 3 (jac)");
  const bool same = pk_same_x(c, x) != 0;
  if (fresh) *fresh = same ? 0 : 1;
  if (!same) {
    c->target[0] = nullptr;
    take_block(c, block);
    if ((rc = pk_prepare_x(c, x))) return rc;
  }
  if ((rc = pk_fetch(c, what, nullptr))) return rc;
  if (what == 0 && f_out) *f_out = c->landed[0][0];
  return 0;
}

// ONE call for all five results of an iterate whose multipliers are known together with x (a solver written against the C
// ABI; Evaluator.cycle): x and lambda are staged like in the callbacks, the whole cycle is ONE launch (pk_cycle), the results
// land like the callbacks' -- [J (changing part) | grad f | g] in `block` (pinned, [J | grad f | g], constants filled in by
// pk_fill_jac_constants), H in `hess` (pinned), f in the context's pinned word -- and the call returns when all of it is
// there.  Reference layout of the Jacobian only.
int pk_callback_cycle(pk_ctx* c, const double* x, const double* lambda, double sigma, double* block, double* hess, double* f_out) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !lambda || !block || !hess) return fail(c, 60, "null host buffer");
  if (c->jac_compact) return fail(c, 68, "pk_callback_cycle: the one-launch cycle writes the reference layout of the Jacobian");
  PK_HIP(c, hipSetDevice(c->device));
  c->x_valid = false;
  if ((rc = stage_lambda(c, lambda))) return rc;
  c->target[0] = nullptr;
  take_block(c, block);
  c->target[4] = hess;
  c->target_visible[4] = !c->host_direct;
  c->target_pinned[4] = true;
  const bool sx = small_x(c), sr = small_results(c) && !c->host_direct;
  if ((rc = stage_upload(c, c->h_xs, c->ev_xs, c->xs_seq, c->xbuf, x, sx ? nullptr : c->d_x, (size_t)c->n, &c->h_x))) return rc;
  c->x_src = sx ? c->h_x : c->d_x;
  for (int k = 0; k < 5; ++k) {
    c->landed[k] = c->target[k] ? c->target[k] : c->h_out[k];
    c->enq[k] = c->done[k] = false;
  }
  c->jac_filled = !c->jconst.empty() && c->target_filled && !c->host_direct && !sr;
  c->early_valid = false;
  double* o[5];
  for (int k = 0; k < 5; ++k) {
    c->stored_direct[k] = k == 0 || (k == 4 && hess_goes_direct(c) && c->target_visible[4]) || (k >= 1 && k <= 3 && sr);
    o[k] = c->stored_direct[k] ? c->landed[k] : (k == 4 ? c->d_H : device_result(c, k));
  }
  *(volatile unsigned long long*)c->landed[0] = (unsigned long long)PK_EMPTY;
  if ((rc = pk_eval_cycle_dev(c, c->x_src, c->lam_src, sigma, o[0], o[1], o[2], o[3], o[4], nullptr))) return rc;
  c->lam_staged = false;
  if ((rc = enqueue_result_copies(c, 0x1Fu))) return rc;
  c->x_valid = true;
  c->cur_J_asked = true;
  if ((rc = wait_result(c, 4)) || (rc = wait_result(c, 3)) || (rc = wait_result(c, 0))) return rc;
  c->done[1] = c->done[2] = true;
  if (f_out) *f_out = c->landed[0][0];
  return 0;
}

// Queue the upload of the multipliers of the next pk_eval_hess_prepared and return: the caller's check of x
// (pk_same_x, a pass over n doubles) then runs while the DMA is in flight.
int pk_stage_lambda(pk_ctx* c, const double* lambda) {
  int rc = ready(c);
  if (rc) return rc;
  if (!lambda) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  return stage_lambda(c, lambda);
}

// Hessian on the x of the last pk_prepare_x (no re-upload of x); vals == NULL: the result stays where it landed;
// lambda == NULL: the multipliers staged by pk_stage_lambda
int pk_eval_hess_prepared(pk_ctx* c, const double* lambda, double sigma, double* vals) {
  int rc = ready(c);
  if (rc) return rc;
  if (!lambda && !c->lam_staged) return fail(c, 60, "null host buffer (no multipliers staged either)");
  if (!c->x_valid) return fail(c, 64, "pk_eval_hess_prepared: no prepared x (pk_prepare_x)");
  PK_HIP(c, hipSetDevice(c->device));
  if (lambda && (rc = stage_lambda(c, lambda))) return rc;
  c->lam_staged = false;
  c->landed[4] = c->target[4] ? c->target[4] : c->h_out[4];
  c->enq[4] = c->done[4] = false;
  c->stored_direct[4] = (c->host_direct || hess_goes_direct(c)) && (!c->target[4] || c->target_visible[4]);
  if ((rc = pk_eval_hess_dev(c, c->x_src, c->lam_src, sigma, c->stored_direct[4] ? c->landed[4] : c->d_H, nullptr))) return rc;
  if ((rc = enqueue_result_copies(c, 1u << 4))) return rc;
  if ((rc = wait_result(c, 4))) return rc;
  if (vals && vals != c->landed[4]) std::memcpy(vals, c->landed[4], sizeof(double) * (size_t)c->nnz_H);
  return 0;
}

// The compact Hessian layout on the x of the last pk_prepare_x: what a solver that was handed the compact structure calls
// instead of pk_eval_hess_prepared -- 6 ... 10 x fewer values over PCIe.  lambda == NULL: the
// multipliers staged by pk_stage_lambda.  vals_pinned = 1: `vals` is device-visible host memory (pk_host_alloc) and the
// DMA writes it directly; 0: the values land in a pinned buffer of the context and are copied on.
int pk_eval_hessc_prepared(pk_ctx* c, const double* lambda, double sigma, double* vals, int vals_pinned) {
  int rc = ready(c);
  if (rc) return rc;
  if (!vals) return fail(c, 60, "null host buffer");
  if (!lambda && !c->lam_staged) return fail(c, 60, "null host buffer (no multipliers staged either)");
  if (!c->x_valid) return fail(c, 64, "pk_eval_hessc_prepared: no prepared x (pk_prepare_x)");
  if (c->nnz_Hc <= 0) return fail(c, 51, "pk_eval_hessc: no compact Hessian layout was supplied to pk_set_problem");
  PK_HIP(c, hipSetDevice(c->device));
  if (lambda && (rc = stage_lambda(c, lambda))) return rc;
  c->lam_staged = false;
  if ((rc = pk_eval_hessc_dev(c, c->x_src, c->lam_src, sigma, c->d_Hc, nullptr))) return rc;
  const size_t bytes = sizeof(double) * (size_t)c->nnz_Hc;
  double* dst = vals;
  if (!vals_pinned) {
    if (!c->h_Hc) PK_HIP(c, hipHostMalloc((void**)&c->h_Hc, bytes, hipHostMallocDefault));
    dst = c->h_Hc;
  }
  const bool hc_by_kernel = bytes <= ((size_t)c->kernel_download << 20) && !(((uintptr_t)dst ^ (uintptr_t)c->d_Hc) & 8);
  if ((rc = copy_async(c, dst, c->d_Hc, (size_t)c->nnz_Hc, hipMemcpyDeviceToHost, hc_by_kernel))) return rc;
  ++c->op_seq;
  c->mark_pending = false;
  if (hc_by_kernel && (rc = enqueue_mark(c))) return rc;
  if (c->spin_wait) {      // (every earlier copy of this iterate has been waited for by its callback)
    if ((rc = wait_results_landed(c))) return rc;
  } else {
    PK_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (!vals_pinned) std::memcpy(vals, c->h_Hc, bytes);
  return 0;
}

// ONE call for the Hessian callback of a host shim (SystemBase.hessian, systembase.py:820-835): the multipliers are staged
// first (their upload, if any, runs while x is compared), a new x is prepared like in pk_callback_x (landing block
// `block`), then the Hessian of the Lagrangian is evaluated on the prepared x and waited for.  compact = 0: reference
// layout, `hess` = pinned landing place of nnz_H values (NULL: the context's buffer); compact = 1: the compact layout
// (pk_eval_hessc), `hess` = pinned landing place of nnz_Hc values (required).
int pk_callback_hess(pk_ctx* c, const double* x, const double* lambda, double sigma, double* block, double* hess, int compact,
                     int* fresh) {
  int rc = ready(c);
  if (rc) return rc;
  if (!x || !lambda) return fail(c, 60, "null host buffer");
  if (compact && !hess) return fail(c, 60, "pk_callback_hess: the compact layout needs a landing array");
  PK_HIP(c, hipSetDevice(c->device));
  if ((rc = stage_lambda(c, lambda))) return rc;
  if (!compact) {
    c->target[4] = hess;
    c->target_visible[4] = !c->host_direct;      // (by contract `hess` is pinned memory the device can address)
    c->target_pinned[4] = hess != nullptr;
  }
  if (fresh) *fresh = 0;
  // The solver's Hessian callback comes on the iterate the x-callbacks just ran on: launch on the prepared x at once and
  // compare x with it WHILE the GPU works (the compare is a pass over n doubles: 10 us at 12k nodes, 90 us at 40k).  A
  // different x discards the launch (its values are overwritten below) and takes the ordinary route.
  if (c->speculative_hess && c->x_valid && c->h_x) {
    const size_t bytes = sizeof(double) * (size_t)c->nnz_Hc;
    if (compact) {
      if ((rc = pk_eval_hessc_dev(c, c->x_src, c->lam_src, sigma, c->d_Hc, nullptr))) return rc;
      const bool hc_by_kernel = bytes <= ((size_t)c->kernel_download << 20) && !(((uintptr_t)hess ^ (uintptr_t)c->d_Hc) & 8);
      if ((rc = copy_async(c, hess, c->d_Hc, (size_t)c->nnz_Hc, hipMemcpyDeviceToHost, hc_by_kernel))) return rc;
      ++c->op_seq;
      c->mark_pending = false;
      if (hc_by_kernel && (rc = enqueue_mark(c))) return rc;
    } else {
      c->landed[4] = c->target[4] ? c->target[4] : c->h_out[4];
      c->enq[4] = c->done[4] = false;
      c->stored_direct[4] = (c->host_direct || hess_goes_direct(c)) && (!c->target[4] || c->target_visible[4]);
      if ((rc = pk_eval_hess_dev(c, c->x_src, c->lam_src, sigma, c->stored_direct[4] ? c->landed[4] : c->d_H, nullptr))) return rc;
      if ((rc = enqueue_result_copies(c, 1u << 4))) return rc;
    }
    const bool same = pk_same_bits(c->h_x, x, (size_t)c->n) != 0;
    const uint64_t seen = c->op_seq;
    if (same) {
      c->lam_staged = false;
      if (!compact) return wait_result(c, 4);
      return wait_results_landed(c);
    }
    PK_HIP(c, hipStreamSynchronize(c->stream));       // (the discarded launch must not write behind the one that follows)
    c->mark_pending = false;
    c->idle_seq = seen;
    c->x_valid = false;
  }
  const bool same = pk_same_x(c, x) != 0;
  if (fresh) *fresh = same ? 0 : 1;
  if (!same) {
    c->target[0] = nullptr;
    take_block(c, block);
    if ((rc = pk_prepare_x(c, x))) return rc;
  }
  if (compact) return pk_eval_hessc_prepared(c, nullptr, sigma, hess, 1);
  return pk_eval_hess_prepared(c, nullptr, sigma, nullptr);
}

// Runs [start[i], stop[i]) of the Jacobian values -- in the layout the shim serves, pk_set_jacobian_layout; every layout keeps
// its own -- that never change with x (ascending, disjoint): the x-results' copy to the
// host skips them from now on.  The landing arrays must hold those values already: pk_fill_jac_constants writes them into
// an array once (the context's own landing buffer is filled here).  n_runs = 0 restores the full copy.
// Reference: the translation part of the Jacobian, phasebase.py:1071-1081, is recomputed and returned by every call there.
int pk_set_jac_constant_runs(pk_ctx* c, int n_runs, const int64_t* start, const int64_t* stop) {
  int rc = ready(c);
  if (rc) return rc;
  if (n_runs < 0 || (n_runs > 0 && (!start || !stop))) return fail(c, 66, "pk_set_jac_constant_runs: bad arguments");
  int64_t at = 0;
  for (int i = 0; i < n_runs; ++i) {
    if (start[i] < at || stop[i] <= start[i] || stop[i] > (int64_t)result_count(c, 3))
      return fail(c, 66, "pk_set_jac_constant_runs: run %d [%lld, %lld) is out of order or out of range", i, (long long)start[i], (long long)stop[i]);
    at = stop[i];
  }
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  c->x_valid = false;
  c->jconst.clear();
  c->jruns.clear();
  at = 0;
  for (int i = 0; i < n_runs; ++i) {
    if (start[i] > at) c->jruns.emplace_back(at, start[i]);
    c->jconst.emplace_back(start[i], stop[i]);
    at = stop[i];
  }
  if (at < (int64_t)result_count(c, 3) || c->jruns.empty()) c->jruns.emplace_back(at, (int64_t)result_count(c, 3));
  if (n_runs == 0) return 0;
  // one evaluation of J (in the layout the shim serves) into the context's device buffer, whatever x it holds -- the
  // constant entries do not depend on it --, from which the constants are taken
  if ((rc = c->jac_compact ? pk_eval_jacc_dev(c, c->d_x, c->d_Jc, nullptr) : pk_eval_jac_dev(c, c->d_x, c->d_J, nullptr))) return rc;
  return pk_fill_jac_constants(c, c->h_out[3]);
}

// the x-independent runs of J (pk_set_jac_constant_runs) -> jac[...]; the other entries of `jac` are not touched
int pk_fill_jac_constants(pk_ctx* c, double* jac) {
  int rc = ready(c);
  if (rc) return rc;
  if (!jac) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  const double* dj = device_result(c, 3);
  for (const auto& r : c->jconst)
    PK_HIP(c, hipMemcpyAsync(jac + r.first, dj + r.first, sizeof(double) * (size_t)(r.second - r.first), hipMemcpyDeviceToHost, c->stream));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

// Which layout the Jacobian of the host shim (pk_prepare_x / pk_fetch(3) / pk_callback_x(3), the J part of a landing block)
// has: 0 the reference's triplets (default), 1 the compact layout of pk_eval_jacc.
int pk_set_jacobian_layout(pk_ctx* c, int compact) {
  int rc = ready(c);
  if (rc) return rc;
  if (compact && c->nnz_Jc <= 0) return fail(c, 52, "pk_set_jacobian_layout: no compact Jacobian layout was supplied to pk_set_problem");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  if ((compact != 0) != c->jac_compact) {      // every layout has constant runs of its own
    c->jruns.swap(c->jruns_other);
    c->jconst.swap(c->jconst_other);
  }
  c->jac_compact = compact != 0;
  c->x_valid = false;
  for (int k = 0; k < 5; ++k) c->target[k] = nullptr;
  return 0;
}

// A/B switches of the host shim (defaults: what measured fastest): "spin_wait" (1: results are awaited
// by polling, 0: hipEventSynchronize), "lambda_direct" (1: the Hessian kernel reads the staged multipliers from pinned
// memory itself, 0: they are uploaded first), "chunk_upload" (1: large inputs are staged and uploaded in chunks),
// "kernel_upload" / "kernel_download" (copy kernels instead of the DMA engine), "split_copy" (grad f | g ahead of J),
// "speculative_hess" (the Hessian is launched before x has been compared with the prepared iterate).
int pk_set_host_option(pk_ctx* c, const char* name, int value) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!name) return fail(c, 67, "pk_set_host_option: null name");
  if (c->have_problem) {
    PK_HIP(c, hipSetDevice(c->device));
    PK_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->x_valid = false;
  c->lam_staged = false;
  if (!std::strcmp(name, "spin_wait")) c->spin_wait = value != 0;
  else if (!std::strcmp(name, "lambda_direct")) c->lambda_direct = value != 0;
  else if (!std::strcmp(name, "chunk_upload")) c->chunk_upload = value != 0;
  else if (!std::strcmp(name, "kernel_upload")) c->kernel_upload = value != 0;
  else if (!std::strcmp(name, "kernel_download")) c->kernel_download = value < 0 ? 0 : (value > 4096 ? 4096 : value);
  else if (!std::strcmp(name, "split_copy")) c->split_copy = value != 0;
  else if (!std::strcmp(name, "speculative_hess")) c->speculative_hess = value != 0;
  else if (!std::strcmp(name, "hess_direct")) c->hess_direct = value != 0;
  else if (!std::strcmp(name, "xpart_single")) c->xpart_single = value != 0;
  else if (!std::strcmp(name, "separate_x")) {
    if (value && c->has_big) return fail(c, 67, "pk_set_host_option: separate_x needs a mesh without intervals of more than 64 points "
                                                "(such a mesh has the fused x-kernel only)");
    c->separate_x = value != 0;
    drop_cycle_graph(c);
  }
  else if (!std::strcmp(name, "small_direct")) c->small_direct = value != 0;
  else if (!std::strcmp(name, "small_x_kb")) c->small_x_kb = value < 0 ? 0 : (value > (1 << 20) ? (1 << 20) : value);
  else if (!std::strcmp(name, "adaptive_prefetch")) { c->adaptive_prefetch = value != 0; c->cur_J_asked = true; }
  else if (!std::strcmp(name, "mark_wait")) { c->mark_wait = value != 0; c->mark_pending = false; }
  else if (!std::strcmp(name, "poll_limit")) { c->poll_limit = value < 0 ? 0 : value; drop_cycle_graph(c); }
  else return fail(c, 67, "pk_set_host_option: unknown option \"%s\"", name);
  return 0;
}

// where result `what` (0..4) of the current iterate landed (valid after its pk_fetch / pk_eval_hess_prepared)
int pk_result_location(pk_ctx* c, int what, double** ptr) {
  int rc = ready(c);
  if (rc) return rc;
  if (what < 0 || what > 4 || !ptr) return fail(c, 62, "pk_result_location: bad arguments");
  *ptr = c->landed[what] ? c->landed[what] : c->h_out[what];
  return 0;
}

// pinned host result buffers of the context: what = 0 f, 1 grad, 2 g, 3 jac, 4 hess
int pk_host_buffer(pk_ctx* c, int what, double** ptr, int64_t* count) {
  int rc = ready(c);
  if (rc) return rc;
  if (what < 0 || what > 4 || !ptr) return fail(c, 62, "pk_host_buffer: bad arguments");
  const int64_t cnt[5] = {1, c->n, c->m, c->nnz_J, c->nnz_H};
  *ptr = c->h_out[what];
  if (count) *count = cnt[what];
  return 0;
}

// ---------------------------------------------------------------- sharded cycles: peer memory + the exchange of the sums
// Device memory of the caller's own (a mailbox, a reassembly buffer).  finegrained = 1: coherent with other GPUs and the
// host WHILE kernels run (flags polled across devices); 0: ordinary device memory.
int pk_device_alloc(pk_ctx* c, size_t bytes, int finegrained, void** out) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!out) return fail(c, 60, "null output pointer");
  PK_HIP(c, hipSetDevice(c->device));
  if (finegrained) PK_HIP(c, hipExtMallocWithFlags(out, bytes ? bytes : 8, hipDeviceMallocFinegrained));
  else PK_HIP(c, hipMalloc(out, bytes ? bytes : 8));
  PK_HIP(c, hipMemset(*out, 0, bytes ? bytes : 8));
  PK_HIP(c, hipDeviceSynchronize());
  return 0;
}

int pk_device_free(pk_ctx* c, void* p) {
  if (!c) return fail(nullptr, 1, "null context");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipDeviceSynchronize());
  if (p) PK_HIP(c, hipFree(p));
  return 0;
}

// Inter-process handle (64 bytes) of a pk_device_alloc allocation, and its mapping in another process (one per GPU).
int pk_ipc_export(pk_ctx* c, void* dptr, void* handle64) {
  if (!c) return fail(nullptr, 1, "null context");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size of the C ABI");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, dptr));
  return 0;
}

int pk_ipc_open(pk_ctx* c, const void* handle64, void** out) {
  if (!c) return fail(nullptr, 1, "null context");
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, sizeof h);
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}

int pk_ipc_close(pk_ctx* c, void* p) {
  if (!c) return fail(nullptr, 1, "null context");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipDeviceSynchronize());
  PK_HIP(c, hipIpcCloseMemHandle(p));
  return 0;
}

// Host memory of the caller's own (e.g. a shared-memory segment several processes map) made a DMA / kernel target:
// page-locks [p, p + bytes) and returns the address the device sees it at.  pk_host_unregister before it is unmapped.
int pk_host_register(pk_ctx* c, void* p, size_t bytes, void** dev_ptr) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!p || !bytes || !dev_ptr) return fail(c, 60, "null host buffer");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  PK_HIP(c, hipHostGetDevicePointer(dev_ptr, p, 0));
  return 0;
}

int pk_host_unregister(pk_ctx* c, void* p) {
  if (!c) return fail(nullptr, 1, "null context");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipDeviceSynchronize());
  PK_HIP(c, hipHostUnregister(p));
  return 0;
}

// Asynchronous copy between any two addresses the device can see (device memory, registered / pinned host memory)
int pk_copy_dev(pk_ctx* c, void* dst, const void* src, size_t bytes, void* stream) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!bytes) return 0;
  PK_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, pick(c, stream)));
  return 0;
}

// A shard whose gradient output is another GPU's buffer keeps the slots shared by all nodes (partial sums) local.
int pk_set_shared_grad_target(pk_ctx* c, double* d_grad_shared) {
  if (!c) return fail(nullptr, 1, "null context");
  c->gshared = d_grad_shared;
  drop_cycle_graph(c);
  return 0;
}

// d_boxes: device array of `world` pointers, entry r = rank r's mailbox as mapped in this process; d_idx: device array
// of the NLP indices of the n_sh shared gradient slots; stride: words per sender slot (>= 1 + n_I + n_sh).
int pk_set_exchange(pk_ctx* c, int world, int rank, const void* d_boxes, const int32_t* d_idx, int n_sh, int stride) {
  int rc = ready(c);
  if (rc) return rc;
  // everything is checked before anything is stored: a refused call leaves the context as it was
  if (world < 1 || world > PK_MAX_RANKS || rank < 0 || rank >= world)
    return fail(c, 90, "pk_set_exchange: %d ranks (at most %d), rank %d", world, PK_MAX_RANKS, rank);
  if (c->md.n_I + n_sh > 512 || stride < 1 + c->md.n_I + n_sh)
    return fail(c, 91, "pk_set_exchange: partial vector of %d doubles (at most 512), slot of %d words", c->md.n_I + n_sh, stride);
  if (n_sh != c->n_gz) return fail(c, 94, "pk_set_exchange: %d shared gradient slots, the problem has %d", n_sh, c->n_gz);
  if (!d_boxes) return fail(c, 93, "pk_set_exchange: no mailbox table");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(c->stream));
  std::vector<unsigned long long*> boxes((size_t)world, nullptr);
  PK_HIP(c, hipMemcpy(boxes.data(), d_boxes, sizeof(void*) * (size_t)world, hipMemcpyDeviceToHost));
  if (!boxes[(size_t)rank]) return fail(c, 93, "pk_set_exchange: this rank's own mailbox is missing from the table");
  // this rank's mailbox starts empty and its cycle count at zero (stale flags of an earlier set-up cannot match: every rank
  // resets here, and the caller's barrier behind the set-up comes before the first flag is raised)
  const size_t words = 2 * (size_t)world * (size_t)stride + PK_XC_STATE;
  PK_HIP(c, hipMemset(boxes[(size_t)rank], 0, sizeof(unsigned long long) * words));
  PK_HIP(c, hipDeviceSynchronize());
  c->xc_box = (const unsigned long long* const*)d_boxes;
  c->xc_own = boxes[(size_t)rank];
  c->xc_idx = d_idx;
  c->xc_world = world; c->xc_rank = rank; c->xc_nsh = n_sh; c->xc_stride = stride;
  drop_cycle_graph(c);
  return 0;
}

// cycles exchanged so far and how many of them gave up waiting for a peer (their sums read NaN on THIS rank, while a late
// peer still got finite ones: a caller checks this before it trusts f across the ranks).  Synchronizes the stream.
int pk_exchange_status(pk_ctx* c, void* stream, int64_t* cycles, int64_t* timed_out) {
  int rc = ready(c);
  if (rc) return rc;
  if (!c->xc_own) return fail(c, 92, "pk_exchange_status: call pk_set_exchange first");
  PK_HIP(c, hipSetDevice(c->device));
  PK_HIP(c, hipStreamSynchronize(pick(c, stream)));
  unsigned long long st[2] = {0, 0};
  PK_HIP(c, hipMemcpy(st, c->xc_own + 2 * (size_t)c->xc_world * (size_t)c->xc_stride, sizeof st, hipMemcpyDeviceToHost));
  if (cycles) *cycles = (int64_t)st[0];
  if (timed_out) *timed_out = (int64_t)st[1];
  return 0;
}

// 1: pk_eval_cycle_dev's single launch exchanges the partial sums itself (its finalize workgroup posts, waits and adds:
// a sharded cycle is ONE launch per GPU); 0: the caller runs pk_exchange_sums_dev behind it (a second launch).
int pk_set_exchange_inline(pk_ctx* c, int enable) {
  if (!c) return fail(nullptr, 1, "null context");
  if (enable && !c->xc_box) return fail(c, 92, "pk_set_exchange_inline: call pk_set_exchange first");
  if (enable && !c->md.sharded)
    return fail(c, 95, "pk_set_exchange_inline: the code object was generated for a single GPU (no exchange code in pk_cycle)");
  c->xc_inline = enable != 0;
  drop_cycle_graph(c);
  return 0;
}

// After the shard's pk_eval_cycle_dev on the same stream: post this rank's partial sums to every peer, take theirs, leave
// the global integrals, the summed shared gradient slots (in d_grad, or the shared-slot target) and -- write_f -- f.
int pk_exchange_sums_dev(pk_ctx* c, const double* d_x, double* d_grad, double* d_f, int epoch, int write_f, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (!c->xc_box) return fail(c, 92, "pk_exchange_sums: call pk_set_exchange first");
  if (epoch < 0) epoch = 0;                   // (0: the exchange counts the cycles itself, in device memory)
  PkArgs A = base_args(c, d_x, nullptr, 0.0);
  A.o_grad = d_grad; A.o_f = d_f;
  A.xc_box = (unsigned long long* const*)c->xc_box; A.xc_idx = c->xc_idx;
  A.xc_world = c->xc_world; A.xc_rank = c->xc_rank; A.xc_epoch = epoch; A.xc_nsh = c->xc_nsh; A.xc_stride = c->xc_stride;
  A.flags = (A.flags & ~F_WRITE_F) | (write_f ? F_WRITE_F : 0);
  return launch(c, K_XCHG, A, 1, 0, pick(c, stream));
}

// dst[dst_off + i] = src[src_off + i] over a device table of n_chunks (src_off, dst_off, len) int64 triples
int pk_copy_runs_dev(pk_ctx* c, const int64_t* d_table, int n_chunks, const double* d_src, double* d_dst, void* stream) {
  int rc = ready(c);
  if (rc) return rc;
  if (n_chunks <= 0) return 0;
  PkArgs A = base_args(c, nullptr, nullptr, 0.0);
  A.rc_table = d_table; A.rc_n = n_chunks; A.rc_src = d_src; A.rc_dst = d_dst;
  return launch(c, K_RUNS, A, (unsigned)(n_chunks < 8192 ? n_chunks : 8192), 0, pick(c, stream));
}

// *d_dst = value, in stream order (d_dst: device address of a 64-bit word, typically of a registered host segment)
int pk_store_word_dev(pk_ctx* c, void* d_dst, int64_t value, void* stream) {
  if (!c) return fail(nullptr, 1, "null context");
  if (!d_dst || ((uintptr_t)d_dst & 7)) return fail(c, 60, "pk_store_word: destination must be an 8-byte aligned device address");
  PK_HIP(c, hipSetDevice(c->device));
  hipLaunchKernelGGL(pk_store_word_kernel, dim3(1), dim3(64), 0, pick(c, stream), (unsigned long long*)d_dst, (unsigned long long)value);
  PK_HIP(c, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- profiling
int pk_profile(pk_ctx* c, int enable) {
  if (!c) return fail(nullptr, 1, "null context");
  c->profiling = enable != 0;
  c->profile_mask = (unsigned)enable;   /* bit k set: time kernel id k */
  return 0;
}

// developer tracing (models generated with POCKIT_AMD_TRACE=1): per-tile s_memtime marks of the last launch
int pk_trace_read(pk_ctx* c, uint64_t* out, int64_t count) {
  int rc = ready(c);
  if (rc) return rc;
  const int64_t need = ((int64_t)c->n_tiles * 3 + 3) * 16;   // [tile][role] records + pk_cycle's three special workgroups
  PK_HIP(c, hipSetDevice(c->device));
  if (!c->d_trace) {
    PK_HIP(c, hipStreamSynchronize(c->stream));
    PK_HIP(c, hipMalloc((void**)&c->d_trace, sizeof(uint64_t) * (size_t)(need ? need : 1)));
    PK_HIP(c, hipMemset(c->d_trace, 0, sizeof(uint64_t) * (size_t)(need ? need : 1)));
    drop_cycle_graph(c);
    return 0;          // first call only arms the buffer
  }
  if (!out || count < need) return fail(c, 72, "pk_trace_read: need room for %lld marks", (long long)need);
  PK_HIP(c, hipDeviceSynchronize());
  PK_HIP(c, hipMemcpy(out, c->d_trace, sizeof(uint64_t) * (size_t)need, hipMemcpyDeviceToHost));
  PK_HIP(c, hipMemset(c->d_trace, 0, sizeof(uint64_t) * (size_t)need));
  return 0;
}

int pk_profile_sampling(pk_ctx* c, int period) {
  if (!c) return fail(nullptr, 1, "null context");
  if (period < 1) return fail(c, 71, "pk_profile_sampling: period must be >= 1");
  c->profile_period = (unsigned)period;
  for (auto& v : c->profile_seen) v = 0;
  return 0;
}

int pk_profile_read(pk_ctx* c, int k, int64_t* launches, double* total_ms) {
  if (!c) return fail(nullptr, 1, "null context");
  if (k < 0 || k >= K_COUNT) return fail(c, 70, "pk_profile_read: bad kernel id %d", k);
  for (auto& ev : c->pending[k]) {
    float ms = 0.f;
    PK_HIP(c, hipEventSynchronize(ev.b));
    PK_HIP(c, hipEventElapsedTime(&ms, ev.a, ev.b));
    c->total_ms[k] += ms;
    c->launches[k] += 1;
    c->free_events.push_back(ev);
  }
  c->pending[k].clear();
  if (launches) *launches = c->launches[k];
  if (total_ms) *total_ms = c->total_ms[k];
  return 0;
}

}  // extern "C"
