// driver.cpp -- TEST INFRASTRUCTURE: drives the host runtime behind the C ABI (pockit_amd/csrc/pk_runtime.cpp, compiled with
// -fsanitize=address,undefined against the host-only HIP stand-in of this directory) through the protocols a solver-side shim
// uses: landing blocks, constant Jacobian runs, the prepared-x callbacks in any order, every switch of the shim, the speculative
// Hessian on a matching and on a new x, pageable targets, the compact layouts, the one-call cycle, the per-callback host entry
// points, the CSR maps, error paths, tear-down.  Every result is checked against what the stand-in's "kernels" write for the x /
// lambda / sigma of THAT call (stale staging buffers, missed copies, wrong offsets and reuse-before-completion show up as wrong
// values; memory errors are the sanitizers' to report).  Exit code 0 = all good.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/pockit_hip.h"
#include "../../pockit_amd/csrc/pk_abi.h"
#include "fake_hip.h"

static int g_checks = 0;
static char g_where[256] = "start";
#define CHECK(cond)                                                                         \
  do {                                                                                      \
    ++g_checks;                                                                             \
    if (!(cond)) {                                                                          \
      std::fprintf(stderr, "driver.cpp:%d: CHECK failed: %s  [%s]\n", __LINE__, #cond, g_where);  \
      { const auto& lg = fake_hip_log(); std::fprintf(stderr, "last operations:");                 \
        for (size_t i_ = lg.size() > 40 ? lg.size() - 40 : 0; i_ < lg.size(); ++i_) std::fprintf(stderr, " %s", lg[i_].c_str()); \
        std::fprintf(stderr, "\n"); }                                                             \
      std::exit(1);                                                                         \
    }                                                                                       \
  } while (0)
#define OK(call)                                                                                        \
  do {                                                                                                  \
    int rc_ = (call);                                                                                   \
    ++g_checks;                                                                                         \
    if (rc_ != 0) {                                                                                     \
      std::fprintf(stderr, "driver.cpp:%d: %s -> %d: %s\n", __LINE__, #call, rc_, pk_last_error(ctx));  \
      std::exit(1);                                                                                     \
    }                                                                                                   \
  } while (0)

static pk_ctx* ctx = nullptr;
static FakeSizes S;
static bool in_runs(const std::vector<std::pair<int64_t, int64_t>>& r, int64_t p) {
  for (auto& q : r)
    if (p >= q.first && p < q.second) return true;
  return false;
}
static void check_x_results(const std::vector<double>& x, const double* f, const double* grad, const double* g, const double* J, bool compact) {
  if (f) CHECK(*f == fake_f(x.data(), S.n));
  if (grad) for (int64_t i = 0; i < S.n; ++i) {
    if (grad[i] != fake_grad(x.data(), S.n, i)) std::fprintf(stderr, "grad[%lld] = %.17g, expected %.17g\n", (long long)i, grad[i], fake_grad(x.data(), S.n, i));
    CHECK(grad[i] == fake_grad(x.data(), S.n, i));
  }
  if (g) for (int64_t j = 0; j < S.m; ++j) CHECK(g[j] == fake_g(x.data(), S.n, j));
  if (J) {
    const int64_t nn = compact ? S.nnz_Jc : S.nnz_J;
    for (int64_t p = 0; p < nn; ++p)
      CHECK(J[p] == fake_jac(x.data(), S.n, p, in_runs(compact ? S.jconst_compact : S.jconst, p)) + (compact ? 0.25 : 0.0));
  }
}
static void check_hess(const std::vector<double>& x, const std::vector<double>& lam, double sigma, const double* H, bool compact) {
  const int64_t nn = compact ? S.nnz_Hc : S.nnz_H;
  for (int64_t p = 0; p < nn; ++p) CHECK(H[p] == fake_hess(x.data(), lam.data(), sigma, S.n, S.m, p) - (compact ? 0.5 : 0.0));
}

int main() {
  S.n = 37; S.m = 23; S.nnz_J = 211; S.nnz_H = 97; S.nnz_Jc = 150; S.nnz_Hc = 41;
  S.jconst = {{0, 64}, {100, 140}};
  S.jconst_compact = {{0, 50}};
  fake_hip_set_sizes(S);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  auto fresh = [&](int k) { std::vector<double> v((size_t)k); for (auto& e : v) e = U(rng); return v; };

  CHECK(pk_create(nullptr, 0) != 0);
  OK(pk_create(&ctx, 0));
  CHECK(pk_eval_f(ctx, nullptr, nullptr) != 0);                 // nothing loaded yet: an error, not a crash
  pk_model_desc md{};
  md.n_phase = 1; md.n_I = 1; md.nred = 1; md.lds_g = md.lds_j = md.lds_h = md.lds_x = md.lds_e = md.lds_jc = 64;
  md.ne_j = md.ne_h = md.ne_a = md.ne_hc = md.ne_jc = 1; md.prepass_f = 1; md.tab_cap = 64;
  const char image[16] = "fake code";
  md.tab_cap = 65;
  CHECK(pk_load_model(ctx, image, sizeof image, &md) != 0);    // refused capacity
  md.tab_cap = 64;
  md.max_phases = PK_HOST_MAX_PHASES + 1;
  CHECK(pk_load_model(ctx, image, sizeof image, &md) != 0);    // more phase records than the kernel arguments hold
  md.max_phases = PK_HOST_MAX_PHASES; md.n_phase = PK_HOST_MAX_PHASES;
  OK(pk_load_model(ctx, image, sizeof image, &md));            // (the most: accepted)
  md.max_phases = 0; md.n_phase = 1;
  OK(pk_load_model(ctx, image, sizeof image, &md));
  PkPhase ph{};
  pk_problem_desc pd{};
  pd.n = S.n; pd.m = S.m; pd.n_phase = 1; pd.nnz_J = S.nnz_J; pd.nnz_H = S.nnz_H; pd.nnz_Jc = S.nnz_Jc; pd.nnz_Hc = S.nnz_Hc;
  pd.phases = &ph;
  PkTile tiles[PK_WAVES_PER_BLOCK] = {};          // one block of empty tiles: every tile kernel gets a (one-workgroup) launch
  for (auto& t : tiles) t.K = 1;
  ph.tile_hi = PK_WAVES_PER_BLOCK;
  pd.tiles = tiles;
  pd.n_tiles = PK_WAVES_PER_BLOCK;
  std::vector<int32_t> jr((size_t)S.nnz_J, 1), hr((size_t)S.nnz_H, 2);
  pd.jac_row = pd.jac_col = jr.data(); pd.hess_row = pd.hess_col = hr.data();
  pd.n_phase = 2;
  CHECK(pk_set_problem(ctx, &pd) != 0);                         // phase count does not match the model
  pd.n_phase = 1;
  OK(pk_set_problem(ctx, &pd));
  OK(pk_set_problem(ctx, &pd));                                 // set again: everything of the first one is released
  std::vector<int32_t> back((size_t)S.nnz_J);
  OK(pk_get_structure(ctx, back.data(), nullptr, nullptr, nullptr));
  CHECK(back[5] == 1);

  // ---- constant runs + landing blocks
  const int64_t lo[2] = {0, 100}, hi[2] = {64, 140}, bad_lo[2] = {50, 20}, bad_hi[2] = {60, 30};
  CHECK(pk_set_jac_constant_runs(ctx, 2, bad_lo, bad_hi) != 0);
  OK(pk_set_jac_constant_runs(ctx, 2, lo, hi));
  const size_t blk_len = (size_t)(S.nnz_J + S.n + S.m);
  std::vector<double*> blocks, hblocks;
  for (int b = 0; b < 3; ++b) {
    void *p = nullptr, *q = nullptr;
    CHECK(pk_host_alloc(sizeof(double) * blk_len, &p) == 0 && pk_host_alloc(sizeof(double) * (size_t)S.nnz_H, &q) == 0);
    blocks.push_back((double*)p);
    hblocks.push_back((double*)q);
    for (size_t i = 0; i < blk_len; ++i) blocks.back()[i] = NAN;
    OK(pk_fill_jac_constants(ctx, blocks.back()));
  }
  const char* options[] = {"spin_wait", "lambda_direct", "chunk_upload", "kernel_upload", "kernel_download", "split_copy", "speculative_hess",
                           "mark_wait", "hess_direct", "xpart_single", "small_direct", "adaptive_prefetch"};
  int defaults[] = {1, 1, 1, 1, 8, 1, 1, 1, 1, 1, 1, 1};
  const int n_options = 12;
  std::vector<double> x = fresh(S.n), lam = fresh(S.m);
  double f = 0.0;
  int is_new = 0;
  // (the fake system is small: with "small_direct" the kernels read x in place and store into the landing blocks themselves;
  //  the walk through the switches runs once in that mode and once with uploads and copies)
  for (int it = 0; it < 156; ++it) {
    defaults[10] = it < 78 ? 1 : 0;
    if (it % 3 == 0) {                                          // walk through every switch, one at a time off its default
      for (int o = 0; o < n_options; ++o) OK(pk_set_host_option(ctx, options[o], defaults[o]));
      const int o = (it / 3) % (n_options + 1);
      if (o < n_options) OK(pk_set_host_option(ctx, options[o], defaults[o] ? 0 : 1));
      OK(pk_set_host_mode(ctx, (it / 3) % 5 != 4, (it / 3) % 7 == 6));      // prefetch off / kernels storing into the targets now and then
    }
    x = fresh(S.n);
    const double sigma = U(rng);
    std::memset(g_where, ' ', 100);
    g_where[100] = 0;
    std::snprintf(g_where, 100, "iterate %d, switch %d off default, prefetch %d, host_direct %d", it, (it / 3) % (n_options + 1),
                  (it / 3) % 5 != 4, (it / 3) % 7 == 6);
    g_where[std::strlen(g_where)] = ' ';
    double* blk = blocks[(size_t)(it % 3)];
    double* hb = hblocks[(size_t)(it % 3)];
    int order[4] = {0, 1, 2, 3};
    std::shuffle(order, order + 4, rng);
    if (it % 5 == 4) {                                          // the Hessian callback sees the new x first
      lam = fresh(S.m);
      OK(pk_callback_hess(ctx, x.data(), lam.data(), sigma, blk, hb, 0, &is_new));
      CHECK(is_new == 1);
      check_hess(x, lam, sigma, hb, false);
    }
    for (int k = 0; k < 4; ++k) {
      OK(pk_callback_x(ctx, order[k], x.data(), blk, &f, &is_new));
      CHECK(is_new == ((k == 0 && it % 5 != 4) ? 1 : 0));
      if (order[k] == 0) CHECK(f == fake_f(x.data(), S.n));
      std::snprintf(g_where + 100, 100, " | x-callback %d (what %d)", k, order[k]);
      if (order[k] == 1) check_x_results(x, nullptr, blk + S.nnz_J, nullptr, nullptr, false);
      if (order[k] == 2) check_x_results(x, nullptr, nullptr, blk + S.nnz_J + S.n, nullptr, false);
      if (order[k] == 3) check_x_results(x, nullptr, nullptr, nullptr, blk, false);
    }
    CHECK(pk_same_x(ctx, x.data()) == 1);
    lam = fresh(S.m);
    OK(pk_callback_hess(ctx, x.data(), lam.data(), sigma, blk, hb, 0, &is_new));
    CHECK(is_new == 0);
    check_hess(x, lam, sigma, hb, false);
    std::snprintf(g_where + 100, 100, " | after the Hessian");
    check_x_results(x, nullptr, blk + S.nnz_J, blk + S.nnz_J + S.n, blk, false);      // still intact after the Hessian
    OK(pk_callback_hess(ctx, x.data(), lam.data(), 0.5 * sigma, nullptr, nullptr, 0, &is_new));      // the context's own buffer
    double* own = nullptr;
    OK(pk_result_location(ctx, 4, &own));
    check_hess(x, lam, 0.5 * sigma, own, false);
  }
  for (int o = 0; o < n_options; ++o) OK(pk_set_host_option(ctx, options[o], defaults[o]));
  OK(pk_set_host_mode(ctx, 1, 0));
  CHECK(pk_set_host_option(ctx, "no such switch", 1) != 0);

  // ---- a line search: rejected trial points ask for f and g only, the accepted one for everything (adaptive prefetch)
  for (int small = 0; small < 2; ++small) {
    OK(pk_set_host_option(ctx, "small_direct", small));
    for (int it = 0; it < 24; ++it) {
      x = fresh(S.n);
      std::snprintf(g_where, 100, "line search %d (small %d)", it, small);
      double* blk = blocks[(size_t)(it % 3)];
      OK(pk_callback_x(ctx, 0, x.data(), blk, &f, &is_new));
      CHECK(is_new == 1 && f == fake_f(x.data(), S.n));
      OK(pk_callback_x(ctx, 2, x.data(), blk, &f, &is_new));
      check_x_results(x, nullptr, nullptr, blk + S.nnz_J + S.n, nullptr, false);
      if (it % 3 != 2) continue;                                  // (two rejected trial points, then an accepted one)
      OK(pk_callback_x(ctx, 1, x.data(), blk, &f, &is_new));
      OK(pk_callback_x(ctx, 3, x.data(), blk, &f, &is_new));
      CHECK(is_new == 0);
      check_x_results(x, nullptr, blk + S.nnz_J, blk + S.nnz_J + S.n, blk, false);
      lam = fresh(S.m);
      OK(pk_callback_hess(ctx, x.data(), lam.data(), 0.75, blk, hblocks[(size_t)(it % 3)], 0, &is_new));
      check_hess(x, lam, 0.75, hblocks[(size_t)(it % 3)], false);
    }
  }
  OK(pk_set_host_option(ctx, "small_direct", 1));

  // ---- all five results from one call (pk_callback_cycle): lands like the callbacks, the iterate becomes the prepared one
  for (int it = 0; it < 16; ++it) {
    if (it % 4 == 0) {
      OK(pk_set_host_option(ctx, "small_direct", it < 8));
      OK(pk_set_host_option(ctx, "mark_wait", it % 8 == 0));
      OK(pk_set_host_option(ctx, "hess_direct", it % 3 != 0));
      OK(pk_set_host_option(ctx, "lambda_direct", it != 4));
    }
    x = fresh(S.n);
    lam = fresh(S.m);
    const double sigma = U(rng);
    std::snprintf(g_where, 100, "one-call cycle %d", it);
    double* blk = blocks[(size_t)(it % 3)];
    double* hb = hblocks[(size_t)(it % 3)];
    for (int64_t p = 0; p < S.nnz_H; ++p) hb[p] = NAN;
    f = NAN;
    OK(pk_callback_cycle(ctx, x.data(), lam.data(), sigma, blk, hb, &f));
    check_x_results(x, &f, blk + S.nnz_J, blk + S.nnz_J + S.n, blk, false);
    check_hess(x, lam, sigma, hb, false);
    OK(pk_callback_x(ctx, 3, x.data(), blocks[(size_t)((it + 1) % 3)], &f, &is_new));      // same x: served from what has landed
    CHECK(is_new == 0);
    lam = fresh(S.m);
    OK(pk_callback_hess(ctx, x.data(), lam.data(), sigma, blocks[(size_t)((it + 1) % 3)], hb, 0, &is_new));
    CHECK(is_new == 0);
    check_hess(x, lam, sigma, hb, false);
    check_x_results(x, nullptr, blk + S.nnz_J, blk + S.nnz_J + S.n, blk, false);
  }
  CHECK(pk_callback_cycle(ctx, x.data(), lam.data(), 1.0, nullptr, hblocks[0], &f) != 0);
  for (int o = 0; o < n_options; ++o) OK(pk_set_host_option(ctx, options[o], defaults[o]));

  // ---- a C-ABI caller's plain arrays as targets: the whole Jacobian is copied, nothing assumed about them
  {
    std::vector<double> tf(1, NAN), tg((size_t)S.n, NAN), tc((size_t)S.m, NAN), tj((size_t)S.nnz_J, NAN), th((size_t)S.nnz_H, NAN);
    OK(pk_set_result_targets(ctx, tf.data(), tg.data(), tc.data(), tj.data(), th.data()));
    x = fresh(S.n);
    OK(pk_prepare_x(ctx, x.data()));
    for (int w = 3; w >= 0; --w) OK(pk_fetch(ctx, w, nullptr));
    OK(pk_eval_hess_prepared(ctx, lam.data(), 1.25, nullptr));
    check_x_results(x, tf.data(), tg.data(), tc.data(), tj.data(), false);
    check_hess(x, lam, 1.25, th.data(), false);
    std::vector<double> copy((size_t)S.nnz_J);
    OK(pk_fetch(ctx, 3, copy.data()));
    check_x_results(x, nullptr, nullptr, nullptr, copy.data(), false);
    OK(pk_set_result_targets(ctx, nullptr, nullptr, nullptr, nullptr, nullptr));
    OK(pk_invalidate_x(ctx));
    CHECK(pk_fetch(ctx, 1, nullptr) != 0);                     // no prepared x any more
  }

  // ---- the compact layouts on the same protocol
  {
    OK(pk_set_jacobian_layout(ctx, 1));
    const int64_t clo[1] = {0}, chi[1] = {50};
    OK(pk_set_jac_constant_runs(ctx, 1, clo, chi));
    void* p = nullptr;
    CHECK(pk_host_alloc(sizeof(double) * (size_t)(S.nnz_Jc + S.n + S.m), &p) == 0);
    double* cblk = (double*)p;
    OK(pk_fill_jac_constants(ctx, cblk));
    std::vector<double> hc((size_t)S.nnz_Hc);
    void* hp = nullptr;
    CHECK(pk_host_alloc(sizeof(double) * (size_t)S.nnz_Hc, &hp) == 0);
    for (int it = 0; it < 6; ++it) {
      x = fresh(S.n);
      for (int w = 0; w < 4; ++w) OK(pk_callback_x(ctx, w, x.data(), cblk, &f, &is_new));
      check_x_results(x, &f, cblk + S.nnz_Jc, cblk + S.nnz_Jc + S.n, cblk, true);
      OK(pk_callback_hess(ctx, x.data(), lam.data(), 0.75, cblk, (double*)hp, 1, &is_new));
      check_hess(x, lam, 0.75, (double*)hp, true);
      OK(pk_eval_hessc_prepared(ctx, lam.data(), 0.25, hc.data(), 0));
      check_hess(x, lam, 0.25, hc.data(), true);
    }
    OK(pk_set_jacobian_layout(ctx, 0));
    x = fresh(S.n);
    OK(pk_callback_x(ctx, 3, x.data(), blocks[0], &f, &is_new));      // the reference layout's runs are back
    check_x_results(x, nullptr, nullptr, nullptr, blocks[0], false);
    std::vector<double> jc((size_t)S.nnz_Jc);
    OK(pk_eval_jacc(ctx, x.data(), jc.data()));
    check_x_results(x, nullptr, nullptr, nullptr, jc.data(), true);
    CHECK(pk_host_free(p) == 0 && pk_host_free(hp) == 0);
  }

  // ---- the one-call cycle and the per-callback host entry points
  {
    x = fresh(S.n);
    std::vector<double> g1((size_t)S.n), c1((size_t)S.m), j1((size_t)S.nnz_J), h1((size_t)S.nnz_H), hc((size_t)S.nnz_Hc);
    double f1 = 0.0;
    OK(pk_eval_cycle(ctx, x.data(), lam.data(), 2.0, &f1, g1.data(), c1.data(), j1.data(), h1.data()));
    check_x_results(x, &f1, g1.data(), c1.data(), j1.data(), false);
    check_hess(x, lam, 2.0, h1.data(), false);
    x = fresh(S.n);
    OK(pk_eval_f(ctx, x.data(), &f1));
    OK(pk_eval_grad(ctx, x.data(), g1.data()));
    OK(pk_eval_g(ctx, x.data(), c1.data()));
    OK(pk_eval_jac(ctx, x.data(), j1.data()));
    OK(pk_eval_hess(ctx, x.data(), lam.data(), -1.0, h1.data()));
    OK(pk_eval_hessc(ctx, x.data(), lam.data(), -1.0, hc.data()));
    check_x_results(x, &f1, g1.data(), c1.data(), j1.data(), false);
    check_hess(x, lam, -1.0, h1.data(), false);
    check_hess(x, lam, -1.0, hc.data(), true);
    CHECK(pk_eval_hess(ctx, x.data(), nullptr, 1.0, h1.data()) != 0);
  }

  // ---- CSR maps: the permutation form and the sliced, padded form for repeated entries
  {
    std::vector<int32_t> perm((size_t)S.nnz_J);
    for (int64_t p = 0; p < S.nnz_J; ++p) perm[(size_t)p] = (int32_t)(S.nnz_J - 1 - p);
    OK(pk_set_csr_map(ctx, 0, nullptr, perm.data(), S.nnz_J, S.nnz_J));
    std::vector<double> csr((size_t)S.nnz_J), j1((size_t)S.nnz_J);
    x = fresh(S.n);
    OK(pk_eval_jac_csr(ctx, x.data(), csr.data()));
    OK(pk_eval_jac(ctx, x.data(), j1.data()));
    for (int64_t p = 0; p < S.nnz_J; ++p) CHECK(csr[(size_t)p] == j1[(size_t)(S.nnz_J - 1 - p)]);
    std::vector<int32_t> hperm((size_t)S.nnz_H), seg;
    for (int64_t p = 0; p < S.nnz_H; ++p) hperm[(size_t)p] = (int32_t)p;
    for (int32_t q = 0; q < S.nnz_H; q += 1 + (q % 3)) seg.push_back(q);      // runs of 1 ... 3 triplets
    const int64_t nu = (int64_t)seg.size();
    seg.push_back((int32_t)S.nnz_H);
    OK(pk_set_csr_map(ctx, 1, seg.data(), hperm.data(), nu, S.nnz_H));
    std::vector<double> hcsr((size_t)nu);
    OK(pk_eval_hess_csr(ctx, x.data(), lam.data(), 1.0, hcsr.data()));
    CHECK(hcsr[0] == -7.0);                                     // (the stand-in's mark for the sliced form)
    hperm[3] = (int32_t)S.nnz_H;
    CHECK(pk_set_csr_map(ctx, 1, seg.data(), hperm.data(), nu, S.nnz_H) != 0);      // index out of range: refused on the host
    CHECK(pk_set_csr_map(ctx, 7, nullptr, perm.data(), 1, 1) != 0);
  }

  // ---- profiling path (timed launches), then tear-down
  OK(pk_profile_sampling(ctx, 2));
  OK(pk_profile(ctx, 1 << 6));
  for (int it = 0; it < 5; ++it) {
    x = fresh(S.n);
    OK(pk_callback_x(ctx, 0, x.data(), blocks[0], &f, &is_new));
  }
  int64_t launches = 0;
  double ms = 0.0;
  OK(pk_profile_read(ctx, 6, &launches, &ms));
  CHECK(launches >= 2);
  OK(pk_profile(ctx, 0));
  OK(pk_sync(ctx, nullptr));       // (the last iterates' copies into the blocks were never asked for: a landing block must outlive them)
  for (auto* b : blocks) CHECK(pk_host_free(b) == 0);
  for (auto* b : hblocks) CHECK(pk_host_free(b) == 0);
  pk_destroy(ctx);
  CHECK(fake_hip_live_allocations() == 0);                      // nothing of the context outlives it
  std::printf("runtime driver: %d checks passed\n", g_checks);
  return 0;
}
