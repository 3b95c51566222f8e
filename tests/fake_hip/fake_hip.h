// fake_hip.h -- TEST INFRASTRUCTURE: what the driver tells the host-only HIP stand-in about the problem (the stand-in's
// "kernels" write values any reader can recompute) and what it can ask back.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

struct FakeSizes {
  int n = 0, m = 0;
  int64_t nnz_J = 0, nnz_H = 0, nnz_Jc = 0, nnz_Hc = 0;
  std::vector<std::pair<int64_t, int64_t>> jconst, jconst_compact;      // x-independent runs of J (reference / compact layout)
};
void fake_hip_set_sizes(const FakeSizes& s);
const std::vector<std::string>& fake_hip_log();
void fake_hip_clear_log();
size_t fake_hip_live_allocations();
double fake_f(const double* x, int n);
double fake_grad(const double* x, int n, int64_t i);
double fake_g(const double* x, int n, int64_t j);
double fake_jac(const double* x, int n, int64_t p, bool constant);
double fake_hess(const double* x, const double* lam, double sigma, int n, int m, int64_t p);
