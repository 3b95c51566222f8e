// copy_kernel_probe.cpp -- developer helper (GPU box): which shape of a device -> pinned-host copy kernel uses the PCIe link
// best (the host shim's result copies of <= 8 MiB go through such a kernel: no cross-engine hand-off behind the kernel that
// produced the data; the DMA engine reaches 56 GB/s on large copies, the first copy kernel 51).
// Variants: grid size, 16-byte elements per lane and iteration (unroll), plain / nontemporal / system-scope stores,
// contiguous-per-workgroup vs grid-strided assignment.  Build: hipcc -O2 --offload-arch=gfx950 tools/copy_kernel_probe.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
typedef double d2 __attribute__((ext_vector_type(2)));

template <int U, int MODE, bool BLOCKED>
__global__ void __launch_bounds__(256) copyk(const d2* __restrict__ src, d2* __restrict__ dst, size_t n) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  if (BLOCKED) {                       // every workgroup owns one contiguous slice
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)blockDim.x * U) {
      d2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) if (i + (size_t)u * blockDim.x < hi) v[u] = src[i + (size_t)u * blockDim.x];
#pragma unroll
      for (int u = 0; u < U; ++u) if (i + (size_t)u * blockDim.x < hi) {
        d2* p = dst + i + (size_t)u * blockDim.x;
        if (MODE == 1) __builtin_nontemporal_store(v[u], p); else *p = v[u];
      }
    }
    return;
  }
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += nthreads * U) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * nthreads < n) v[u] = src[i + u * nthreads];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * nthreads < n) {
      d2* p = dst + i + u * nthreads;
      if (MODE == 1) __builtin_nontemporal_store(v[u], p); else *p = v[u];
    }
  }
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t MAXB = 64u << 20;
  char *d, *h;
  CK(hipMalloc((void**)&d, MAXB));
  CK(hipHostMalloc((void**)&h, MAXB, hipHostMallocDefault));
  CK(hipMemset(d, 1, MAXB));
  auto run = [&](auto kern, unsigned grid, size_t bytes, bool d2h) -> double {
    std::vector<double> t;
    for (int rep = 0; rep < 40; ++rep) {
      auto t0 = clk::now();
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, (const d2*)(d2h ? d : h), (d2*)(d2h ? h : d), bytes / 16);
      while (hipStreamQuery(st) == hipErrorNotReady) { }
      t.push_back(us(t0, clk::now()));
    }
    return med(t);
  };
  for (size_t bytes : {(size_t)1536064, (size_t)2879776, (size_t)4895400, (size_t)6447528}) {
    {   // the DMA engine for comparison
      std::vector<double> t;
      for (int rep = 0; rep < 40; ++rep) {
        auto t0 = clk::now();
        CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st));
        while (hipStreamQuery(st) == hipErrorNotReady) { }
        t.push_back(us(t0, clk::now()));
      }
      double m = med(t);
      printf("%9zu bytes  DMA engine D2H %7.1f us %5.1f GB/s\n", bytes, m, bytes / m / 1e3);
    }
    for (unsigned grid : {32u, 64u, 128u, 256u, 512u, 1024u, 2048u}) {
      double a = run(copyk<1, 0, false>, grid, bytes, true), b = run(copyk<2, 0, false>, grid, bytes, true),
             c = run(copyk<4, 0, false>, grid, bytes, true), e = run(copyk<1, 1, false>, grid, bytes, true),
             f = run(copyk<4, 1, false>, grid, bytes, true), g = run(copyk<1, 0, true>, grid, bytes, true),
             hh = run(copyk<4, 0, true>, grid, bytes, true), k = run(copyk<4, 1, true>, grid, bytes, true);
      printf("   grid %4u: strided u1 %6.1f u2 %6.1f u4 %6.1f | nt u1 %6.1f u4 %6.1f | blocked u1 %6.1f u4 %6.1f nt-u4 %6.1f   best %.1f GB/s\n", grid, a, b, c, e,
             f, g, hh, k, bytes / std::min({a, b, c, e, f, g, hh, k}) / 1e3);
    }
  }
  // host -> device (x upload): 0.77 MB and 4.8 MB
  for (size_t bytes : {(size_t)768064, (size_t)4800096}) {
    for (unsigned grid : {64u, 256u, 1024u}) {
      double a = run(copyk<1, 0, false>, grid, bytes, false), c = run(copyk<4, 0, false>, grid, bytes, false), g = run(copyk<4, 0, true>, grid, bytes, false);
      printf("%9zu bytes H2D grid %4u: strided u1 %6.1f u4 %6.1f blocked u4 %6.1f\n", bytes, grid, a, c, g);
    }
  }
  return 0;
}
