// persist_probe.cpp -- developer helper (GPU box): what a RESIDENT-WAVES form of the cycle could save.
// A persistent pk_cycle would keep its ~1000 waves on the chip across iterates and start an iterate on a doorbell in device
// memory instead of a launch.  Before building it: the floor of both ways of starting and finishing one iterate of a grid of
// the cycle's shape (255 workgroups x 256 threads; 1000 for the 40k-node shape), with NO work in it.
//   A. back-to-back launches of an empty kernel: launch + drain per iterate (what pk_cycle pays today);
//   B. one resident launch, `iters` iterates: every workgroup polls a doorbell word (agent-scope load) until it shows the
//      iterate's number, does nothing, and arrives at a counter (agent-scope atomic add); the workgroup that arrives last
//      rings the doorbell for the next iterate -- the cheapest possible "all outputs of iterate k exist before k + 1 starts";
//   C. the same with the doorbell rung by the HOST (a store into fine-grained device memory through the host mapping is not
//      available here: the host writes pinned host memory the GPU polls over PCIe) after it has seen the arrival counter,
//      i.e. a solver-in-the-loop resident form.
// Bounded polls everywhere (a missed doorbell ends the kernel instead of hanging the GPU).
// Build: hipcc -O2 --offload-arch=gfx950 tools/persist_probe.cpp -o /tmp/persist_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }

__global__ void empty_kernel(unsigned long long* sink) {
  if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFFu) sink[0] = 1;
}

__global__ void resident(unsigned long long* bell, unsigned long long* arrived, int iters, int self_ring, unsigned long long* failed) {
  for (int it = 1; it <= iters; ++it) {
    if (threadIdx.x == 0) {
      long tries = 0;
      while (__hip_atomic_load(bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)it && ++tries < (1L << 22)) __builtin_amdgcn_s_sleep(1);
      if (tries >= (1L << 22)) { __hip_atomic_store(failed, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
    __syncthreads();
    if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;
    // (an iterate's work would be here)
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long n = __hip_atomic_fetch_add(arrived, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1ull;
      if (self_ring && n == (unsigned long long)it * gridDim.x)
        __hip_atomic_store(bell, (unsigned long long)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long *d_words, *h_words;
  CK(hipExtMallocWithFlags((void**)&d_words, 4096, hipDeviceMallocFinegrained));
  CK(hipHostMalloc((void**)&h_words, 4096, hipHostMallocDefault));
  for (unsigned grid : {255u, 1000u}) {
    // A. back-to-back empty launches
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d_words);
    CK(hipStreamSynchronize(st));
    const int N = 5000;
    auto t0 = clk::now();
    for (int k = 0; k < N; ++k) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d_words);
    CK(hipStreamSynchronize(st));
    const double a = us(t0, clk::now()) / N;
    // B. resident, self-ringing (device memory doorbell)
    const int iters = 2000;
    CK(hipMemset(d_words, 0, 4096));
    unsigned long long one = 1;
    CK(hipMemcpy(d_words, &one, 8, hipMemcpyHostToDevice));          // bell = 1: the first iterate may start
    CK(hipDeviceSynchronize());
    t0 = clk::now();
    hipLaunchKernelGGL(resident, dim3(grid), dim3(256), 0, st, d_words, d_words + 16, iters, 1, d_words + 32);
    CK(hipStreamSynchronize(st));
    const double b = us(t0, clk::now()) / iters;
    unsigned long long out[40];
    CK(hipMemcpy(out, d_words, sizeof out, hipMemcpyDeviceToHost));
    // C. resident, the host rings (pinned host memory polled over PCIe), after it has seen every workgroup arrive
    const int hiters = 300;
    for (int i = 0; i < 64; ++i) h_words[i] = 0;
    h_words[0] = 1;
    t0 = clk::now();
    hipLaunchKernelGGL(resident, dim3(grid), dim3(256), 0, st, h_words, h_words + 16, hiters, 0, h_words + 32);
    bool ok = true;
    for (int it = 1; it <= hiters && ok; ++it) {
      const auto tw = clk::now();
      while (((volatile unsigned long long*)h_words)[16] < (unsigned long long)it * grid)
        if (us(tw, clk::now()) > 2e6) { ok = false; break; }
      ((volatile unsigned long long*)h_words)[0] = (unsigned long long)(it + 1);
    }
    if (!ok) ((volatile unsigned long long*)h_words)[32] = 1;          // let the kernel leave
    CK(hipStreamSynchronize(st));
    const double c = us(t0, clk::now()) / hiters;
    printf("grid %4u x 256: A empty launches back to back %.2f us per iterate | B resident, device doorbell %.2f us per iterate%s | "
           "C resident, host in the loop %.2f us per iterate%s\n", grid, a, b, out[32] ? " (a poll timed out)" : "", c,
           (ok && !h_words[32]) ? "" : " (timed out)");
  }
  return 0;
}
