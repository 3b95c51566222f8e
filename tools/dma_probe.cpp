// dma_probe.cpp -- developer helper (GPU box): what the pieces of the host-landed NLP-callback cycle cost on THIS box's
// PCIe link, so that the host shim (pk_runtime.cpp) can be shaped by measurements instead of guesses:
//   1. one pinned DMA + event wait, D2H and H2D, for the transfer sizes of the C3 / C5 cycles;
//   2. the same bytes as k back-to-back DMAs (what chunked / run-wise copies pay per extra operation);
//   3. host memcpy of the staging sizes (pageable -> pinned), whole and chunk-pipelined with the H2D;
//   4. a kernel reading its input straight from pinned host memory instead of an H2D in front of it;
//   5. wake-up latency: hipEventSynchronize vs. spinning on a pinned word the kernel stores with system scope;
//   6. D2H on a second stream while an H2D runs on the first (full duplex?).
// Build: hipcc -O2 --offload-arch=gfx950 tools/dma_probe.cpp -o gpurun_out/dma_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

__global__ void sum_kernel(const double* __restrict__ in, size_t n, double* out, volatile unsigned long long* flag, unsigned long long tag) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
  if (acc == 12345.678) out[1] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = acc;
    if (flag) __hip_atomic_store((unsigned long long*)flag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t st, st2;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  hipEvent_t ev, ev2;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  const size_t MAXB = 128u << 20;
  char *d, *h, *pg;
  CK(hipMalloc((void**)&d, MAXB));
  CK(hipHostMalloc((void**)&h, MAXB, hipHostMallocDefault));
  pg = (char*)malloc(MAXB);
  memset(pg, 1, MAXB); memset(h, 2, MAXB);
  CK(hipMemset(d, 0, MAXB));
  const int REP = 60;
  // ---- 1. single DMA + event wait
  const size_t sizes[] = {8, 768064, 1536128, 2879776, 4895400, 6047400, 6447528, 7583528, 47359088, 57278608};
  printf("1. one DMA + hipEventSynchronize\n%12s %10s %10s %10s %10s\n", "bytes", "D2H us", "GB/s", "H2D us", "GB/s");
  for (size_t b : sizes) {
    double r[2];
    for (int dir = 0; dir < 2; ++dir) {
      std::vector<double> t;
      for (int k = 0; k < REP; ++k) {
        auto t0 = clk::now();
        if (dir == 0) CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st)); else CK(hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t.push_back(us(t0, clk::now()));
      }
      r[dir] = med(t);
    }
    printf("%12zu %10.1f %10.1f %10.1f %10.1f\n", b, r[0], b / r[0] / 1e3, r[1], b / r[1] / 1e3);
  }
  // ---- 2. the same bytes as k back-to-back DMAs on one stream
  printf("\n2. D2H of B bytes as k back-to-back DMAs (one event wait at the end), us\n%12s", "bytes");
  const int ks[] = {1, 2, 3, 4, 8, 16};
  for (int k : ks) printf(" %8d", k);
  printf("\n");
  for (size_t b : {(size_t)768064, (size_t)2879776, (size_t)6447528, (size_t)47359088}) {
    printf("%12zu", b);
    for (int k : ks) {
      std::vector<double> t;
      for (int rep = 0; rep < REP; ++rep) {
        auto t0 = clk::now();
        const size_t c = (b / k) & ~(size_t)7;
        for (int i = 0; i < k; ++i) CK(hipMemcpyAsync(h + i * c, d + i * c, i == k - 1 ? b - (size_t)i * c : c, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t.push_back(us(t0, clk::now()));
      }
      printf(" %8.1f", med(t));
    }
    printf("\n");
  }
  printf("   H2D likewise\n");
  for (size_t b : {(size_t)768064, (size_t)4800096}) {
    printf("%12zu", b);
    for (int k : ks) {
      std::vector<double> t;
      for (int rep = 0; rep < REP; ++rep) {
        auto t0 = clk::now();
        const size_t c = (b / k) & ~(size_t)7;
        for (int i = 0; i < k; ++i) CK(hipMemcpyAsync(d + i * c, h + i * c, i == k - 1 ? b - (size_t)i * c : c, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t.push_back(us(t0, clk::now()));
      }
      printf(" %8.1f", med(t));
    }
    printf("\n");
  }
  // ---- 3. staging: memcpy pageable -> pinned, then H2D; whole vs chunk-pipelined
  printf("\n3. stage (memcpy pageable -> pinned) + H2D + wait, us: memcpy alone | whole | pipelined in k chunks\n");
  for (size_t b : {(size_t)768064, (size_t)4800096}) {
    std::vector<double> tm;
    for (int rep = 0; rep < REP; ++rep) {
      pg[rep] ^= 1;
      auto t0 = clk::now();
      memcpy(h, pg, b);
      tm.push_back(us(t0, clk::now()));
    }
    printf("%12zu memcpy %7.1f |", b, med(tm));
    for (int k : {1, 2, 3, 4, 6, 8}) {
      std::vector<double> t;
      for (int rep = 0; rep < REP; ++rep) {
        pg[rep] ^= 1;
        auto t0 = clk::now();
        const size_t c = (b / k) & ~(size_t)63;
        for (int i = 0; i < k; ++i) {
          const size_t len = i == k - 1 ? b - (size_t)i * c : c;
          memcpy(h + i * c, pg + i * c, len);
          CK(hipMemcpyAsync(d + i * c, h + i * c, len, hipMemcpyHostToDevice, st));
        }
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t.push_back(us(t0, clk::now()));
      }
      printf(" k=%d %7.1f", k, med(t));
    }
    printf("\n");
  }
  // H2D straight from pageable memory (the runtime stages it itself)
  for (size_t b : {(size_t)768064, (size_t)4800096}) {
    std::vector<double> t;
    for (int rep = 0; rep < REP; ++rep) {
      pg[rep] ^= 1;
      auto t0 = clk::now();
      CK(hipMemcpyAsync(d, pg, b, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(ev, st));
      CK(hipEventSynchronize(ev));
      t.push_back(us(t0, clk::now()));
    }
    printf("%12zu H2D from pageable memory %7.1f\n", b, med(t));
  }
  // ---- 4. kernel reads its input from pinned host memory vs H2D + kernel on device memory
  printf("\n4. kernel summing n doubles: H2D + kernel(device) + wait | kernel(pinned host) + wait, us\n");
  double* dout;
  CK(hipMalloc((void**)&dout, 64));
  for (size_t b : {(size_t)768064, (size_t)4800096}) {
    std::vector<double> ta, tb;
    for (int rep = 0; rep < REP; ++rep) {
      auto t0 = clk::now();
      CK(hipMemcpyAsync(d, h, b, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, st, (const double*)d, b / 8, dout, nullptr, 0ull);
      CK(hipEventRecord(ev, st));
      CK(hipEventSynchronize(ev));
      ta.push_back(us(t0, clk::now()));
      t0 = clk::now();
      hipLaunchKernelGGL(sum_kernel, dim3(256), dim3(256), 0, st, (const double*)h, b / 8, dout, nullptr, 0ull);
      CK(hipEventRecord(ev, st));
      CK(hipEventSynchronize(ev));
      tb.push_back(us(t0, clk::now()));
    }
    printf("%12zu  %8.1f | %8.1f\n", b, med(ta), med(tb));
  }
  // ---- 5. wake-up latency: event wait vs spinning on a pinned word
  {
    volatile unsigned long long* flag = (volatile unsigned long long*)h;
    std::vector<double> ta, tb, tc;
    for (int rep = 0; rep < 200; ++rep) {
      auto t0 = clk::now();
      hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, st, (const double*)d, (size_t)64, dout, nullptr, 0ull);
      CK(hipEventRecord(ev, st));
      CK(hipEventSynchronize(ev));
      ta.push_back(us(t0, clk::now()));
      *flag = 0;
      t0 = clk::now();
      hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, st, (const double*)d, (size_t)64, dout, flag, (unsigned long long)(rep + 1));
      while (*flag != (unsigned long long)(rep + 1)) { }
      tb.push_back(us(t0, clk::now()));
      CK(hipStreamSynchronize(st));
      t0 = clk::now();
      hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, st, (const double*)d, (size_t)64, dout, nullptr, 0ull);
      CK(hipStreamSynchronize(st));
      tc.push_back(us(t0, clk::now()));
    }
    printf("\n5. tiny kernel + wait: event sync %.1f us | spin on a pinned word %.1f us | stream sync %.1f us\n", med(ta), med(tb), med(tc));
    // D2H completion seen by spinning on the LAST word of the destination (prefilled with a sentinel)
    for (size_t b : {(size_t)768064, (size_t)6447528}) {
      std::vector<double> t1, t2;
      unsigned long long* last = (unsigned long long*)(h + b - 8);
      for (int rep = 0; rep < REP; ++rep) {
        CK(hipMemsetAsync(d + b - 8, 0x11, 8, st));
        CK(hipStreamSynchronize(st));
        *(volatile unsigned long long*)last = 0xDEADBEEFull;
        auto t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        while (*(volatile unsigned long long*)last == 0xDEADBEEFull) { }
        t1.push_back(us(t0, clk::now()));
        CK(hipStreamSynchronize(st));
        t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t2.push_back(us(t0, clk::now()));
      }
      printf("   D2H %zu bytes: spin on the last word %.1f us | event %.1f us\n", b, med(t1), med(t2));
    }
  }
  // ---- 6. full duplex: D2H of 6.4 MB on stream 2 while an H2D of 0.77 MB runs on stream 1
  {
    std::vector<double> ta, tb;
    const size_t bd = 6447528, bu = 768064;
    for (int rep = 0; rep < REP; ++rep) {
      auto t0 = clk::now();
      CK(hipMemcpyAsync(h, d, bd, hipMemcpyDeviceToHost, st));
      CK(hipEventRecord(ev, st));
      CK(hipEventSynchronize(ev));
      CK(hipMemcpyAsync(d + (64u << 20), h + (64u << 20), bu, hipMemcpyHostToDevice, st));
      CK(hipEventRecord(ev, st));
      CK(hipEventSynchronize(ev));
      ta.push_back(us(t0, clk::now()));
      t0 = clk::now();
      CK(hipMemcpyAsync(h, d, bd, hipMemcpyDeviceToHost, st));
      CK(hipEventRecord(ev, st));
      CK(hipMemcpyAsync(d + (64u << 20), h + (64u << 20), bu, hipMemcpyHostToDevice, st2));
      CK(hipEventRecord(ev2, st2));
      CK(hipEventSynchronize(ev));
      CK(hipEventSynchronize(ev2));
      tb.push_back(us(t0, clk::now()));
    }
    printf("\n6. D2H 6.4 MB then H2D 0.77 MB, serial %.1f us | on two streams at once %.1f us\n", med(ta), med(tb));
  }

  // ---- 7. how the host learns that a D2H has finished: hipEventSynchronize | spinning on hipEventQuery | spinning on
  //         hipStreamQuery | a stream write-value behind the copy + spinning on that pinned word
  {
    unsigned long long* word = (unsigned long long*)(h + (100u << 20));
    void* dword = nullptr;
    bool have_wv = hipHostGetDevicePointer(&dword, word, 0) == hipSuccess;
    for (size_t b : {(size_t)8, (size_t)768064, (size_t)6447528}) {
      std::vector<double> t1, t2, t3, t4;
      for (int rep = 0; rep < REP; ++rep) {
        auto t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t1.push_back(us(t0, clk::now()));
        t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) { }
        t2.push_back(us(t0, clk::now()));
        t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        while (hipStreamQuery(st) == hipErrorNotReady) { }
        t3.push_back(us(t0, clk::now()));
        if (have_wv) {
          *(volatile unsigned long long*)word = 0;
          t0 = clk::now();
          CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
          if (hipStreamWriteValue64(st, dword, (uint64_t)(rep + 1), 0) != hipSuccess) { have_wv = false; (void)hipGetLastError(); CK(hipStreamSynchronize(st)); continue; }
          while (*(volatile unsigned long long*)word != (unsigned long long)(rep + 1)) { }
          t4.push_back(us(t0, clk::now()));
          CK(hipStreamSynchronize(st));
        }
      }
      printf("7. D2H %8zu bytes: event sync %.1f | spin hipEventQuery %.1f | spin hipStreamQuery %.1f | write-value + spin %.1f us\n", b, med(t1), med(t2),
             med(t3), t4.empty() ? -1.0 : med(t4));
    }
  }
  // ---- 8. a copy KERNEL (16 bytes per lane) instead of the DMA engine: pinned host -> device and device -> pinned host
  {
    for (size_t b : {(size_t)768064, (size_t)2879776, (size_t)6447528}) {
      std::vector<double> t1, t2;
      for (int rep = 0; rep < REP; ++rep) {
        auto t0 = clk::now();
        hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, st, (const double2*)h, (double2*)d, b / 16);
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) { }
        t1.push_back(us(t0, clk::now()));
        t0 = clk::now();
        hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, st, (const double2*)d, (double2*)h, b / 16);
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) { }
        t2.push_back(us(t0, clk::now()));
      }
      printf("8. copy kernel %8zu bytes: host -> device %.1f us (%.1f GB/s) | device -> host %.1f us (%.1f GB/s)\n", b, med(t1), b / med(t1) / 1e3,
             med(t2), b / med(t2) / 1e3);
    }
  }
  // ---- 7. how the host learns that a D2H has finished: hipEventSynchronize | spinning on hipEventQuery | spinning on
  //         hipStreamQuery | a stream write-value behind the copy + spinning on that pinned word
  {
    unsigned long long* word = (unsigned long long*)(h + (100u << 20));
    void* dword = nullptr;
    bool have_wv = hipHostGetDevicePointer(&dword, word, 0) == hipSuccess;
    for (size_t b : {(size_t)8, (size_t)768064, (size_t)6447528}) {
      std::vector<double> t1, t2, t3, t4;
      for (int rep = 0; rep < REP; ++rep) {
        auto t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
        t1.push_back(us(t0, clk::now()));
        t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) { }
        t2.push_back(us(t0, clk::now()));
        t0 = clk::now();
        CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
        while (hipStreamQuery(st) == hipErrorNotReady) { }
        t3.push_back(us(t0, clk::now()));
        if (have_wv) {
          *(volatile unsigned long long*)word = 0;
          t0 = clk::now();
          CK(hipMemcpyAsync(h, d, b, hipMemcpyDeviceToHost, st));
          if (hipStreamWriteValue64(st, dword, (uint64_t)(rep + 1), 0) != hipSuccess) { have_wv = false; (void)hipGetLastError(); CK(hipStreamSynchronize(st)); continue; }
          while (*(volatile unsigned long long*)word != (unsigned long long)(rep + 1)) { }
          t4.push_back(us(t0, clk::now()));
          CK(hipStreamSynchronize(st));
        }
      }
      printf("7. D2H %8zu bytes: event sync %.1f | spin hipEventQuery %.1f | spin hipStreamQuery %.1f | write-value + spin %.1f us\n", b, med(t1), med(t2),
             med(t3), t4.empty() ? -1.0 : med(t4));
    }
  }
  // ---- 8. a copy KERNEL (16 bytes per lane) instead of the DMA engine: pinned host -> device and device -> pinned host
  {
    for (size_t b : {(size_t)768064, (size_t)2879776, (size_t)6447528}) {
      std::vector<double> t1, t2;
      for (int rep = 0; rep < REP; ++rep) {
        auto t0 = clk::now();
        hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, st, (const double2*)h, (double2*)d, b / 16);
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) { }
        t1.push_back(us(t0, clk::now()));
        t0 = clk::now();
        hipLaunchKernelGGL(copy_kernel, dim3(512), dim3(256), 0, st, (const double2*)d, (double2*)h, b / 16);
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) { }
        t2.push_back(us(t0, clk::now()));
      }
      printf("8. copy kernel %8zu bytes: host -> device %.1f us (%.1f GB/s) | device -> host %.1f us (%.1f GB/s)\n", b, med(t1), b / med(t1) / 1e3,
             med(t2), b / med(t2) / 1e3);
    }
  }
  return 0;
}
