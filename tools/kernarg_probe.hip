// GPU box: does a kernel take more than 4 KB of arguments by value on this stack?  (PkArgs holds the phase records by value.)
// Build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/kernarg_probe tools/kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { int head; int v[N]; };
template <int N> __global__ void probe(Big<N> b, long long* out) {
  long long s = 0;
  for (int k = threadIdx.x; k < N; k += blockDim.x) s += b.v[k];
  atomicAdd(reinterpret_cast<unsigned long long*>(out), static_cast<unsigned long long>(s));
}
template <int N> int run(long long* d) {
  Big<N> b; b.head = 0; long long want = 0;
  for (int k = 0; k < N; ++k) { b.v[k] = 3 * k + 1; want += b.v[k]; }
  long long zero = 0, got = -1;
  if (hipMemcpy(d, &zero, 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe<N>, dim3(1), dim3(64), 0, 0, b, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(&got, d, 8, hipMemcpyDeviceToHost);
  std::printf("%6zu bytes of arguments: %s (%lld, want %lld)\n", sizeof(Big<N>) + 8, e == hipSuccess && got == want ? "ok" : hipGetErrorString(e), got, want);
  return !(e == hipSuccess && got == want);
}
int main() {
  long long* d; if (hipMalloc(&d, 8) != hipSuccess) return 2;
  int bad = 0;
  bad += run<512>(d); bad += run<1023>(d); bad += run<2048>(d); bad += run<4096>(d); bad += run<8192>(d);
  return bad;
}
