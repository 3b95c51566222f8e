// fake_hip.cpp -- TEST INFRASTRUCTURE: host-only HIP stand-in (see hip/hip_runtime.h here).  Streams defer their work; the
// "kernels" of a loaded code object are tiny host functions that write RECOGNIZABLE values computed from the inputs they
// are handed at the moment they run (so stale inputs, missed copies and wrong offsets show up in the outputs).
#include "hip/hip_runtime.h"

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../pockit_amd/csrc/pk_abi.h"
#include "fake_hip.h"

thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;

struct FakeEvent {
  uint64_t ticket = 0;      // position in its stream's history; 0 = never recorded (complete)
  FakeStream* stream = nullptr;
};
struct FakeStream {
  std::deque<std::function<void()>> q;
  uint64_t enq = 0, done = 0;
  void push(std::function<void()> f) { q.push_back(std::move(f)); ++enq; }
  bool step() {
    if (q.empty()) return false;
    auto f = std::move(q.front());
    q.pop_front();
    f();
    ++done;
    return true;
  }
  void drain() { while (step()) { } }
};
struct FakeFunction {
  std::string name;
};
struct FakeModule {
  std::map<std::string, FakeFunction> fn;
};

static std::set<FakeStream*> g_streams;
static std::set<void*> g_host, g_dev;
static FakeSizes g_sizes;
static std::vector<std::string> g_log;
static int g_default_steps = 1;      // queued operations a poll lets run

void fake_hip_set_sizes(const FakeSizes& s) { g_sizes = s; }
const std::vector<std::string>& fake_hip_log() { return g_log; }
void fake_hip_clear_log() { g_log.clear(); }
size_t fake_hip_live_allocations() { return g_host.size() + g_dev.size(); }

void fake_hip_enqueue(hipStream_t s, std::function<void()> fn) { s->push(std::move(fn)); }

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorNotReady ? "not ready" : "fake error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceSynchronize() {
  for (auto* s : g_streams) s->drain();
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new FakeStream(); g_streams.insert(*s); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { s->drain(); g_streams.erase(s); delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { s->drain(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) {
  for (int k = 0; k < g_default_steps; ++k) s->step();
  return s->q.empty() ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new FakeEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { e->stream = s; e->ticket = s->enq; return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) {
  if (!e->stream || e->stream->done >= e->ticket) return hipSuccess;
  e->stream->step();
  return e->stream->done >= e->ticket ? hipSuccess : hipErrorNotReady;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
  while (e->stream && e->stream->done < e->ticket) e->stream->step();
  return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

static void* alloc(std::set<void*>& book, size_t bytes) {
  void* p = std::malloc(bytes ? bytes : 8);
  std::memset(p, 0xA5, bytes ? bytes : 8);      // uninitialised "device" memory is garbage, not zeros
  book.insert(p);
  return p;
}
hipError_t hipMalloc(void** p, size_t bytes) { *p = alloc(g_dev, bytes); return hipSuccess; }
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  if (!g_dev.erase(p)) { std::fprintf(stderr, "fake_hip: hipFree of an unknown pointer\n"); std::abort(); }
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { *p = alloc(g_host, bytes); return hipSuccess; }
hipError_t hipHostFree(void* p) {
  if (!g_host.erase(p)) { std::fprintf(stderr, "fake_hip: hipHostFree of an unknown pointer\n"); std::abort(); }
  std::free(p);
  return hipSuccess;
}
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p) {
  // only exact allocation starts are known here: enough for the runtime's "may a kernel store into this target" question
  if (g_host.count(const_cast<void*>(p))) { attr->type = hipMemoryTypeHost; return hipSuccess; }
  if (g_dev.count(const_cast<void*>(p))) { attr->type = hipMemoryTypeDevice; return hipSuccess; }
  attr->type = hipMemoryTypeUnregistered;
  return hipErrorInvalidValue;
}
hipError_t hipMemset(void* p, int value, size_t bytes) { std::memset(p, value, bytes); return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  for (auto* s : g_streams) s->drain();
  std::memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
  g_log.push_back(kind == hipMemcpyHostToDevice ? "h2d" : kind == hipMemcpyDeviceToHost ? "d2h" : "copy");
  s->push([=]() { std::memcpy(dst, src, bytes); });
  return hipSuccess;
}

hipError_t hipModuleLoadData(hipModule_t* m, const void*) { *m = new FakeModule(); return hipSuccess; }
hipError_t hipModuleUnload(hipModule_t m) { delete m; return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name) {
  m->fn[name].name = name;
  *f = &m->fn[name];
  return hipSuccess;
}

// ---- the "kernels": values any reader can recompute from (x, lambda, sigma, position)
double fake_f(const double* x, int n) { return x[0] + 2.0 * x[n - 1]; }
double fake_grad(const double* x, int n, int64_t i) { return 2.0 * x[i % n] + 1.0; }
double fake_g(const double* x, int n, int64_t j) { return x[(j + 1) % n] - 3.0; }
double fake_jac(const double* x, int n, int64_t p, bool constant) { return constant ? (p % 2 ? 1.0 : -1.0) : 0.5 * x[p % n] + (double)p; }
double fake_hess(const double* x, const double* lam, double sigma, int n, int m, int64_t p) {
  return lam[p % m] * sigma + x[(p + 3) % n] + (double)p;
}
static bool is_const(const std::vector<std::pair<int64_t, int64_t>>& runs, int64_t p) {
  for (auto& r : runs)
    if (p >= r.first && p < r.second) return true;
  return false;
}

static void run_kernel(const std::string& name, const std::vector<char>& argbuf) {
  const FakeSizes& S = g_sizes;
  const bool cyc = name == "pk_cycle" || name == "pk_cyclec";
  const size_t off = cyc ? PK_CYCLE_ARGS_OFFSET : 0;
  if (argbuf.size() < off + sizeof(PkArgs)) return;      // (the host-side copy kernel goes another way)
  PkArgs A;
  std::memcpy(&A, argbuf.data() + off, sizeof A);
  const int n = S.n, m = S.m;
  const bool x_part = name == "pk_xall" || cyc;
  if ((name == "pk_fin" || cyc) && (A.flags & 1) && A.o_f) A.o_f[0] = fake_f(A.x, n);
  if ((x_part || name == "pk_grad") && A.o_grad)
    for (int64_t i = 0; i < n; ++i) A.o_grad[i] = fake_grad(A.x, n, i);
  if ((x_part || name == "pk_g") && A.o_g)
    for (int64_t j = 0; j < m; ++j) A.o_g[j] = fake_g(A.x, n, j);
  // (pk_cycle serves the compact layouts itself when its flags say so: bit 9 the Jacobian, bit 8 the Hessian)
  const bool cyc_cj = name == "pk_cyclec" && (A.flags & 512), cyc_ch = name == "pk_cyclec" && (A.flags & 256);
  if ((x_part || name == "pk_jac") && A.o_jac && !cyc_cj)
    for (int64_t p = 0; p < S.nnz_J; ++p) A.o_jac[p] = fake_jac(A.x, n, p, is_const(S.jconst, p));
  if ((name == "pk_jacc" || cyc_cj) && A.o_jac)
    for (int64_t p = 0; p < S.nnz_Jc; ++p) A.o_jac[p] = fake_jac(A.x, n, p, is_const(S.jconst_compact, p)) + 0.25;
  if ((name == "pk_hess" || cyc) && A.o_hess && !cyc_ch)
    for (int64_t p = 0; p < S.nnz_H; ++p) A.o_hess[p] = fake_hess(A.x, A.lam, A.sigma, n, m, p);
  if ((name == "pk_hessc" || cyc_ch) && A.o_hess)
    for (int64_t p = 0; p < S.nnz_Hc; ++p) A.o_hess[p] = fake_hess(A.x, A.lam, A.sigma, n, m, p) - 0.5;
  if (name == "pk_csr" && A.csr_out) {
    for (int p = 0; p < A.n_csr; ++p) A.csr_out[p] = A.csr_seg ? -7.0 : A.csr_in[A.csr_perm[p]];
  }
}

hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t s,
                                 void**, void** extra) {
  std::vector<char> buf;
  if (extra && extra[0] == HIP_LAUNCH_PARAM_BUFFER_POINTER && extra[2] == HIP_LAUNCH_PARAM_BUFFER_SIZE) {
    const size_t sz = *static_cast<size_t*>(extra[3]);
    buf.assign(static_cast<const char*>(extra[1]), static_cast<const char*>(extra[1]) + sz);
  }
  const std::string name = f->name;
  g_log.push_back(name);
  (void)gx;
  s->push([name, buf]() { run_kernel(name, buf); });
  return hipSuccess;
}
hipError_t hipExtModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, size_t shmem,
                                    hipStream_t s, void** params, void** extra, hipEvent_t start, hipEvent_t stop, unsigned) {
  hipEventRecord(start, s);
  hipError_t e = hipModuleLaunchKernel(f, gx / (bx ? bx : 1), gy, gz, bx, by, bz, (unsigned)shmem, s, params, extra);
  hipEventRecord(stop, s);
  return e;
}

// graphs are not part of what the sanitized build exercises: capture is refused (the runtime reports it and goes on)
hipError_t hipStreamBeginCapture(hipStream_t, int) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) { std::memset(h, 0, sizeof *h); std::memcpy(h, &p, sizeof p); return hipSuccess; }
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) { std::memcpy(p, &h, sizeof *p); return hipSuccess; }
hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
