// hip_runtime.h -- TEST INFRASTRUCTURE: a host-only stand-in for the dozen HIP runtime calls pockit_amd/csrc/pk_runtime.cpp
// uses, so that the host runtime (pinned rings, double-buffered staging, prepared-x protocol, copy batching, polling waits)
// can be built with -fsanitize=address,undefined and driven on the CPU (tests/fake_hip/driver.cpp, tests/test_runtime_sanitized.py).
// Never on a GPU, never part of the product.
//
// Semantics that matter for the protocol are kept: work enqueued on a stream is DEFERRED -- it runs only when the host
// waits for it or polls (one queued operation per hipStreamQuery / hipEventQuery call) -- so a host that touches a staging
// buffer before its upload has run, or reads a result before its copy, sees wrong data exactly as it would on the device.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <tuple>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorNotSupported = 801 };

struct FakeStream;
struct FakeEvent;
struct FakeModule;
struct FakeFunction;
typedef FakeStream* hipStream_t;
typedef FakeEvent* hipEvent_t;
typedef FakeModule* hipModule_t;
typedef FakeFunction* hipFunction_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 {
  double x, y;
};
extern thread_local dim3 blockIdx, threadIdx, blockDim, gridDim;

enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3, hipMemoryTypeUnregistered = 0 };
struct hipPointerAttribute_t {
  int type;
};
struct hipIpcMemHandle_t {
  char reserved[64];
};

#define hipHostMallocDefault 0u
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
#define hipDeviceMallocFinegrained 1u
#define hipHostRegisterMapped 2u
#define hipHostRegisterPortable 1u
#define hipIpcMemLazyEnablePeerAccess 1u
#define hipStreamCaptureModeThreadLocal 1
#define HIP_LAUNCH_PARAM_BUFFER_POINTER ((void*)0x01)
#define HIP_LAUNCH_PARAM_BUFFER_SIZE ((void*)0x02)
#define HIP_LAUNCH_PARAM_END ((void*)0x03)
#define __global__
#define __launch_bounds__(x)

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceSynchronize();
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned flags);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* attr, const void* p);
hipError_t hipMemset(void* p, int value, size_t bytes);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipModuleLoadData(hipModule_t* m, const void* image);
hipError_t hipModuleUnload(hipModule_t m);
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name);
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                 unsigned shmem, hipStream_t s, void** params, void** extra);
hipError_t hipExtModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                    size_t shmem, hipStream_t s, void** params, void** extra, hipEvent_t start, hipEvent_t stop,
                                    unsigned flags);
hipError_t hipStreamBeginCapture(hipStream_t s, int mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t e);
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p);
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned flags);
hipError_t hipIpcCloseMemHandle(void* p);

// a host function enqueued like a kernel (hipLaunchKernelGGL of a __global__ function compiled for the host)
void fake_hip_enqueue(hipStream_t s, std::function<void()> fn);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                     \
  do {                                                                                                  \
    const dim3 g_ = (grid), b_ = (block);                                                               \
    auto args_ = std::make_tuple(__VA_ARGS__); /* by value, NOW: as a real launch takes them */          \
    fake_hip_enqueue((stream), [=]() {                                                                  \
      gridDim = g_;                                                                                     \
      blockDim = b_;                                                                                    \
      for (unsigned bx_ = 0; bx_ < g_.x; ++bx_)                                                         \
        for (unsigned tx_ = 0; tx_ < b_.x; ++tx_) {                                                     \
          blockIdx = dim3(bx_, 0, 0);                                                                   \
          threadIdx = dim3(tx_, 0, 0);                                                                  \
          std::apply(kernel, args_);                                                                    \
        }                                                                                               \
    });                                                                                                 \
  } while (0)
