// TEST-ONLY stand-in for <hip/hip_runtime_api.h>: just enough of the HIP runtime API, backed by malloc and executed
// synchronously, to compile csrc/engine.cpp's HOST logic (stream pool, tick assembler, segment arena, pipelined ingest
// bookkeeping) for ThreadSanitizer / AddressSanitizer runs on a CPU - GPU sanitizers are not available on the pool.
// Never on the product's include path (cutter_vad_amd/_build.py builds with hipcc against the real ROCm headers); used only
// by tools/san_tick/run.sh and tests/test_tick_sanitizers.py.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct fakeStream *hipStream_t;
typedef struct fakeEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
    int multiProcessorCount;
};

extern "C" int fake_hip_fail_after;     // > 0: the n-th hipMemcpyAsync from now fails once (error-path tests)

static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof *p);
    std::strcpy(p->name, "host stand-in (sanitizer build)");
    std::strcpy(p->gcnArchName, "gfx950:stand-in");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stand-in failure"; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemset(void *p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t);
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
static inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(1)); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
