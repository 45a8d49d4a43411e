// tests/emu/hip/hip_runtime.h -- a CPU stand-in for <hip/hip_runtime.h>, used ONLY to build
// tests/emu/libairmodes_emu.so: the unmodified product sources (gr-air-modes_amd/csrc/*.hip)
// compiled with g++ so that kernel index logic, barriers, wave collectives and the host
// control flow can be exercised by the "-m 'not gpu'" tests in a container without a GPU.
//
// TEST INFRASTRUCTURE ONLY.  The product package never loads this library; it is not a
// fallback.  Every workgroup runs as blockDim.x cooperative fibers (ucontext) on one OS
// thread: __syncthreads() and the wave collectives (__ballot, __shfl*) yield until the
// whole workgroup / 64-lane wave has arrived.  Device memory is host memory.
#pragma once
#define AM_HIP_EMULATION 1   /* am_is_emulated() reports it */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <functional>
#include <chrono>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline void __threadfence_system() {}
static inline void __threadfence() {}
#define __builtin_readcyclecounter() 0ull
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __constant__
#define __launch_bounds__(...)
// a clock that always advances (kernels may wait on it)
static inline long long clock64() { static long long t = 0; return t += 1000; }
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::dyn_smem());

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct hipemu_stream *hipStream_t;
struct hipemu_event { double t_ms; };          // launches run synchronously: an event is the wall clock at its record
typedef struct hipemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hipemu {
void *dyn_smem();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void barrier();
int barrier_or(int pred);
unsigned long long ballot(int pred);
uint64_t shuffle(uint64_t v, int src_lane_rel, int width, int mode);   // mode 0 idx, 1 down, 2 up, 3 xor
} // namespace hipemu

inline void __syncthreads() { hipemu::barrier(); }
inline int __syncthreads_or(int pred) { return hipemu::barrier_or(pred); }
// wave-level compiler fence in the product = a real rendezvous of the wave's fibers here
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
// (fibers of one OS thread never interleave inside this)
#define __hip_atomic_compare_exchange_strong(p, expected, desired, so, fo, scope) ((*(p) == *(expected)) ? (*(p) = (desired), true) : (*(expected) = *(p), false))
// only used on values that are already the same in every lane of the wave
#define __builtin_amdgcn_readfirstlane(x) (x)
#ifndef __clang__
#define __builtin_nontemporal_load(p) (*(p))
#endif
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::ballot(0))
inline unsigned long long __ballot(int pred) { return hipemu::ballot(pred); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <class T> inline T hipemu_shfl(T v, int arg, int width, int mode)
{
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    raw = hipemu::shuffle(raw, arg, width, mode);
    T out;
    memcpy(&out, &raw, sizeof(T));
    return out;
}
template <class T> inline T __shfl(T v, int lane, int width = 64) { return hipemu_shfl(v, lane, width, 0); }
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) { return hipemu_shfl(v, (int)d, width, 1); }
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) { return hipemu_shfl(v, (int)d, width, 2); }
template <class T> inline T __shfl_xor(T v, int m, int width = 64) { return hipemu_shfl(v, m, width, 3); }

// fibers of one OS thread never interleave inside these, so plain read-modify-write is atomic
template <class T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }

inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 3; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
// (every pointer is "device memory" here)
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void *devicePointer; void *hostPointer; };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) { a->type = hipMemoryTypeDevice; a->device = 0; a->devicePointer = (void *)p; a->hostPointer = (void *)p; return hipSuccess; }
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event{0.0}; return hipSuccess; }
#define hipEventDisableSystemFence 0x20000000
#define hipEventDisableTiming 0x2
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new hipemu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t)
{
    if (e) e->t_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }     // launches run to completion synchronously
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (a && b) ? (float)(b->t_ms - a->t_ms) : 0.0f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 2; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
