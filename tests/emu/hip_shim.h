// TEST INFRASTRUCTURE ONLY — never part of the product, never loaded by it.
//
// hip_shim.h: lets g++ compile csrc/mi355sat.hip + device/kernels.hip.h into
// tests/emu/libmi355sat_emu.so, a *wavefront emulator* build used by the CPU-side
// tests (`-m "not gpu"`) to exercise the kernels' logic without a GPU and to run
// them under ASan/UBSan (GPU sanitizers are not available on the pool).
//
// Model: one workgroup at a time; its 64 lanes are ucontext fibers that run
// sequentially and rendezvous at every wave collective (ballot, shuffle,
// readfirstlane) and at every wave/LDS fence, which is where the real hardware's
// lockstep matters to the algorithm.  Between two rendezvous points lane 0 runs
// to completion before lane 1 starts, so code that relies on "all lanes load
// before any lane stores" within one segment must have a fence in between (it
// does; see kernels.hip.h).  Timing and memory-ordering behaviour of the real
// machine are NOT modelled: GPU tests remain the authority.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define MI355SAT_EMU 1

// ---- language shims --------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local      /* (two emulated solvers may run in two host threads) */

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline int2 make_int2(int a, int b) { return int2{a, b}; }

namespace emu {
struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void collective_begin(uint64_t contribution, int site);   // publish + rendezvous
uint64_t collective_read(int lane);                        // value published by `lane`
void fence_rendezvous();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
}  // namespace emu
#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

using std::max;
using std::min;

static inline unsigned long long __ballot(int p) {
    emu::collective_begin(p ? 1 : 0, 1);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) m |= (unsigned long long)(emu::collective_read(l) & 1) << l;
    return m;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline long long __double_as_longlong(double d) { long long x; __builtin_memcpy(&x, &d, 8); return x; }
static inline double __longlong_as_double(long long x) { double d; __builtin_memcpy(&d, &x, 8); return d; }
static inline int __shfl(int v, int src, int /*w*/ = 64) {
    emu::collective_begin((uint64_t)(uint32_t)v, 2);
    return (int)(uint32_t)emu::collective_read(src & 63);
}
static inline int __shfl_xor(int v, int mask, int /*w*/ = 64) {
    emu::collective_begin((uint64_t)(uint32_t)v, 3);
    return (int)(uint32_t)emu::collective_read(((int)emu::t_threadIdx.x ^ mask) & 63);
}
static inline unsigned long long __shfl_xor(unsigned long long v, int mask, int /*w*/ = 64) {
    emu::collective_begin(v, 4);
    return emu::collective_read(((int)emu::t_threadIdx.x ^ mask) & 63);
}
static inline int __shfl_up(int v, int delta, int /*w*/ = 64) {
    emu::collective_begin((uint64_t)(uint32_t)v, 5);
    int src = (int)emu::t_threadIdx.x - delta;
    return src < 0 ? v : (int)(uint32_t)emu::collective_read(src);
}
static inline int __builtin_amdgcn_readfirstlane(int v) {
    emu::collective_begin((uint64_t)(uint32_t)v, 6);
    return (int)(uint32_t)emu::collective_read(0);
}
#define __builtin_amdgcn_fence(order, scope) emu::fence_rendezvous()
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
unsigned long long emu_ticks_100mhz();
// s_memrealtime is a scalar instruction: one value per wave.  The fibers of a wave run one after the other
// here, so each would read a later clock and the lanes could disagree about "slice over" - lane 0's reading
// is handed to all of them.
static inline unsigned long long emu_uniform_ticks() {
    emu::collective_begin(emu_ticks_100mhz(), 7);
    return emu::collective_read(0);
}
#define __builtin_amdgcn_s_memrealtime() emu_uniform_ticks()
#ifndef __clang__
static inline unsigned long long __builtin_readcyclecounter() { return __builtin_ia32_rdtsc(); }
#endif
template <class T>
static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T>
static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
template <class T>
static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T>
static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T>
static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
// (the kernels call these on plain and on volatile-qualified pointers)
template <class T>
static inline T __hip_atomic_load(const volatile T* p, int, int) { return *p; }
template <class T, class V>
static inline T __hip_atomic_fetch_or(volatile T* p, V v, int, int) { T o = *p; *p = o | (T)v; return o; }
template <class T, class V>
static inline T __hip_atomic_fetch_and(volatile T* p, V v, int, int) { T o = *p; *p = o & (T)v; return o; }
template <class T, class V>
static inline T __hip_atomic_fetch_add(volatile T* p, V v, int, int) { T o = *p; *p = o + (T)v; return o; }
template <class T, class V>
static inline bool __hip_atomic_compare_exchange_strong(volatile T* p, T* expected, V v, int, int, int) {
    T o = *p;
    if (o == *expected) { *p = (T)v; return true; }
    *expected = o;
    return false;
}
// dynamic LDS: one static 160 KiB block per (sequentially executed) workgroup
#define HIP_DYNAMIC_SHARED(type, var) static thread_local type var[163840 / sizeof(type)];

// ---- runtime shims ------------------------------------------------------------------
typedef int hipError_t;
#define hipSuccess 0
typedef void* hipStream_t;
typedef struct emu_event* hipEvent_t;
struct emu_event { double t; };
enum { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
#define hipStreamNonBlocking 0
#define hipHostMallocMapped 0
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, int) { *s = (void*)1; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
double emu_now();
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0}; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e3); return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n, int) { *p = calloc(1, n); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, int) {
    for (size_t r = 0; r < h; r++) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return 0;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)8 << 30; return 0; }
static inline hipError_t hipGetLastError() { return 0; }
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
