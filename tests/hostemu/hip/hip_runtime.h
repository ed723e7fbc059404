// TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT.
//
// A stand-in <hip/hip_runtime.h> that lets the *unmodified* HIP kernel sources under
// pytorch-dense-correspondence_amd/csrc/ be compiled for the host (x86-64, ROCm clang++) and
// executed lane by lane, so that index arithmetic, LDS layouts, MFMA fragment mappings and barrier
// placement can be checked on a machine that has no GPU.  It is only ever put on the include path
// by tests/hostemu/build_emu.py; the shipped library is built by hipcc against the real header and
// never sees this file.  It is slow (thousands of times slower than a CU) and models no caches,
// no memory ordering and no hazards -- passing here proves logic, not hardware behaviour.
//
// Execution model: one OS thread per concurrently running workgroup; inside it every work-item is
// a cooperative fiber (hand-written x86-64 context switch).  __syncthreads() and the wave-level
// operations (shuffles, MFMA) are rendezvous points at which fibers yield to one another.
// Wavefront = 64 lanes.  MFMA fragment maps follow /opt/skills/guides/cdna_hip_programming.md
// section 3: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31],
// C/D: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)   (32x32x2 f32)
// and A[l&15][k = l>>4], B[k = l>>4][l&15], C/D: col = lane&15, row = 4*(lane>>4) + reg (16x16x4 f32).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __constant__ static

typedef int hipError_t;
typedef void* hipStream_t;
static const hipError_t hipSuccess = 0;
static const hipError_t hipErrorInvalidValue = 1;
inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess(emu)" : "hipError(emu)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
typedef struct hipemu_event { double t; }* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
// (the emulation executes every launch synchronously: streams and cross-stream dependencies are no-ops)
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event{0.0}; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)new int(0); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete (int*)s; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

namespace hipemu {

struct Idx { unsigned x, y, z; };
extern thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

extern "C" void hipemu_switch(void** save_sp, void* new_sp);

enum { RUNNABLE = 0, WAIT_BLOCK = 1, DONE = 2 };

// LDS-DMA (`buffer_load ... lds`) in flight: 16 bytes captured at issue, written to LDS when the issuing work-item's
// counted vmcnt wait retires it (the LATEST moment the hardware allows) -- or at issue (HIPEMU_LDS_DMA=early, the earliest)
struct PendingDma {
    void* dst;
    unsigned char data[16];
};
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE;
    unsigned tid = 0;
    std::vector<PendingDma> dma;   // FIFO, oldest first
};

struct WaveState {
    int arrived = 0;
    unsigned gen = 0;
    int lanes = 64;
    uint64_t buf_a[64];
    uint64_t buf_b[64];
    unsigned char wide_a[64][16];   // 16-byte MFMA operands (8 x f16 per lane)
    unsigned char wide_b[64][16];
};

struct Worker {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    void* sched_sp = nullptr;
    int cur = -1;
    int nthreads = 0;
    int blk_arrived = 0;
    unsigned blk_gen = 0;
    const std::function<void()>* body = nullptr;
    static const size_t kStack = 192 * 1024;
};
extern thread_local Worker* t_worker;

inline void yield_to_scheduler() {
    Worker* w = t_worker;
    Fiber& f = w->fibers[w->cur];
    hipemu_switch(&f.sp, w->sched_sp);
}

inline void block_barrier() {
    Worker* w = t_worker;
    unsigned gen = w->blk_gen;
    if (++w->blk_arrived == w->nthreads) {
        w->blk_arrived = 0;
        w->blk_gen++;
        for (auto& f : w->fibers) if (f.state == WAIT_BLOCK) f.state = RUNNABLE;
        return;
    }
    Fiber& f = w->fibers[w->cur];
    f.state = WAIT_BLOCK;
    while (w->blk_gen == gen) yield_to_scheduler();
    f.state = RUNNABLE;
}

inline WaveState& my_wave() { Worker* w = t_worker; return w->waves[w->cur >> 6]; }
inline int lane_id() { return t_worker->cur & 63; }

inline void wave_barrier() {
    WaveState& ws = my_wave();
    unsigned gen = ws.gen;
    if (++ws.arrived == ws.lanes) { ws.arrived = 0; ws.gen++; return; }
    while (ws.gen == gen) yield_to_scheduler();
}

template <class T> inline T wave_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "exchange up to 8 bytes");
    WaveState& ws = my_wave();
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    ws.buf_a[lane_id()] = raw;
    wave_barrier();
    uint64_t got = ws.buf_a[src_lane & 63];
    if ((src_lane & 63) >= ws.lanes) got = raw;
    wave_barrier();
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);

template <class F> inline void launch(dim3 grid, dim3 block, F&& f) {
    std::function<void()> body(std::forward<F>(f));
    run_grid(grid, block, body);
}

}  // namespace hipemu

#define threadIdx (::hipemu::t_threadIdx)
#define blockIdx (::hipemu::t_blockIdx)
#define blockDim (::hipemu::t_blockDim)
#define gridDim (::hipemu::t_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { ::hipemu::block_barrier(); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() {}

template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = ::hipemu::lane_id();
    return ::hipemu::wave_exchange(v, l ^ mask);
}
template <class T> static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int l = ::hipemu::lane_id();
    int src = l + (int)delta;
    if ((l % width) + (int)delta >= width) src = l;
    return ::hipemu::wave_exchange(v, src);
}
template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = ::hipemu::lane_id();
    return ::hipemu::wave_exchange(v, (l / width) * width + (src % width));
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) {
        int p = ::hipemu::wave_exchange(pred, i);
        if (p && i < ::hipemu::my_wave().lanes) m |= 1ull << i;
    }
    return m;
}

// ---- atomics (workgroups run on different OS threads) ----
static inline float atomicAdd(float* p, float v) {
    uint32_t* ip = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        float f;
        memcpy(&f, &old, 4);
        float nf = f + v;
        uint32_t nv;
        memcpy(&nv, &nf, 4);
        if (__atomic_compare_exchange_n(ip, &old, nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}
static inline double atomicAdd(double* p, double v) {
    uint64_t* ip = reinterpret_cast<uint64_t*>(p);
    uint64_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    for (;;) {
        double f;
        memcpy(&f, &old, 8);
        double nf = f + v;
        uint64_t nv;
        memcpy(&nv, &nf, 8);
        if (__atomic_compare_exchange_n(ip, &old, nv, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) {
    unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long expected, unsigned long long desired) {
    __atomic_compare_exchange_n(p, &expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;   // (the value found, like the device function)
}
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) {
    return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST);
}
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
// agent-scope relaxed atomic loads / stores (the cooperative batch-norm launches: statistics exchanged between workgroups)
#define __HIP_MEMORY_SCOPE_AGENT 4
static inline unsigned long long hipemu_atomic_load(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
static inline float hipemu_atomic_load(const float* p) {
    uint32_t u = __atomic_load_n(reinterpret_cast<const uint32_t*>(p), __ATOMIC_SEQ_CST);
    float f; memcpy(&f, &u, 4); return f;
}
static inline void hipemu_atomic_store(float* p, float v) {
    uint32_t u; memcpy(&u, &v, 4);
    __atomic_store_n(reinterpret_cast<uint32_t*>(p), u, __ATOMIC_SEQ_CST);
}
#define __hip_atomic_load(p, order, scope) hipemu_atomic_load(p)
#define __hip_atomic_store(p, v, order, scope) hipemu_atomic_store(p, v)
#define __builtin_amdgcn_s_sleep(n) std::this_thread::yield()
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline long long __double2ll_rn(double x) { return llrint(x); }   // (default rounding mode: to nearest even, as v_cvt does)
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#define unsafeAtomicAdd atomicAdd
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
template <class T> static inline T min(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max(T a, T b) { return a > b ? a : b; }

// ---- MFMA (f32 in / f32 accumulate), k-ordered fmaf chain like the hardware ----
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));

static inline hipemu_f32x16 hipemu_mfma_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    using namespace ::hipemu;
    WaveState& ws = my_wave();
    int l = lane_id();
    memcpy(&ws.buf_a[l], &a, 4);
    memcpy(&ws.buf_b[l], &b, 4);
    wave_barrier();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &ws.buf_a[row + 32 * k], 4);
            memcpy(&bv, &ws.buf_b[col + 32 * k], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wave_barrier();
    return c;
}
static inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    using namespace ::hipemu;
    WaveState& ws = my_wave();
    int l = lane_id();
    memcpy(&ws.buf_a[l], &a, 4);
    memcpy(&ws.buf_b[l], &b, 4);
    wave_barrier();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &ws.buf_a[row + 16 * k], 4);
            memcpy(&bv, &ws.buf_b[col + 16 * k], 4);
            acc = fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wave_barrier();
    return c;
}
// v_mfma_f32_32x32x16_f16: lane (i = l & 31, h = l >> 5) holds A[i][8h..8h+7] and B[8h..8h+7][i]; fp32 accumulate.
typedef _Float16 hipemu_h8 __attribute__((ext_vector_type(8)));
static inline hipemu_f32x16 hipemu_mfma_32x32x16f16(hipemu_h8 a, hipemu_h8 b, hipemu_f32x16 c, int, int, int) {
    using namespace ::hipemu;
    WaveState& ws = my_wave();
    int l = lane_id();
    memcpy(ws.wide_a[l], &a, 16);
    memcpy(ws.wide_b[l], &b, 16);
    wave_barrier();
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h) {
            _Float16 av[8], bv[8];
            memcpy(av, ws.wide_a[row + 32 * h], 16);
            memcpy(bv, ws.wide_b[col + 32 * h], 16);
            for (int k = 0; k < 8; ++k) acc += (float)av[k] * (float)bv[k];
        }
        c[r] = acc;
    }
    wave_barrier();
    return c;
}
// v_mfma_f32_16x16x32_f16: lane (i = l & 15, q = l >> 4) holds A[i][8q..8q+7] and B[8q..8q+7][i]; C/D: col = l & 15,
// row = 4 (l >> 4) + reg; fp32 accumulate.
static inline hipemu_f32x4 hipemu_mfma_16x16x32f16(hipemu_h8 a, hipemu_h8 b, hipemu_f32x4 c, int, int, int) {
    using namespace ::hipemu;
    WaveState& ws = my_wave();
    int l = lane_id();
    memcpy(ws.wide_a[l], &a, 16);
    memcpy(ws.wide_b[l], &b, 16);
    wave_barrier();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int q = 0; q < 4; ++q) {
            _Float16 av[8], bv[8];
            memcpy(av, ws.wide_a[row + 16 * q], 16);
            memcpy(bv, ws.wide_b[col + 16 * q], 16);
            for (int k = 0; k < 8; ++k) acc += (float)av[k] * (float)bv[k];
        }
        c[r] = acc;
    }
    wave_barrier();
    return c;
}
// raw buffer resources (stride 0): loads whose byte offset (voffset; the scalar offset is NOT range-checked, as on the
// hardware) reaches past num_records return zeros
struct hipemu_buffer_rsrc { const char* base; unsigned num_records; };
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
static inline hipemu_buffer_rsrc hipemu_make_buffer_rsrc(const void* p, short, int num_records, int) {
    return hipemu_buffer_rsrc{(const char*)p, (unsigned)num_records};
}
static inline hipemu_u32x4 hipemu_raw_buffer_load_b128(hipemu_buffer_rsrc r, int voffset, int soffset, int) {
    hipemu_u32x4 v = {0u, 0u, 0u, 0u};
    for (int i = 0; i < 4; ++i)
        if ((unsigned long long)(unsigned)voffset + 4ull * i + 4ull <= r.num_records)
            { unsigned w; memcpy(&w, r.base + (long long)(unsigned)voffset + (unsigned)soffset + 4 * i, 4); v[i] = w; }
    return v;
}
static inline hipemu_u32x2 hipemu_raw_buffer_load_b64(hipemu_buffer_rsrc r, int voffset, int soffset, int) {
    hipemu_u32x2 v = {0u, 0u};
    for (int i = 0; i < 2; ++i)
        if ((unsigned long long)(unsigned)voffset + 4ull * i + 4ull <= r.num_records)
            { unsigned w; memcpy(&w, r.base + (long long)(unsigned)voffset + (unsigned)soffset + 4 * i, 4); v[i] = w; }
    return v;
}
static inline void hipemu_raw_buffer_store_b128(hipemu_u32x4 v, hipemu_buffer_rsrc r, int voffset, int soffset, int) {
    if ((unsigned long long)(unsigned)voffset + 16ull <= r.num_records)
        memcpy(const_cast<char*>(r.base) + (long long)(unsigned)voffset + (unsigned)soffset, &v, 16);
}
#define __builtin_amdgcn_raw_buffer_store_b128 hipemu_raw_buffer_store_b128
#define __builtin_amdgcn_s_waitcnt(x) (hipemu_wait_vmcnt(((x) & 0xF) | (((x) >> 14) << 4)), __atomic_thread_fence(__ATOMIC_SEQ_CST))
#define __amdgpu_buffer_rsrc_t hipemu_buffer_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, n, flags) hipemu_make_buffer_rsrc((const void*)(p), stride, n, flags)
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu_raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_load_b64 hipemu_raw_buffer_load_b64
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu_mfma_32x32x16f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 hipemu_mfma_16x16x32f16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32
// ---- LDS-DMA + counted waits + raw barrier (conv_hl_kernels.hip)
static inline bool hipemu_dma_early() {
    static const int early = [] { const char* e = getenv("HIPEMU_LDS_DMA"); return (e && !strcmp(e, "early")) ? 1 : 0; }();
    return early != 0;
}
static inline void hipemu_buffer_load_lds(hipemu_buffer_rsrc r, void* lds_base, int size, int voffset, int soffset) {
    if (size != 16) { fprintf(stderr, "hipemu: LDS-DMA of %d bytes not modelled\n", size); abort(); }
    ::hipemu::Worker* w = ::hipemu::t_worker;
    ::hipemu::Fiber& f = w->fibers[w->cur];
    ::hipemu::PendingDma d;
    d.dst = (char*)lds_base + 16 * ::hipemu::lane_id();
    const hipemu_u32x4 v = hipemu_raw_buffer_load_b128(r, voffset, soffset, 0);
    memcpy(d.data, &v, 16);
    if (hipemu_dma_early()) memcpy(d.dst, d.data, 16);
    else f.dma.push_back(d);
}
static inline void hipemu_wait_vmcnt(int n) {   // retire the oldest LDS-DMA pieces until at most n are in flight
    ::hipemu::Worker* w = ::hipemu::t_worker;
    ::hipemu::Fiber& f = w->fibers[w->cur];
    while ((int)f.dma.size() > n) {
        memcpy(f.dma.front().dst, f.dma.front().data, 16);
        f.dma.erase(f.dma.begin());
    }
}
// ds_read_b64_tr_b16 (semantics measured on the MI355X, tools/tr_probe.bin): the 16 lanes of a group supply 8-byte runs
// S[s][0..3]; lane i of the group receives S[4 j + (i >> 2)][i & 3], j = 0..3
typedef short hipemu_s4 __attribute__((ext_vector_type(4)));
static inline hipemu_s4 hipemu_ds_read_tr16_b64(const void* p) {
    using namespace ::hipemu;
    WaveState& ws = my_wave();
    const int l = lane_id();
    memcpy(&ws.buf_a[l], p, 8);
    wave_barrier();
    const int g = l >> 4, i = l & 15;
    hipemu_s4 out;
    for (int j = 0; j < 4; ++j) {
        short v[4];
        memcpy(v, &ws.buf_a[16 * g + 4 * j + (i >> 2)], 8);
        out[j] = v[i & 3];
    }
    wave_barrier();
    return out;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) hipemu_ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, lp, sz, vo, so, a, b) hipemu_buffer_load_lds(rs, (void*)(lp), sz, vo, so)
#define DCN_WAIT_VMCNT(n) hipemu_wait_vmcnt(n)
#define DCN_WAIT_LGKMCNT0() ((void)0)
#define DCN_OPAQUE_INT(v) ((void)(v))
#define __builtin_amdgcn_s_barrier() ::hipemu::block_barrier()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
static inline int hipemu_readfirstlane(int v) { return ::hipemu::wave_exchange(v, 0); }
#define __builtin_amdgcn_readfirstlane hipemu_readfirstlane
