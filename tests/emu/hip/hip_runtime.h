// CPU emulation shim for <hip/hip_runtime.h>  --  TEST INFRASTRUCTURE ONLY.
//
// tests/emu/build_emu.py compiles the *unmodified* product sources in stgcn_amd/csrc/ with the host
// clang++ and `-I tests/emu` so that this header is found instead of the ROCm one.  Every
// workgroup is executed as 256 cooperative fibers (a minimal x86-64 context switch) on one host thread; __syncthreads(),
// wave shuffles and the f32 MFMA are emulated as rendezvous points, the MFMA with the documented
// gfx950 lane->element maps (cdna_hip_programming.md section 3) and an fmaf chain in k order, so index
// math, fragment layouts, masking and barrier placement of the kernels are exercised on the CPU
// before any GPU minute is spent.  It is not a performance model and it never ships.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
// residency queries of the launch heuristics: the emulator runs workgroups one after another, any capacity will do
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return 0; }
template <typename K>
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* nb, K, int threads, size_t) { *nb = 2048 / threads; return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyDeviceToHost = 2 };
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }   // launches run synchronously

// events are host timestamps (launches run synchronously): the library's per-launch timer then reports emulation time per kernel
typedef double* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new double(0.0); return 0; }
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }   // launches run synchronously
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    if (e) *e = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    return 0;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (a && b) ? (float)(*b - *a) : 0.f; return 0; }

namespace emu {
constexpr int kWave = 64;
constexpr size_t kLdsBytes = 160 * 1024;
struct Fiber {
    void* sp = nullptr;                 // saved stack pointer (emu_switch)
    std::vector<char> stack;
    bool done = false;
    unsigned tid = 0;
    const unsigned* wait_gen = nullptr; // blocked at a barrier until *wait_gen != wait_val (the scheduler does not resume it before)
    unsigned wait_val = 0;
};
struct WaveState {
    float a[kWave], b[kWave];
    float a8[kWave][8], b8[kWave][8];   // operands of the 32-deep bf16 MFMA
    uint32_t u[kWave];
    int arrived = 0;
    unsigned gen = 0;
};
struct State {
    dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
    void* main_sp = nullptr;
    std::vector<Fiber> fibers;          // pool: only grows; a launch uses the first nthreads
    Fiber* cur = nullptr;
    std::function<void()> body;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    int nthreads = 0;
    int live = 0;                       // threads of the running workgroup that have not returned yet
    std::vector<WaveState> waves;
    long n_mfma = 0;
    // workgroups that wait for PEERS of their own launch (symmetric exchanges: a tile needs the row statistics of tiles that have not run
    // yet): the waiting workgroup is abandoned and run again from its start after the rest of the grid; the counter bumps it already made
    // are not repeated (pub_skip of them are skipped in the re-run), its stores are idempotent
    bool defer_req = false;
    int pub_seen = 0, pub_skip = 0;
    long ticket_fixed = -1, ticket_taken = -1;   // a re-run workgroup keeps the ticket of its abandoned run
    // test setting of the bounded waits (stgcn_set_chain_spin_ticks < 0): a wait that no re-run can complete "times out" like the device's
    // instead of aborting the emulation -- set by the waiting workgroup (may) and by the launch loop once nothing else makes progress (now)
    bool peer_may_give_up = false, peer_give_up = false;
    // LDS race check (build_emu.py --race): per-thread counts of the workgroup / wave barriers passed, the name of the running kernel
    std::vector<unsigned> bpass, wpass;
    const char* kname = "?";
    size_t lds_used = kLdsBytes;
    long races = 0, races_intra = 0;
};
extern State g;
void yield();
void block_barrier();
void wave_barrier();
bool run_block();   // false: the workgroup deferred itself (peer_defer) and has to be run again
bool same_bytes(const void* a, const void* b, size_t n);   // (uninstrumented compare, LDS race check build)
void race_launch_end();   // LDS race check build: abort if the launch that just ended had a cross-wave LDS race (no-op otherwise)
// called by ONE thread of a workgroup whose peers have not all run yet; does not return
[[noreturn]] void peer_defer();
// counts the calls of the running workgroup; true = this bump was already made by an abandoned run of the same workgroup
static inline bool peer_publish_done() { return g.pub_seen++ < g.pub_skip; }
static inline unsigned peer_ticket(unsigned* word) {
    if (g.ticket_fixed >= 0) return (unsigned)g.ticket_fixed;
    g.ticket_taken = (long)(*word)++;
    return (unsigned)g.ticket_taken;
}
}  // namespace emu

// The one dynamic-LDS array every kernel (all live in namespace stgcn) declares as
// `extern __shared__ float stgcn_smem[]`; defined in emu_runtime.cpp.
namespace stgcn { extern float stgcn_smem[]; }

#define threadIdx (emu::g.threadIdx_)
#define blockIdx (emu::g.blockIdx_)
#define blockDim (emu::g.blockDim_)
#define gridDim (emu::g.gridDim_)

static inline void __syncthreads() { emu::block_barrier(); }

typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: A[i][k] held by lane i + 16k, B[k][j] by lane j + 16k,
// D[i][j] for i = 4*(lane>>4) + r (r = 0..3), j = lane & 15.
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
    const unsigned t = emu::g.threadIdx_.x;
    emu::WaveState& w = emu::g.waves[t / emu::kWave];
    const int lane = t % emu::kWave;
    w.a[lane] = a;
    w.b[lane] = b;
    emu::wave_barrier();
    emu_f32x4 d = c;
    const int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.a[i + 16 * k], w.b[j + 16 * k], acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    if (lane == 0) emu::g.n_mfma++;
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32((a), (b), (c))

// v_mfma_f32_16x16x32_bf16: lane l holds 8 bf16 of A[i = l & 15][k = 8 * (l >> 4) + e] and of B[k][j = l & 15]
// (cdna_hip_programming.md section 3); C/D as the f32 form.  Products of bf16 values are exact in fp32; the
// accumulation order of the hardware is not specified, an fmaf chain in k order stands in for it.
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short emu_u16x8 __attribute__((ext_vector_type(8)));
static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
    const unsigned t = emu::g.threadIdx_.x;
    emu::WaveState& w = emu::g.waves[t / emu::kWave];
    const int lane = t % emu::kWave;
    const emu_u16x8 ua = __builtin_bit_cast(emu_u16x8, a), ub = __builtin_bit_cast(emu_u16x8, b);
    for (int e = 0; e < 8; ++e) {
        const uint32_t xa = (uint32_t)ua[e] << 16, xb = (uint32_t)ub[e] << 16;
        memcpy(&w.a8[lane][e], &xa, 4);
        memcpy(&w.b8[lane][e], &xb, 4);
    }
    emu::wave_barrier();
    emu_f32x4 d = c;
    const int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int e = 0; e < 8; ++e) acc = fmaf(w.a8[i + 16 * kg][e], w.b8[j + 16 * kg][e], acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    if (lane == 0) emu::g.n_mfma++;
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu_mfma_f32_16x16x32_bf16((a), (b), (c))
// v_mfma_f32_16x16x16_bf16: lane l holds 4 bf16 of A[i = l & 15][k = 4 * (l >> 4) + e] and of B[k][j = l & 15]; C/D as the f32 form
typedef short emu_s16x4 __attribute__((ext_vector_type(4)));
static inline emu_f32x4 emu_mfma_f32_16x16x16bf16_1k(emu_s16x4 a, emu_s16x4 b, emu_f32x4 c) {
    const unsigned t = emu::g.threadIdx_.x;
    emu::WaveState& w = emu::g.waves[t / emu::kWave];
    const int lane = t % emu::kWave;
    for (int e = 0; e < 4; ++e) {
        const uint32_t xa = (uint32_t)(unsigned short)a[e] << 16, xb = (uint32_t)(unsigned short)b[e] << 16;
        memcpy(&w.a8[lane][e], &xa, 4);
        memcpy(&w.b8[lane][e], &xb, 4);
    }
    emu::wave_barrier();
    emu_f32x4 d = c;
    const int j = lane & 15;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int e = 0; e < 4; ++e) acc = fmaf(w.a8[i + 16 * kg][e], w.b8[j + 16 * kg][e], acc);
        d[r] = acc;
    }
    emu::wave_barrier();
    if (lane == 0) emu::g.n_mfma++;
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, x, y, z) emu_mfma_f32_16x16x16bf16_1k((a), (b), (c))
// global_load_lds: lane i copies `size` bytes from its own global pointer to (wave-uniform LDS base) + size * i + offset
static inline void emu_global_load_lds(const void* g, void* lds_base, unsigned size, int off) {
    char* dst = static_cast<char*>(lds_base) + (emu::g.threadIdx_.x % emu::kWave) * size + off;
#ifdef STGCN_EMU_RACE
    // LDS race check: a copy that leaves the destination as it is (the pipelined GEMM's copy slots past the last block repeat that block:
    // "same bytes to the same place" from another wave, benign by construction) is not an access at all
    if (emu::same_bytes(dst, g, size)) return;
#endif
    memcpy(dst, g, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), (size), (off))
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
static inline long long wall_clock64() { static long long t = 0; return ++t; }
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()   // (the point where the lanes of a wave meet: LDS exchange inside a wave)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)

template <typename T>
static inline T emu_shfl_from(T v, int src_lane) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    const unsigned t = emu::g.threadIdx_.x;
    emu::WaveState& w = emu::g.waves[t / emu::kWave];
    const int lane = t % emu::kWave;
    uint32_t bits;
    memcpy(&bits, &v, 4);
    w.u[lane] = bits;
    emu::wave_barrier();
    uint32_t r = w.u[src_lane & 63];
    emu::wave_barrier();
    T out;
    memcpy(&out, &r, 4);
    return out;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu_shfl_from(v, (int)(emu::g.threadIdx_.x % 64) ^ mask); }
template <typename T> static inline T __shfl_down(T v, int d, int width = 64) { (void)width; int l = emu::g.threadIdx_.x % 64; return emu_shfl_from(v, l + d < 64 ? l + d : l); }
template <typename T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return emu_shfl_from(v, src); }

#define __expf(x) expf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32); }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }

template <typename... KArgs, typename... Args>
static inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    if (shmem > emu::kLdsBytes) { fprintf(stderr, "emu: LDS request %zu > 160 KiB\n", shmem); abort(); }
    if (block.x % 64 != 0 || block.y != 1 || block.z != 1) { fprintf(stderr, "emu: block must be 1-D multiple of 64\n"); abort(); }
    emu::g.gridDim_ = grid;
    emu::g.blockDim_ = block;
    emu::g.nthreads = (int)block.x;
    emu::g.lds_used = shmem;
    emu::g.body = [=]() { kernel(static_cast<KArgs>(args)...); };
    struct Deferred { dim3 b; int pubs; long ticket; };
    std::vector<Deferred> again;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                emu::g.blockIdx_ = dim3(bx, by, bz);
                emu::g.pub_skip = 0;
                emu::g.ticket_fixed = emu::g.ticket_taken = -1;
                if (!emu::run_block()) again.push_back(Deferred{dim3(bx, by, bz), emu::g.pub_seen, emu::g.ticket_taken});
            }
    while (!again.empty()) {   // workgroups that waited for peers: from their start again, now that the rest of the grid has run
        std::vector<Deferred> next;
        bool progress = false;
        for (const Deferred& d : again) {
            emu::g.blockIdx_ = d.b;
            emu::g.pub_skip = d.pubs;
            emu::g.ticket_fixed = d.ticket;
            emu::g.ticket_taken = d.ticket;
            if (emu::run_block()) progress = true;
            else {
                if (emu::g.pub_seen > d.pubs) progress = true;
                next.push_back(Deferred{d.b, emu::g.pub_seen > d.pubs ? emu::g.pub_seen : d.pubs, emu::g.ticket_taken});
            }
        }
        if (!progress) {
            if (emu::g.peer_may_give_up && !emu::g.peer_give_up) emu::g.peer_give_up = true;   // (the next round's waits run out)
            else { fprintf(stderr, "emu: deadlock: %zu workgroups wait for peer counters that nobody will complete\n", next.size()); abort(); }
        }
        again.swap(next);
    }
    emu::g.peer_may_give_up = emu::g.peer_give_up = false;
    emu::g.pub_skip = 0;
    emu::g.ticket_fixed = -1;
    emu::race_launch_end();
}
