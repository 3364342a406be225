// Mechanism probe for "chained launches" (DESIGN.md section 3c): two DEPENDENT stages of the ST block in ONE launch, the consumer stage
// starting slab by slab while the producer stage is still walking the time axis.  Synthetic stand-ins with the C2 block-1 geometry:
//   producers : 256 workgroups x 512 threads, workgroup p owns (window b, node tile j) pairs of an equal split and walks T steps; per step it is
//               busy for `prod_us`, then one wave writes the 16 x 16 fp32 tile of slab (b, t) with 16-byte write-through (sc1) stores, drains
//               its stores and bumps ready[b * T + t]                                  (= tc1_fwd_kernel publishing A tiles)
//   consumers : one workgroup of 256 threads per (slab, part); waits for ready[slab] == TILES, stages the 13 KiB slab with sc1 buffer loads
//               (no L1 hit possible on a line another CU rewrote), checks EVERY word against the value of THIS launch, is busy `cons_us`
//                                                                                   (= gconv_fwd_kernel staging A)
// Variants:  two   = two launches, plain stores / loads (what the library does today)
//            chain = one launch, virtual block index from an atomic ticket (producers first: a waiting workgroup only ever waits for
//                    workgroups that already run), unused waves of the narrower role exit at once, counters reset by the last finisher
//            chain_acq = the same with ONE agent-scope acquire fence + plain loads in the consumer instead of sc1 loads
// Every launch writes a different payload (device epoch word bumped by the last finisher) and consumers pre-read their slab with PLAIN loads
// before waiting (L1-warm: the stale-line hazard of MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/chain_probe.hip -o /tmp/chain_probe && /tmp/chain_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int B = 32, T = 6, TILES = 13, N = 207, PARTS = 4, NPROD = 256;
constexpr int SLABS = B * T, ITEMS = B * TILES, NCONS = SLABS * PARTS;
constexpr long long kSpinLimit = 100000000ll;   // 1 s of the 100 MHz wall clock

struct Ctl {
    unsigned* ticket; unsigned* done; unsigned* ready; unsigned* error; unsigned* epoch; unsigned* bad;
    long long* stamps;   // [4]: min producer start, max producer end, min consumer start, max consumer end (wall clock)
    int cstride;         // words between the arrival counters of consecutive slabs (1: six counters per 128-byte line; 32: a line each)
};

__device__ __forceinline__ void busy_us(float us) {
    const long long t0 = wall_clock64(), ticks = (long long)(us * 100.0f);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ unsigned payload(unsigned epoch, int slab, int row, int q) { return epoch * 2654435761u + (unsigned)(slab * 4096 + row * 4 + q) * 40503u; }

__device__ __forceinline__ void st4_sc1(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

template <bool CHAIN>
__device__ void producer_body(float* A, Ctl c, int vb, float prod_us) {
    const unsigned epoch = __hip_atomic_load((gu32*)c.epoch, RLX_AGENT);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long units = (long)ITEMS * T, lo = units * vb / NPROD, hi = units * (vb + 1) / NPROD;
    if (tid == 0) atomicMin((unsigned long long*)&c.stamps[0], (unsigned long long)wall_clock64());
    for (long u = lo; u < hi; ++u) {
        const int item = (int)(u / T), t = (int)(u - (long)item * T), b = item / TILES, j = item - b * TILES, slab = b * T + t;
        if (wave != 4) busy_us(prod_us);   // (the storing wave is an "E" wave: it is not on the critical path of the step)
        __syncthreads();
        if (wave == 4) {   // one "E" wave writes the tile: lane -> (row = lane >> 2, quad = lane & 3)
            const int row = j * 16 + (lane >> 2), q = lane & 3;
            if (row < N) {
                f32x4 v;
                for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(float, payload(epoch, slab, row, q * 4 + i) & 0x3fffffffu);
                float* p = A + ((size_t)slab * N + row) * 16 + q * 4;
                if (CHAIN) st4_sc1(p, v); else *reinterpret_cast<f32x4*>(p) = v;
            }
            if (CHAIN) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing wave drains (write-through stores: acknowledged by memory)
                if (lane == 0) __hip_atomic_fetch_add((gu32*)c.ready + slab * c.cstride, 1u, RLX_AGENT);
            }
        }
    }
    if (tid == 0) atomicMax((unsigned long long*)&c.stamps[1], (unsigned long long)wall_clock64());
}

template <int MODE, int SLEEP = 8, int REP = 1>   // 0: plain loads (separate launch), 1: wait + sc1 loads, 2: wait + acquire fence + plain loads
__device__ void consumer_body(const float* A, float* out, Ctl c, int vb, float cons_us, float* lds) {
    const unsigned epoch = __hip_atomic_load((gu32*)c.epoch, RLX_AGENT);
    const int tid = threadIdx.x, slab = vb / PARTS;
    const float* As = A + (size_t)slab * N * 16;
    if (tid == 0) atomicMin((unsigned long long*)&c.stamps[2], (unsigned long long)wall_clock64());
    if (MODE != 0) {
        // L1-warm: touch the slab with plain loads BEFORE it is complete (values of the previous launch, or half-written lines)
        float s = 0.f;
        for (int i = tid; i < N * 4; i += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(As + (size_t)i * 4); s += v[0]; }
        lds[tid] = s;
        if (tid == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load((gu32*)c.ready + slab * c.cstride, RLX_AGENT) < (unsigned)TILES) {
                for (int r = 0; r < REP; ++r) __builtin_amdgcn_s_sleep(SLEEP);
                if (wall_clock64() - t0 > kSpinLimit) { __hip_atomic_store((gu32*)c.error, 1u, RLX_AGENT); break; }
            }
            if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    unsigned bad = 0;
    float s = 0.f;
    if (MODE == 1) {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)As, 0, N * 64, 0x00020000);
        for (int i0 = tid; i0 < N * 4; i0 += 4 * 256) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i < N * 4 ? i : N * 4 - 1) * 16, 0, 16 /* sc1 */);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256;
                if (i < N * 4)
                    for (int k = 0; k < 4; ++k) { bad += v[u][k] != (payload(epoch, slab, i >> 2, (i & 3) * 4 + k) & 0x3fffffffu); s += __builtin_bit_cast(float, v[u][k]); }
            }
        }
    } else {
        for (int i = tid; i < N * 4; i += 256) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(As + (size_t)i * 4);
            for (int k = 0; k < 4; ++k) { bad += v[k] != (payload(epoch, slab, i >> 2, (i & 3) * 4 + k) & 0x3fffffffu); s += __builtin_bit_cast(float, v[k]); }
        }
    }
    if (bad) atomicAdd(c.bad, bad);
    busy_us(cons_us);
    __syncthreads();
    out[(size_t)vb * 256 + tid] = s + lds[tid & 63];
    if (tid == 0) atomicMax((unsigned long long*)&c.stamps[3], (unsigned long long)wall_clock64());
}

__device__ __forceinline__ void last_finisher_reset(Ctl c, unsigned total) {
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add((gu32*)c.done, 1u, RLX_AGENT);
        if (d == total - 1) {
            for (int i = 0; i < SLABS; ++i) __hip_atomic_store((gu32*)c.ready + i * c.cstride, 0u, RLX_AGENT);
            __hip_atomic_store((gu32*)c.ticket, 0u, RLX_AGENT);
            __hip_atomic_store((gu32*)c.done, 0u, RLX_AGENT);
            __hip_atomic_fetch_add((gu32*)c.epoch, 1u, RLX_AGENT);
        }
    }
}

template <int CMODE, int SLEEP, int REP>
__global__ __launch_bounds__(512) void chain_kernel(float* A, float* out, Ctl c, float prod_us, float cons_us) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) reinterpret_cast<unsigned*>(smem)[300] = __hip_atomic_fetch_add((gu32*)c.ticket, 1u, RLX_AGENT);
    __syncthreads();
    const int vb = __builtin_amdgcn_readfirstlane((int)reinterpret_cast<unsigned*>(smem)[300]);
    if (vb < NPROD) {
        producer_body<true>(A, c, vb, prod_us);
    } else {
        if (threadIdx.x >= 256) return;   // the consumer role is 4 waves wide: the other 4 leave (a finished wave is not counted by s_barrier)
        consumer_body<CMODE, SLEEP, REP>(A, out, c, vb - NPROD, cons_us, smem);
    }
    last_finisher_reset(c, NPROD + NCONS);
}
__global__ __launch_bounds__(512) void prod_kernel(float* A, Ctl c, float prod_us) { producer_body<false>(A, c, (int)blockIdx.x, prod_us); }
__global__ __launch_bounds__(256) void cons_kernel(const float* A, float* out, Ctl c, float cons_us) {
    extern __shared__ float smem[];
    consumer_body<0>(A, out, c, (int)blockIdx.x, cons_us, smem);
}   // (the two-launch variant keeps one payload: nothing can be stale across a kernel boundary)
__global__ void any_k1(long long* st, float us) { if (threadIdx.x == 0) atomicMin((unsigned long long*)&st[0], (unsigned long long)wall_clock64()); busy_us(us); if (threadIdx.x == 0) atomicMax((unsigned long long*)&st[1], (unsigned long long)wall_clock64()); }
__global__ void any_k2(long long* st) { if (threadIdx.x == 0) atomicMin((unsigned long long*)&st[2], (unsigned long long)wall_clock64()); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const float prod_us = argc > 1 ? atof(argv[1]) : 3.5f, cons_us = argc > 2 ? atof(argv[2]) : 5.0f;
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    float *A, *out; unsigned* words; long long* stamps;
    CK(hipMalloc(&A, (size_t)SLABS * N * 16 * 4 + 4096)); CK(hipMalloc(&out, (size_t)NCONS * 256 * 4));
    CK(hipMalloc(&words, (SLABS * 32 + 64) * 4)); CK(hipMalloc(&stamps, 64));
    CK(hipMemset(words, 0, (SLABS * 32 + 64) * 4)); CK(hipMemset(A, 0, (size_t)SLABS * N * 16 * 4 + 4096));
    Ctl c{words, words + 1, words + 32, words + 2, words + 3, words + 4, stamps, 1};
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset_stamps = [&]() { long long h[4] = {0x7fffffffffffffffll, 0, 0x7fffffffffffffffll, 0}; CK(hipMemcpy(stamps, h, 32, hipMemcpyHostToDevice)); };
    auto report = [&](const char* name, float ms) {
        unsigned h[8]; long long hs[4];
        CK(hipMemcpy(h, words, 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs, stamps, 32, hipMemcpyDeviceToHost));
        printf("%-16s %8.2f us per step | wrong words %u, spin give-ups %u | last launch: producers %.1f us, consumers start %+.1f us after the first producer, end %+.1f us after the last producer\n",
               name, 1e3 * ms / reps, h[4], h[2], (hs[1] - hs[0]) / 100.0, (hs[2] - hs[0]) / 100.0, (hs[3] - hs[1]) / 100.0);
        unsigned z = 0; CK(hipMemcpy(words + 4, &z, 4, hipMemcpyHostToDevice));
    };
    const size_t lds = 28672;   // what tc1_fwd_kernel asks for: every workgroup of the chained launch reserves it
    for (int pass = 0; pass < 2; ++pass) {
        // ---- two launches --------------------------------------------------------------------------------------------------------
        for (int w = 0; w < 2; ++w) {
            if (w == 1) CK(hipEventRecord(e0, st));
            for (int i = 0; i < (w ? reps : 10); ++i) {
                if (i == reps - 1) { CK(hipStreamSynchronize(st)); reset_stamps(); }
                hipLaunchKernelGGL(prod_kernel, dim3(NPROD), dim3(512), lds, st, A, c, prod_us);
                hipLaunchKernelGGL(cons_kernel, dim3(NCONS), dim3(256), 13568, st, (const float*)A, out, c, cons_us);
            }
            if (w == 1) CK(hipEventRecord(e1, st));
        }
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        report("two", ms);
        // ---- chained: sc1 loads; counter stride 1 / 32 words; s_sleep 8 / 32 / 127 / 10 x 127 between polls; acquire-fence variant ------
        for (int var = 0; var < 7; ++var) {
            const int stride = (var == 0 || var == 6) ? 1 : 32, slp = var <= 1 ? 8 : var == 2 ? 32 : var == 3 ? 127 : var == 4 ? 1270 : var == 5 ? -8 : 127;
            c.cstride = stride;
            for (int w = 0; w < 2; ++w) {
                if (w == 1) CK(hipEventRecord(e0, st));
                for (int i = 0; i < (w ? reps : 10); ++i) {
                    if (i == reps - 1) { CK(hipStreamSynchronize(st)); reset_stamps(); }
                    const dim3 g(NPROD + NCONS), b(512);
                    if (slp == 8) hipLaunchKernelGGL((chain_kernel<1, 8, 1>), g, b, lds, st, A, out, c, prod_us, cons_us);
                    else if (slp == 32) hipLaunchKernelGGL((chain_kernel<1, 32, 1>), g, b, lds, st, A, out, c, prod_us, cons_us);
                    else if (slp == 127) hipLaunchKernelGGL((chain_kernel<1, 127, 1>), g, b, lds, st, A, out, c, prod_us, cons_us);
                    else if (slp == 1270) hipLaunchKernelGGL((chain_kernel<1, 127, 10>), g, b, lds, st, A, out, c, prod_us, cons_us);
                    else hipLaunchKernelGGL((chain_kernel<2, 8, 1>), g, b, lds, st, A, out, c, prod_us, cons_us);
                }
                if (w == 1) CK(hipEventRecord(e1, st));
            }
            CK(hipStreamSynchronize(st));
            CK(hipGetLastError());
            CK(hipEventElapsedTime(&ms, e0, e1));
            char nm[64];
            snprintf(nm, sizeof nm, "chain s%d %s%d", stride, slp < 0 ? "acq" : "z", slp < 0 ? 8 : slp);
            report(nm, ms);
        }
    }
    // ---- hipExtAnyOrderLaunch on this device: does the second kernel start before the first has ended? -----------------------------------
    for (int flag = 0; flag <= 1; ++flag) {
        reset_stamps();
        hipLaunchKernelGGL(any_k1, dim3(128), dim3(256), 0, st, stamps, 50.0f);
        hipExtLaunchKernelGGL(any_k2, dim3(128), dim3(256), 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, stamps);
        hipError_t e = hipStreamSynchronize(st);
        long long hs[4]; CK(hipMemcpy(hs, stamps, 32, hipMemcpyDeviceToHost));
        printf("hipExtLaunchKernel flags=%d (%s): second kernel starts %.1f us after the first one STARTED (first runs %.1f us)\n", flag, hipGetErrorString(e),
               (hs[2] - hs[0]) / 100.0, (hs[1] - hs[0]) / 100.0);
    }
    return 0;
}
