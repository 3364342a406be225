// Device-side helpers shared by all STGCN kernels (gfx950 / CDNA4, wave64).
//
// MFMA conventions used throughout (cdna_hip_programming.md section 3, v_mfma_f32_16x16x4_f32):
//   A operand: lane l holds A[row = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][col = l & 15]
//   C/D      : lane l, reg r holds D[row = 4 * (l >> 4) + r][col = l & 15]
// "16-chunk trick": a lane loads 4 consecutive k values (one 16-byte load) and feeds component s
// to MFMA step s, so MFMA step s of chunk kc contracts k = kc*16 + 4*(l>>4) + s.  Both operands use
// the same permutation of k, so the sum is unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- optional per-phase cycle stamps (diagnostic build -DSTGCN_PHASE_TIMING only; compiled out otherwise) -------
#ifdef STGCN_PHASE_TIMING
#ifdef STGCN_PHASE_WALL
#define STGCN_PHASE_CLOCK() wall_clock64()   // 100 MHz, one time base for the whole device: workgroup timelines
#else
#define STGCN_PHASE_CLOCK() clock64()        // shader cycles, per-XCD base: phase durations inside a workgroup
#endif
__device__ long long stgcn_phase_buf[4096 * 16];
__device__ int stgcn_phase_kid;
#define STGCN_PHASE(kid, i)                                                                                     \
    do {                                                                                                        \
        if (stgcn_phase_kid == (kid) && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096)                \
            stgcn_phase_buf[blockIdx.x * 16 + (i)] = STGCN_PHASE_CLOCK();                                       \
    } while (0)
// busy-time accumulators of one role of a wave-specialised kernel (cycles between leaving a barrier and reaching the next one)
#define STGCN_ACC_DECL() long long stgcn_acc_ = 0, stgcn_t0_ = 0, stgcn_acc2_ = 0, stgcn_t2_ = 0
#define STGCN_ACC_BEGIN() stgcn_t0_ = clock64()
#define STGCN_ACC_END() stgcn_acc_ += clock64() - stgcn_t0_
#define STGCN_ACC2_BEGIN() stgcn_t2_ = clock64()
#define STGCN_ACC2_END() stgcn_acc2_ += clock64() - stgcn_t2_
#define STGCN_ACC_STORE(kid, i, cond)                                                                           \
    do {                                                                                                        \
        if (stgcn_phase_kid == (kid) && (cond) && blockIdx.x < 4096) stgcn_phase_buf[blockIdx.x * 16 + (i)] = stgcn_acc_; \
    } while (0)
#define STGCN_ACC2_STORE(kid, i, cond)                                                                          \
    do {                                                                                                        \
        if (stgcn_phase_kid == (kid) && (cond) && blockIdx.x < 4096) stgcn_phase_buf[blockIdx.x * 16 + (i)] = stgcn_acc2_; \
    } while (0)
#else
#define STGCN_PHASE(kid, i) ((void)0)
#define STGCN_ACC_DECL() ((void)0)
#define STGCN_ACC_BEGIN() ((void)0)
#define STGCN_ACC_END() ((void)0)
#define STGCN_ACC_STORE(kid, i, cond) ((void)0)
#define STGCN_ACC2_BEGIN() ((void)0)
#define STGCN_ACC2_END() ((void)0)
#define STGCN_ACC2_STORE(kid, i, cond) ((void)0)
#endif

namespace stgcn {

constexpr int kThreads = 256;   // 4 waves per workgroup, one per SIMD
constexpr int kTileRows = 64;   // rows of a flat row tile (4 MFMA m-tiles)
constexpr int kSegMax = 128;    // K columns staged in LDS per segment

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// 16-byte WRITE-THROUGH store (global_store_dwordx4 ... sc1) for bulk outputs that the same kernel never reads again.  A plain store
// leaves its line dirty in the XCD's L2 and the kernel boundary then pays for the write-back (MI355X_MICROARCH.md "boundary": + dirty
// bytes / 6 TB/s); an sc1 store leaves L2 while the kernel is still computing and costs the same per instruction.  Measured on one box
// (profiles/r2-31_wt_store_ab.txt, two alternating runs each): plain 0.4197, epilogue stores write-through 0.4155, in-loop stores as
// well 0.4132 ms per step -- about 1 %, not the 50 us the dirty-byte rule would predict for 300 MB per step.  The compiler does not count inline-asm memory operations in vmcnt: every later
// s_waitcnt it places is therefore conservative (it waits for these stores as well), never too short.  STGCN_WT_STORES=0 at build
// time falls back to plain stores (A/B); the CPU emulator always uses the plain store.
#ifndef STGCN_WT_STORES
#define STGCN_WT_STORES 1
#endif
__device__ __forceinline__ void st4_wt(float* p, f32x4 v) {
#if STGCN_WT_STORES && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
    st4(p, v);
#endif
}
// the same for stores inside a time-stepping loop, whose wave goes on to wait for later loads (STGCN_WT_STORES >= 2: measured slower)
__device__ __forceinline__ void st4_wt2(float* p, f32x4 v) {
#if STGCN_WT_STORES >= 2 && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
    st4(p, v);
#endif
}
// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4, gfx950): lane i of the wave copies its 16 bytes at g to
// lds_base + 16 * i, where lds_base must be WAVE-UNIFORM (it travels in M0) -- the LDS image of one instruction is always the
// 1 KiB lane-linear block, any permutation has to be applied to the per-lane source address.  Completion is counted in vmcnt;
// __syncthreads() drains it before the barrier.
__device__ __forceinline__ void glds16(const void* g, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}
// XCD-aware work placement (speed only, never correctness): workgroup b is observed to run on XCD b % 8 and every XCD
// has its own L2, so work items that re-read the same rows (the Kt taps of neighbouring time steps, the m-chunks of a
// weight-gradient row chunk) are given to ONE XCD as a contiguous range: workgroup b of n takes item
// start(b % 8) + b / 8, where XCD x owns n/8 (+1 if x < n % 8) consecutive items.
#ifndef STGCN_XCD_MAP
#define STGCN_XCD_MAP 1
#endif
__device__ __forceinline__ int xcd_item(int b, int n) {
#if STGCN_XCD_MAP
    const int x = b & 7, per = n >> 3, rem = n & 7;
    return x * per + (x < rem ? x : rem) + (b >> 3);
#else
    return b;
#endif
}

// Integer division by a launch-time value that is almost always a power of two (channel counts, float4 columns of a
// row): an integer divide is ~25 VALU instructions per element on this ISA and the staging / epilogue loops do one per
// 16-byte load.  sh = pow2_shift(d) once (uniform), then fast_div(x, d, sh) per element (x >= 0).
__device__ __forceinline__ int pow2_shift(int d) { return (d > 0 && (d & (d - 1)) == 0) ? 31 - __builtin_clz((unsigned)d) : -1; }
__device__ __forceinline__ int fast_div(int x, int d, int sh) { return sh >= 0 ? (x >> sh) : x / d; }

// v_exp_f32 + v_rcp_f32 (1 ulp each), 4 instructions.  (__frcp_rn / "1.0f / x" expand to the 11-instruction IEEE divide
// sequence, __expf to a range-reduced polynomial: together they were a quarter of the gated conv's row pass.)
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

// tanh(x) = 2 sigmoid(2x) - 1 on the same two hardware instructions (abs. error ~4e-7): libm's tanhf is a branchy ~60-instruction
// call that the gated kernels would pay per element in forward AND backward (and that spilled the role-specialised kernels)
__device__ __forceinline__ float tanh_f(float x) { return 2.0f * sigmoid_f(2.0f * x) - 1.0f; }

// ---- gate math (reference model/layers.py:105 GLU, :109 GTU) --------------------------------
// forward: h = act(u) * s ; act = identity (glu) or tanh (gtu)
__device__ __forceinline__ float gate_fwd(float u, float s, int act) { return (act == 0 ? u : tanh_f(u)) * s; }
// backward: returns dU, dQ for upstream dh
__device__ __forceinline__ void gate_bwd(float dh, float u, float s, int act, float& du, float& dq) {
    if (act == 0) {
        du = dh * s;
        dq = dh * u * s * (1.0f - s);
    } else {
        const float th = tanh_f(u);
        du = dh * s * (1.0f - th * th);
        dq = dh * th * s * (1.0f - s);
    }
}

// ---- block-wide sum of two values (256 threads) ----------------------------------------------
// red must point to >= 8 floats of LDS.  All threads must call.
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();   // protect red from a previous use
    if (lane == 0) {
        red[wave] = a;
        red[4 + wave] = b;
    }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
}

// ---- Philox4x32-10 counter-based RNG (dropout mask; regenerated in backward, never stored) ---
struct Philox4 { uint32_t x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
    uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 r = {c0, c1, c2, c3};
    return r;
}
// keep-scale factors for the 4 consecutive elements [4*idx4, 4*idx4+4): 0 (dropped) or 1/(1-p).
// thresh = floor(p * 2^32) clamped; an element is kept iff its 32-bit draw >= thresh.
__device__ __forceinline__ f32x4 dropout_scale4(uint64_t idx4, uint64_t seed, uint64_t offset, uint32_t thresh, float scale) {
    const Philox4 r = philox4x32_10(idx4, offset, seed);
    f32x4 k;
    k[0] = r.x >= thresh ? scale : 0.f;
    k[1] = r.y >= thresh ? scale : 0.f;
    k[2] = r.z >= thresh ? scale : 0.f;
    k[3] = r.w >= thresh ? scale : 0.f;
    return k;
}

}  // namespace stgcn
