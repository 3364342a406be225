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
#include <type_traits>

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

// Tuning knobs whose A/B runs are over (each lost or tied at least twice, profiles/HISTORY_*): read from the environment only in
// -DSTGCN_EXPERIMENTS builds; the product build compiles their defaults in (every live knob doubles a tested surface).
#ifdef STGCN_EXPERIMENTS
#define STGCN_EXP_ENV(name) getenv(name)
#else
#define STGCN_EXP_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace stgcn {

constexpr int kThreads = 256;   // 4 waves per workgroup, one per SIMD
constexpr int kTileRows = 64;   // rows of a flat row tile (4 MFMA m-tiles)
#ifndef STGCN_GC_RING1
#define STGCN_GC_RING1 2   // r3-28: 2 / 3 / 4 chunks -> C2 gconv_bwd@0 24.8 / 27.0 / 30.5 us (registers: 7 / 6 / 5 waves per SIMD), forward equal
#endif
// operator-fragment chunks a wave of the slab-resident graph-conv kernels keeps in flight, by node tiles per wave (register budget)
constexpr int gc_ring(int maxq) { return maxq <= 1 ? STGCN_GC_RING1 : 2; }
constexpr int kSegMax = 128;    // K columns staged in LDS per segment

#ifndef STGCN_TS_DBG
#define STGCN_TS_DBG 0   // timing experiments only (wrong results): 1 = activation loads (ldx*) return an opaque zero, 2 = no MFMAs, 3 = no activation stores (stx*)
#endif
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
#if STGCN_TS_DBG == 2
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
#if STGCN_TS_DBG
__device__ __forceinline__ f32x4 dbg_opaque4() {
    f32x4 v;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 0\n\tv_mov_b32 %3, 0" : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]));
    return v;
}
__device__ __forceinline__ void dbg_keep4(f32x4 v) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
#endif
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

// ================================================================================================
// Storage / arithmetic type of the activations ("ET"): float (BASELINE.json configs[0], [1], [3]) or bf16 (configs[2], [4]).
//   float : fp32 tensors in HBM, exact fp32 matrix products (v_mfma_f32_16x16x4_f32, four per 16-deep product step)
//   bf16  : every activation / saved tensor / activation gradient that crosses a kernel boundary is stored as bf16 (RNE), every
//           matrix-product operand is rounded to bf16 where it enters the matrix cores (ONE v_mfma_f32_16x16x16_bf16 per 16-deep step:
//           a lane's 4 consecutive k values of the "16-chunk" permutation are exactly that instruction's operand layout), and
//           everything else -- accumulators, gates, LayerNorm statistics and parameters, master weights, gradient partials, the
//           optimizer -- stays fp32.  oracle/stblock_stages.py restates these rounding points (QuantBf16).
// Kernels take `typename ET`; tensors keep their `float*` slots in the argument structs (opaque base addresses: the plan sizes
// them in 4-byte units) and are re-typed with et_ptr<ET>() at the top of the kernel.
// ================================================================================================
struct bf16 { unsigned short v; };
// float storage, "bf16x3" matrix products: every operand is split x = hi + lo into two bf16 where it enters the matrix cores and a product
// is formed as hi*hi + hi*lo + lo*hi (three v_mfma_f32_16x16x16_bf16, fp32 accumulation; the dropped lo*lo term and the split residual
// are ~2^-16 relative to the product).  Three bf16 MFMAs of 17.5 cycles replace four fp32-input MFMAs of 32 cycles
// (profiles/r3-04_mfma_issue_rate.txt): 2.4 x fewer matrix-pipe cycles.  Used for the BACKWARD kernels of the fp32 configurations only
// (stgcn_set_bwd_precision): their parity bar is 1e-3 relative on gradients, 50 x above this error; the forward (1e-4 absolute on
// activations) keeps exact fp32 products.
struct f32x { float v; };
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));   // operand of v_mfma_f32_16x16x32_bf16
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_bits_rne(float x) {   // round-to-nearest-even; NaN stays NaN (quiet bit set), +-inf stays +-inf
    const unsigned u = __builtin_bit_cast(unsigned, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // (the rounding add would carry a large NaN payload into the sign: -0)
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two floats -> two bf16 packed little-endian (v_cvt_pk_bf16_f32 on gfx950; the emulator rounds in software: same RNE result)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
#else
    return bf16_bits_rne(a) | (bf16_bits_rne(b) << 16);
#endif
}
__device__ __forceinline__ u32x2_t pack_bf16x4(f32x4 v) {
    u32x2_t r;
    r[0] = pack_bf16x2(v[0], v[1]);
    r[1] = pack_bf16x2(v[2], v[3]);
    return r;
}
__device__ __forceinline__ f32x4 unpack_bf16x4(u32x2_t r) {
    f32x4 v;
    v[0] = __builtin_bit_cast(float, r[0] << 16);
    v[1] = __builtin_bit_cast(float, r[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, r[1] << 16);
    v[3] = __builtin_bit_cast(float, r[1] & 0xffff0000u);
    return v;
}
__device__ __forceinline__ float bf16_round(float x) { return __builtin_bit_cast(float, bf16_bits_rne(x) << 16); }

template <typename ET> __device__ __forceinline__ const ET* et_ptr(const float* p) { return reinterpret_cast<const ET*>(p); }
template <typename ET> __device__ __forceinline__ ET* et_ptr(float* p) { return reinterpret_cast<ET*>(p); }

// 4 consecutive elements <-> f32x4 (16-byte / 8-byte accesses), one element <-> float
#if STGCN_TS_DBG == 1
__device__ __forceinline__ f32x4 ldx4(const float* p) { (void)p; return dbg_opaque4(); }
#else
__device__ __forceinline__ f32x4 ldx4(const float* p) { return ld4(p); }
#endif
__device__ __forceinline__ f32x4 ldx4(const f32x* p) { return ld4(reinterpret_cast<const float*>(p)); }
__device__ __forceinline__ void stx4(f32x* p, f32x4 v) { st4(reinterpret_cast<float*>(p), v); }
__device__ __forceinline__ float ldx1(const f32x* p) { return p->v; }
__device__ __forceinline__ void stx1(f32x* p, float v) { p->v = v; }
#if STGCN_TS_DBG == 1
__device__ __forceinline__ f32x4 ldx4(const bf16* p) { (void)p; return dbg_opaque4(); }
#else
__device__ __forceinline__ f32x4 ldx4(const bf16* p) { return unpack_bf16x4(*reinterpret_cast<const u32x2_t*>(p)); }
#endif
#if STGCN_TS_DBG == 3
__device__ __forceinline__ void stx4(float* p, f32x4 v) { (void)p; dbg_keep4(v); }
#else
__device__ __forceinline__ void stx4(float* p, f32x4 v) { st4(p, v); }
#endif
#if STGCN_TS_DBG == 3
__device__ __forceinline__ void stx4(bf16* p, f32x4 v) { (void)p; dbg_keep4(v); }
#else
__device__ __forceinline__ void stx4(bf16* p, f32x4 v) { *reinterpret_cast<u32x2_t*>(p) = pack_bf16x4(v); }
#endif
// What a 4-element load leaves in registers, and its conversion to fp32 -- SEPARATE steps for values requested ahead of their use: ldx4
// unpacks (bf16) at the load, and an unpack or a mask right behind a load makes the compiler wait for the load there, so a "prefetched"
// ldx4 result is no prefetch at all (r3-33: without its activation loads the bf16 tc1_bwd launch of C3 takes 66 of 105 us).
template <typename ET> struct Raw4 { f32x4 v; };
template <> struct Raw4<bf16> { u32x2_t v; };
template <typename ET> __device__ __forceinline__ Raw4<ET> ldraw4(const ET* p) {
    Raw4<ET> r;
    if constexpr (sizeof(ET) == 2) r.v = *reinterpret_cast<const u32x2_t*>(p);
    else r.v = ld4(reinterpret_cast<const float*>(p));
#if STGCN_TS_DBG == 1
    if constexpr (sizeof(ET) == 2) { const f32x4 z = dbg_opaque4(); r.v[0] = __builtin_bit_cast(unsigned, z[0]); r.v[1] = __builtin_bit_cast(unsigned, z[1]); }
    else r.v = dbg_opaque4();
#endif
    return r;
}
template <typename ET> __device__ __forceinline__ f32x4 cvt4(const Raw4<ET>& r) {
    if constexpr (sizeof(ET) == 2) return unpack_bf16x4(r.v);
    else return r.v;
}
// workgroup barrier that does not wait for outstanding memory operations of the calling wave (the emulator's copies are synchronous)
__device__ __forceinline__ void barrier_only() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// orders LDS accesses between the lanes of ONE wave: a wave's LDS instructions execute in order, so the hardware needs nothing; the compiler
// must not move LDS accesses across this point (and the emulator lets the wave's other lanes catch up here)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename ET> struct Raw1 { float v; };               // one element, raw (see Raw4)
template <> struct Raw1<bf16> { unsigned short v; };
template <typename ET> __device__ __forceinline__ Raw1<ET> ldraw1(const ET* p) {
    Raw1<ET> r;
    if constexpr (sizeof(ET) == 2) r.v = *reinterpret_cast<const unsigned short*>(p);
    else r.v = *reinterpret_cast<const float*>(p);
    return r;
}
template <typename ET> __device__ __forceinline__ float cvt1(const Raw1<ET>& r) {
    if constexpr (sizeof(ET) == 2) return __builtin_bit_cast(float, (unsigned)r.v << 16);
    else return r.v;
}
// 8 consecutive elements (16 B of bf16: one dwordx4 access; 32 B of fp32: two)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ldx8(const float* p, f32x4& a, f32x4& b) { a = ld4(p); b = ld4(p + 4); }
__device__ __forceinline__ void stx8(float* p, f32x4 a, f32x4 b) { st4(p, a); st4(p + 4, b); }
__device__ __forceinline__ void ldx8(const bf16* p, f32x4& a, f32x4& b) {
    const u32x4_t r = *reinterpret_cast<const u32x4_t*>(p);
    u32x2_t lo, hi;
    lo[0] = r[0]; lo[1] = r[1]; hi[0] = r[2]; hi[1] = r[3];
    a = unpack_bf16x4(lo);
    b = unpack_bf16x4(hi);
}
__device__ __forceinline__ void stx8(bf16* p, f32x4 a, f32x4 b) {
    const u32x2_t lo = pack_bf16x4(a), hi = pack_bf16x4(b);
    u32x4_t r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
    *reinterpret_cast<u32x4_t*>(p) = r;
}
__device__ __forceinline__ float ldx1(const float* p) { return *p; }
__device__ __forceinline__ float ldx1(const bf16* p) { return __builtin_bit_cast(float, (unsigned)p->v << 16); }
__device__ __forceinline__ void stx1(float* p, float v) { *p = v; }
__device__ __forceinline__ void stx1(bf16* p, float v) { p->v = (unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu); }   // (v_cvt_pk_bf16_f32 on the device: RNE, NaN-preserving)
// write-through variants (see st4_wt)
#if STGCN_TS_DBG == 3
__device__ __forceinline__ void stx4_wt(float* p, f32x4 v) { (void)p; dbg_keep4(v); }
#else
__device__ __forceinline__ void stx4_wt(float* p, f32x4 v) { st4_wt(p, v); }
#endif
__device__ __forceinline__ void stx4_wt(f32x* p, f32x4 v) { st4_wt(reinterpret_cast<float*>(p), v); }
__device__ __forceinline__ void stx4_wt2(f32x* p, f32x4 v) { st4_wt2(reinterpret_cast<float*>(p), v); }
__device__ __forceinline__ void stx4_wt(bf16* p, f32x4 v) {
#if STGCN_TS_DBG == 3
    (void)p; dbg_keep4(v); return;
#endif
    const u32x2_t r = pack_bf16x4(v);
#if STGCN_WT_STORES && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
#else
    *reinterpret_cast<u32x2_t*>(p) = r;
#endif
}
#if STGCN_TS_DBG == 3
__device__ __forceinline__ void stx4_wt2(float* p, f32x4 v) { (void)p; dbg_keep4(v); }
#else
__device__ __forceinline__ void stx4_wt2(float* p, f32x4 v) { st4_wt2(p, v); }
#endif
__device__ __forceinline__ void stx4_wt2(bf16* p, f32x4 v) {
#if STGCN_TS_DBG == 3
    (void)p; dbg_keep4(v); return;
#endif
    const u32x2_t r = pack_bf16x4(v);
#if STGCN_WT_STORES >= 2 && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
#else
    *reinterpret_cast<u32x2_t*>(p) = r;
#endif
}
// the value a tensor of type ET holds after v was stored to it (identity for float): kernels that go on USING a value they also
// store (row partials of a gradient they write, ..) use the stored value, so that every consumer of the tensor sees one number
template <typename ET> __device__ __forceinline__ float et_round(float v) {
    if constexpr (sizeof(ET) == 2) return bf16_round(v);
    else return v;
}
template <typename ET> __device__ __forceinline__ f32x4 et_round4(f32x4 v) {
    if constexpr (sizeof(ET) == 2) return unpack_bf16x4(pack_bf16x4(v));
    else return v;
}

// ---- dropout encoding of a block output (round 5) --------------------------------------------------------------------------------
// The backward reads a block's dropout mask OFF ITS OUTPUT y instead of regenerating it (~100 VALU instructions of Philox per 4 elements).
// Rounds 4's test "kept iff y != 0" took a kept element whose LayerNorm output is an exact zero (gamma = beta = 0 for it: zero-initialised
// affine parameters) for dropped -- its gamma could then never leave zero.  Now the two zeros are told apart by their SIGN:
//     dropped element          -> stored as -0.0
//     kept element, value == 0 -> stored as +0.0: the keep-scale multiply is fma(v, keep_scale, +0.0), and (-0.0) + (+0.0) = +0.0 in
//                                 round-to-nearest (the same instruction count as the plain multiply; not folded without fast-math)
// Numerically nothing changes (-0.0 == +0.0 everywhere downstream; the reference's own dropped elements are +-0 by the sign of the value the
// mask multiplies), and the mask test is two compares.  (bf16 storage: a kept value below the smallest bf16 subnormal, |v| < 2^-134, could still
// round to -0.0 at the store -- that needs gamma and beta themselves denormal.)
__device__ __forceinline__ float drop_encode(float v, float keep_scale, bool kept) { return kept ? __builtin_fmaf(v, keep_scale, 0.f) : -0.f; }
// dropped iff y is -0.0, i.e. kept iff y != 0 or its sign bit is clear (this form -- a float compare and a signed-integer compare against 0 -- instead
// of one compare against the literal 0x80000000 keeps tc2_bwd_kernel's C2 instance inside its 128 registers: the literal form spilled 3 dwords)
__device__ __forceinline__ bool drop_kept(float y) { return (y != 0.f) | (__builtin_bit_cast(int, y) >= 0); }   // (pass a scalar copy, not a vector element expression)

// ---- one 16-deep matrix-product step on a wave's 16 x 16 accumulator tile ----------------------------------------------------------
// Operand fragments: a lane's 4 consecutive k values (k = 4 * (lane >> 4) + s) of one row (A) / column (B).
//   Mma<float>: frag = f32x4, mma = 4 x v_mfma_f32_16x16x4_f32 (step s contracts component s)
//   Mma<bf16> : frag = 4 bf16 (one VGPR pair), mma = 1 x v_mfma_f32_16x16x16_bf16
template <typename ET> struct Mma;
template <> struct Mma<float> {
    typedef f32x4 frag;
    static __device__ __forceinline__ frag cvt(f32x4 v) { return v; }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = mfma4(a[s], b[s], c);
        return c;
    }
    // two accumulators sharing the A operand, MFMAs interleaved step by step (independent chains)
    static __device__ __forceinline__ void mma_b2(const frag& a, const frag& b0, const frag& b1, f32x4& c0, f32x4& c1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = mfma4(a[s], b0[s], c0);
            c1 = mfma4(a[s], b1[s], c1);
        }
    }
    // two accumulators sharing the B operand
    static __device__ __forceinline__ void mma_a2(const frag& a0, const frag& a1, const frag& b, f32x4& c0, f32x4& c1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = mfma4(a0[s], b[s], c0);
            c1 = mfma4(a1[s], b[s], c1);
        }
    }
    // one product spread over two accumulators (even / odd steps): two dependent chains instead of one
    static __device__ __forceinline__ void mma_split(const frag& a, const frag& b, f32x4& c0, f32x4& c1) {
        c0 = mfma4(a[0], b[0], c0);
        c1 = mfma4(a[1], b[1], c1);
        c0 = mfma4(a[2], b[2], c0);
        c1 = mfma4(a[3], b[3], c1);
    }
    // two independent products, MFMAs interleaved step by step
    static __device__ __forceinline__ void mma_2x(const frag& a0, const frag& b0, f32x4& c0, const frag& a1, const frag& b1, f32x4& c1) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c0 = mfma4(a0[s], b0[s], c0);
            c1 = mfma4(a1[s], b1[s], c1);
        }
    }
    // two products into ONE accumulator, steps interleaved
    static __device__ __forceinline__ void mma_ab2(const frag& a0, const frag& b0, const frag& a1, const frag& b1, f32x4& c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            c = mfma4(a0[s], b0[s], c);
            c = mfma4(a1[s], b1[s], c);
        }
    }
};
template <> struct Mma<bf16> {
    typedef s16x4 frag;
    static __device__ __forceinline__ frag cvt(f32x4 v) { return __builtin_bit_cast(s16x4, pack_bf16x4(v)); }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) {
        #if STGCN_TS_DBG == 2
        asm volatile("" ::"v"(a), "v"(b));
        return c;
#else
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
#endif
    }
    static __device__ __forceinline__ void mma_b2(const frag& a, const frag& b0, const frag& b1, f32x4& c0, f32x4& c1) {
        c0 = mma(a, b0, c0);
        c1 = mma(a, b1, c1);
    }
    static __device__ __forceinline__ void mma_a2(const frag& a0, const frag& a1, const frag& b, f32x4& c0, f32x4& c1) {
        c0 = mma(a0, b, c0);
        c1 = mma(a1, b, c1);
    }
    static __device__ __forceinline__ void mma_split(const frag& a, const frag& b, f32x4& c0, f32x4& c1) { c0 = mma(a, b, c0); }
    static __device__ __forceinline__ void mma_2x(const frag& a0, const frag& b0, f32x4& c0, const frag& a1, const frag& b1, f32x4& c1) {
        c0 = mma(a0, b0, c0);
        c1 = mma(a1, b1, c1);
    }
    static __device__ __forceinline__ void mma_ab2(const frag& a0, const frag& b0, const frag& a1, const frag& b1, f32x4& c) {
        c = mma(a0, b0, c);
        c = mma(a1, b1, c);
    }
};
// acc[i][j] += A_i x B_j over one 16-deep step for a WM x NT block of accumulator tiles (fp32: step-major order, every accumulator
// sees its 4 MFMAs WM * NT issue slots apart)
template <typename ET, int WM, int NT>
__device__ __forceinline__ void mma_tile(f32x4 (&acc)[WM][NT], const f32x4 (&a)[WM], const f32x4 (&b)[NT]) {
    if constexpr (std::is_same<ET, float>::value) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma4(a[i][s], b[j][s], acc[i][j]);
    } else {
        typename Mma<ET>::frag fa[WM], fb[NT];
#pragma unroll
        for (int i = 0; i < WM; ++i) fa[i] = Mma<ET>::cvt(a[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = Mma<ET>::cvt(b[j]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = Mma<ET>::mma(fa[i], fb[j], acc[i][j]);
    }
}
template <> struct Mma<f32x> {
    struct frag { s16x4 h, l; };
    static __device__ __forceinline__ frag cvt(f32x4 v) {
        const u32x2_t hp = pack_bf16x4(v);
        const f32x4 hv = unpack_bf16x4(hp);
        frag f;
        f.h = __builtin_bit_cast(s16x4, hp);
        f.l = __builtin_bit_cast(s16x4, pack_bf16x4(v - hv));
        return f;
    }
    static __device__ __forceinline__ f32x4 mma(const frag& a, const frag& b, f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.l, b.h, c, 0, 0, 0);   // (small terms first)
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b.l, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b.h, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_b2(const frag& a, const frag& b0, const frag& b1, f32x4& c0, f32x4& c1) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.l, b0.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.l, b1.h, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b0.l, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b1.l, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b0.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b1.h, c1, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_a2(const frag& a0, const frag& a1, const frag& b, f32x4& c0, f32x4& c1) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0.l, b.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1.l, b.h, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0.h, b.l, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1.h, b.l, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0.h, b.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1.h, b.h, c1, 0, 0, 0);
    }
    // one product spread over two accumulators: the two small terms in c1, the main term in c0
    static __device__ __forceinline__ void mma_split(const frag& a, const frag& b, f32x4& c0, f32x4& c1) {
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.l, b.h, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.h, b.l, c1, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_2x(const frag& a0, const frag& b0, f32x4& c0, const frag& a1, const frag& b1, f32x4& c1) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0.l, b0.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1.l, b1.h, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0.h, b0.l, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1.h, b1.l, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a0.h, b0.h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a1.h, b1.h, c1, 0, 0, 0);
    }
    static __device__ __forceinline__ void mma_ab2(const frag& a0, const frag& b0, const frag& a1, const frag& b1, f32x4& c) {
        c = mma(a0, b0, c);
        c = mma(a1, b1, c);
    }
};
// four scalars gathered into a fragment (strided LDS reads)
__device__ __forceinline__ f32x4 gather4(const float* p, int stride) {
    f32x4 v;
    v[0] = p[0];
    v[1] = p[stride];
    v[2] = p[2 * stride];
    v[3] = p[3 * stride];
    return v;
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
        du = dh * s * __builtin_fmaf(-th, th, 1.0f);   // (explicit: the function is inlined at several call sites that must round alike)
        dq = dh * th * s * (1.0f - s);
    }
}

// ================================================================================================
// Chained launches: consecutive, DEPENDENT stages of a block run as roles of ONE launch; a consumer workgroup
// starts as soon as the slab it needs is complete instead of after the producer kernel's last workgroup, and its prologue (weights,
// operator fragments) overlaps the producer's tail.  Protocol (MI355X_MICROARCH.md "inter-workgroup visibility", recipe R1):
//   * a workgroup's role and item come from an atomic TICKET, not from blockIdx: tickets are handed out in start order and the stages
//     are numbered producer first, so a workgroup only ever waits for workgroups that are already running -- no assumption about
//     dispatch order or residency (a consumer that holds a compute unit can never starve the producer it waits for);
//   * the producer stores the hand-off tensor WRITE-THROUGH (16-byte sc1 stores), every storing wave drains its stores
//     (s_waitcnt vmcnt(0)), a workgroup barrier (or a single storing wave) orders them before ONE lane bumps the slab's arrival counter
//     (relaxed, agent scope);
//   * the consumer's lane 0 polls that ONE word (relaxed, s_sleep between polls, bounded: a give-up sets a sticky error word instead of
//     hanging the device), a workgroup barrier follows, and the hand-off tensor is then read with sc1 loads (never served from this
//     CU's L1, which another CU's stores do not refresh);
//   * the LAST workgroup to finish zeroes ticket and counters (kernels of one stream do not overlap, so the next launch finds them
//     clean -- also under hipGraph replay, where no host code runs between launches).
// The CPU emulator runs the workgroups of a launch one after another in ticket order: every counter is complete when it is read.
// ================================================================================================
constexpr int kChainHdr = 4;   // words: [0] ticket, [1] finished workgroups, [2] sticky error (1 + counter index of a wait that gave up), [3] spare
struct ChainCtl {
    unsigned* words;   // device words of this launch: header, then `ncount` arrival counters (null: the stage runs as its own launch)
    int ncount;
    unsigned total;    // workgroups of the launch
    long long spin;    // bound of one wait in ticks of the 100 MHz wall clock (launchers: g_chain_spin_ticks, stgcn_set_chain_spin_ticks)
};
constexpr long long kChainSpinTicks = 200000000ll;   // default bound: 2 s of the 100 MHz wall clock
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) unsigned chain_gu32;
__device__ __forceinline__ unsigned chain_ld(const unsigned* p) { return __hip_atomic_load((const chain_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_st(unsigned* p, unsigned v) { __hip_atomic_store((chain_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned chain_add(unsigned* p, unsigned v) { return __hip_atomic_fetch_add((chain_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every wave that stored hand-off data calls this before the barrier / the counter bump (inline asm: the compiler cannot drop it)
__device__ __forceinline__ void chain_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ unsigned chain_ld(const unsigned* p) { return *p; }
__device__ __forceinline__ void chain_st(unsigned* p, unsigned v) { *p = v; }
__device__ __forceinline__ unsigned chain_add(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
__device__ __forceinline__ void chain_drain_stores() {}
#endif
// virtual block index of this workgroup (all threads call; `slot` = one LDS word the launcher reserved behind the roles' own LDS)
__device__ __forceinline__ int chain_enter(const ChainCtl& c, unsigned* slot) {
    if (threadIdx.x == 0) *slot = chain_add(c.words, 1u);
    __syncthreads();
    return __builtin_amdgcn_readfirstlane((int)*slot);
}
// one lane, after the hand-off stores of the whole workgroup are drained and ordered before this call
__device__ __forceinline__ void chain_publish(const ChainCtl& c, int idx, unsigned n = 1u) { chain_add(c.words + kChainHdr + idx, n); }
// thread `leader` waits until counter idx has reached `expected` (all threads of the role call: a workgroup barrier follows)
__device__ __forceinline__ void chain_wait(const ChainCtl& c, int idx, unsigned expected) {
    if (threadIdx.x == 0) {
        const unsigned* p = c.words + kChainHdr + idx;
#if defined(__HIP_DEVICE_COMPILE__)
        if (chain_ld(p) < expected) {
            const long long t0 = wall_clock64();
            while (chain_ld(p) < expected) {
                __builtin_amdgcn_s_sleep(24);   // ~0.7 us between polls: hundreds of waiting workgroups share the counters' memory channels with the producers' bumps (tools/ubench/chain_probe.hip)
                if (wall_clock64() - t0 > c.spin) {   // never hang the device: flag it and go on (the results of this launch are void)
                    chain_st(c.words + 2, 1u + (unsigned)idx);
                    break;
                }
            }
        }
#elif !defined(__HIPCC__)
        if (chain_ld(p) < expected) {   // (emulator: producers ran to completion before this workgroup started -- a short count is a protocol bug)
            fprintf(stderr, "emu: chained launch: counter %d holds %u, expected %u (missing publish?)\n", idx, chain_ld(p), expected);
            abort();
        }
#endif
    }
    __syncthreads();
}
// end of a role body (every wave that is still alive calls): the last workgroup of the launch to get here re-arms the control words
__device__ __forceinline__ void chain_exit(const ChainCtl& c) {
    __syncthreads();   // no wave of this workgroup publishes after the count below
    if (threadIdx.x == 0) {
        if (chain_add(c.words + 1, 1u) == c.total - 1) {
            for (int i = 0; i < c.ncount; ++i) chain_st(c.words + kChainHdr + i, 0u);
            chain_st(c.words, 0u);
            chain_st(c.words + 1, 0u);
        }
    }
}
// ---- symmetric exchange between the workgroups of ONE launch: every workgroup publishes, then waits for its PEERS (the row statistics of
// a LayerNorm slab that several row tiles share).  The launcher guarantees that the whole grid is resident at once (grid <= occupancy x
// compute units, nothing else on the stream), so every peer is running or about to; the wait is bounded like chain_wait.  The emulator
// abandons a workgroup whose peers have not run yet and runs it again after the rest of the grid (tests/emu: peer_defer), without
// repeating its counter bumps.
__device__ __forceinline__ void chain_publish_peer(const ChainCtl& c, int idx, unsigned n = 1u) {
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(__HIPCC__)
    if (emu::peer_publish_done()) return;
#endif
    chain_add(c.words + kChainHdr + idx, n);
}
// TEST setting (stgcn_set_chain_spin_ticks < 0): the launch's first item withholds its arrival, so that its peers' waits -- bounded by
// |spin| ticks -- run out for certain and the give-up path (sticky word, poisoned results) can be tested on the device and in the emulator
__device__ __forceinline__ bool chain_withhold(const ChainCtl& c, long item) { return c.spin < 0 && item == 0; }
// item of this workgroup by start order (as chain_enter; the emulator hands a re-run workgroup the ticket its abandoned run drew)
__device__ __forceinline__ int chain_enter_peer(const ChainCtl& c, unsigned* slot) {
    if (threadIdx.x == 0) {
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(__HIPCC__)
        *slot = emu::peer_ticket(c.words);
#else
        *slot = chain_add(c.words, 1u);
#endif
    }
    __syncthreads();
    return __builtin_amdgcn_readfirstlane((int)*slot);
}
// ONE thread polls (no barrier here: the caller follows its last poll with __syncthreads()).  Returns false if the wait gave up (sticky error
// word set): the caller must make that visible in what it computes -- a result built on missing peer data is garbage, and a NaN is louder.
__device__ __forceinline__ bool chain_poll_peer(const ChainCtl& c, int idx, unsigned expected) {
    const unsigned* p = c.words + kChainHdr + idx;
#if defined(__HIP_DEVICE_COMPILE__)
    if (chain_ld(p) < expected) {
        const long long t0 = wall_clock64();
        while (chain_ld(p) < expected) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > (c.spin < 0 ? -c.spin : c.spin)) {
                chain_st(c.words + 2, 1u + (unsigned)idx);
                return false;
            }
        }
    }
#elif !defined(__HIPCC__)
    if (chain_ld(p) < expected) {
        if (c.spin < 0) {   // (test setting, see chain_withhold: a wait that no re-run of the grid completes runs out like the device's)
            if (emu::g.peer_give_up) {
                chain_st(c.words + 2, 1u + (unsigned)idx);
                return false;
            }
            emu::g.peer_may_give_up = true;
        }
        emu::peer_defer();
    }
#else
    (void)p; (void)expected;
#endif
    return true;
}
// ---- one-word exchange between the P workgroups that share a LayerNorm slab (tc2_ln_fwd_kernel, PP > 1): a peer's (mean, M2) travels as
// ONE 64-bit word whose top bit (the sign of M2 >= 0, never set by a value) says "written" -- no counter, no drain, no second round trip.
// The words are zero when the launch starts (pack launch / the launch's last workgroup, chain_exit_slots).
__device__ __forceinline__ unsigned long long peer_word(float mean, float M2) {
    return ((unsigned long long)(__builtin_bit_cast(unsigned, M2) | 0x80000000u) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, mean);
}
__device__ __forceinline__ float peer_word_mean(unsigned long long w) { return __builtin_bit_cast(float, (unsigned)w); }
__device__ __forceinline__ float peer_word_m2(unsigned long long w) { return __builtin_bit_cast(float, (unsigned)(w >> 32) & 0x7fffffffu); }
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) unsigned long long chain_gu64;
__device__ __forceinline__ unsigned long long chain_ld64(const unsigned long long* p) { return __hip_atomic_load((const chain_gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store((chain_gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#else
__device__ __forceinline__ unsigned long long chain_ld64(const unsigned long long* p) { return *p; }
__device__ __forceinline__ void chain_st64(unsigned long long* p, unsigned long long v) { *p = v; }
#endif
// ONE thread waits for a peer's word (bounded like chain_poll_peer; a give-up sets the sticky word and returns 0: the caller poisons its result)
__device__ __forceinline__ unsigned long long chain_poll_word(const ChainCtl& c, const unsigned long long* p, int idx) {
    unsigned long long w = chain_ld64(p);
#if defined(__HIP_DEVICE_COMPILE__)
    if (!(w >> 63)) {
        const long long t0 = wall_clock64();
        while (!((w = chain_ld64(p)) >> 63)) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > (c.spin < 0 ? -c.spin : c.spin)) {
                chain_st(c.words + 2, 1u + (unsigned)idx);
                return 0ull;
            }
        }
    }
#elif !defined(__HIPCC__)
    if (!(w >> 63)) {
        if (c.spin < 0) {   // (test setting: a wait that no re-run of the grid completes runs out like the device's)
            if (emu::g.peer_give_up) {
                chain_st(c.words + 2, 1u + (unsigned)idx);
                return 0ull;
            }
            emu::g.peer_may_give_up = true;
        }
        emu::peer_defer();
    }
#endif
    return w;
}
// one (x, y) pair of a hand-off array, written through / read past this CU's L1 (base wave-uniform, idx = element index)
__device__ __forceinline__ void st2_wt(float2* p, float2 v) {
#if STGCN_WT_STORES && defined(__HIP_DEVICE_COMPILE__)
    const u32x2_t r = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
#else
    *p = v;
#endif
}
__device__ __forceinline__ float2 ld2_sc1(const float2* base, long nelem, int idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(nelem * 8), 0x00020000);
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, idx * 8, 0, 16);
    const unsigned vx = v[0], vy = v[1];   // (scalar copies: __builtin_bit_cast of an ext-vector ELEMENT reads element 0 -- clang, host and device)
    return make_float2(__builtin_bit_cast(float, vx), __builtin_bit_cast(float, vy));
#else
    (void)nelem;
    return base[idx];
#endif
}
// 4 consecutive elements of a hand-off tensor, never from this CU's L1 (sc1 buffer load; `base` must be wave-uniform: kernel arguments and
// the virtual block index only).  off = element offset from base.  LIMIT: the descriptor's size and the byte offset are 32-bit --
// the window [base, base + nelem) must stay below 2 GiB; callers rebase per slab (the head's row partials: N elements; the experimental
// chained forward: one slab / KT slabs), a whole-tensor window on a big B * T * N would clamp silently.
template <typename ET> __device__ __forceinline__ Raw4<ET> ldraw4_sc1(const ET* base, long nelem, int off) {
    Raw4<ET> r;
#if defined(__HIP_DEVICE_COMPILE__)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(nelem * (long)sizeof(ET)), 0x00020000);
    if constexpr (sizeof(ET) == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off * 2, 0, 16);
        r.v[0] = v[0]; r.v[1] = v[1];
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off * 4, 0, 16);
        r.v = __builtin_bit_cast(f32x4, v);
    }
#else
    (void)nelem;
    r = ldraw4(base + off);
#endif
    return r;
}

// ---- "bf16x6": fp32-accurate matrix products on the bf16 matrix pipe (round 6) -----------------------------------------------------------
// An fp32 value splits EXACTLY into three bf16 numbers, x = h + m + l (8 + 8 + 8 significand bits; each residual is exact in fp32).  A product
// of two such values is 9 terms; the three smallest (m l, l m, l l: below 2^-32 of the product) are dropped, the other six are formed EXACTLY
// by bf16 multiplies and accumulated in fp32 by the matrix pipe: the same 2^-24-per-product accuracy class as v_mfma_f32_16x16x4_f32, at
// 3 x 17.5 cycles per 16-deep step instead of 4 x 32 -- and a bf16 MFMA does not exclude the SIMD's VALU (profiles/r6-01_valu_mfma_overlap.txt).
// Two terms share one 32-deep instruction: slots 0..3 / 4..7 of a lane's operand hold a lane's 4 k values of two PLANES,
//     (Ah|Al)(Bl|Bh) = Ah Bl + Al Bh ,  (Ah|Am)(Bm|Bm) = Ah Bm + Am Bm ,  (Ah|Am)(Bh|Bh) = Ah Bh + Am Bh      (small terms first).
// Round 4's "bf16x6" split every streamed fragment at its CONSUMER and lost to the split's VALU time; here operands are split ONCE where
// they are produced (stationary weights at kernel start, activation tiles when they are staged into LDS) and read back as planes.
struct Frag3 { s16x4 h, m, l; };
__device__ __forceinline__ Frag3 split3(f32x4 v) {
    Frag3 f;
    const u32x2_t hp = pack_bf16x4(v);
    const f32x4 r1 = v - unpack_bf16x4(hp);
    const u32x2_t mp = pack_bf16x4(r1);
    const f32x4 r2 = r1 - unpack_bf16x4(mp);
    f.h = __builtin_bit_cast(s16x4, hp);
    f.m = __builtin_bit_cast(s16x4, mp);
    f.l = __builtin_bit_cast(s16x4, pack_bf16x4(r2));
    return f;
}
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 cat8(s16x4 a, s16x4 b) {
    const s16x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ f32x4 mma3(const Frag3& a, const Frag3& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a.h, a.l), cat8(b.l, b.h), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a.h, a.m), cat8(b.m, b.m), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a.h, a.m), cat8(b.h, b.h), c, 0, 0, 0);
}
// two A operands against one B (P and Q half of a gated conv): the B-side operands are formed once
__device__ __forceinline__ void mma3_a2(const Frag3& a0, const Frag3& a1, const Frag3& b, f32x4& c0, f32x4& c1) {
    const bf16x8 blh = cat8(b.l, b.h), bmm = cat8(b.m, b.m), bhh = cat8(b.h, b.h);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a0.h, a0.l), blh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a1.h, a1.l), blh, c1, 0, 0, 0);
    const bf16x8 a0hm = cat8(a0.h, a0.m), a1hm = cat8(a1.h, a1.m);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0hm, bmm, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1hm, bmm, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0hm, bhh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1hm, bhh, c1, 0, 0, 0);
}
// the same triple as one 16-byte (h | m) group + an 8-byte l group (two LDS planes): a consumer gets (h | m) -- one 32-deep operand as it stands --
// with ONE ds_read_b128, and the compiler cannot fuse the reads of neighbouring planes into ds_read2_b64 (half the LDS rate, 32 banks)
__device__ __forceinline__ void st_frag3_hml(short* phm, short* pl, const Frag3& f) {
    *reinterpret_cast<s16x8*>(phm) = s16x8{f.h[0], f.h[1], f.h[2], f.h[3], f.m[0], f.m[1], f.m[2], f.m[3]};
    *reinterpret_cast<s16x4*>(pl) = f.l;
}
#if defined(__HIP_DEVICE_COMPILE__)
#define STGCN_ON_DEVICE 1   // (constant of `if constexpr` choices between a hand-scheduled device form and the portable form the host emulator runs)
#else
#define STGCN_ON_DEVICE 0
#endif
#ifndef STGCN_MW_ASM
#define STGCN_MW_ASM 1   // build-time A/B switch: 0 = tc1_bwd's weight-gradient waves in the compiler-scheduled form (used to bisect pass r6-53, profiles/r6_experiments.md)
#endif
// A[m][k = 4 consecutive rows] fragment of a ROW-MAJOR bf16 tile in LDS (element (row, col) at p0[row * ld + col]): lane (l15, g) receives column
// col0 + l15 of rows 4g .. 4g + 3.  On the device one ds_read_b64_tr_b16: every lane supplies the address of 4 contiguous elements -- lane i of a
// 16-lane group: row 4g + (i >> 2), columns col0 + 4 (i & 3) .. + 3 -- and receives element (c & 3) of the lanes 4j + (c >> 2), j = 0 .. 3
// (tools/ubench/tr_read.hip).  Conflict-free when ld is an odd multiple of 8 dwords (8 rows x 8 dwords = the 64 banks).
__device__ __forceinline__ s16x4 ld_tr4(const short* p0, int ld, int g, int l15) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + (4 * g + (l15 >> 2)) * ld + 4 * (l15 & 3)));
#else
    return s16x4{p0[(4 * g) * ld + l15], p0[(4 * g + 1) * ld + l15], p0[(4 * g + 2) * ld + l15], p0[(4 * g + 3) * ld + l15]};
#endif
}
#if defined(__HIP_DEVICE_COMPILE__)
// LDS byte address of a pointer into shared memory (the operand of a hand-written ds_read)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
#endif
// one plane triple of 4 consecutive k values in LDS: planes `pstride` shorts apart, 8-byte accesses
__device__ __forceinline__ void st_frag3(short* p, int pstride, const Frag3& f) {
    *reinterpret_cast<s16x4*>(p) = f.h;
    *reinterpret_cast<s16x4*>(p + pstride) = f.m;
    *reinterpret_cast<s16x4*>(p + 2 * pstride) = f.l;
}
__device__ __forceinline__ Frag3 ld_frag3(const short* p, int pstride) {
    Frag3 f;
    f.h = *reinterpret_cast<const s16x4*>(p);
    f.m = *reinterpret_cast<const s16x4*>(p + pstride);
    f.l = *reinterpret_cast<const s16x4*>(p + 2 * pstride);
    return f;
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
