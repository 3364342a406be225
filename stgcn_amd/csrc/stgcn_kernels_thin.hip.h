// The THIN first layer of an ST block (K = Kt * c_in <= 4 taps, c0 = 64, c1 = 16: the 1-channel input of STGCN's first block), round 5.
//
// hazdzz/STGCN model/layers.py:87-120 (TemporalConvLayer, c_in = 1: Align zero-pads the residual), :222-231 (GraphConvLayer's Align 64 -> 16).
//
// A 1-channel, 3-tap convolution has 4.6 MB of algorithmic traffic at the C2 size and next to no arithmetic; rounds 1-4 ran it through the
// general row-tile kernels (tconv_fwd_kernel: K padded to 16, five workgroup barriers per 64-row tile; thin_tc1_bwd_kernel: four), and the
// two launches took 13.9 + 19.7 us of the 373 us step -- 7 % and 8 % of either roofline: pure barrier-step latency chains.
// Here ONE WAVE owns a 16-row tile and nothing it does needs another wave:
//   * the conv itself is ONE v_mfma_f32_16x16x4_f32 per 16 output channels: Z^T[o][row] = W^T[o][k] x^T[k][row] with k = the (up to) 4 taps
//     -- A = one VGPR per o-tile, stationary; B = the lane's own tap of its row; the bias rides in as the C operand;
//   * D leaves lane (g, l15) with channels 16 mt + 4 g + r of row l15, P and Q of a channel in the same lane: sigmoid, gate (and in backward
//     the gate backward) run on registers, and the lane's h values are already the B operand of the Align product A^T = Wa^T h^T, whose D is
//     4 consecutive output channels of one row = one 16-byte store.  Forward: no LDS at all, no barrier;
//   * backward: dH^T = Wa dA^T the same way; the weight gradients contract over ROWS, i.e. need the D-layout values transposed -- through a
//     wave-private LDS tile (wave_lds_sync only); dW_eff, db_eff (as the "tap" column of ones), dWa accumulate in MFMA accumulators over all
//     tiles of the wave; the four waves of a workgroup are combined once, at the end (a fixed-order tree through LDS: the kernel's only workgroup barriers).
// Every global load of a tile is requested one tile ahead (raw, clamped, unconditional).
#pragma once
#include <type_traits>
#include "stgcn_device.hip.h"
#include "stgcn_kernels_fwd.hip.h"

namespace stgcn {

constexpr int kThinTaps = 4;   // K = Kt * c_in <= 4

// the (b, t, n) decomposition of a flat output row and the source row of its tap 0 (in units of source ROWS of C elements)
__device__ __forceinline__ size_t thin_row_base(const TapSrc& ts, long R, unsigned per_b, size_t xbs) {
    const unsigned Ru = (unsigned)R, b = Ru / per_b, rem = Ru - b * per_b;
    return (size_t)b * xbs + rem;
}

// W_eff[k][o] of the thin layer from the reference's own parameter (what pack_kernel's PK_TCONV_DENSE writes, tconv_weff with C < 64: the
// zero-padding Align adds the input channel to output channel o = ch of the P half at the last tap); k >= Kt * C: 0
__device__ __forceinline__ float thin_weff(const float* cw, int C, int Kt, int k, int o) {
    if (k >= Kt * C) return 0.f;
    const int tap = k / C, ch = k - tap * C;
    float v = cw[((size_t)o * C + ch) * Kt + tap];
    if (tap == Kt - 1 && o == ch) v += 1.0f;   // (o < 64 always holds for o == ch < C <= 4)
    return v;
}

// conv operands of the 8 output-channel tiles (P: 0..3, Q: 4..7) and the product itself, by storage type:
//   float / f32x storage: exact fp32, one 4-deep MFMA per tile (lane group g holds tap g);  bf16: one 16-deep bf16 MFMA (group 0 holds all taps)
template <typename ET> struct ThinConv {
    float w[8];
    __device__ __forceinline__ void load(const float* Wd, int K, int g, int l15) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) w[mt] = g < K ? Wd[(size_t)g * 128 + 16 * mt + l15] : 0.f;
    }
    // the same operands straight from the reference's conv weight (128, C, Kt, 1): W_eff[k = tap * C + ch][o] (thin_weff)
    __device__ __forceinline__ void load_native(const float* cw, int C, int Kt, int g, int l15) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) w[mt] = thin_weff(cw, C, Kt, g, 16 * mt + l15);
    }
    // xk: the K taps of this lane's row (zeros beyond K)
    __device__ __forceinline__ void run(const f32x4& xk, int g, f32x4 (&acc)[8]) const {
        const float xb = g == 0 ? xk[0] : g == 1 ? xk[1] : g == 2 ? xk[2] : xk[3];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt] = mfma4(w[mt], xb, acc[mt]);
    }
};
template <> struct ThinConv<bf16> {
    Mma<bf16>::frag w[8];
    __device__ __forceinline__ void load(const float* Wd, int K, int g, int l15) {
        (void)K;   // (rows K .. 15 of the dense pack are zeros)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = Wd[(size_t)(4 * g + s) * 128 + 16 * mt + l15];
            w[mt] = Mma<bf16>::cvt(v);
        }
    }
    __device__ __forceinline__ void load_native(const float* cw, int C, int Kt, int g, int l15) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = thin_weff(cw, C, Kt, 4 * g + s, 16 * mt + l15);
            w[mt] = Mma<bf16>::cvt(v);
        }
    }
    __device__ __forceinline__ void run(const f32x4& xk, int g, f32x4 (&acc)[8]) const {
        const Mma<bf16>::frag xb = Mma<bf16>::cvt(g == 0 ? xk : zero4());
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt] = Mma<bf16>::mma(w[mt], xb, acc[mt]);
    }
};

// ================================================================================================
// Forward: A[row][0..15] = Align(GLU / GTU(tmp_conv1(x)))  -- the gate inputs are not stored (the backward recomputes them, recompute_tc1)
// ================================================================================================
struct ThinFwdArgs {
    TapSrc ts;           // x through Kt taps, dir = +1 (strided / indexed windows allowed)
    const float* Wd;     // dense W_eff [16][128] (PK_TCONV_DENSE; rows >= K are zeros)
    const float* bias;   // b_eff [128]
    const float* Wap;    // PK_ALIGN_FWD fragments: K = 64 (4 chunks), 16 columns
    const float* ba;     // [16]
    float* A;            // [rows][16]
    // native = 1 (round 6, the launch fused with the model's weight pack: pack_thin_fwd_kernel): the operands come straight from the
    // reference's parameters -- nothing this kernel reads is written by the pack role of the same launch
    int native;
    const float* cw;     // tmp_conv1.causal_conv.weight (128, C, Kt, 1)
    const float* cb;     // tmp_conv1.causal_conv.bias (128) or null
    const float* aw;     // graph_conv.align.align_conv.weight (16, 64, 1, 1)
    const float* ab;     // graph_conv.align.align_conv.bias (16) or null
};
// pack role and first-layer role of ONE launch share the window index of a captured step (TapSrc.idx_dev = one of the pack's step counters):
// every first-layer wave reads the OLD index, forms old + inc itself and reports in; the pack role's counter thread bumps the word in place
// once `expected` waves have reported in g_pack_readers (and re-arms the counts).  on == 0: not fused / no shared counter.
struct PackSync {
    int on;
    const long* idx_ptr;
    unsigned expected;
    long inc, mod;
};

// vb / nb: index and count of the workgroups of this role (a launch of its own: blockIdx.x / gridDim.x)
template <typename ET, int ACT>
__device__ __forceinline__ void thin_tc1_fwd_body(const ThinFwdArgs& a, const int vb, const int nb, const PackSync& sy) {
    typedef Mma<ET> MM;
    const int lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    const long wave_id = (long)vb * 4 + (threadIdx.x >> 6), nwaves = (long)nb * 4;
    const long rows = a.ts.rows, tiles = (rows + 15) >> 4;
    const int K = a.ts.taps * a.ts.C, C = a.ts.C, N = a.ts.N;
    const unsigned per_b = (unsigned)(a.ts.Tdst * N);
    const size_t xbs = (size_t)tap_bstride(a.ts);
    ET* const A_ = et_ptr<ET>(a.A);

    ThinConv<ET> cw;
    f32x4 bz[8];          // bias of the lane's channels 16 mt + 4 g + r, P then Q: the C operand of the conv
    typename MM::frag waf[4];   // Align: A[m = j = l15][k = c = 16 kc + 4 g + s] = Wa[c][j]
    f32x4 ba4;
    if (a.native) {       // (uniform) what pack_kernel writes for this layer, read off the parameters themselves
        cw.load_native(a.cw, C, a.ts.taps, g, l15);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) bz[mt] = a.cb ? ld4(a.cb + 16 * mt + 4 * g) : zero4();
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) waf[kc] = MM::cvt(ld4(a.aw + (size_t)l15 * 64 + 16 * kc + 4 * g));
        ba4 = a.ab ? ld4(a.ab + 4 * g) : zero4();
    } else {
        cw.load(a.Wd, K, g, l15);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) bz[mt] = ld4(a.bias + 16 * mt + 4 * g);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) waf[kc] = MM::cvt(ld4(a.Wap + ((size_t)kc * 64 + lane) * 4));
        ba4 = ld4(a.ba + 4 * g);
    }

    // (the operand loads above are in flight while the index arrives: one wait covers both)
    const bool shared_idx = sy.on && a.ts.idx_dev == sy.idx_ptr;   // (uniform)
    const ET* xsrc = shared_idx ? et_ptr<ET>(a.ts.src) : tap_base<ET>(a.ts);
    if (shared_idx) {   // the window index is bumped by the pack role of THIS launch: see PackSync
        const long old = (long)chain_ld64(reinterpret_cast<const unsigned long long*>(a.ts.idx_dev));
        chain_drain_stores();                       // (s_waitcnt vmcnt(0): the index has arrived before this wave reports in)
        if (lane == 0) chain_add(g_pack_readers + (int)(wave_id & (kPackSyncWords - 1)) * kPackSyncStride, 1u);
        const long v = old + sy.inc;
        xsrc = et_ptr<ET>(a.ts.src) + (sy.mod > 0 ? v % sy.mod : v) * a.ts.idx_stride;
    }
    // the K taps of row (tile, l15): scalar loads, raw, requested one tile ahead
    auto request = [&](long t, Raw1<ET> (&xr)[kThinTaps]) __attribute__((always_inline)) {
        const long R0 = (t < tiles ? t : tiles - 1) * 16 + l15, R = R0 < rows ? R0 : rows - 1;
        const size_t base = thin_row_base(a.ts, R, per_b, xbs);
#pragma unroll
        for (int k = 0; k < kThinTaps; ++k) {
            const int kk = k < K ? k : K - 1, tap = kk / C, ch = kk - tap * C;
            xr[k] = ldraw1(xsrc + (base + (size_t)tap * N) * C + ch);
        }
    };
    Raw1<ET> xn[kThinTaps];
    request(wave_id, xn);
    for (long t = wave_id; t < tiles; t += nwaves) {
        f32x4 xk;
#pragma unroll
        for (int k = 0; k < kThinTaps; ++k) xk[k] = k < K ? cvt1(xn[k]) : 0.f;
        request(t + nwaves, xn);   // (past the last tile: a valid address, never used)
        f32x4 acc[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt] = bz[mt];
        cw.run(xk, g, acc);
        f32x4 out = zero4();
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            f32x4 h;
#pragma unroll
            for (int s = 0; s < 4; ++s) h[s] = gate_fwd(acc[kc][s], sigmoid_f(acc[4 + kc][s]), ACT);   // (compile-time: no branch per element)
            out = MM::mma(waf[kc], MM::cvt(h), out);   // B[k = c = 16 kc + 4 g + s][n = row = l15]
        }
        const long R = t * 16 + l15;
        if (R < rows) stx4_wt(A_ + (size_t)R * 16 + 4 * g, out + ba4);   // D[m = j = 4 g + r][n = row]: 16 contiguous bytes
    }
}

template <typename ET, int ACT>
__global__ __launch_bounds__(256) void thin_tc1_fwd_kernel(ThinFwdArgs a) {
    thin_tc1_fwd_body<ET, ACT>(a, (int)blockIdx.x, (int)gridDim.x, PackSync{0, nullptr, 0u, 0, 0});
}
// The model's weight pack and the thin first layer of its first block as ONE launch (round 6): the two are independent once the layer
// reads its operands off the parameters (ThinFwdArgs.native), both are latency chains on a mostly idle device (6.4 + 8.0 us at C2), and a
// kernel boundary of their own costs another 1.8 us.  Workgroups [0, n_thin) run the layer, [n_thin, ..) the pack jobs (the layer first:
// the pack role's counter thread waits for the layer's waves to have read the window index, PackSync -- and the CPU emulator, which runs
// workgroups in index order, then finds them done).
template <typename ET, int ACT>
__global__ __launch_bounds__(256) void pack_thin_fwd_kernel(PackArgs p, ThinFwdArgs t, int n_thin, PackSync sy) {
#ifdef STGCN_DIAG_FUSE   // (timing experiments only: 1 = the pack role returns at once, 2 = the layer role does)
    if (STGCN_DIAG_FUSE == 1 && (int)blockIdx.x >= n_thin) return;
    if (STGCN_DIAG_FUSE == 2 && (int)blockIdx.x < n_thin) return;
#endif
    if ((int)blockIdx.x < n_thin) thin_tc1_fwd_body<ET, ACT>(t, (int)blockIdx.x, n_thin, sy);   // (uniform per workgroup)
    else pack_body(p, (int)blockIdx.x - n_thin, sy);
}

// ================================================================================================
// Backward: dA -> dH = dA Wa^T -> gate backward on the recomputed gate inputs -> dZ (to memory only if an input gradient is needed),
// per-workgroup partials  dWa [64][16] | dba [16] | dW_eff [16 rows][128] | db_eff [128]   (the layout of thin_tc1_bwd_kernel)
// ================================================================================================
struct ThinBwdArgs;   // (stgcn_kernels_bwd.hip.h: shared with the row-tile kernel this one replaces)

constexpr int kThinLdT = 68, kThinLdD = 20, kThinLdX = 16;
// floats per wave: transposition tile | dA tile | tap tile | bias of the 128 conv outputs | the 4 Align fragments (64 lanes x 4 floats each)
constexpr int kThinWaveLds = 16 * kThinLdT + 16 * kThinLdD + 16 * kThinLdX + 128 + 4 * 256;
inline size_t thin_bwd2_lds_bytes() { return (size_t)4 * kThinWaveLds * sizeof(float); }   // (>= the 2 x 12 x 64 float4 of the final combine)
static_assert(4 * kThinWaveLds >= 2 * 12 * 64 * 4 + 2 * 16, "the end-of-kernel combine reuses the waves' tiles");

// Register budget (pass r5-01: the first version held every per-tile value at once -- 186 VGPRs + 80 AGPRs, ONE wave per SIMD, so the 2048
// waves of the launch ran in two rounds and it was slower than the kernel it replaced): per-tile constants that are touched once per tile (the
// conv bias, the Align fragments) live in the wave's LDS, the gate backward runs channel tile by channel tile and its h values go straight to
// the transposition tile; what stays in registers across tiles is the 48 accumulator registers and the 8 conv weights.
template <typename ET, int ACT, typename ARGS>
__device__ __forceinline__ void thin_tc1_bwd2_body(const ARGS& a) {
    typedef Mma<ET> MM;
    constexpr bool kF32 = sizeof(ET) == 4;   // float / f32x storage
    typedef typename std::conditional<kF32, float, bf16>::type ST;   // storage type of x / dA / dZ
    extern __shared__ float stgcn_smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    float* const Tt = stgcn_smem + wave * kThinWaveLds;   // [16 rows][68]  values in D layout written row major, read back transposed
    float* const Dt = Tt + 16 * kThinLdT;                 // [16 rows][20]  dA tile
    float* const Xt = Dt + 16 * kThinLdD;                 // [16 rows][16]  taps 0 .. K-1, a column of ones (-> db_eff), zeros
    float* const Bs = Xt + 16 * kThinLdX;                 // [128]          b_eff
    float* const Ws = Bs + 128;                           // [4][64 lanes][4]  PK_ALIGN_BWD fragments
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    const long rows = a.rows, tiles = (rows + 15) >> 4;
    const int K = a.ts.taps * a.ts.C, C = a.ts.C, N = a.ts.N;
    const unsigned per_b = (unsigned)(a.ts.Tdst * N);
    const size_t xbs = (size_t)tap_bstride(a.ts);
    const ST* const xsrc = tap_base<ST>(a.ts);
    const ST* const dA_ = et_ptr<ST>(a.dA);
    ST* const dZ_ = et_ptr<ST>(a.dZ);

    ThinConv<ST> cw;
    cw.load(a.Wd, K, g, l15);
    {   // wave-private constants -> LDS
        const f32x4 b0 = ld4(a.bias + 4 * (lane & 31));
        f32x4 wf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) wf[mt] = ld4(a.WaT + ((size_t)mt * 64 + lane) * 4);   // dH^T = Wa dA^T: A[m = c = 16 mt + l15][k = j = 4 g + s] = Wa[c][j]
        if (lane < 32) st4(Bs + 4 * lane, b0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) st4(Ws + (mt * 64 + lane) * 4, wf[mt]);
        for (int i = lane; i < 16 * kThinLdX; i += 64) Xt[i] = 0.f;   // (columns K + 1 .. 15 stay zero)
    }
    wave_lds_sync();

    f32x4 accW[8], accA[4], dba = zero4();   // dW_eff^T[o][k] (column K: db_eff), dWa[c][j], dba (lane-local over rows)
#pragma unroll
    for (int i = 0; i < 8; ++i) accW[i] = zero4();
#pragma unroll
    for (int i = 0; i < 4; ++i) accA[i] = zero4();

    struct Req { Raw4<ST> da; Raw1<ST> x[kThinTaps]; };
    auto request = [&](long t, Req& q) __attribute__((always_inline)) {
        const long R0 = (t < tiles ? t : tiles - 1) * 16 + l15, R = R0 < rows ? R0 : rows - 1;
        q.da = ldraw4(dA_ + (size_t)R * 16 + 4 * g);
        const size_t base = thin_row_base(a.ts, R, per_b, xbs);
#pragma unroll
        for (int k = 0; k < kThinTaps; ++k) {
            const int kk = k < K ? k : K - 1, tap = kk / C, ch = kk - tap * C;
            q.x[k] = ldraw1(xsrc + (base + (size_t)tap * N) * C + ch);
        }
    };
    Req rq;
    request(wave_id, rq);
    for (long t = wave_id; t < tiles; t += nwaves) {
        const long R = t * 16 + l15;
        const bool rv = R < rows;
        const f32x4 da = rv ? cvt4(rq.da) : zero4();   // rows beyond the tensor: dA = 0 makes every contribution vanish
        f32x4 xk;
#pragma unroll
        for (int k = 0; k < kThinTaps; ++k) xk[k] = k < K ? cvt1(rq.x[k]) : 0.f;
        request(t + nwaves, rq);
        dba += da;
        // tiles the row contractions read transposed: dA[row][j], x[row][k] (+ the column of ones)
        st4(Dt + l15 * kThinLdD + 4 * g, da);
        Xt[l15 * kThinLdX + g] = g < K ? (g == 0 ? xk[0] : g == 1 ? xk[1] : g == 2 ? xk[2] : xk[3]) : (g == K ? 1.f : 0.f);
        if (g == 0) Xt[l15 * kThinLdX + 4] = K == 4 ? 1.f : 0.f;
        // recomputed gate inputs z = [u | q] of channels 16 mt + 4 g + r, row l15
        f32x4 z[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) z[mt] = ld4(Bs + 16 * mt + 4 * g);
        cw.run(xk, g, z);
        // channel tile by channel tile: dH = Wa dA^T, gate backward; z becomes [dU | dQ], h goes to the transposition tile
        {
            const typename MM::frag db = MM::cvt(da);   // B[k = j = 4 g + s][n = row = l15]
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 dh = MM::mma(MM::cvt(ld4(Ws + (mt * 64 + lane) * 4)), db, zero4());
                f32x4 hq;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float u = z[mt][s], sg = sigmoid_f(z[4 + mt][s]);
                    float du, dq;
                    gate_bwd(dh[s], u, sg, ACT, du, dq);
                    hq[s] = gate_fwd(u, sg, ACT);
                    z[mt][s] = du;
                    z[4 + mt][s] = dq;
                }
                st4(Tt + l15 * kThinLdT + 16 * mt + 4 * g, hq);
            }
        }
        if (a.dZ && rv) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) stx4_wt(dZ_ + (size_t)R * 128 + 16 * mt + 4 * g, z[mt]);
        }
        // ---- row contractions: the D-layout values go through the wave's tile, row major, and come back as A operands [m][k = row] ----
        // dWa[c][j] += sum_rows h[row][c] dA[row][j]
        wave_lds_sync();
        {
            const typename MM::frag bd = MM::cvt(gather4(Dt + (4 * g) * kThinLdD + l15, kThinLdD));   // B[k = row = 4 g + s][n = j = l15]
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
                accA[ct] = MM::mma(MM::cvt(gather4(Tt + (4 * g) * kThinLdT + 16 * ct + l15, kThinLdT)), bd, accA[ct]);
        }
        const typename MM::frag bx = MM::cvt(gather4(Xt + (4 * g) * kThinLdX + l15, kThinLdX));       // B[k = row = 4 g + s][n = tap = l15]
        // dW_eff^T[o][k] += sum_rows dZ[row][o] x[row][k]   (P half, then Q half through the same tile)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            wave_lds_sync();   // the tile's previous readers are done
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) st4(Tt + l15 * kThinLdT + 16 * mt + 4 * g, z[4 * half + mt]);
            wave_lds_sync();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                accW[4 * half + mt] = MM::mma(MM::cvt(gather4(Tt + (4 * g) * kThinLdT + 16 * mt + l15, kThinLdT)), bx, accW[4 * half + mt]);
        }
        wave_lds_sync();       // (the next tile rewrites Dt / Xt / Tt)
    }
    // dba[j]: the lane's quad j = 4 g .. 4 g + 3 summed over the 16 rows of the group
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = dba[i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        dba[i] = v;
    }
    // ---- the four waves' accumulators in a fixed order: (w0 + w2) + (w1 + w3) -------------------------------------------------------
    f32x4* const cmb = reinterpret_cast<f32x4*>(stgcn_smem);   // [2][12][64] float4, then [2][16] floats
    float* const cmb_ba = stgcn_smem + 2 * 12 * 64 * 4;
    __syncthreads();           // every wave is done with its tiles
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cmb[((wave - 2) * 12 + i) * 64 + lane] = accW[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) cmb[((wave - 2) * 12 + 8 + i) * 64 + lane] = accA[i];
        if (l15 == 0) st4(cmb_ba + (wave - 2) * 16 + 4 * g, dba);
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) accW[i] += cmb[(wave * 12 + i) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) accA[i] += cmb[(wave * 12 + 8 + i) * 64 + lane];
        dba += ld4(cmb_ba + wave * 16 + 4 * g);
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) cmb[i * 64 + lane] = accW[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) cmb[(8 + i) * 64 + lane] = accA[i];
        if (l15 == 0) st4(cmb_ba + 4 * g, dba);
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) accW[i] += cmb[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) accA[i] += cmb[(8 + i) * 64 + lane];
    dba += ld4(cmb_ba + 4 * g);
    constexpr int c0 = 64, NC = 128;
    float* const part = a.part + (size_t)blockIdx.x * (c0 * 16 + 16 + 16 * NC + NC);
    // dWa[c = 16 ct + 4 g + r][j = l15]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(16 * ct + 4 * g + r) * 16 + l15] = accA[ct][r];
    if (l15 == 0) st4(part + c0 * 16 + 4 * g, dba);
    // dW_eff[k = l15][o = 16 mt + 4 g + r] (16-byte stores), column k = K: db_eff[o]
    if (l15 < K) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) st4(part + c0 * 16 + 16 + (size_t)l15 * NC + 16 * mt + 4 * g, accW[mt]);
    } else if (l15 == K) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) st4(part + c0 * 16 + 16 + 16 * NC + 16 * mt + 4 * g, accW[mt]);
    }
}

}  // namespace stgcn
