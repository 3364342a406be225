// Forward kernels of the fused ST-Conv block (gfx950).  See DESIGN.md section 3 for the data flow.
//
// All activations are channels-last (B, T, N, C) fp32.  "rows" are flattened (b, t, n) triples.
// Reference arithmetic: hazdzz/STGCN model/layers.py:87-120 (TemporalConvLayer), :143-172
// (ChebGraphConv), :194-206 (GraphConv), :222-231 (GraphConvLayer), :250-258 (STConvBlock).
#pragma once
#include "stgcn_device.hip.h"

namespace stgcn {

// ================================================================================================
// Weight packing: parameters in the reference's native layouts -> MFMA B-operand fragment order.
// Packed layout of a [K x NC] matrix W (K padded to KP = 16*KCH, NC = 16*NTT):
//     Wp[((nt*KCH + kc)*64 + lane)*4 + s] = W[kc*16 + 4*(lane>>4) + s][nt*16 + (lane&15)]
// so a wave fetches one (nt, kc) fragment with a single coalesced 1 KiB load (16 B per lane).
// ================================================================================================
enum PackKind { PK_TCONV_FWD = 0, PK_TCONV_BWD = 1, PK_TCONV_BIAS = 2, PK_ALIGN_FWD = 3, PK_ALIGN_BWD = 4, PK_ALIGN_BIAS = 5,
                PK_LIN_FWD = 6, PK_LIN_BWD = 7,
                PK_TCONV_DENSE = 8, PK_ALIGN_DENSE = 9,
                PK_TCONV_BWDT = 10,
                PK_ZERO = 11 };   // n zero words at dst (control words of the chained launches: re-armed by the launch that opens the step)   // one-step transposed conv as a dense GEMM: K = NC (o), cols = tap*Cin + i   // W_eff row major [KP][NC] (used to recompute a cheap first-layer conv in backward)   // nn.Linear weight (out = Cout, in = Cin): y = x W^T / dx = dy W

struct PackJob {
    int kind;
    int n;            // number of floats to produce
    float* dst;
    const float* w;   // conv weight (2*Cout, Cin, Kt, 1)   | align weight (c1, c0, 1, 1)
    const float* b;   // conv bias (2*Cout) or null          | align bias (c1) or null
    const float* aw;  // temporal-layer Align conv weight (Cout, Cin, 1, 1) (used iff Cin > Cout)
    const float* ab;  // temporal-layer Align conv bias (Cout) or null
    int Cin, Cout, Kt, KCH, gated;
};
constexpr int kMaxPackJobs = 40;   // 2-3 ST blocks x 10 jobs + the head (stgcn_prepack)
constexpr int kMaxStepCounters = 4;
struct PackArgs {
    PackJob job[kMaxPackJobs];
    int start[kMaxPackJobs + 1];   // prefix sums of workgroup counts
    int njobs;
    // device-side step counters advanced by the first launch of a training step (stgcn_prepack): *ptr = (*ptr + inc) [% mod]
    int ncounters;
    long* cptr[kMaxStepCounters];
    long cinc[kMaxStepCounters], cmod[kMaxStepCounters];
};

// W_eff[tap*Cin + i][o]: causal-conv weight with the residual branch Align(x)[:, :, Kt-1:] folded in
// (layers.py:88-89 and :14-23): the residual only touches tap Kt-1 of the first Cout outputs.
__device__ __forceinline__ float tconv_weff(const PackJob& j, int tap, int i, int o) {
    float v = j.w[((size_t)o * j.Cin + i) * j.Kt + tap];
    if (tap == j.Kt - 1 && o < j.Cout) {
        if (j.Cin > j.Cout) v += j.aw[(size_t)o * j.Cin + i];
        else if (i == o) v += 1.0f;
    }
    return v;
}

// counts of first-layer waves that have read the window index in a launch shared with the pack role (PackSync, stgcn_kernels_thin.hip.h):
// kPackSyncWords words 256 bytes apart -- device-scope atomics on ONE address retire at ~15 ns each on MI355X (2072 waves: 31 us, measured,
// pass r6-12), on 32 addresses of different L2 channels the same reports take ~1 us.  Zero between launches (the pack role's counter wave
// re-arms them).  One array per process: two models stepping CONCURRENTLY on different streams of one process would share it -- not a
// configuration of this path (one process per GPU, one step at a time).
constexpr int kPackSyncWords = 32, kPackSyncStride = 64;
__device__ unsigned g_pack_readers[kPackSyncWords * kPackSyncStride];
// bid = workgroup index within the pack role; sy: see PackSync (on == 0: a launch of its own)
template <typename SY>
__device__ __forceinline__ void pack_body(const PackArgs& a, const int bid, const SY& sy) {
    if (bid == 0 && threadIdx.x < 64) {   // the counter wave
        for (int k = 0; k < a.ncounters; ++k) {
            if (sy.on && a.cptr[k] == sy.idx_ptr) {
                // the first-layer role of this launch reads this counter: bump it only when all of its waves have the old value (bounded:
                // 2 s of the 100 MHz wall clock -- a device that cannot start ~500 small workgroups in that time has other problems).
                // Lane l < kPackSyncWords polls word l; the wave sums.
                const int l = (int)threadIdx.x;
                unsigned* const word = g_pack_readers + (l < kPackSyncWords ? l : 0) * kPackSyncStride;
#if defined(__HIP_DEVICE_COMPILE__)
                const long long t0 = wall_clock64();
                for (;;) {
                    unsigned c = l < kPackSyncWords ? chain_ld(word) : 0u;
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
                    if (c >= sy.expected || wall_clock64() - t0 > kChainSpinTicks) break;
                    __builtin_amdgcn_s_sleep(4);
                }
#endif
                if (l < kPackSyncWords) chain_st(word, 0u);
            }
            if (threadIdx.x == 0) {
                const long v = *a.cptr[k] + a.cinc[k];
                *a.cptr[k] = a.cmod[k] > 0 ? v % a.cmod[k] : v;
            }
        }
    }
    int jb = 0;
    while (jb + 1 < a.njobs && bid >= a.start[jb + 1]) ++jb;
    const PackJob& j = a.job[jb];
    const int e = (bid - a.start[jb]) * kThreads + (int)threadIdx.x;
    if (e >= j.n) return;
    const int NC = j.gated ? 2 * j.Cout : j.Cout;
    float v = 0.f;
    if (j.kind == PK_ZERO) {
        v = 0.f;
    } else if (j.kind == PK_TCONV_BIAS) {
        v = j.b ? j.b[e] : 0.f;
        if (j.Cin > j.Cout && e < j.Cout && j.ab) v += j.ab[e];
    } else if (j.kind == PK_ALIGN_BIAS) {
        v = (j.Cin > j.Cout && j.b) ? j.b[e] : 0.f;
    } else if (j.kind == PK_TCONV_DENSE) {
        const int kidx = e / NC, o = e - kidx * NC;
        if (kidx < j.Kt * j.Cin) v = tconv_weff(j, kidx / j.Cin, kidx % j.Cin, o);
    } else if (j.kind == PK_ALIGN_DENSE) {   // Wa[i][jj] row major, i < Cin (= c0), jj < Cout (= c1)
        const int i = e / j.Cout, jj = e - i * j.Cout;
        v = (j.Cin > j.Cout) ? j.w[(size_t)jj * j.Cin + i] : (i == jj ? 1.f : 0.f);
    } else {
        const int s = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
        const int kc = rest % j.KCH, nt = rest / j.KCH;
        const int kidx = kc * 16 + 4 * (lane >> 4) + s, col = nt * 16 + (lane & 15);
        if (j.kind == PK_TCONV_FWD) {          // K = Kt*Cin, cols = NC
            if (kidx < j.Kt * j.Cin) v = tconv_weff(j, kidx / j.Cin, kidx % j.Cin, col);
        } else if (j.kind == PK_TCONV_BWD) {   // K = Kt*NC, cols = Cin (padded to 16)
            if (col < j.Cin) v = tconv_weff(j, kidx / NC, col, kidx % NC);
        } else if (j.kind == PK_TCONV_BWDT) {  // K = NC (o), cols = Kt*Cin: B[o][tap*Cin + i] = W_eff[tap*Cin + i][o]
            if (kidx < NC && col < j.Kt * j.Cin) v = tconv_weff(j, col / j.Cin, col % j.Cin, kidx);
        } else if (j.kind == PK_ALIGN_FWD) {   // A = H @ Wa : K = c0 (=Cin), cols = c1 (=Cout)
            if (kidx < j.Cin && col < j.Cout) v = (j.Cin > j.Cout) ? j.w[(size_t)col * j.Cin + kidx] : (kidx == col ? 1.f : 0.f);
        } else if (j.kind == PK_ALIGN_BWD) {   // dH = dA @ Wa^T : K = c1, cols = c0
            if (kidx < j.Cout && col < j.Cin) v = (j.Cin > j.Cout) ? j.w[(size_t)kidx * j.Cin + col] : (kidx == col ? 1.f : 0.f);
        } else if (j.kind == PK_LIN_FWD) {     // K = in (Cin), cols = out (Cout): B[k][o] = weight[o][k]
            if (kidx < j.Cin && col < j.Cout) v = j.w[(size_t)col * j.Cin + kidx];
        } else {                                // PK_LIN_BWD: K = out, cols = in: B[o][i] = weight[o][i]
            if (kidx < j.Cout && col < j.Cin) v = j.w[(size_t)kidx * j.Cin + col];
        }
    }
    j.dst[e] = v;
}
struct PackNoSync { int on; const long* idx_ptr; unsigned expected; };
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) { pack_body(a, (int)blockIdx.x, PackNoSync{0, nullptr, 0u}); }

// The graph shift operator is constant, so the Chebyshev polynomials T_k(L) (T_0 = I, T_1 = L, T_k = 2 L T_{k-1} - T_{k-2},
// layers.py:153-161 applied to the operator instead of the activations) are formed ONCE per model and rewritten into MFMA
// B-operand fragment order, zero padded to NP = roundup(N, 16):
//     Tf[k-1][((ht*KCH + kc)*64 + lane)*4 + s] = T_k[ht*16 + (lane&15)][kc*16 + 4*(lane>>4) + s]        k = 1 .. terms-1
// (and the same for T_k^T, used by backward).  Every term of the graph conv is then an independent X0 x T_k^T product: no
// recursion through LDS, no barrier between terms, one pass over the staged X0 for all terms.  One (node tile, k chunk)
// fragment is a contiguous 1 KiB wave load (8 full cache lines).
// dense, zero padded copy: D[h][i] = L[h][i]
__global__ __launch_bounds__(256) void gso_dense_kernel(const float* L, int N, int NP, float* D) {
    const int e = (int)blockIdx.x * kThreads + (int)threadIdx.x;
    if (e >= NP * NP) return;
    const int h = e / NP, i = e - h * NP;
    D[e] = (h < N && i < N) ? L[(size_t)h * N + i] : 0.f;
}
// out = 2 * L * Tm1 - Tm2 (dense NP x NP; Tm2 == nullptr stands for the identity on the first N rows); fp64 accumulation
__global__ __launch_bounds__(256) void cheb_next_kernel(const float* L, const float* Tm1, const float* Tm2, int N, int NP, float* out) {
    const int e = (int)blockIdx.x * kThreads + (int)threadIdx.x;
    if (e >= NP * NP) return;
    const int h = e / NP, i = e - h * NP;
    double acc = 0.0;
    for (int m = 0; m < NP; ++m) acc += (double)L[(size_t)h * NP + m] * (double)Tm1[(size_t)m * NP + i];
    const double prev = Tm2 ? (double)Tm2[e] : ((h == i && h < N) ? 1.0 : 0.0);
    out[e] = (float)(2.0 * acc - prev);
}
// dense padded D -> fragment order of D and of D^T
__global__ __launch_bounds__(256) void gso_frag_kernel(const float* D, int NP, float* Tf, float* TTf) {
    const int e = (int)blockIdx.x * kThreads + (int)threadIdx.x;
    if (e >= NP * NP) return;
    const int KCH = NP >> 4;
    const int s = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
    const int kc = rest % KCH, ht = rest / KCH;
    const int h = ht * 16 + (lane & 15), i = kc * 16 + 4 * (lane >> 4) + s;
    Tf[e] = D[(size_t)h * NP + i];
    TTf[e] = D[(size_t)i * NP + h];
}

// ================================================================================================
// Row-tile im2col staging (shared by the forward conv GEMM, the backward-data GEMM).
// A "tap source" is a (B, Tsrc, N, C) tensor viewed as the implicit matrix whose row (b, t, n),
// t < Tdst, is the concatenation over taps k of src[b, t + dir*k, n, :]  (dir = +1 forward conv,
// dir = -1 transposed conv; out-of-range taps read as zeros).
// ================================================================================================
struct TapSrc {
    const float* src;
    int C, taps, N, Tsrc, Tdst, dir;
    long rows;   // B * Tdst * N
    // Windows that are NOT laid out back to back (device-side windowing): consecutive windows b
    // start bstride rows apart (0 = dense, Tsrc * N) -- e.g. N rows when window b is rows [s + b, s + b + Tsrc) of the
    // resident (time, N) series, which the reference replicates 12x into a (num, 1, n_his, N) tensor (dataloader.py:32-47)
    // -- and the whole source is shifted by *idx_dev * idx_stride floats (the position of a captured training step in the
    // resident series; idx_dev == nullptr: no shift).
    long bstride;
    const long* idx_dev;
    long idx_stride;
};
// (idx_stride counts ELEMENTS of the source's storage type)
template <typename ET = float>
__device__ __forceinline__ const ET* tap_base(const TapSrc& ts) { return et_ptr<ET>(ts.src) + (ts.idx_dev ? *ts.idx_dev * ts.idx_stride : 0); }
__device__ __forceinline__ long tap_bstride(const TapSrc& ts) { return ts.bstride ? ts.bstride : (long)ts.Tsrc * ts.N; }

// per-tile row bookkeeping in LDS: rowbase[r] = flat source row of tap 0, rowt[r] = t (or -2^20 if the
// row is beyond the tensor, which makes every tap invalid)
template <int TR = kTileRows>
__device__ __forceinline__ void tile_rowinfo(const TapSrc& ts, long tile_row0, int* rowbase, int* rowt) {
    const int r = threadIdx.x;
    if (r < TR) {
        const long R = tile_row0 + r;
        int base = 0, t = -(1 << 20);
        if (R < ts.rows) {   // rows < 2^31 (checked on the host): 32-bit divisions only
            const unsigned per_b = (unsigned)(ts.Tdst * ts.N), Ru = (unsigned)R;
            const unsigned b = Ru / per_b;
            const unsigned rem = Ru - b * per_b;
            t = (int)(rem / (unsigned)ts.N);
            base = (int)(b * (unsigned)tap_bstride(ts) + rem);   // = (b*Tsrc + t)*N + n for dense windows
        }
        rowbase[r] = base;
        rowt[r] = t;
    }
}

// Stage columns [k0, k0 + kseg) of the implicit matrix for the TR rows of the tile into At[TR][lda].
template <int TR = kTileRows, int THREADS = kThreads, typename ET = float>
__device__ __forceinline__ void tile_load_segment(const TapSrc& ts, const int* rowbase, const int* rowt, int k0, int kseg,
                                                  float* At, int lda) {
    const int K = ts.taps * ts.C, csh = pow2_shift(ts.C);
    const ET* const src = tap_base<ET>(ts);
    if ((ts.C & 3) == 0) {
        const int q4 = kseg >> 2, qsh = pow2_shift(q4);
        for (int idx = threadIdx.x; idx < TR * q4; idx += THREADS) {
            const int r = fast_div(idx, q4, qsh), q = idx - r * q4;
            const int kidx = k0 + 4 * q;
            f32x4 v = zero4();
            if (kidx < K) {
                const int tap = fast_div(kidx, ts.C, csh), ch = kidx - tap * ts.C;
                const int tt = rowt[r] + ts.dir * tap;
                if (tt >= 0 && tt < ts.Tsrc) v = ldx4(src + ((size_t)(rowbase[r] + ts.dir * tap * ts.N)) * ts.C + ch);
            }
            st4(At + r * lda + 4 * q, v);
        }
    } else {   // narrow inputs (C = 1 for the first block): scalar gather of the K valid columns only (one round of loads per
               // thread for 64 rows x 3 taps instead of one per 256 of the 16 padded columns), zeros in the padding
        const int kv = (K - k0) < kseg ? (K - k0 > 0 ? K - k0 : 0) : kseg, padw = kseg - kv;
        for (int idx = threadIdx.x; idx < TR * padw; idx += THREADS) {
            const int r = idx / padw, q = idx - r * padw;
            At[r * lda + kv + q] = 0.f;
        }
        for (int idx = threadIdx.x; idx < TR * kv; idx += THREADS) {
            const int r = idx / kv, q = idx - r * kv;
            const int kidx = k0 + q;
            const int tap = fast_div(kidx, ts.C, csh), ch = kidx - tap * ts.C;
            const int tt = rowt[r] + ts.dir * tap;
            float v = 0.f;
            if (tt >= 0 && tt < ts.Tsrc) v = ldx1(src + ((size_t)(rowbase[r] + ts.dir * tap * ts.N)) * ts.C + ch);
            At[r * lda + q] = v;
        }
    }
}

// Workgroups that share a CU start together and would run their load / MFMA / epilogue phases in lockstep (the MFMA
// pipe then idles during every load phase).  Block b, b+256, b+512, .. land on the same CU; delaying them by a
// different fraction of a phase interleaves the phases.  (experiment knob: -DSTGCN_STAGGER=<units of 2048 cycles>)
#ifndef STGCN_STAGGER
#define STGCN_STAGGER 0
#endif
// STGCN_ABL: timing-only ablations (results are WRONG): 1 gconv without operator loads, 2 gconv without term MFMAs,
// 3 tconv_fwd without U/S/A stores, 4 tconv_fwd without tile loads, 5 tconv_fwd without MFMAs, 6 gconv without X0 staging + stores
#ifndef STGCN_ABL
#define STGCN_ABL 0
#endif
// STGCN_PIPE_FWD / STGCN_PIPE_BWD: software-pipelined K-segment loop (weights -> registers, next tile segment in flight
// during the MFMAs).  Measured on MI355X (profiles/): helps the transposed conv (3+ segments), hurts the forward conv.
#ifndef STGCN_PIPE_FWD
#define STGCN_PIPE_FWD 0
#endif
#ifndef STGCN_PIPE_BWD
#define STGCN_PIPE_BWD 1
#endif
__device__ __forceinline__ void stagger_start() {
#if STGCN_STAGGER > 0
    const int k = ((int)(blockIdx.x >> 8) & 3) * STGCN_STAGGER;
    for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(32);
#endif
}

// Split staging: global -> registers (in flight during the MFMA loop) and registers -> LDS.
template <int TR, int THREADS>
struct TileRegs {
    f32x4 v[(TR * (kSegMax / 4) + THREADS - 1) / THREADS];
};
template <int TR, int THREADS, typename ET = float>
__device__ __forceinline__ void tile_prefetch_segment(const TapSrc& ts, const int* rowbase, const int* rowt, int k0, int kseg,
                                                      TileRegs<TR, THREADS>& regs) {
    constexpr int NV = (TR * (kSegMax / 4) + THREADS - 1) / THREADS;
    const int K = ts.taps * ts.C, q4 = kseg >> 2, qsh = pow2_shift(q4), csh = pow2_shift(ts.C);
    const ET* const src = tap_base<ET>(ts);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * THREADS;
        f32x4 v = zero4();
        if (idx < TR * q4) {
            const int r = fast_div(idx, q4, qsh), q = idx - r * q4;
            const int kidx = k0 + 4 * q;
            if (kidx < K) {
                const int tap = fast_div(kidx, ts.C, csh), ch = kidx - tap * ts.C;
                const int tt = rowt[r] + ts.dir * tap;
                if (tt >= 0 && tt < ts.Tsrc) v = ldx4(src + ((size_t)(rowbase[r] + ts.dir * tap * ts.N)) * ts.C + ch);
            }
        }
        regs.v[i] = v;
    }
}
template <int TR, int THREADS>
__device__ __forceinline__ void tile_commit_segment(int kseg, const TileRegs<TR, THREADS>& regs, float* At, int lda) {
    constexpr int NV = (TR * (kSegMax / 4) + THREADS - 1) / THREADS;
    const int q4 = kseg >> 2, qsh = pow2_shift(q4);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = threadIdx.x + i * THREADS;
        if (idx < TR * q4) {
            const int r = fast_div(idx, q4, qsh), q = idx - r * q4;
            st4(At + r * lda + 4 * q, regs.v[i]);
        }
    }
}

// Weight fragments of a whole segment (<= 8 chunks) held in registers: loaded BEFORE the next segment's tile prefetch is
// issued, so that waiting for them (in-order vmcnt) does not drain the prefetch.
template <int NT>
struct SegWeights {
    f32x4 b[kSegMax / 16][NT];
};
template <int NT>
__device__ __forceinline__ void seg_load_weights(SegWeights<NT>& w, int kcs, const float* Wp, int kc0, int KCH, int nt0, int nts) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int kc = 0; kc < kSegMax / 16; ++kc)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            w.b[kc][j] = kc < kcs ? ld4(Wp + ((size_t)((nt0 + j * nts) * KCH + kc0 + kc) * 64 + lane) * 4) : zero4();
}
template <int WM, int NT, typename ET = float>
__device__ __forceinline__ void seg_mma_w(f32x4 (&acc)[WM][NT], const float* At, int lda, int mt0, int kcs, const SegWeights<NT>& w) {
    const int lane = threadIdx.x & 63;
    const float* arow = At + (mt0 * 16 + (lane & 15)) * lda + 4 * (lane >> 4);
#pragma unroll
    for (int kc = 0; kc < kSegMax / 16; ++kc) {
        if (kc < kcs) {
            f32x4 a[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = ld4(arow + i * 16 * lda + kc * 16);
            mma_tile<ET, WM, NT>(acc, a, w.b[kc]);
        }
    }
}

// acc[i][j] += A_tile(m-tile mt0+i) x Wp(n-tile nt0 + j*nts) over kcs 16-wide chunks of this segment.
template <int WM, int NT, typename ET = float>
__device__ __forceinline__ void seg_mma(f32x4 (&acc)[WM][NT], const float* At, int lda, int mt0, int kcs, const float* Wp,
                                        int kc0, int KCH, int nt0, int nts) {
    const int lane = threadIdx.x & 63;
    const float* arow = At + (mt0 * 16 + (lane & 15)) * lda + 4 * (lane >> 4);
    for (int kc = 0; kc < kcs; ++kc) {
        f32x4 b[NT], a[WM];
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = ld4(Wp + ((size_t)((nt0 + j * nts) * KCH + kc0 + kc) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = ld4(arow + i * 16 * lda + kc * 16);
        mma_tile<ET, WM, NT>(acc, a, b);
    }
}

// Weight fragments of a WHOLE (small) K held in registers, requested in one batch at kernel start: kernels with a couple
// of workgroups per CU (the head's fc GEMMs) cannot hide one L2 round trip per K chunk behind other waves.
template <int NT, int KMAX>
struct PreW {
    f32x4 b[KMAX][NT];
};
template <int NT, int KMAX>
__device__ __forceinline__ void pre_load_weights(PreW<NT, KMAX>& w, const float* Wp, int KCH, int nt0, int nts) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int kc = 0; kc < KMAX; ++kc)
#pragma unroll
        for (int j = 0; j < NT; ++j) w.b[kc][j] = kc < KCH ? ld4(Wp + ((size_t)((nt0 + j * nts) * KCH + kc) * 64 + lane) * 4) : zero4();
}
template <int WM, int NT, int KMAX, typename ET = float>
__device__ __forceinline__ void pre_mma(f32x4 (&acc)[WM][NT], const float* At, int lda, int mt0, int KCH, const PreW<NT, KMAX>& w) {
    const int lane = threadIdx.x & 63;
    const float* arow = At + (mt0 * 16 + (lane & 15)) * lda + 4 * (lane >> 4);
#pragma unroll
    for (int kc = 0; kc < KMAX; ++kc) {
        if (kc < KCH) {
            f32x4 a[WM];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = ld4(arow + i * 16 * lda + kc * 16);
            mma_tile<ET, WM, NT>(acc, a, w.b[kc]);
        }
    }
}

// LDS carve for the row-tile GEMM kernels (floats): [rowbase 64 ints][rowt 64 ints][4 scratch words][At 64 x (kSegMax+4)]
constexpr int kTileHdr = 132;
// (tconv_fwd re-uses At as its [64][NC + 4] epilogue tile: tile_lds_floats(NC))
constexpr int kLdaMax = kSegMax + 4;
constexpr int kTileLdsFloats = kTileHdr + kTileRows * kLdaMax;
inline int tile_lds_floats(int nc, int rows = kTileRows) { return kTileHdr + rows * ((nc > kSegMax ? nc : kSegMax) + 4); }

// ================================================================================================
// F1: gated temporal convolution  Z = im2col(x) @ W_eff + b_eff ; U = Z[:, :Cout] ; S = sigmoid(Z[:, Cout:])
//     H = act(U) * S   (layers.py:87-109, residual folded into W_eff)
//     optional epilogue: A = H @ Wa + ba  (GraphConvLayer's Align, layers.py:223)
// grid = ceil(rows / 64); each wave owns n-tiles {w + 4j}, so P channel c and Q channel c + Cout
// meet in the same lane (GLU pairing without data movement).
// ================================================================================================
struct TconvFwdArgs {
    TapSrc ts;            // x viewed through Kt taps, dir = +1
    const float* Wp;      // packed W_eff, K = Kt*Cin (KCH chunks), NC = 2*Cout
    const float* bias;    // b_eff[2*Cout]
    int KCH, Cout, act;
    float* U;             // [rows][Cout]  (nullable)
    float* S;             // [rows][Cout]  (nullable)
    float* H;             // [rows][Cout]  (nullable)
    const float* Wap;     // packed Wa: K = Cout, cols = c1 (nullable -> no align epilogue)
    const float* ba;      // [c1]
    float* A;             // [rows][c1]
    int c1;
    float2* rowstat;      // [rows] (mean over the row's Cout channels of H, sum of squared deviations) or null
};

// TM = m-tiles (16 rows) per tile; WAVES = 4 or 8.  With 8 waves the two halves of the workgroup (wave >> 2) own
// the two halves of the tile's rows: twice the waves in flight per CU at the same LDS footprint.
template <int NT, int TM, int WAVES, typename ET>
__global__ __launch_bounds__(WAVES * 64) void tconv_fwd_kernel(TconvFwdArgs a) {
    constexpr int TR = TM * 16, THREADS = WAVES * 64, WM = TM / (WAVES / 4);   // rows per tile, m-tiles per wave
    typedef Mma<ET> MM;
    ET* const U_ = et_ptr<ET>(a.U);
    ET* const S_ = et_ptr<ET>(a.S);
    ET* const H_ = et_ptr<ET>(a.H);
    ET* const A_ = et_ptr<ET>(a.A);
    extern __shared__ float stgcn_smem[];
    int* rowbase = reinterpret_cast<int*>(stgcn_smem);
    int* rowt = rowbase + 64;
    float* At = stgcn_smem + kTileHdr;
    const int wv = threadIdx.x >> 6, wave = wv & 3, mt0 = (wv >> 2) * WM, lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    const long row0 = (long)xcd_item(blockIdx.x, gridDim.x) * TR;

    STGCN_PHASE(1, 0);
    stagger_start();
    tile_rowinfo<TR>(a.ts, row0, rowbase, rowt);
    __syncthreads();
    STGCN_PHASE(1, 1);

    f32x4 acc[WM][NT];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4();

    const int KP = a.KCH * 16;
#if STGCN_PIPE_FWD
    if ((a.ts.C & 3) == 0) {
        // software pipeline over the K segments (single LDS buffer): weight fragments of segment s -> registers, then the
        // tile loads of segment s+1 are issued and stay in flight while the MFMAs of segment s run.
        TileRegs<TR, THREADS> regs;
        int kseg = KP < kSegMax ? KP : kSegMax;
        tile_prefetch_segment<TR, THREADS, ET>(a.ts, rowbase, rowt, 0, kseg, regs);
        tile_commit_segment<TR, THREADS>(kseg, regs, At, kseg + 4);
        for (int k0 = 0; k0 < KP; k0 += kSegMax) {
            kseg = (KP - k0) < kSegMax ? (KP - k0) : kSegMax;
            const int kn = k0 + kSegMax, ksegn = (KP - kn) < kSegMax ? (KP - kn) : kSegMax;
            SegWeights<NT> w;
            seg_load_weights<NT>(w, kseg >> 4, a.Wp, k0 >> 4, a.KCH, wave, 4);
            if (kn < KP) tile_prefetch_segment<TR, THREADS, ET>(a.ts, rowbase, rowt, kn, ksegn, regs);
            __syncthreads();                       // At (segment k0) committed by every thread
            seg_mma_w<WM, NT, ET>(acc, At, kseg + 4, mt0, kseg >> 4, w);
            if (kn < KP) {
                __syncthreads();                   // every wave done reading At
                tile_commit_segment<TR, THREADS>(ksegn, regs, At, ksegn + 4);
            }
        }
    } else
#endif
    for (int k0 = 0; k0 < KP; k0 += kSegMax) {
        const int kseg = (KP - k0) < kSegMax ? (KP - k0) : kSegMax;
        if (k0 > 0) __syncthreads();   // previous segment fully consumed
#if STGCN_ABL == 4
        for (int idx = threadIdx.x; idx < TR * (kseg + 4); idx += THREADS) At[idx] = 0.5f;
#else
        tile_load_segment<TR, THREADS, ET>(a.ts, rowbase, rowt, k0, kseg, At, kseg + 4);
#endif
        __syncthreads();
        STGCN_PHASE(1, 2 + 2 * (k0 / kSegMax));
#if STGCN_ABL == 5
        acc[0][0][0] += At[threadIdx.x];
#else
        seg_mma<WM, NT, ET>(acc, At, kseg + 4, mt0, kseg >> 4, a.Wp, k0 >> 4, a.KCH, wave, 4);
#endif
        STGCN_PHASE(1, 3 + 2 * (k0 / kSegMax));
    }

    // ---- epilogue: accumulators -> LDS tile Zt[64][NC + 4] -> row-major float4 pass (coalesced U/S/H stores) ----
    const int Cout = a.Cout, NC = 2 * Cout, ldz = NC + 4;
    const bool do_align = a.Wap != nullptr;
    float* Zt = At;
    __syncthreads();   // every wave is done reading At
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (wave + 4 * j) * 16 + l15;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Zt[((mt0 + i) * 16 + 4 * g + r) * ldz + col] = acc[i][j][r];
    }
    __syncthreads();
    const int c4n = Cout >> 2, c4sh = pow2_shift(c4n);
    for (int idx = threadIdx.x; idx < TR * c4n; idx += THREADS) {
        const int row = fast_div(idx, c4n, c4sh), c4 = idx - row * c4n;
        const long R = row0 + row;
        const f32x4 p = ld4(Zt + row * ldz + 4 * c4), q = ld4(Zt + row * ldz + Cout + 4 * c4);
        const f32x4 bp = ld4(a.bias + 4 * c4), bq = ld4(a.bias + Cout + 4 * c4);
        f32x4 u, sg, h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = p[i] + bp[i];
            sg[i] = sigmoid_f(q[i] + bq[i]);
            h[i] = gate_fwd(u[i], sg[i], a.act);
        }
#if STGCN_ABL == 3
        if (R < a.ts.rows && u[0] == 12345.678f) stx4(U_ + (size_t)R * Cout + 4 * c4, sg);
#else
        if (R < a.ts.rows) {
            const size_t o = (size_t)R * Cout + 4 * c4;
            if (a.U) stx4_wt(U_ + o, u);
            if (a.S) stx4_wt(S_ + o, sg);
            if (a.H) stx4_wt(H_ + o, h);
        }
#endif
        if (a.rowstat) {   // per-row LayerNorm partials: the c4n lanes holding one row are contiguous in the wave
            float sr = (h[0] + h[1]) + (h[2] + h[3]);
            for (int m = c4n >> 1; m >= 1; m >>= 1) sr += __shfl_xor(sr, m);
            const float mr = sr / (float)Cout;
            float d2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) d2 += (h[i] - mr) * (h[i] - mr);
            for (int m = c4n >> 1; m >= 1; m >>= 1) d2 += __shfl_xor(d2, m);
            if (c4 == 0 && R < a.ts.rows) a.rowstat[R] = make_float2(mr, d2);
        }
        if (do_align) st4(Zt + row * ldz + 4 * c4, h);   // H tile in place of the P half
    }
    STGCN_PHASE(1, 12);
    if (!do_align) return;
    __syncthreads();
    const int ldh = ldz;

    // ---- align epilogue: A[TR x c1] = H[TR x Cout] @ Wa + ba ; wave wv (< TM) owns rows 16wv..16wv+15 -------
    if (wv >= TM) return;
    const int KCHa = Cout >> 4;
    for (int nt = 0; nt < (a.c1 >> 4); ++nt) {
        f32x4 c0 = zero4(), c1v = zero4();
        f32x4 bw[8];   // the Align weights of this column tile, requested together (Cout <= 128: at most 8 chunks; one L2 round trip, not KCHa)
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) bw[kc] = ld4(a.Wap + ((size_t)(nt * KCHa + (kc < KCHa ? kc : KCHa - 1)) * 64 + lane) * 4);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            if (kc < KCHa) {
                const f32x4 av = ld4(At + (wv * 16 + l15) * ldh + kc * 16 + 4 * g);
                MM::mma_split(MM::cvt(av), MM::cvt(bw[kc]), c0, c1v);
            }
        }
        const int col = nt * 16 + l15;
        const float bb = a.ba[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long R = row0 + wv * 16 + 4 * g + r;
            if (R < a.ts.rows) stx1(A_ + (size_t)R * a.c1 + col, c0[r] + c1v[r] + bb);
        }
    }
    STGCN_PHASE(1, 13);
}

// ================================================================================================
// Batched whole-tile staging (used by the wide-output kernels below): every global load of the im2col tile [TR x K] (all
// taps) is ISSUED before the first wait, SB 16-byte loads per thread in flight, row coordinates computed in registers (no
// LDS row table, no barrier before the loads).  The row-tile kernel above stages with a load -> wait -> LDS-store loop,
// which the many co-resident workgroups of the ST-block shapes hide; a variant of it with these batched loads and the
// weights of a whole K round in registers was faster only at tiny batch sizes (profiles/r19_r33_experiments.md v42).
// ================================================================================================
// (b, t, n) of a flat output row of a tap source, two 32-bit divisions (once per thread)
struct RowCoord { int b, t, n; };
__device__ __forceinline__ RowCoord row_coord(const TapSrc& ts, long R) {
    const unsigned per_b = (unsigned)(ts.Tdst * ts.N), Ru = (unsigned)R;
    RowCoord c;
    c.b = (int)(Ru / per_b);
    const unsigned rem = Ru - (unsigned)c.b * per_b;
    c.t = (int)(rem / (unsigned)ts.N);
    c.n = (int)(rem - (unsigned)c.t * (unsigned)ts.N);
    return c;
}
// coordinates of row R0 + r from those of R0 (r < 64: a few compare / subtract steps instead of divisions)
__device__ __forceinline__ RowCoord row_advance(const TapSrc& ts, RowCoord c, int r) {
    c.n += r;
    while (c.n >= ts.N) { c.n -= ts.N; ++c.t; }
    while (c.t >= ts.Tdst) { c.t -= ts.Tdst; ++c.b; }
    return c;
}

// Whole tile of a forward (dir = +1) tap source -> At[TR][lda], columns [0, KP): batches of SB loads per thread in flight.
template <int TR, int SB, int THREADS = kThreads, typename ET = float>
__device__ __forceinline__ void stage_tile_fwd(const TapSrc& ts, long row0, int KP, float* At, int lda) {
    const int tid = threadIdx.x, K = ts.taps * ts.C;
    const RowCoord c0 = row_coord(ts, row0 < ts.rows ? row0 : 0);
    const ET* const src = tap_base<ET>(ts);
    const long bs = tap_bstride(ts);
    if ((ts.C & 3) == 0) {
        const int c4n = ts.C >> 2, c4sh = pow2_shift(c4n), per_tap = TR * c4n, total = ts.taps * per_tap;
        for (int base = 0; base < total; base += THREADS * SB) {
            f32x4 v[SB];
            int dst[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int idx = base + i * THREADS + tid;
                v[i] = zero4();
                dst[i] = -1;
                if (idx < total) {
                    int tap = 0, rem = idx;
                    while (rem >= per_tap) { rem -= per_tap; ++tap; }
                    const int r = fast_div(rem, c4n, c4sh), c4 = rem - r * c4n;
                    dst[i] = r * lda + tap * ts.C + 4 * c4;
                    if (row0 + r < ts.rows) {
                        const RowCoord c = row_advance(ts, c0, r);
                        v[i] = ldx4(src + ((size_t)c.b * bs + (size_t)(c.t + tap) * ts.N + c.n) * ts.C + 4 * c4);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SB; ++i)
                if (dst[i] >= 0) st4(At + dst[i], v[i]);
        }
        if (K < KP) {   // zero the K padding (K not a multiple of 16)
            const int padw = KP - K;
            for (int idx = tid; idx < TR * padw; idx += THREADS) {
                const int r = idx / padw, q = idx - r * padw;
                At[r * lda + K + q] = 0.f;
            }
        }
    } else {   // narrow inputs (C = 1 for the first block): scalar gather of the K valid columns, zeros elsewhere
        const int total = TR * KP;
        for (int base = 0; base < total; base += THREADS * SB) {
            float v[SB];
            int dst[SB];
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const int idx = base + i * THREADS + tid;
                v[i] = 0.f;
                dst[i] = -1;
                if (idx < total) {
                    const int r = idx / KP, q = idx - r * KP;
                    dst[i] = r * lda + q;
                    if (q < K && row0 + r < ts.rows) {
                        const int tap = q / ts.C, ch = q - tap * ts.C;
                        const RowCoord c = row_advance(ts, c0, r);
                        v[i] = ldx1(src + ((size_t)c.b * bs + (size_t)(c.t + tap) * ts.N + c.n) * ts.C + ch);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SB; ++i)
                if (dst[i] >= 0) At[dst[i]] = v[i];
        }
    }
}

inline int tconv2_lds_floats(int kp, int nc, int rows) { return rows * ((kp > nc ? kp : nc) + 4); }

#ifdef STGCN_EXPERIMENTS   // round-1 experiment variants (never launched by the default build): -DSTGCN_EXPERIMENTS + STGCN_TCONV_V=3
// ================================================================================================
// F1 (v3, "time-complete tiles"): one workgroup owns 16 consecutive nodes of one window b for ALL time steps.
// The input tile X[b, 0..Tsrc-1, n0..n0+15, :] (Tsrc*16 rows of C floats) is read from HBM exactly once into LDS; the
// Kt taps of the temporal conv are row shifts INSIDE that tile (A operand of output step t, tap k = LDS rows
// (t + k)*16 .. +16): no im2col copy, no re-reads of neighbouring time steps through L2, identical work per
// workgroup (row tiles that straddle slabs made v1's workgroups differ 3x in backward).  Every wave keeps the
// weight fragments of its n-tiles for the WHOLE K in registers (loaded once per workgroup: B * ceil(N/16) = 416
// workgroups at C2 instead of 1242 tiles re-fetching 96 KB each) and walks the output steps in groups of MG m-tiles;
// each group's accumulators go through a small LDS tile for the bias / sigmoid / GLU row pass with coalesced 16-byte
// U, S stores, the LayerNorm row partials and the optional Align(c0 -> c1) GEMM, exactly like v1.
// Template: WAVES (4 or 8), NT n-tiles per wave (NC = 16 * WAVES * NT), KCW = K/16 chunks held in registers, MG m-tiles
// per group.  Requires C % 16 == 0 (the 1-channel first layer stays on v1).
// ================================================================================================
inline size_t tconv3_lds_bytes(int Tsrc, int C, int NC, int MG) { return ((size_t)Tsrc * 16 * (C + 4) + (size_t)MG * 16 * (NC + 4)) * sizeof(float); }

#ifndef STGCN_V3_STAGGER
#define STGCN_V3_STAGGER 0
#endif
template <int WAVES, int NT, int KCW, int MG>
__global__ __launch_bounds__(WAVES * 64) void tconv_fwd3_kernel(TconvFwdArgs a, int node_tiles) {
    constexpr int THREADS = WAVES * 64, NC = 16 * WAVES * NT, COUT = NC / 2, C4N = COUT / 4, LDZ = NC + 4;
    constexpr int NIT = MG * 16 * C4N / THREADS;   // row-pass iterations per group
    static_assert(MG * 16 * C4N % THREADS == 0 && THREADS % C4N == 0, "row pass must tile the workgroup");
    extern __shared__ float stgcn_smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const TapSrc& ts = a.ts;
    const int C = ts.C, ldx = C + 4, Tsrc = ts.Tsrc, Tdst = ts.Tdst, N = ts.N, cpt = C >> 4;   // cpt: K chunks per tap
    const int item = xcd_item(blockIdx.x, gridDim.x);
    const int b = item / node_tiles, n0 = (item - b * node_tiles) * 16;
    float* Xt = stgcn_smem;                       // [Tsrc*16][ldx]
    float* Zt = Xt + Tsrc * 16 * ldx;             // [MG*16][LDZ]
    const int KCH = a.KCH;
    const bool do_align = a.Wap != nullptr;

    STGCN_PHASE(a.Wap ? 1 : 7, 0);
    // ---- 1. weights of the whole K, bias, align weights -> registers (requested first) ---------------------------
    f32x4 wb[KCW][NT];
#pragma unroll
    for (int q = 0; q < KCW; ++q)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            wb[q][j] = q < KCH ? ld4(a.Wp + ((size_t)((wave + WAVES * j) * KCH + q) * 64 + lane) * 4) : zero4();
    const int c4 = tid & (C4N - 1);
    const f32x4 bp = ld4(a.bias + 4 * c4), bq = ld4(a.bias + COUT + 4 * c4);
    f32x4 wa[COUT / 16];
    float bal = 0.f;
    if (do_align && wave < MG) {
#pragma unroll
        for (int kc = 0; kc < COUT / 16; ++kc) wa[kc] = ld4(a.Wap + ((size_t)kc * 64 + lane) * 4);
        bal = a.ba[l15];
    }

    STGCN_PHASE(a.Wap ? 1 : 7, 1);
    // ---- 2. the input tile: all time steps of 16 nodes, once -----------------------------------------------------
    {
        const int c4n = C >> 2, c4sh = pow2_shift(c4n), total = Tsrc * 16 * c4n;
        const float* xb = tap_base(ts) + (size_t)b * tap_bstride(ts) * C;
        for (int base = 0; base < total; base += THREADS * 8) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + i * THREADS + tid;
                v[i] = zero4();
                if (idx < total) {
                    const int row = fast_div(idx, c4n, c4sh), q4 = idx - row * c4n, t = row >> 4, nn = row & 15;
                    if (n0 + nn < N) v[i] = ld4(xb + ((size_t)t * N + n0 + nn) * C + 4 * q4);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + i * THREADS + tid;
                if (idx < total) {
                    const int row = fast_div(idx, c4n, c4sh), q4 = idx - row * c4n;
                    st4(Xt + row * ldx + 4 * q4, v[i]);
                }
            }
        }
    }
    __syncthreads();
    STGCN_PHASE(a.Wap ? 1 : 7, 2);
#if STGCN_V3_STAGGER > 0
    // workgroups beyond the first 256 are the second residents of their CU: half a group period behind the first, so that
    // one workgroup's row pass runs beside the other's MFMAs (experiment knob, units of 2048 cycles)
    if (blockIdx.x >= 256)
        for (int i = 0; i < STGCN_V3_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
#endif

    // ---- 3. output steps in groups of MG ----------------------------------------------------------------------------
    for (int mg = 0; mg < Tdst; mg += MG) {
        f32x4 acc[MG][NT];
#pragma unroll
        for (int i = 0; i < MG; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = zero4();
#pragma unroll
        for (int q = 0; q < KCW; ++q) {
            if (q < KCH) {
                const int tap = q / cpt, kc = q - tap * cpt;
                f32x4 av[MG];
#pragma unroll
                for (int i = 0; i < MG; ++i) {
                    int t = mg + i;
                    if (t >= Tdst) t = Tdst - 1;   // padding m-tile of the last group: computed, never stored
                    av[i] = ld4(Xt + ((t + tap) * 16 + l15) * ldx + kc * 16 + 4 * g);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < MG; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma4(av[i][s], wb[q][j][s], acc[i][j]);
            }
        }
        if (mg == 0) STGCN_PHASE(a.Wap ? 1 : 7, 3);
        if (mg > 0) __syncthreads();   // the previous group's readers of Zt are done
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (wave + WAVES * j) * 16 + l15;
#pragma unroll
            for (int i = 0; i < MG; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Zt[(i * 16 + 4 * g + r) * LDZ + col] = acc[i][j][r];
        }
        __syncthreads();
        if (mg == 0) STGCN_PHASE(a.Wap ? 1 : 7, 4);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = (tid + it * THREADS) / C4N, t = mg + (row >> 4), nn = row & 15;
            const bool valid = t < Tdst && n0 + nn < N;
            const size_t R = ((size_t)b * Tdst + t) * N + n0 + nn;
            const f32x4 p = ld4(Zt + row * LDZ + 4 * c4), q = ld4(Zt + row * LDZ + COUT + 4 * c4);
            f32x4 u, sg, h;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i] = p[i] + bp[i];
                sg[i] = sigmoid_f(q[i] + bq[i]);
                h[i] = gate_fwd(u[i], sg[i], a.act);
            }
            if (valid) {
                const size_t o = R * COUT + 4 * c4;
                if (a.U) st4_wt(a.U + o, u);
                if (a.S) st4_wt(a.S + o, sg);
                if (a.H) st4_wt(a.H + o, h);
            }
            if (a.rowstat) {   // per-row LayerNorm partials: the C4N lanes holding one row are contiguous in the wave
                float sr = (h[0] + h[1]) + (h[2] + h[3]);
#pragma unroll
                for (int m = C4N >> 1; m >= 1; m >>= 1) sr += __shfl_xor(sr, m);
                const float mr = sr / (float)COUT;
                float d2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) d2 += (h[i] - mr) * (h[i] - mr);
#pragma unroll
                for (int m = C4N >> 1; m >= 1; m >>= 1) d2 += __shfl_xor(d2, m);
                if (c4 == 0 && valid) a.rowstat[R] = make_float2(mr, d2);
            }
            if (do_align) st4(Zt + row * LDZ + 4 * c4, h);   // H tile in place of the P half
        }
        if (mg == 0) STGCN_PHASE(a.Wap ? 1 : 7, 5);
        if (do_align) {
            __syncthreads();
            if (wave < MG && mg + wave < Tdst) {   // wave w: A[16 x 16] = H[m-tile w] @ Wa + ba
                f32x4 c0 = zero4(), c1v = zero4();
#pragma unroll
                for (int kc = 0; kc < COUT / 16; ++kc) {
                    const f32x4 av = ld4(Zt + (wave * 16 + l15) * LDZ + kc * 16 + 4 * g);
                    c0 = mfma4(av[0], wa[kc][0], c0);
                    c1v = mfma4(av[1], wa[kc][1], c1v);
                    c0 = mfma4(av[2], wa[kc][2], c0);
                    c1v = mfma4(av[3], wa[kc][3], c1v);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nn = 4 * g + r;
                    if (n0 + nn < N) a.A[(((size_t)b * Tdst + mg + wave) * N + n0 + nn) * a.c1 + l15] = c0[r] + c1v[r] + bal;
                }
            }
        }
        if (mg == 0) STGCN_PHASE(a.Wap ? 1 : 7, 6);
    }
    STGCN_PHASE(a.Wap ? 1 : 7, 7);
}


#endif  // STGCN_EXPERIMENTS

// ================================================================================================
// LayerNorm-backward row partials in the epilogue of the kernel that PRODUCES the gradient dy of a LayerNorm output
// (SURVEY.md 8a row a6): the slab constants c1 = mean(g), c2 = mean(g * xhat), g = dropout_mask * dy * gamma, need all N rows
// of a (b, t) slab, so the consumer (tc2_bwd_kernel / ln_gate_bwd_kernel) takes per-ROW sums; forming them where dy is still
// on chip saves the ln_bwd_rowstats_kernel launch and its re-read of dy.
// ================================================================================================
struct LnRowstatOut {
    float2* rowstat;      // [slabs*N] (sum g, sum g*xhat) per row; null = epilogue disabled
    // two ways to form g * xhat: from the LayerNorm's OUTPUT y = mask * (xhat * gamma + beta) (ST blocks: nothing of the forward has to be
    // kept for it, y is the consumer's own input) -- or from the saved gate inputs U, S with the slab statistics (the head's LayerNorm)
    const float* y;       // [slabs*N][C] or null
    const float* beta;    // [N][C] (with y)
    const float* U;       // [slabs*N][C] saved gate inputs of the layer in front of the LayerNorm (y == null)
    const float* S;
    const float* gamma;   // [N][C]
    const float* mean;    // [slabs]
    const float* rstd;
    int N, C, act, training;
    int mask_from_y;      // training, y given: an element was dropped iff y is -0.0 (drop_encode: the forward stores dropped elements as -0.0 and kept
                          // exact zeros as +0.0) instead of regenerating the Philox mask: ~100 VALU instructions per 4 elements less in the consumer's
                          // epilogue, which in the bf16 configurations is what its time steps wait for
    float keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
};
// contribution of 4 consecutive channels (c .. c+3) of row (slab, node): returns (sum g, sum g*xhat) of those 4
// position of the hooked LayerNorm's dropout stream for this launch, read ONCE per kernel: dereferencing offset_dev where the mask is formed
// put a dependent scalar load (and an s_waitcnt vmcnt(0) behind whatever had just been requested) into every row group / time step
__device__ __forceinline__ uint64_t ln_rowstat_offset(const LnRowstatOut& o) {
    return (o.rowstat && o.training) ? o.offset + (o.offset_dev ? *o.offset_dev : 0) : 0;
}
template <typename ET = float>
__device__ __forceinline__ float2 ln_rowstat4(const LnRowstatOut& o, uint64_t off, f32x4 dy, long slab, int node, int c) {
    const size_t e = ((size_t)slab * o.N + node) * o.C + c;
    const f32x4 ga = ld4(o.gamma + (size_t)node * o.C + c);
    f32x4 k = {1.f, 1.f, 1.f, 1.f};
    if (o.training && !(o.y && o.mask_from_y)) {
        k = dropout_scale4((uint64_t)slab * (((uint64_t)o.N * o.C) >> 2) + (((uint64_t)node * o.C + c) >> 2), o.seed, off, o.thresh, o.keep_scale);
    }
    float s1 = 0.f, s2 = 0.f;
    if (o.y) {   // (uniform) sum g = sum mask dy gamma ; sum g xhat = sum_kept dy (y - keep_scale * beta)
        const f32x4 y = ldx4(et_ptr<ET>(o.y) + e), be = ld4(o.beta + (size_t)node * o.C + c);
        const float ks = o.training ? o.keep_scale : 1.f;
        if (o.training && o.mask_from_y) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float yi = y[i];
                k[i] = drop_kept(yi) ? o.keep_scale : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s1 += dy[i] * k[i] * ga[i];
            if (k[i] > 0.f) s2 += dy[i] * (y[i] - ks * be[i]);
        }
        return make_float2(s1, s2);
    }
    const f32x4 u = ldx4(et_ptr<ET>(o.U) + e), s = ldx4(et_ptr<ET>(o.S) + e);
    const float mean = o.mean[slab], rstd = o.rstd[slab];
#pragma unroll
    for (int i = 0; i < 4; ++i) dy[i] *= k[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xh = (gate_fwd(u[i], s[i], o.act) - mean) * rstd;
        const float gg = dy[i] * ga[i];
        s1 += gg;
        s2 += gg * xh;
    }
    return make_float2(s1, s2);
}

// ================================================================================================
// F1 (v4, wide outputs / few rows: the output head): row tile of 16*TM rows x NC = 256 columns, 8 waves (wave w owns
// n-tiles w and w + 8), the whole im2col tile in LDS, and the weight fragments STREAMED in rounds of KC chunks through two
// register buffers: the loads of round r+1 are in flight while the MFMAs of round r run.  With 16 x 256 outputs per tile a
// workgroup moves 256 KB of weights through its CU's vector-memory path (~25 B/clk measured) for 8 K MFMA cycles: the head
// conv is bound by that stream and by the L2 round trip per K chunk of v1 (profiles/r19_r33_experiments.md v44), so the
// tile is 32 rows (half the weight bytes per row) and no MFMA ever waits for a load that was not requested a round ahead.
// PLAIN = false: gated epilogue (bias, sigmoid, GLU, U / S stores, LayerNorm row partials) = the head's forward conv.
// PLAIN = true : out[(b*outT + col / outC) * N + n][col % outC] = acc : the head's transposed conv, which for T1 = 1 is
//                ONE dense GEMM dZ[B*N x NC] @ B[NC x Ko*Cin] (weights packed by PK_TCONV_BWDT) instead of Ko masked taps.
// ================================================================================================
// PLAIN with lnb.dy set: the A tile is not read from memory but FORMED in the staging pass -- LayerNorm backward + gate backward of the
// head (the former ln_gate_bwd launch): dH = rstd * (dy * gamma - c1 - xhat * c2), [dU | dQ] = gate'(dH) with the slab constants c1, c2
// rebuilt from the row partials of the tile's one or two slabs; the tile also goes to dZ (the conv weight gradient reads it) and the
// per-element products dy * xhat to dgam[slab] (reduced over the batch by the final reduction; dbeta is reduced from dy itself).
struct LnGateBwdIn {
    const float* dy;      // [slabs*N][C]  gradient of the LayerNorm output; null = the tile is read through f.ts as usual
    const float* U;       // [slabs*N][C]  saved gate inputs
    const float* S;
    const float* gamma;   // [N][C]
    const float* mean;    // [slabs]
    const float* rstd;
    const float2* rowstat;   // [slabs*N]  (sum g, sum g * xhat) per row
    float* dZ;            // [slabs*N][2C]
    float* dgam;          // [slabs][N][C]
    float* dbet;          // [slabs][N][C] fp32 copy of dy for the dbeta reduction, or null: dbeta is reduced from the fp32 dy tensor itself
    int N, C, act;
};
struct Tconv4Args {
    TconvFwdArgs f;     // ts, Wp, bias, KCH, Cout (NC = 2*Cout), act, U, S, rowstat (gated) -- H, align fields unused
    float* out;         // PLAIN: destination tensor (B, outT, N, outC)
    int outT, outC;
    LnRowstatOut rs;    // PLAIN: LayerNorm-backward row partials of the layer whose output `out` is the gradient of (rs.rowstat != null)
    LnGateBwdIn lnb;    // PLAIN: see above
};

// HEADF (gated form only): the rest of the head's forward in the SAME launch -- LayerNorm([N, 128]) + fc1 + ReLU + dropout + fc2
// (layers.py:278-282) on the tile's gated rows while they are in LDS (the former fc_fwd launch: its re-read of U / S, its weight
// prologue and one launch ramp).  A LayerNorm slab (T1 = 1: the N rows of one window) spans several row tiles, so the tiles exchange
// their row statistics: write-through row partials -> arrival counter of the window -> wait for the window's other tiles (chain_*_peer,
// stgcn_device.hip.h; the launcher checks that the whole grid is resident) -> slab statistics from the N partials (sc1 loads).
struct HeadFcTail {
    ChainCtl chain;       // header + one arrival counter per window (T1 = 1: slab = window)
    const float* gamma;   // [N][128] LayerNorm weight / bias
    const float* beta;
    float* yln;           // [rows][128] LayerNorm output (the fc1 weight gradient reads it)
    float* mean;          // [windows]
    float* rstd;
    float eps;
    const float* W1p;     // packed PK_LIN_FWD: K = 128, cols = 128
    const float* b1;      // [128] or null
    const float* w2;      // [128]
    const float* b2;      // [1] or null
    float* hd;            // [rows][128] dropout(relu(fc1)), saved for backward
    float* out;           // [rows]
    int training;
    float keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
    int use_ticket;       // tiles by start order (chain_enter) instead of by blockIdx: grids beyond one resident round
};

template <int TM, int KC, bool PLAIN, typename ET, bool HEADF = false>
__device__ __forceinline__ void tconv_fwd4_body(const Tconv4Args& aa, const HeadFcTail* ht) {
    static_assert(!(PLAIN && HEADF), "the fused head tail follows the gated epilogue");
    constexpr int WAVES = 8, NT = 2, THREADS = 512, TR = TM * 16, NC = 256, COUT = 128, LDZ = NC + 4;
    const TconvFwdArgs& a = aa.f;
    ET* const U_ = et_ptr<ET>(a.U);
    ET* const S_ = et_ptr<ET>(a.S);
    ET* const H_ = et_ptr<ET>(a.H);
    ET* const out_ = et_ptr<ET>(aa.out);
    extern __shared__ float stgcn_smem[];
    float* At = stgcn_smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    long row0 = (long)xcd_item(blockIdx.x, gridDim.x) * TR;
    const int KCH = a.KCH, KP = KCH * 16, lda = KP + 4;

    f32x4 wA[KC][NT], wB[KC][NT];
    auto load_round = [&](f32x4 (&w)[KC][NT], int kc0) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                w[kc][j] = (kc0 + kc < KCH) ? ld4(a.Wp + ((size_t)((wave + WAVES * j) * KCH + kc0 + kc) * 64 + lane) * 4) : zero4();
    };
    load_round(wA, 0);
    if constexpr (HEADF) {
        // more tiles than the device holds at once: the tile comes from a TICKET (start order), so that a tile only ever waits for tiles that
        // run already or start as soon as earlier ones end (their peers lie at most two windows ahead: the launcher checks that span)
        if (ht->use_ticket) row0 = (long)chain_enter_peer(ht->chain, reinterpret_cast<unsigned*>(stgcn_smem + TR * LDZ + 4 * WAVES)) * TR;
    }
    bool staged = false;
    if constexpr (PLAIN) {
        if (aa.lnb.dy) {   // uniform
            staged = true;
            const LnGateBwdIn& b = aa.lnb;
            const int N = b.N, C = b.C, c4n = C >> 2;
            const float inv_n = 1.0f / ((float)N * (float)C);
            const long rows = a.ts.rows, rend = row0 + TR < rows ? row0 + TR : rows;
            float* red = At + TR * lda;   // 16 floats behind the tile
            for (long slab = row0 / N; slab * N < rend; ++slab) {
                float x = 0.f, y = 0.f;
                for (int r = tid; r < N; r += THREADS) {
                    const float2 v = b.rowstat[slab * N + r];
                    x += v.x;
                    y += v.y;
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    x += __shfl_xor(x, m);
                    y += __shfl_xor(y, m);
                }
                __syncthreads();   // red free (previous slab)
                if (lane == 0) {
                    red[wave] = x;
                    red[WAVES + wave] = y;
                }
                __syncthreads();
                float c1 = 0.f, c2 = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) {
                    c1 += red[w];
                    c2 += red[WAVES + w];
                }
                c1 *= inv_n;
                c2 *= inv_n;
                const float mean = b.mean[slab], rstd = b.rstd[slab];
                for (int idx = tid; idx < TR * c4n; idx += THREADS) {
                    const int row = idx / c4n, c4 = idx - row * c4n;
                    const long R = row0 + row;
                    if (R >= rows || R / N != slab) continue;
                    const int n = (int)(R - slab * N);
                    const size_t e = (size_t)R * C + 4 * c4;
                    const f32x4 dy = ldx4(et_ptr<ET>(b.dy) + e), u = ldx4(et_ptr<ET>(b.U) + e), sg = ldx4(et_ptr<ET>(b.S) + e),
                                ga = ld4(b.gamma + (size_t)n * C + 4 * c4);
                    f32x4 du, dq, dg;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xh = (gate_fwd(u[i], sg[i], b.act) - mean) * rstd;
                        const float dh = rstd * (dy[i] * ga[i] - c1 - xh * c2);
                        dg[i] = dy[i] * xh;
                        float du_, dq_;
                        gate_bwd(dh, u[i], sg[i], b.act, du_, dq_);
                        du[i] = du_;
                        dq[i] = dq_;
                    }
                    st4(At + row * lda + 4 * c4, du);
                    st4(At + row * lda + C + 4 * c4, dq);
                    stx4_wt(et_ptr<ET>(b.dZ) + (size_t)R * 2 * C + 4 * c4, du);
                    stx4_wt(et_ptr<ET>(b.dZ) + (size_t)R * 2 * C + C + 4 * c4, dq);
                    st4_wt(b.dgam + e, dg);
                    if (b.dbet) st4_wt(b.dbet + e, dy);
                }
            }
            for (int idx = tid; idx < TR * 2 * c4n; idx += THREADS) {   // rows past the end of the last tile
                const int row = idx / (2 * c4n), c4 = idx - row * 2 * c4n;
                if (row0 + row >= rows) st4(At + row * lda + 4 * c4, zero4());
            }
        }
    }
    if (!staged) stage_tile_fwd<TR, 4, THREADS, ET>(a.ts, row0, KP, At, lda);
    __syncthreads();

    f32x4 acc[TM][NT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4();
    const float* arow = At + l15 * lda + 4 * g;
    auto mma_round = [&](const f32x4 (&w)[KC][NT], int kc0) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            if (kc0 + kc < KCH) {
                f32x4 av[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = ld4(arow + i * 16 * lda + (kc0 + kc) * 16);
                mma_tile<ET, TM, NT>(acc, av, w[kc]);
            }
        }
    };
    for (int kc0 = 0; kc0 < KCH; kc0 += 2 * KC) {
        if (kc0 + KC < KCH) load_round(wB, kc0 + KC);
        mma_round(wA, kc0);
        if (kc0 + 2 * KC < KCH) load_round(wA, kc0 + 2 * KC);
        if (kc0 + KC < KCH) mma_round(wB, kc0 + KC);
    }

    // ---- epilogue through LDS: Zt[TR][NC + 4] -> row-major 16-byte stores ------------------------------------------
    float* Zt = At;
    __syncthreads();   // every wave is done reading At
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (wave + WAVES * j) * 16 + l15;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Zt[(i * 16 + 4 * g + r) * LDZ + col] = acc[i][j][r];
    }
    __syncthreads();
    if constexpr (PLAIN) {
        constexpr int Q4 = NC / 4, NIT = TR * Q4 / THREADS;   // float4 columns of the full row
        const int q = tid & (Q4 - 1);
        const int tap = (4 * q) / aa.outC, ci = 4 * q - tap * aa.outC;
        const RowCoord c0 = row_coord(a.ts, row0 < a.ts.rows ? row0 : 0);
        const uint64_t hoff = ln_rowstat_offset(aa.rs);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = (tid + it * THREADS) / Q4;
            const bool rin = row0 + row < a.ts.rows;
            const RowCoord c = row_advance(a.ts, c0, rin ? row : 0);     // (b, t = 0, n) of the dZ row
            const f32x4 v = et_round4<ET>(ld4(Zt + row * LDZ + 4 * q));   // (the hook below sees what the tensor holds)
            if (rin) stx4_wt(out_ + (((size_t)c.b * aa.outT + tap) * a.ts.N + c.n) * aa.outC + ci, v);
            if (aa.rs.rowstat) {   // uniform; the outC / 4 lanes holding one output row are consecutive and aligned
                const long slab = (long)c.b * aa.outT + tap;
                float2 p = rin ? ln_rowstat4<ET>(aa.rs, hoff, v, slab, c.n, ci) : make_float2(0.f, 0.f);
                for (int m = aa.outC >> 3; m >= 1; m >>= 1) {
                    p.x += __shfl_xor(p.x, m);
                    p.y += __shfl_xor(p.y, m);
                }
                if (rin && ci == 0) aa.rs.rowstat[slab * a.ts.N + c.n] = p;
            }
        }
    } else {
        constexpr int C4N = COUT / 4, NIT = TR * C4N / THREADS;
        const int c4 = tid & (C4N - 1);
        const f32x4 bp = ld4(a.bias + 4 * c4), bq = ld4(a.bias + COUT + 4 * c4);
        f32x4 uk[HEADF ? NIT : 1], sk[HEADF ? NIT : 1];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = (tid + it * THREADS) / C4N;
            const long R = row0 + row;
            const f32x4 p = ld4(Zt + row * LDZ + 4 * c4), q = ld4(Zt + row * LDZ + COUT + 4 * c4);
            f32x4 u, sg, h;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i] = p[i] + bp[i];
                sg[i] = sigmoid_f(q[i] + bq[i]);
                h[i] = gate_fwd(u[i], sg[i], a.act);
            }
            if constexpr (HEADF) {   // (stored behind the publish of the row partials: the peers wait for those 8 bytes per row only)
                uk[it] = u;
                sk[it] = sg;
            } else if (R < a.ts.rows) {
                const size_t o = (size_t)R * COUT + 4 * c4;
                if (a.U) stx4_wt(U_ + o, u);
                if (a.S) stx4_wt(S_ + o, sg);
                if (a.H) stx4_wt(H_ + o, h);
            }
            if constexpr (HEADF) {   // (in place: this thread alone reads p / q of (row, c4))
                // the LayerNorm acts on what the saved tensors hold (the backward rebuilds xhat from U / S): with bf16 storage, the gate of the ROUNDED inputs
                f32x4 hs = h;
                if constexpr (sizeof(ET) == 2) {
                    const f32x4 ur = et_round4<ET>(u), sr_ = et_round4<ET>(sg);
#pragma unroll
                    for (int i = 0; i < 4; ++i) hs[i] = gate_fwd(ur[i], sr_[i], a.act);
                }
                st4(Zt + row * LDZ + 4 * c4, hs);
            }
            if (a.rowstat) {   // per-row LayerNorm partials: the C4N lanes holding one row are contiguous in the wave
                float sr = (h[0] + h[1]) + (h[2] + h[3]);
#pragma unroll
                for (int m = C4N >> 1; m >= 1; m >>= 1) sr += __shfl_xor(sr, m);
                const float mr = sr / (float)COUT;
                float d2 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) d2 += (h[i] - mr) * (h[i] - mr);
#pragma unroll
                for (int m = C4N >> 1; m >= 1; m >>= 1) d2 += __shfl_xor(d2, m);
                if (c4 == 0 && R < a.ts.rows) {
                    if constexpr (HEADF) st2_wt(a.rowstat + R, make_float2(mr, d2));
                    else a.rowstat[R] = make_float2(mr, d2);
                }
            }
        }
        if constexpr (HEADF) {
            // ---- publish this tile's row partials, request what the tail needs, wait for the windows' other tiles -------------------
            const HeadFcTail& f = *ht;
            const long rows = a.ts.rows;
            const int N = a.ts.N;
            const long rend = row0 + TR < rows ? row0 + TR : rows;
            const int s_lo = (int)(row0 / N), s_hi = (int)((rend - 1) / N);   // the launcher admits N >= TR only: one or two windows
            chain_drain_stores();
            __syncthreads();
            if (tid == 0 && !chain_withhold(f.chain, row0))
                for (int sw = s_lo; sw <= s_hi; ++sw) chain_publish_peer(f.chain, sw);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const long R = row0 + (tid + it * THREADS) / C4N;
                if (R < rows) {
                    const size_t o = (size_t)R * COUT + 4 * c4;
                    if (a.U) stx4_wt(U_ + o, uk[it]);
                    if (a.S) stx4_wt(S_ + o, sk[it]);
                }
            }
            PreW<1, 8> w1;   // fc1 weight fragments of the whole K = 128: wave w owns output-channel tile w (the conv's weight rounds are done)
            pre_load_weights<1, 8>(w1, f.W1p, 8, wave, 8);
            f32x4 ga[NIT], be[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = (tid + it * THREADS) / C4N;
                const long R = row0 + row < rows ? row0 + row : rows - 1;
                const int n = (int)(R % N);
                ga[it] = ld4(f.gamma + (size_t)n * COUT + 4 * c4);
                be[it] = ld4(f.beta + (size_t)n * COUT + 4 * c4);
            }
            const f32x4 w2 = ld4(f.w2 + 4 * c4);
            f32x4 b1 = zero4();
            if (f.b1) b1 = ld4(f.b1 + 4 * c4);
            const float b2 = f.b2 ? f.b2[0] : 0.f;
            const uint64_t off = f.offset + (f.offset_dev ? *f.offset_dev : 0);
            float* const gaveup = Zt + TR * LDZ + 4 * WAVES + 1;   // (LDS word behind the reduction scratch and the ticket word)
            if (tid == 0) {
                bool ok = true;
                for (int sw = s_lo; sw <= s_hi; ++sw) {
                    const long w0 = (long)sw * N, w1r = w0 + N < rows ? w0 + N : rows;
                    ok &= chain_poll_peer(f.chain, sw, (unsigned)((w1r - 1) / TR - w0 / TR + 1));
                }
                // a wait that gave up (a peer tile never arrived: the device was not this launch's alone for seconds): the tile's predictions
                // become NaN -- the loss of the step says so -- instead of numbers normalised with incomplete statistics
                *gaveup = ok ? 0.f : __builtin_nanf("");
            }
            __syncthreads();
            const float poison = *gaveup;
            // ---- slab statistics of the one or two windows (as slab_stats_from_rows: sums about the first row's mean) ---------------
            float* red = Zt + TR * LDZ;   // 4 * WAVES floats behind the tile
            float mean2[2] = {0.f, 0.f}, rstd2[2] = {1.f, 1.f};
            {
                const bool two = s_hi > s_lo;   // (uniform)
                const float2* rs0 = a.rowstat + (size_t)s_lo * N;
                const float2* rs1 = a.rowstat + (size_t)s_hi * N;
                const float xa = ld2_sc1(rs0, N, 0).x, xb = two ? ld2_sc1(rs1, N, 0).x : 0.f;
                float sa1 = 0.f, sa2 = 0.f, sb1 = 0.f, sb2 = 0.f;
                for (int r = tid; r < N; r += THREADS) {   // both windows' partials requested together
                    const float2 va = ld2_sc1(rs0, N, r);
                    float2 vb = make_float2(0.f, 0.f);
                    if (two) vb = ld2_sc1(rs1, N, r);
                    const float da = va.x - xa, db = vb.x - xb;
                    sa1 += da;
                    sa2 += va.y + (float)COUT * da * da;
                    sb1 += db;
                    sb2 += vb.y + (float)COUT * db * db;
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    sa1 += __shfl_xor(sa1, m);
                    sa2 += __shfl_xor(sa2, m);
                    sb1 += __shfl_xor(sb1, m);
                    sb2 += __shfl_xor(sb2, m);
                }
                if (lane == 0) {
                    red[wave] = sa1;
                    red[WAVES + wave] = sa2;
                    red[2 * WAVES + wave] = sb1;
                    red[3 * WAVES + wave] = sb2;
                }
                __syncthreads();
                sa1 = sa2 = sb1 = sb2 = 0.f;
#pragma unroll
                for (int wv = 0; wv < WAVES; ++wv) {
                    sa1 += red[wv];
                    sa2 += red[WAVES + wv];
                    sb1 += red[2 * WAVES + wv];
                    sb2 += red[3 * WAVES + wv];
                }
                const float nf = (float)N, nc = (float)N * (float)COUT;
                const float dma = sa1 / nf, dmb = sb1 / nf;
                mean2[0] = xa + dma;
                mean2[1] = xb + dmb;
                rstd2[0] = 1.0f / sqrtf(fmaxf(sa2 - nc * dma * dma, 0.f) / nc + f.eps);
                rstd2[1] = 1.0f / sqrtf(fmaxf(sb2 - nc * dmb * dmb, 0.f) / nc + f.eps);
                if (tid == 0) {   // the tile that holds a window's first row keeps its statistics for the backward
                    if ((long)s_lo * N >= row0) {
                        f.mean[s_lo] = mean2[0];
                        f.rstd[s_lo] = rstd2[0];
                    }
                    if (two) {    // (a second window always starts inside the tile)
                        f.mean[s_hi] = mean2[1];
                        f.rstd[s_hi] = rstd2[1];
                    }
                }
            }
            // ---- LayerNorm in place (columns 0 .. 127 of the tile), yln to memory ----------------------------------------------------
            ET* const yln_ = et_ptr<ET>(f.yln);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = (tid + it * THREADS) / C4N;
                const long R = row0 + row;
                f32x4 o = zero4();
                if (R < rows) {
                    const int wi = (int)(R / N) - s_lo;
                    const float mean = wi ? mean2[1] : mean2[0], rstd = wi ? rstd2[1] : rstd2[0];
                    const f32x4 h = ld4(Zt + row * LDZ + 4 * c4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (h[i] - mean) * rstd * ga[it][i] + be[it][i];
                    stx4_wt(yln_ + (size_t)R * COUT + 4 * c4, o);
                }
                st4(Zt + row * LDZ + 4 * c4, o);
            }
            __syncthreads();
            // ---- fc1 on the matrix cores: [TR x 128] @ [128 x 128], result to columns 128 .. 255 of the tile --------------------------
            f32x4 acc1[TM][1];
#pragma unroll
            for (int i = 0; i < TM; ++i) acc1[i][0] = zero4();
            pre_mma<TM, 1, 8, ET>(acc1, Zt, LDZ, 0, 8, w1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Zt[(i * 16 + 4 * g + r) * LDZ + COUT + wave * 16 + l15] = acc1[i][0][r];   // (nobody reads these columns any more)
            __syncthreads();
            // ---- bias, ReLU, dropout, fc2 (a row's 32 lanes are half a wave) ---------------------------------------------------------------
            ET* const hd_ = et_ptr<ET>(f.hd);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = (tid + it * THREADS) / C4N;
                const long R = row0 + row;
                f32x4 h = ld4(Zt + row * LDZ + COUT + 4 * c4);
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = fmaxf(h[i] + b1[i], 0.f);
                if (f.training) {
                    const f32x4 k = dropout_scale4((uint64_t)R * C4N + c4, f.seed, off, f.thresh, f.keep_scale);
#pragma unroll
                    for (int i = 0; i < 4; ++i) h[i] *= k[i];
                }
                float pd = h[0] * w2[0] + h[1] * w2[1] + h[2] * w2[2] + h[3] * w2[3];
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) pd += __shfl_xor(pd, m);
                if (R < rows) {
                    stx4_wt(hd_ + (size_t)R * COUT + 4 * c4, h);
                    if (c4 == 0) f.out[R] = pd + b2 + poison;
                }
            }
            chain_exit(f.chain);
        }
    }
}
template <int TM, int KC, bool PLAIN, typename ET>
__global__ __launch_bounds__(512) void tconv_fwd4_kernel(Tconv4Args aa) {
    tconv_fwd4_body<TM, KC, PLAIN, ET, false>(aa, nullptr);
}
template <int TM, int KC, typename ET>
__global__ __launch_bounds__(512) void head_fwd_kernel(Tconv4Args aa, HeadFcTail ht) {
    tconv_fwd4_body<TM, KC, false, ET, true>(aa, &ht);
}

// ================================================================================================
// F2: graph convolution on one (b, t) slab  X0 = A[slab] (N x 16)
//     X_k = T_k(L) X0, k = 1 .. terms-1                (layers.py:147-161; Kipf: X1 = L X0, layers.py:198)
//     Y  = sum_k Xk Wk + bias                          (layers.py:165-168 / :199-202)
//     G  = relu(Y + X0)                                (layers.py:229, 253)
// One workgroup per slab.  X0 lives transposed in LDS (XT[c][node]) so that it is the MFMA A operand (rows = 16
// channels, k = nodes); the fragments of the precomputed polynomials T_k come straight from L2 (16-B loads, requested
// two chunks ahead); all terms share ONE pass over X0: per k-chunk one LDS read feeds the MFMAs of two terms.
// D = X_k^T tile[c][h] leaves each lane with 4 consecutive channels of one node, which is at once the store layout
// and the A operand of the 16x16 weight contraction.  No barrier after the staging of X0.
// The wave count is a launch parameter (blockDim.x / 64 <= MAXW): one wave per node tile up to 16 tiles (MAXQ = 1, e.g.
// 13 waves for the 207-node graph: measured faster than 8 waves x 2 tiles), 8 waves x MAXQ tiles beyond.
// ================================================================================================
struct GconvFwdArgs {
    // chained launch (ChainCtl of the launch, stgcn_device.hip.h): counter index bases, -1 = not chained on that side
    int chain_in;        // counter chain_in + slab counts the node tiles of A[slab] that are complete (chain_expect of them)
    unsigned chain_expect;
    int chain_out;       // counter chain_out + slab: bumped once per part when this part's rows of G[slab] are written (through)
    const float* A;      // [slabs][N][16]
    const float* Lp;     // fragment-packed T_1 .. T_{Ks-1} (stgcn_gso_prepare), NP*NP floats each
    const float* W;      // cheb: [Ks][16][16] ; kipf: [16][16]
    const float* bias;   // [16] or null
    float* Xk;           // [Ks-1][slabs][N][16]   (X1..X_{Ks-1}, saved for backward; nullable)
    float* G;            // [slabs][N][16]
    int N, NP, Ks, kipf; // Ks = number of terms (kipf: 2)
    int parts;           // workgroups per slab: part p owns node tiles p, p + parts, ... (grid = slabs * parts)
    long slabs;
    float* XT;           // tiled path only: two bf16 operand-form buffers (plan: ws_XT), used when g_gc_precision > 0
};

// SP = (b, t) slabs per workgroup (2: every operator fragment a wave loads multiplies the X chunks of two slabs; opt-in, STGCN_GC_SP=2:
// measured equal / slower, the loop was never bound by the volume of that stream).
inline size_t gconv_fwd_lds_bytes(int NP, int sp, int waves, bool chain_out) {   // X0 transposed (+ one 16 x 20 transposition tile per wave for written-through G rows)
    return ((size_t)sp * 16 * (NP + 4) + (chain_out ? (size_t)waves * 16 * 20 : 0)) * sizeof(float);
}
// B16P (bf16 activations only): the operator products on v_mfma_f32_16x16x32_bf16 from the operator's bf16 fragment PLANE (stgcn_gso_prepare
// writes it behind the fp32 fragments for stgcn_kernels_gcslab16.hip.h: F[((ht * KC32 + kc) * 64 + lane) * 8 + j] =
// bf16(T_k[ht*16 + (lane & 15)][kc*32 + 8*(lane >> 4) + j])) and a bf16 copy of X0^T in LDS: half the operator bytes per product, half the
// matrix instructions, no conversion of the fragments in the loop.  The values are those of the 16-deep bf16 form (bf16(T_k) is what
// Mma<bf16>::cvt makes of the fp32 fragment); only the accumulation order inside a 32-deep instruction differs.
__host__ __device__ inline int gc_np32(int N) { return (N + 31) / 32 * 32; }
inline size_t gconv_fwd_b16p_lds_bytes(int NP, int N) { return (size_t)16 * (NP + 4) * 4 + (size_t)16 * (gc_np32(N) + 8) * 2; }
__host__ __device__ inline size_t gc_plane_floats(int NP, int N) { return (size_t)(NP / 16) * (gc_np32(N) / 32) * 64 * 4; }   // one plane of one term (gs16_term_floats = two)
// bid = (slab group, part) index of this workgroup, THREADS = threads of the role (the calling waves: threadIdx.x < THREADS)
template <int MAXQ, int MAXW, typename ET, int SP, bool B16P = false>
__device__ __forceinline__ void gconv_fwd_body(const GconvFwdArgs& a, const int bid, const int THREADS, const ChainCtl& chain) {
    static_assert(!B16P || (sizeof(ET) == 2 && SP == 1), "the bf16-plane products are the bf16-activation form, one slab per workgroup");
    typedef Mma<ET> MM;
    extern __shared__ float stgcn_smem[];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int P = a.parts, part = (int)((unsigned)bid % (unsigned)P);
    const long slab0 = (long)((unsigned)bid / (unsigned)P) * SP;   // slabs slab0 .. slab0 + SP - 1 (the last group may be short)
    const bool cin = SP == 1 && chain.words && a.chain_in >= 0, cout = SP == 1 && chain.words && a.chain_out >= 0;
    if (cin) chain_wait(chain, a.chain_in + (int)slab0, a.chain_expect);   // every node tile of A[slab0] has been written (through) by its producer
    // node tile of (wave w, slot q) = part + P * (w + nwaves * q) = wave + WAVES * q with the two names below
    const int wave = part + P * __builtin_amdgcn_readfirstlane(tid >> 6), WAVES = P * (THREADS >> 6);   // (scalar: the tile tests below are branches, not exec masks)
    const int N = a.N, NP = a.NP, LDX = NP + 4, HT = NP >> 4, KCH = NP >> 4;
    const size_t MSZ = (size_t)NP * NP;
    float* const XT0 = stgcn_smem;   // X0 transposed: [SP][16][LDX]
    const int NP32 = gc_np32(N), KC32 = NP32 >> 5, LDB = NP32 + 8;               // (B16P)
    unsigned short* const XH = reinterpret_cast<unsigned short*>(XT0 + SP * 16 * LDX);   // (B16P) bf16 X0^T [16][LDB], rows 16-byte aligned

    STGCN_PHASE(4, 0);
    ET* const Xk_ = et_ptr<ET>(a.Xk);
    ET* const G_ = et_ptr<ET>(a.G);
    bool live[SP];
#pragma unroll
    for (int j = 0; j < SP; ++j) {
        live[j] = slab0 + j < a.slabs;
        const ET* Asl = et_ptr<ET>(a.A) + (size_t)(live[j] ? slab0 + j : slab0) * N * 16;
        // (four requests in flight per thread, raw, addresses clamped: one load -> four LDS stores per trip waited for every load in turn)
        for (int idx0 = tid; idx0 < NP * 4; idx0 += 4 * THREADS) {
            Raw4<ET> rw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * THREADS, n = idx >> 2, c4 = idx & 3;
                const int eo = (n < N ? n : N - 1) * 16 + c4 * 4;
                rw[u] = cin ? ldraw4_sc1(Asl, (long)N * 16, eo) : ldraw4(Asl + eo);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + u * THREADS, n = idx >> 2, c4 = idx & 3;
                if (idx < NP * 4) {
                    const f32x4 v = n < N ? cvt4(rw[u]) : zero4();
#pragma unroll
                    for (int i = 0; i < 4; ++i) XT0[(j * 16 + c4 * 4 + i) * LDX + n] = v[i];
                    if constexpr (B16P) {   // (the values ARE bf16 numbers: the upper halves of their fp32 form)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float vi = v[i];   // (a copy: __builtin_bit_cast of the vector ELEMENT expression reads element 0 on the host compiler)
                            XH[(c4 * 4 + i) * LDB + n] = (unsigned short)(__builtin_bit_cast(unsigned, vi) >> 16);
                        }
                    }
                }
            }
        }
    }
    if constexpr (B16P) {   // zero the plane's padding columns NP .. NP32 + 7 (they meet zero operator columns, but 0 * NaN is NaN)
        for (int idx = tid; idx < 16 * (LDB - NP); idx += THREADS) XH[(idx / (LDB - NP)) * LDB + NP + idx % (LDB - NP)] = 0;
    }
    __syncthreads();
    STGCN_PHASE(4, 1);

    f32x4 yacc[MAXQ][SP], res[MAXQ][SP];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int ht = wave + WAVES * q;
#pragma unroll
        for (int j = 0; j < SP; ++j) {
            yacc[q][j] = zero4();
            // residual X0[h = ht*16 + 4g + r][j = l15]  (D layout of the weight contraction)
            res[q][j] = ht < HT ? ld4(XT0 + (j * 16 + l15) * LDX + ht * 16 + 4 * g) : zero4();
        }
    }
    // weight fragment B[kk = c][col = j] = W_k[c = 4g + s][j = l15]
    auto wfrag = [&](int k) {
        f32x4 wf = zero4();
        if (!(a.kipf && k == 0)) {
            const float* Wk = a.W + (a.kipf ? 0 : (size_t)k * 256);
#pragma unroll
            for (int s = 0; s < 4; ++s) wf[s] = Wk[(4 * g + s) * 16 + l15];
        }
        return wf;
    };
    {   // term 0: X0 W0
        const typename MM::frag wf = MM::cvt(wfrag(0));
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int ht = wave + WAVES * q;
            if (ht < HT) {
                const int h = ht * 16 + l15;
#pragma unroll
                for (int j = 0; j < SP; ++j) yacc[q][j] = MM::mma(MM::cvt(gather4(XT0 + (j * 16 + 4 * g) * LDX + h, LDX)), wf, yacc[q][j]);
            }
        }
    }
    // terms k0, k0+1 together: acc1 = X0^T-tile products with T_k0, acc2 with T_{k0+1}.
    // Operator fragments: a ring of RG chunks in registers with STATIC indices, and a loop body without branches.  The former
    // p <- n <- load rotation went through register moves behind an s_waitcnt vmcnt(0), and the tile / term tests inside the loop were
    // branches at which the compiler's wait-count bookkeeping starts over: either way every iteration waited for the load it had just
    // issued (2.5 k cycles per chunk for 256 cycles of MFMAs; r3-25: 15 of the 23 us of the C2 block-0 launch).  Hence TWO (both terms of
    // the pair exist) and NQ (tiles this wave owns: slots q < NQ) are compile-time constants of the body, chosen once per wave.
    int nq = 0;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) nq += wave + WAVES * q < HT ? 1 : 0;
    auto term_pair = [&](int k0, auto two_tag, auto nq_tag) __attribute__((always_inline)) {
        constexpr bool TWO = decltype(two_tag)::value;
        constexpr int NQ = decltype(nq_tag)::value;
        constexpr int RG = MAXQ == 2 ? 4 : gc_ring(MAXQ);
        // B16P: the hi plane of term k0 (each term: hi plane, lo plane) behind the fp32 fragments of all terms
        const size_t PSZ = gc_plane_floats(NP, N);
        const float* T1 = B16P ? a.Lp + (size_t)(a.Ks - 1) * MSZ + (size_t)(k0 - 1) * 2 * PSZ : a.Lp + (size_t)(k0 - 1) * MSZ;
        const float* T2 = T1 + (B16P ? 2 * PSZ : MSZ);
        const int NCH = B16P ? KC32 : KCH;             // chunks of the k loop (32 or 16 nodes each)
        const typename MM::frag wf1 = MM::cvt(wfrag(k0)), wf2 = MM::cvt(TWO ? wfrag(k0 + 1) : zero4());
        STGCN_PHASE(4, 2 * k0);
        f32x4 acc1[NQ][SP], acc2[NQ][SP], r1[RG][NQ], r2[RG][NQ];
        int fo[NQ];      // this lane's float offset into the fragments of tile q, chunk 0 (< 2^31: at most 7 terms of 512 x 512)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                acc1[q][j] = zero4();
                acc2[q][j] = zero4();
            }
            fo[q] = ((wave + WAVES * q) * NCH * 64 + lane) * 4;
#pragma unroll
            for (int d = 0; d < RG; ++d) {
                const int dc = d < NCH ? d : NCH - 1;   // (graphs of fewer than RG chunks: a valid address, never used)
                r1[d][q] = ld4(T1 + fo[q] + 256 * dc);
                if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * dc);
            }
        }
        auto chunk = [&](int kc, int d, auto load_tag) __attribute__((always_inline)) {   // d = kc % RG, as a constant after unrolling
          if constexpr (B16P) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, ld4(reinterpret_cast<const float*>(XH + l15 * LDB + kc * 32 + 8 * g)));   // A[c = l15][node = kc*32 + 8g + j]
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                acc1[q][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, r1[d][q]), acc1[q][0], 0, 0, 0);
                if (TWO) acc2[q][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(bf16x8, r2[d][q]), acc2[q][0], 0, 0, 0);
                if (decltype(load_tag)::value) {   // this slot's next occupant
                    r1[d][q] = ld4(T1 + fo[q] + 256 * (kc + RG));
                    if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * (kc + RG));
                }
            }
          } else {
            typename MM::frag af[SP];
#pragma unroll
            for (int j = 0; j < SP; ++j) af[j] = MM::cvt(ld4(XT0 + (j * 16 + l15) * LDX + kc * 16 + 4 * g));   // A[c = l15][node = kc*16 + 4g + s]
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const typename MM::frag bf1 = MM::cvt(r1[d][q]), bf2 = MM::cvt(TWO ? r2[d][q] : zero4());
#pragma unroll
                for (int j = 0; j < SP; ++j) {
                    if (TWO) MM::mma_b2(af[j], bf1, bf2, acc1[q][j], acc2[q][j]);
                    else acc1[q][j] = MM::mma(af[j], bf1, acc1[q][j]);
                }
                if (decltype(load_tag)::value) {   // this slot's next occupant
                    r1[d][q] = ld4(T1 + fo[q] + 256 * (kc + RG));
                    if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * (kc + RG));
                }
            }
          }
            __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the refills of all RG slots to the end of the unrolled body)
        };
        int kc0 = 0;
        for (; kc0 + 2 * RG <= NCH; kc0 += RG) {   // steady state: RG chunks, each refills its slot
#pragma unroll
            for (int d = 0; d < RG; ++d) chunk(kc0 + d, d, std::true_type());
        }
        for (; kc0 < NCH; kc0 += RG) {             // last chunks: refill only while there is something left to fetch
#pragma unroll
            for (int d = 0; d < RG; ++d) {
                if (kc0 + d < NCH) {
                    if (kc0 + d + RG < NCH) chunk(kc0 + d, d, std::true_type());
                    else chunk(kc0 + d, d, std::false_type());
                }
            }
        }
        STGCN_PHASE(4, 2 * k0 + 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ht = wave + WAVES * q;
            const int h = ht * 16 + l15;   // acc[r] = X_k[h][c = 4g + r]
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                if (a.Xk && h < N && live[j]) {
                    stx4_wt(Xk_ + (((size_t)(k0 - 1) * a.slabs + slab0 + j) * N + h) * 16 + 4 * g, acc1[q][j]);
                    if (TWO) stx4_wt(Xk_ + (((size_t)k0 * a.slabs + slab0 + j) * N + h) * 16 + 4 * g, acc2[q][j]);
                }
                if (TWO) MM::mma_ab2(MM::cvt(acc1[q][j]), wf1, MM::cvt(acc2[q][j]), wf2, yacc[q][j]);
                else yacc[q][j] = MM::mma(MM::cvt(acc1[q][j]), wf1, yacc[q][j]);
            }
        }
    };
    auto term_pair_nq = [&](int k0, auto two_tag) __attribute__((always_inline)) {
        if (nq == 1) term_pair(k0, two_tag, std::integral_constant<int, 1>());
        if constexpr (MAXQ >= 2) { if (nq == 2) term_pair(k0, two_tag, std::integral_constant<int, 2>()); }
        if constexpr (MAXQ >= 3) { if (nq == 3) term_pair(k0, two_tag, std::integral_constant<int, 3>()); }
        if constexpr (MAXQ >= 4) { if (nq == 4) term_pair(k0, two_tag, std::integral_constant<int, 4>()); }
    };
    for (int k0 = 1; k0 < a.Ks; k0 += 2) {
        if (k0 + 1 < a.Ks) term_pair_nq(k0, std::true_type());
        else term_pair_nq(k0, std::false_type());
    }

    const float bb = a.bias ? a.bias[l15] : 0.f;
    if (cout) {
        // hand-off form: the tile goes through a wave-private LDS tile so that a lane writes 4 channels of one node (16 bytes, write-through);
        // the storing waves drain, the workgroup meets, ONE lane bumps the slab's counter
        float* const tw = stgcn_smem + 16 * LDX + (tid >> 6) * (16 * 20);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int ht = wave + WAVES * q;
            if (ht < HT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tw[(4 * g + r) * 20 + l15] = fmaxf(yacc[q][0][r] + bb + res[q][0][r], 0.f);
                wave_lds_sync();
                const int h = ht * 16 + (lane >> 2);
                const f32x4 v = ld4(tw + (lane >> 2) * 20 + 4 * (lane & 3));
                if (h < N) stx4_wt(G_ + ((size_t)slab0 * N + h) * 16 + 4 * (lane & 3), v);
                wave_lds_sync();
            }
        }
        chain_drain_stores();
        __syncthreads();
        if (tid == 0) chain_publish(chain, a.chain_out + (int)slab0);
        STGCN_PHASE(4, 15);
        return;
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int ht = wave + WAVES * q;
        if (ht < HT) {
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                if (!live[j]) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int h = ht * 16 + 4 * g + r;
                    if (h < N) stx1(G_ + ((size_t)(slab0 + j) * N + h) * 16 + l15, fmaxf(yacc[q][j][r] + bb + res[q][j][r], 0.f));
                }
            }
        }
    }
    STGCN_PHASE(4, 15);
}
template <int MAXQ, int MAXW, typename ET, int SP>
__global__ __launch_bounds__(MAXW * 64) void gconv_fwd_kernel(GconvFwdArgs a) {
    gconv_fwd_body<MAXQ, MAXW, ET, SP>(a, (int)blockIdx.x, (int)blockDim.x, ChainCtl{nullptr, 0, 0u});
}
template <int MAXQ, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void gconv_fwd_b16p_kernel(GconvFwdArgs a) {
    gconv_fwd_body<MAXQ, MAXW, bf16, 1, true>(a, (int)blockIdx.x, (int)blockDim.x, ChainCtl{nullptr, 0, 0u});
}

#ifdef STGCN_EXPERIMENTS   // operator-stationary graph conv (opt-in, STGCN_GC_REG=<workgroups per CU>)
// ================================================================================================
// F2, persistent operator-stationary form (round 6): ONE workgroup per compute unit walks several (b, t) slabs.
// The slab-per-workgroup kernel above runs its 1280 / 768 workgroups of C2 as exactly one resident round: all five workgroups of a CU
// stage X0 together (matrix pipe idle), multiply together and store together -- 19.5 us for 7.2 us of matrix time.  Here
//   * wave w of workgroup (set, part) owns node tile part + 4 w for the whole launch: the fragments of T_1 .. T_{Ks-1} of THAT tile
//     (KCH chunks of 1 KiB per term: 26 KiB for the 207-node graph at Ks = 3) are copied ONCE into a wave-private LDS region and
//     every product reads them from there (the operator crosses L2 -> CU once per workgroup, not once per slab);
//   * the workgroup walks slabs set, set + S, set + 2 S, ..: X0 of the NEXT slab is requested into registers before the products of the
//     current one and written to the other LDS buffer behind them -- one barrier per slab, and it does NOT drain vmcnt (barrier_only):
//     round 2's register-stationary experiment (gconv_fwd_reg_kernel below) put __syncthreads() right behind its prefetch, i.e. waited for
//     the load it had just issued and for the previous slab's write-through stores in every iteration (4.4 us per slab for 1.8 us of MFMAs);
//   * the epilogue is the slab kernel's (X_k tiles are at once store layout and A operand of the 16 x 16 weight contraction).
// grid = 4 * S workgroups of 256 threads (S slab sets, parts = 4: up to 16 node tiles = 256 nodes), LDS = 4 waves x (Ks - 1) x KCH KiB + two
// transposed X0 buffers (C2: 133 KB: one workgroup per CU).  Same arithmetic, same summation order per element as gconv_fwd_kernel.
// MEASURED SLOWER (pass r6-09, profiles/r6-09_gconv_fwd_persistent.txt: C2 24.7 + 18.3 us against 21.6 + 14.9 for the slab kernel, results equal,
// all stage tests green): with ONE wave per SIMD the ~300 VALU / SALU instructions a wave spends per slab outside its 116 MFMAs (X0 commit,
// fetch addresses, epilogue stores) issue at one per ~7.5 cycles (tools/ubench/overlap.hip) and nothing fills the gaps -- 6.5 k cycles per
// slab for 3.7 k of matrix time -- while the slab kernel's five co-resident workgroups per CU give every SIMD five waves to interleave.  A second
// wave group per workgroup does not fit (104 KB of fragments + 4 X0 buffers = the whole LDS) and five slabs per set cut 2 + 2 + 1.  Opt-in
// (-DSTGCN_EXPERIMENTS, STGCN_GC_PERS=1), like round 2's register-stationary form below.
// ================================================================================================
inline size_t gconv_fwd_pers_lds_bytes(int NP, int terms) { return ((size_t)4 * (terms - 1) * (NP / 16) * 256 + (size_t)2 * 16 * (NP + 4)) * sizeof(float); }
template <typename ET>
__global__ __launch_bounds__(256) void gconv_fwd_pers_kernel(GconvFwdArgs a, int S) {
    typedef Mma<ET> MM;
    extern __shared__ float stgcn_smem[];
    constexpr int THREADS = 256, NV = 4;   // float4 per thread and slab (NP * 4 <= 1024)
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = (int)(blockIdx.x & 3u), set = (int)(blockIdx.x >> 2);
    const int N = a.N, NP = a.NP, LDX = NP + 4, HT = NP >> 4, KCH = NP >> 4, NT = a.Ks - 1;   // NT operator terms (1 or 2)
    const size_t MSZ = (size_t)NP * NP;
    const int ht = part + 4 * w;           // this wave's node tile
    const bool own = ht < HT;              // (wave-uniform)
    float* const Fr = stgcn_smem + (size_t)w * NT * KCH * 256;      // [NT][KCH][64 lanes][4]: this wave's operator fragments
    float* const XTb = stgcn_smem + (size_t)4 * NT * KCH * 256;     // [2][16][LDX]: X0 transposed, two buffers
    ET* const Xk_ = et_ptr<ET>(a.Xk);
    ET* const G_ = et_ptr<ET>(a.G);
    const ET* const A_ = et_ptr<ET>(a.A);

    // ---- X0 slabs: registers (one slab ahead) -> LDS, transposed [c][node] ------------------------------------------------------
    Raw4<ET> xv[NV];
    auto fetch = [&](long slab) __attribute__((always_inline)) {   // (unconditional, clamped: a branch around a load resets the compiler's wait counts)
        const ET* Asl = A_ + (size_t)(slab < a.slabs ? slab : a.slabs - 1) * N * 16;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS, n = idx >> 2, c4 = idx & 3;
            xv[i] = ldraw4(Asl + (size_t)(n < N ? n : N - 1) * 16 + c4 * 4);
        }
    };
    auto commit = [&](float* XT) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS, n = idx >> 2, c4 = idx & 3;
            if (idx < NP * 4) {
                const f32x4 v = n < N ? cvt4(xv[i]) : zero4();
#pragma unroll
                for (int j = 0; j < 4; ++j) XT[(c4 * 4 + j) * LDX + n] = v[j];
            }
        }
    };
    fetch(set);
    // ---- this wave's operator fragments -> its LDS region (lane-linear 16-byte stores: conflict free), in batches of 8 loads ---------
    if (own) {
        for (int c0 = 0; c0 < NT * KCH; c0 += 8) {
            f32x4 fv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u < NT * KCH ? c0 + u : NT * KCH - 1, k = c / KCH, kc = c - k * KCH;
                fv[u] = ld4(a.Lp + (size_t)k * MSZ + ((size_t)(ht * KCH + kc) * 64 + lane) * 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c0 + u < NT * KCH) st4(Fr + (size_t)(c0 + u) * 256 + lane * 4, fv[u]);
        }
    }
    // weight fragments B[kk = c][col = j] = W_k[c = 4g + s][j = l15] of the (up to) three terms, bias
    typename MM::frag wf[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        f32x4 v = zero4();
        if (k <= NT && !(a.kipf && k == 0)) {
            const float* Wk = a.W + (a.kipf ? 0 : (size_t)k * 256);
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = Wk[(4 * g + s) * 16 + l15];
        }
        wf[k] = MM::cvt(v);
    }
    const float bb = a.bias ? a.bias[l15] : 0.f;
    commit(XTb);
    barrier_only();

    int it = 0;
    for (long slab = set; slab < a.slabs; slab += S, ++it) {
        const float* const XT0 = XTb + (it & 1) * 16 * LDX;
        fetch(slab + S);                   // the next slab of this set (clamped beyond the end: never committed)
        if (own) {
            const int h = ht * 16 + l15;
            const f32x4 res = ld4(XT0 + l15 * LDX + ht * 16 + 4 * g);   // residual X0[h = ht*16 + 4g + r][j = l15] (D layout of the weight contraction)
            f32x4 yacc = MM::mma(MM::cvt(gather4(XT0 + (4 * g) * LDX + h, LDX)), wf[0], zero4());   // term 0: X0 W0
            f32x4 acc1 = zero4(), acc2 = zero4();
            if (NT == 2) {
#pragma unroll 4
                for (int kc = 0; kc < KCH; ++kc) {
                    const typename MM::frag af = MM::cvt(ld4(XT0 + l15 * LDX + kc * 16 + 4 * g));   // A[c = l15][node = kc*16 + 4g + s]
                    MM::mma_b2(af, MM::cvt(ld4(Fr + (size_t)kc * 256 + lane * 4)), MM::cvt(ld4(Fr + (size_t)(KCH + kc) * 256 + lane * 4)), acc1, acc2);
                }
            } else {
#pragma unroll 4
                for (int kc = 0; kc < KCH; ++kc)
                    acc1 = MM::mma(MM::cvt(ld4(XT0 + l15 * LDX + kc * 16 + 4 * g)), MM::cvt(ld4(Fr + (size_t)kc * 256 + lane * 4)), acc1);
            }
            // acc[r] = X_k[h][c = 4g + r]: store layout and A operand of the weight contraction
            if (a.Xk && h < N) {
                stx4_wt(Xk_ + ((size_t)slab * N + h) * 16 + 4 * g, acc1);
                if (NT == 2) stx4_wt(Xk_ + (((size_t)a.slabs + slab) * N + h) * 16 + 4 * g, acc2);
            }
            if (NT == 2) MM::mma_ab2(MM::cvt(acc1), wf[1], MM::cvt(acc2), wf[2], yacc);
            else yacc = MM::mma(MM::cvt(acc1), wf[1], yacc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = ht * 16 + 4 * g + r;
                if (hh < N) stx1(G_ + ((size_t)slab * N + hh) * 16 + l15, fmaxf(yacc[r] + bb + res[r], 0.f));
            }
        }
        if (slab + S < a.slabs) commit(XTb + ((it & 1) ^ 1) * 16 * LDX);   // (uniform) the other buffer: its last readers are behind the previous barrier
        barrier_only();
    }
}

// ================================================================================================
// F2 (operator-stationary variant): the fragments of T_1 .. T_{Ks-1} a wave needs for ITS node tile (KCH chunks per term,
// 1 KiB each: 26 KiB for the 207-node graph, Ks = 3) are loaded into registers ONCE and the workgroup then walks `spw`
// slabs: per MFMA of the slab-per-workgroup kernel a wave pulls 256 B of operator through its CU's vector-memory path
// (~25 B/clk measured, i.e. the four SIMDs together are fed at 78 % of what their MFMAs consume, and every slab re-reads
// the whole 340 KB operator from L2: 110 MB per launch at C2).  Here the operator crosses the L1 once per workgroup
// (256 workgroups x 104 KiB), the per-slab traffic is the 13 KiB X0 slab (prefetched into registers one slab ahead,
// double-buffered in LDS) and the MFMAs are fed from registers (B) and LDS (A).
// grid = parts * groups; workgroup (part, grp) owns node tiles part + parts * wave and slabs grp*spw .. +spw.
// Template: KCM >= KCH chunks per term held in registers, NTERM = Ks - 1 (1 or 2) operator terms.
// ================================================================================================
template <int KCM, int NTERM>
__global__ __launch_bounds__(256) void gconv_fwd_reg_kernel(GconvFwdArgs a, int spw) {
    extern __shared__ float stgcn_smem[];
    constexpr int THREADS = 256;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int P = a.parts, part = (int)(blockIdx.x % (unsigned)P);
    const long grp = blockIdx.x / (unsigned)P;
    const int N = a.N, NP = a.NP, LDX = NP + 4, HT = NP >> 4, KCH = NP >> 4;
    const size_t MSZ = (size_t)NP * NP;
    const int ht = part + P * (tid >> 6);   // this wave's node tile
    const bool own = ht < HT;
    const long s0 = grp * spw;
    long s1 = s0 + spw;
    if (s1 > a.slabs) s1 = a.slabs;

    // ---- operator fragments of this wave's tile -> registers, weight fragments, bias -------------------------------
    f32x4 Tr[NTERM][KCM];
#pragma unroll
    for (int k = 0; k < NTERM; ++k)
#pragma unroll
        for (int kc = 0; kc < KCM; ++kc)
            Tr[k][kc] = (own && kc < KCH) ? ld4(a.Lp + (size_t)k * MSZ + ((size_t)(ht * KCH + kc) * 64 + lane) * 4) : zero4();
    f32x4 wf[NTERM + 1];   // B[kk = c][col = j] = W_k[c = 4g + s][j = l15]
#pragma unroll
    for (int k = 0; k <= NTERM; ++k) {
        wf[k] = zero4();
        if (!(a.kipf && k == 0)) {
            const float* Wk = a.W + (a.kipf ? 0 : (size_t)k * 256);
#pragma unroll
            for (int s = 0; s < 4; ++s) wf[k][s] = Wk[(4 * g + s) * 16 + l15];
        }
    }
    const float bb = a.bias ? a.bias[l15] : 0.f;

    // ---- X0 slabs: registers (one slab ahead) -> LDS (two buffers), transposed [c][node] -------------------------------
    constexpr int NV = 4;   // float4 per thread per slab (NP * 4 <= 1024, i.e. N <= 256)
    f32x4 xv[NV];
    auto fetch = [&](long slab) {
        const float* Asl = a.A + (size_t)slab * N * 16;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS, n = idx >> 2, c4 = idx & 3;
            xv[i] = (idx < NP * 4 && n < N) ? ld4(Asl + (size_t)n * 16 + c4 * 4) : zero4();
        }
    };
    auto commit = [&](float* XT) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * THREADS, n = idx >> 2, c4 = idx & 3;
            if (idx < NP * 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) XT[(c4 * 4 + j) * LDX + n] = xv[i][j];
            }
        }
    };
    if (s0 < s1) {
        fetch(s0);
        commit(stgcn_smem);
    }
    for (long slab = s0; slab < s1; ++slab) {
        float* const XT0 = stgcn_smem + ((slab - s0) & 1) * 16 * LDX;
        float* const XTn = stgcn_smem + (((slab - s0) & 1) ^ 1) * 16 * LDX;
        if (slab + 1 < s1) fetch(slab + 1);
        __syncthreads();   // XT0 complete; every wave is past its reads of XTn (previous slab)
        if (own) {
            const int h = ht * 16 + l15;
            // residual X0[h = ht*16 + 4g + r][j = l15]  (D layout of the weight contraction)
            const f32x4 res = ld4(XT0 + l15 * LDX + ht * 16 + 4 * g);
            f32x4 yacc = zero4();
#pragma unroll
            for (int s = 0; s < 4; ++s) yacc = mfma4(XT0[(4 * g + s) * LDX + h], wf[0][s], yacc);   // term 0: X0 W0
            f32x4 acc[NTERM];
#pragma unroll
            for (int k = 0; k < NTERM; ++k) acc[k] = zero4();
#pragma unroll
            for (int kc = 0; kc < KCM; ++kc) {
                if (kc < KCH) {
                    const f32x4 af = ld4(XT0 + l15 * LDX + kc * 16 + 4 * g);   // A[c = l15][node = kc*16 + 4g + s]
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int k = 0; k < NTERM; ++k) acc[k] = mfma4(af[s], Tr[k][kc][s], acc[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < NTERM; ++k) {   // acc[k][r] = X_{k+1}[h][c = 4g + r]
                if (a.Xk && h < N) st4_wt(a.Xk + (((size_t)k * a.slabs + slab) * N + h) * 16 + 4 * g, acc[k]);
#pragma unroll
                for (int s = 0; s < 4; ++s) yacc = mfma4(acc[k][s], wf[k + 1][s], yacc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = ht * 16 + 4 * g + r;
                if (hh < N) a.G[((size_t)slab * N + hh) * 16 + l15] = fmaxf(yacc[r] + bb + res[r], 0.f);
            }
        }
        if (slab + 1 < s1) commit(XTn);
    }
}

#endif  // STGCN_EXPERIMENTS

// ================================================================================================
// F4: LayerNorm over the joint [N, C] axes of each (b, t) slab (biased variance, eps 1e-12,
//     layers.py:246/255) of H = act(U) * S, followed by inverted dropout (layers.py:256).
//     Slab statistics come from the per-row partials (mean_r, M2_r) the conv epilogue wrote:
//         mu = mean_r averaged over rows ;  M2 = sum_r M2_r + C * sum_r (mean_r - mu)^2   (exact pairwise merge)
//     so the normalisation itself is a fully parallel streaming kernel: grid = (chunks, slabs).
// ================================================================================================
struct LnFwdArgs {
    const float* U;      // [slabs][n]   n = N*C
    const float* S;
    const float* gamma;  // [n]
    const float* beta;
    const float2* rowstat;   // [slabs*N]
    float* y;            // [slabs][n]
    float* mean;         // [slabs]
    float* rstd;
    int n, N, C, act, training, per;   // per = float4 columns per chunk
    int stats_ready;     // mean / rstd already hold this launch's statistics (ln_slab_stats_kernel ran: slabs of many chunks)
    float eps, keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;   // optional device-side step counter added to `offset` (graph replay safe)
};

// mean / rstd of one slab from its row partials; all 256 threads must call; red: >= 8 floats of LDS
__device__ __forceinline__ void slab_stats_from_rows(const float2* rs, int N, int C, float eps, float* red, float& mean, float& rstd) {
    // ONE pass and one block reduction: sums about the first row's mean x0 (every row mean is within O(sigma) of the slab mean, so the
    // correction term below cancels nothing that matters):  mu = x0 + S1 / N ,  M2 = S2 - N C (mu - x0)^2  with
    // S1 = sum_r (x_r - x0) ,  S2 = sum_r M2_r + C (x_r - x0)^2.  (The two-pass form read the partials twice with a dependent second pass.)
    // Cancellation in M2 is bounded: (mu - x0)^2 <= N * Var(row means) <= N * sigma^2, so the relative error of M2 is at most ~N * 2^-24
    // (207 nodes: 1e-5, 8192 nodes: 5e-4 in the worst case of ONE outlying first row) against the 1e-3 bar on rstd.
    const float x0 = rs[0].x;
    float s1 = 0.f, s2 = 0.f;
    int r0 = 0;
    if ((reinterpret_cast<uintptr_t>(rs) & 15) == 0) {   // two rows per 16-byte load, four loads in flight (slabs of thousands of rows: C5)
        const f32x4* rs4 = reinterpret_cast<const f32x4*>(rs);
        const int N2 = N >> 1;
        float t1 = 0.f, t2 = 0.f;
        int q = threadIdx.x;
        for (; q + 3 * kThreads < N2; q += 4 * kThreads) {
            const f32x4 a = rs4[q], b = rs4[q + kThreads], c = rs4[q + 2 * kThreads], e = rs4[q + 3 * kThreads];
            const float da0 = a[0] - x0, da1 = a[2] - x0, db0 = b[0] - x0, db1 = b[2] - x0;
            const float dc0 = c[0] - x0, dc1 = c[2] - x0, de0 = e[0] - x0, de1 = e[2] - x0;
            s1 += (da0 + da1) + (db0 + db1);
            t1 += (dc0 + dc1) + (de0 + de1);
            s2 += (a[1] + a[3]) + (b[1] + b[3]) + (float)C * ((da0 * da0 + da1 * da1) + (db0 * db0 + db1 * db1));
            t2 += (c[1] + c[3]) + (e[1] + e[3]) + (float)C * ((dc0 * dc0 + dc1 * dc1) + (de0 * de0 + de1 * de1));
        }
        for (; q < N2; q += kThreads) {
            const f32x4 a = rs4[q];
            const float d0 = a[0] - x0, d1 = a[2] - x0;
            s1 += d0 + d1;
            s2 += (a[1] + a[3]) + (float)C * (d0 * d0 + d1 * d1);
        }
        s1 += t1;
        s2 += t2;
        r0 = N2 * 2;
    }
    for (int r = r0 + threadIdx.x; r < N; r += kThreads) {
        const float2 v = rs[r];
        const float d = v.x - x0;
        s1 += d;
        s2 += v.y + (float)C * d * d;
    }
    block_sum2(s1, s2, red);
    const float dm = s1 / (float)N;
    mean = x0 + dm;
    const float m2 = s2 - (float)N * (float)C * dm * dm;
    rstd = 1.0f / sqrtf(fmaxf(m2, 0.f) / ((float)N * (float)C) + eps);
}

// slab statistics once per slab (grid = slabs) for slabs of many chunks: every workgroup of ln_norm_kernel otherwise re-derives them from
// all N row partials -- at 8192 nodes 64 KiB of reads and two block reductions in front of a 4-quads-per-thread payload (C5: 253 us)
__global__ __launch_bounds__(256) void ln_slab_stats_kernel(LnFwdArgs a) {
    extern __shared__ float stgcn_smem[];
    const long slab = blockIdx.x;
    float mean, rstd;
    slab_stats_from_rows(a.rowstat + (size_t)slab * a.N, a.N, a.C, a.eps, stgcn_smem, mean, rstd);
    if (threadIdx.x == 0) {
        a.mean[slab] = mean;
        a.rstd[slab] = rstd;
    }
}

template <typename ET>
__global__ __launch_bounds__(256) void ln_norm_kernel(LnFwdArgs a) {
    extern __shared__ float stgcn_smem[];
    const long slab = blockIdx.y;
    const int chunk = blockIdx.x, tid = threadIdx.x, n4 = a.n >> 2;
    float mean, rstd;
    if (a.stats_ready) {   // (uniform)
        mean = a.mean[slab];
        rstd = a.rstd[slab];
    } else {
        slab_stats_from_rows(a.rowstat + (size_t)slab * a.N, a.N, a.C, a.eps, stgcn_smem, mean, rstd);
        if (chunk == 0 && tid == 0) {
            a.mean[slab] = mean;
            a.rstd[slab] = rstd;
        }
    }
    const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
    const ET* U = et_ptr<ET>(a.U) + (size_t)slab * a.n;
    const ET* S = et_ptr<ET>(a.S) + (size_t)slab * a.n;
    ET* y = et_ptr<ET>(a.y) + (size_t)slab * a.n;
    int q1 = (chunk + 1) * a.per;
    if (q1 > n4) q1 = n4;
#pragma unroll 2
    for (int q = chunk * a.per + tid; q < q1; q += kThreads) {
        const f32x4 u = ldx4(U + 4 * q), s = ldx4(S + 4 * q), ga = ld4(a.gamma + 4 * q), be = ld4(a.beta + 4 * q);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (gate_fwd(u[i], s[i], a.act) - mean) * rstd * ga[i] + be[i];
        if (a.training) {
            const f32x4 k = dropout_scale4((uint64_t)slab * n4 + q, a.seed, off, a.thresh, a.keep_scale);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = drop_encode(o[i], k[i], k[i] > 0.f);   // dropped: -0.0, kept zero: +0.0
        }
        stx4(y + 4 * q, o);
    }
}

// keep-scale mask exactly as ln_norm_kernel draws it (test / debugging aid; also used by the oracle
// comparison in training mode): out[e] in {0, 1/(1-p)}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* out, long n4, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                                                           uint32_t thresh, float keep_scale) {
    const long q = (long)blockIdx.x * kThreads + threadIdx.x;
    if (q >= n4) return;
    st4(out + 4 * q, dropout_scale4((uint64_t)q, seed, offset + (offset_dev ? *offset_dev : 0), thresh, keep_scale));
}

}  // namespace stgcn
