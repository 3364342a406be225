// Slab-resident graph convolution with the operator products on the bf16 matrix cores ("bf16x3", gfx950).
//
// Phase stamps of gconv_fwd_kernel at C2 (profiles/r2u_phases_gconv.txt): 17.5 k of a workgroup's 28 k cycles are the k loop of the
// T_k(L) X0 products, and that loop runs at the CU's fp32-MFMA rate (five co-resident workgroups x 4 waves x 104 MFMAs x 32 cycles on
// four matrix pipes).  v_mfma_f32_16x16x4_f32 delivers 1/16 of the bf16 rate, so the products are formed as in the tiled path
// (stgcn_kernels_gctile.hip.h, "bf16x3"): both operands split x = hi + lo into two bf16 and
//     T X ~= Th Xh + Th Xl + Tl Xh          (3 x v_mfma_f32_16x16x32_bf16, fp32 accumulation)
// the dropped Tl Xl term and the split residuals are ~2^-17 relative per product, at 3/16 of the MFMA cycles.  NOT the default: on the
// METR-LA operator with unit-variance inputs the block output moves by up to 1.2e-4 abs against the exact-fp32 kernels (X_2 = T_2(L) X0
// carries ~1e-5 per element into tmp_conv2 and the LayerNorm) -- outside the 1e-4 parity bar of the fp32 configurations
// (tests/test_emu_gcslab16.py, tests/test_gpu_gcslab16.py).  Only the forward kernel exists in this form (measured: -2.8 us per launch).
// The 16x16 weight contractions, the residual, bias and ReLU stay in fp32 exactly as in gconv_fwd_kernel / gconv_bwd_kernel.
//
// Operator storage (stgcn_gso_prepare, behind the fp32 fragments of every term): per term k two planes (hi, lo) in the A/B fragment
// order of the 32-deep MFMA, NP32 = roundup(N, 32), KC32 = NP32 / 32, HT = NP / 16:
//     F[((ht * KC32 + kc) * 64 + lane) * 8 + j] = bf16( T_k[ht*16 + (lane & 15)][kc*32 + 8*(lane >> 4) + j] )      j = 0 .. 7
// so a wave fetches one (node tile, k chunk) fragment of one plane with one coalesced 1 KiB load, like the fp32 fragments.
#pragma once
#include "stgcn_kernels_bwd.hip.h"
#include "stgcn_kernels_fwd.hip.h"
#include "stgcn_kernels_gctile.hip.h"

namespace stgcn {

inline int gs16_np32(int N) { return (N + 31) / 32 * 32; }
// floats one term's two planes occupy (HT x KC32 fragments of 64 lanes x 16 bytes, twice)
inline size_t gs16_term_floats(int NP, int N) { return (size_t)(NP / 16) * (gs16_np32(N) / 32) * 64 * 4 * 2; }

// dense padded D (NP x NP) -> bf16 hi / lo fragment planes of D (Tf) and of D^T (TTf)
__global__ __launch_bounds__(256) void gso_frag16_kernel(const float* D, int NP, int KC32, float* Tf, float* TTf) {
    const long e = (long)blockIdx.x * kThreads + threadIdx.x;      // one bf16 element of one plane
    const long per = (long)(NP >> 4) * KC32 * 512;
    if (e >= per) return;
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long rest = e >> 9;
    const int kc = (int)(rest % KC32), ht = (int)(rest / KC32);
    const int h = ht * 16 + (lane & 15), i = kc * 32 + 8 * (lane >> 4) + j;
    const float v = i < NP ? D[(size_t)h * NP + i] : 0.f, vt = i < NP ? D[(size_t)i * NP + h] : 0.f;
    unsigned short* f = reinterpret_cast<unsigned short*>(Tf);
    unsigned short* ft = reinterpret_cast<unsigned short*>(TTf);
    const unsigned hh = bf16_rne(v), ht_ = bf16_rne(vt);
    f[e] = (unsigned short)hh;
    f[per + e] = (unsigned short)bf16_rne(v - bf16_to_f32(hh));
    ft[e] = (unsigned short)ht_;
    ft[per + e] = (unsigned short)bf16_rne(vt - bf16_to_f32(ht_));
}

__device__ __forceinline__ bf16x8 ld_bf8(const void* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ f32x4 mfma_bf(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// LDS (bytes): fp32 X0^T [16][NP + 4] | bf16 planes Xh, Xl [16][LDB] each, LDB = NP32 + 8 bf16 (rows 16-byte aligned, the 16 rows of a
// fragment read land on 16 different 4-bank groups)
inline size_t gconv_fwd16_lds_bytes(int NP, int N) { return (size_t)16 * (NP + 4) * 4 + (size_t)2 * 16 * (gs16_np32(N) + 8) * 2; }

// ================================================================================================
// F2 (bf16x3): gconv_fwd_kernel with the T_k X0 products on v_mfma_f32_16x16x32_bf16.  Same work split (one workgroup per slab and
// part, one wave per node tile and slot), same outputs (X_k saved for backward, G = relu(sum_k X_k W_k + b + X0)).
// a.Lp points at the fp32 fragments; the bf16 planes of term k follow them at Lp + (Ks - 1) * NP * NP + (k - 1) * gs16_term_floats.
// ================================================================================================
template <int MAXQ, int MAXW>
__global__ __launch_bounds__(MAXW * 64) void gconv_fwd16_kernel(GconvFwdArgs a) {
    extern __shared__ float stgcn_smem[];
    const int THREADS = blockDim.x, tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int P = a.parts, part = (int)(blockIdx.x % (unsigned)P);
    const long slab = blockIdx.x / (unsigned)P;
    const int wave = part + P * (tid >> 6), WAVES = P * (THREADS >> 6);
    const int N = a.N, NP = a.NP, LDX = NP + 4, HT = NP >> 4, NP32 = (N + 31) / 32 * 32, KC32 = NP32 >> 5, LDB = NP32 + 8;
    float* const XT0 = stgcn_smem;                                                // X0 transposed, fp32: [16][LDX]
    unsigned short* const XH = reinterpret_cast<unsigned short*>(XT0 + 16 * LDX); // bf16 hi plane [16][LDB]
    unsigned short* const XL = XH + 16 * LDB;                                     // bf16 lo plane

    // ---- stage X0: a thread takes two neighbouring nodes x 4 channels (two 16-byte loads, all requested before the first LDS store) ----
    const float* Asl = a.A + (size_t)slab * N * 16;
    constexpr int kStageIt = 4;                       // pairs per thread: NP32 / 2 * 4 <= kStageIt * THREADS (checked by the launcher)
    f32x4 v0[kStageIt], v1[kStageIt];
#pragma unroll
    for (int it = 0; it < kStageIt; ++it) {
        const int idx = tid + it * THREADS, n = (idx >> 2) * 2, c4 = idx & 3;
        v0[it] = (idx < NP32 * 2 && n < N) ? ld4(Asl + (size_t)n * 16 + c4 * 4) : zero4();
        v1[it] = (idx < NP32 * 2 && n + 1 < N) ? ld4(Asl + (size_t)(n + 1) * 16 + c4 * 4) : zero4();
    }
#pragma unroll
    for (int it = 0; it < kStageIt; ++it) {
        const int idx = tid + it * THREADS, n = (idx >> 2) * 2, c4 = idx & 3;
        if (idx < NP32 * 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = c4 * 4 + i;
                const float x0 = v0[it][i], x1 = v1[it][i];
                if (n < NP) XT0[c * LDX + n] = x0;
                if (n + 1 < NP) XT0[c * LDX + n + 1] = x1;
                const unsigned h0 = bf16_rne(x0), h1 = bf16_rne(x1);
                const unsigned l0 = bf16_rne(x0 - bf16_to_f32(h0)), l1 = bf16_rne(x1 - bf16_to_f32(h1));
                *reinterpret_cast<unsigned*>(XH + c * LDB + n) = h0 | (h1 << 16);
                *reinterpret_cast<unsigned*>(XL + c * LDB + n) = l0 | (l1 << 16);
            }
        }
    }
    __syncthreads();

    f32x4 yacc[MAXQ], res[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        yacc[q] = zero4();
        const int ht = wave + WAVES * q;
        res[q] = ht < HT ? ld4(XT0 + l15 * LDX + ht * 16 + 4 * g) : zero4();   // residual X0[h = ht*16 + 4g + r][j = l15]
    }
    auto wfrag = [&](int k) {   // B[kk = c][col = j] = W_k[c = 4g + s][j = l15]
        f32x4 wf = zero4();
        if (!(a.kipf && k == 0)) {
            const float* Wk = a.W + (a.kipf ? 0 : (size_t)k * 256);
#pragma unroll
            for (int s = 0; s < 4; ++s) wf[s] = Wk[(4 * g + s) * 16 + l15];
        }
        return wf;
    };
    {   // term 0: X0 W0 (fp32)
        const f32x4 wf = wfrag(0);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int ht = wave + WAVES * q;
            if (ht < HT) {
                const int h = ht * 16 + l15;
#pragma unroll
                for (int s = 0; s < 4; ++s) yacc[q] = mfma4(XT0[(4 * g + s) * LDX + h], wf[s], yacc[q]);
            }
        }
    }
    const size_t TSZ = (size_t)HT * KC32 * 64 * 4;            // floats per plane
    const float* const Lb = a.Lp + (size_t)(a.Ks - 1) * NP * NP;
    // terms k0, k0 + 1 in one pass over the k chunks: acc1 = T_k0 X0, acc2 = T_{k0+1} X0 (as D^T tiles [c][h])
    for (int k0 = 1; k0 < a.Ks; k0 += 2) {
        const bool two = k0 + 1 < a.Ks;
        const float* T1h = Lb + (size_t)(k0 - 1) * 2 * TSZ;
        const float* T1l = T1h + TSZ;
        const float* T2h = T1l + TSZ;
        const float* T2l = T2h + TSZ;
        const f32x4 wf1 = wfrag(k0), wf2 = two ? wfrag(k0 + 1) : zero4();
        f32x4 acc1[MAXQ], acc2[MAXQ];
        // operator fragments of the current chunk (c) and one chunk ahead (n): 4 x 16 B per slot in flight behind the MFMAs of a chunk
        f32x4 c1h[MAXQ], c1l[MAXQ], c2h[MAXQ], c2l[MAXQ], n1h[MAXQ], n1l[MAXQ], n2h[MAXQ], n2l[MAXQ];
        auto fetch = [&](int kc, f32x4 (&h1)[MAXQ], f32x4 (&l1)[MAXQ], f32x4 (&h2)[MAXQ], f32x4 (&l2)[MAXQ]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int ht = wave + WAVES * q;
                const size_t o = ((size_t)(ht * KC32 + kc) * 64 + lane) * 4;
                const bool in = ht < HT && kc < KC32;
                h1[q] = in ? ld4(T1h + o) : zero4();
                l1[q] = in ? ld4(T1l + o) : zero4();
                h2[q] = (in && two) ? ld4(T2h + o) : zero4();
                l2[q] = (in && two) ? ld4(T2l + o) : zero4();
            }
        };
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            acc1[q] = zero4();
            acc2[q] = zero4();
        }
        fetch(0, c1h, c1l, c2h, c2l);
        fetch(1, n1h, n1l, n2h, n2l);
        for (int kc = 0; kc < KC32; ++kc) {
            const bf16x8 ah = ld_bf8(XH + l15 * LDB + kc * 32 + 8 * g), al = ld_bf8(XL + l15 * LDB + kc * 32 + 8 * g);   // A[c = l15][node = kc*32 + 8g + j]
            bf16x8 b1h[MAXQ], b1l[MAXQ], b2h[MAXQ], b2l[MAXQ];
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                b1h[q] = __builtin_bit_cast(bf16x8, c1h[q]); b1l[q] = __builtin_bit_cast(bf16x8, c1l[q]);
                b2h[q] = __builtin_bit_cast(bf16x8, c2h[q]); b2l[q] = __builtin_bit_cast(bf16x8, c2l[q]);
                c1h[q] = n1h[q]; c1l[q] = n1l[q]; c2h[q] = n2h[q]; c2l[q] = n2l[q];
            }
            fetch(kc + 2, n1h, n1l, n2h, n2l);
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                if (wave + WAVES * q < HT) {   // small terms first (fp32 accumulation order: lo products, then hi * hi)
                    acc1[q] = mfma_bf(al, b1h[q], acc1[q]);
                    if (two) acc2[q] = mfma_bf(al, b2h[q], acc2[q]);
                    acc1[q] = mfma_bf(ah, b1l[q], acc1[q]);
                    if (two) acc2[q] = mfma_bf(ah, b2l[q], acc2[q]);
                    acc1[q] = mfma_bf(ah, b1h[q], acc1[q]);
                    if (two) acc2[q] = mfma_bf(ah, b2h[q], acc2[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int ht = wave + WAVES * q;
            if (ht < HT) {
                const int h = ht * 16 + l15;   // acc[r] = X_k[h][c = 4g + r]
                if (a.Xk && h < N) {
                    st4_wt(a.Xk + (((size_t)(k0 - 1) * a.slabs + slab) * N + h) * 16 + 4 * g, acc1[q]);
                    if (two) st4_wt(a.Xk + (((size_t)k0 * a.slabs + slab) * N + h) * 16 + 4 * g, acc2[q]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    yacc[q] = mfma4(acc1[q][s], wf1[s], yacc[q]);
                    if (two) yacc[q] = mfma4(acc2[q][s], wf2[s], yacc[q]);
                }
            }
        }
    }

    const float bb = a.bias ? a.bias[l15] : 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int ht = wave + WAVES * q;
        if (ht < HT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int h = ht * 16 + 4 * g + r;
                if (h < N) a.G[((size_t)slab * N + h) * 16 + l15] = fmaxf(yacc[q][r] + bb + res[q][r], 0.f);
            }
        }
    }
}

}  // namespace stgcn
