// "Time-stepping" kernels of the fused ST-Conv block (gfx950): one workgroup owns 16 consecutive nodes of ONE window b for
// ALL time steps and walks the time axis.  A temporal convolution only mixes the time steps of one node, so everything that
// hangs on a temporal conv in backward -- LayerNorm + dropout + gate backward, the weight gradient, the transposed conv --
// can be done on one 16-row tile per step with the Kt most recent tiles kept in an LDS ring: the gate-gradient tensor dZ
// (the largest tensor of the block, rows x 2c) is never written to memory, every saved activation is read exactly once, and
// three launches (ln_gate_bwd, tconv_bwd_weight, tconv_bwd_data) become one.
//
// MFMA use (v_mfma_f32_16x16x4_f32, conventions of stgcn_device.hip.h):
//   weight gradient  dW[(tap, i)][o] += sum_rows X[t + tap][row][i] dZ[t][row][o]   A = X^T tile (k = the 16 rows), B = dZ tile
//   transposed conv  dX^T[i][row] = sum_tap sum_o W_eff[(tap, i)][o] dZ[t - tap][row][o]
//                    A = the WEIGHTS, held in registers for the whole kernel ("weights stationary"), B = dZ tiles from LDS;
//                    D leaves a lane with 4 consecutive channels of one row = a 16-byte store.
// Reference arithmetic: hazdzz/STGCN model/layers.py:87-120 (TemporalConvLayer), :246/255-256 (LayerNorm, Dropout).
#pragma once
#include <type_traits>
#include "stgcn_device.hip.h"
#include "stgcn_kernels_fwd.hip.h"

namespace stgcn {

// ================================================================================================
// K1: backward of  tmp_conv2 -> LayerNorm([N, c2]) -> Dropout  (layers.py:254-256) for one (window, 16-node tile):
//     per output step t2:  dy -> (dropout mask, LN backward with the slab constants c1, c2) -> dH -> gate backward -> dZ2 tile
//     dW_eff2 += im2col(G)^T dZ2, db_eff2 += sum dZ2, dgamma += dy_m * xhat, dbeta += dy_m   (per-workgroup partials)
//     dYg[t1] = relu'(G[t1]) * sum_tap dZ2[t1 - tap] W_eff2[tap]^T                           (the gradient entering the graph conv)
// grid = B * ceil(N / 16) workgroups of 256 threads; requires c1 == 16 (tmp_conv2 input channels).
// The per-slab constants c1 = mean(g), c2 = mean(g * xhat) come from the per-row partials of ln_bwd_rowstats_kernel (or of the
// kernel that produced dy) -- they need all N nodes of a slab, which no node tile sees.
// ================================================================================================
struct Tc2BwdArgs {
    const float* y;           // [B][T2][N][C2]  the block's output (training: an element was dropped iff y is -0.0, drop_encode -- no Philox in the time step) or null
    const float* dy;          // [B][T2][N][C2]
    const float* U;           // [B][T2][N][C2]  saved gate inputs of tmp_conv2 (RECOMP = false)
    const float* S;
    const float* Wp;          // packed W_eff2 (PK_TCONV_FWD fragments; RECOMP = true: the gate inputs are RECOMPUTED from the G tiles, not read)
    const float* bias;        // b_eff2 [2*C2]
    const float* gamma;       // [N][C2]
    const float* mean;        // [B*T2]
    const float* rstd;
    const float2* rowstat;    // [B*T2*N]  (sum g, sum g*xhat) per row
    const float2* slabconst;  // [B*T2] (c1, c2) or null: rebuilt from rowstat by every workgroup
    const float* G;           // [B][T1][N][16]  tmp_conv2 input (relu output, also the relu mask)
    const float* Wd;          // [Kt*16][2*C2]   dense W_eff of tmp_conv2 (PK_TCONV_DENSE)
    float* dYg;               // [B][T1][N][16]
    float* part;              // [wgs][Kt*16*NC + NC]  dW_eff2 (transposed: [NC][Kt*16]) | db_eff2 partials
    float* dgam_part;         // [B][N*C2]
    float* dbet_part;
    int B, T1, T2, N, act, training, node_tiles;
    float keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
};

constexpr int kTsMaxT = 32;   // time steps of G kept in LDS (host falls back to the unfused kernels beyond)
inline size_t tc2_bwd_lds_bytes(int C2, int Kt, int T1, int T2, bool recomp = true) {
    return ((size_t)(Kt + 1) * 16 * (2 * C2 + 4) + (size_t)T1 * 16 * 20 + 2 * 4 * 16 * 20 + 4 * (size_t)T2 + (recomp ? 2 * 16 * (2 * C2 + 4) : 0)) * sizeof(float);
}

// Wave specialisation: a workgroup is 8 waves = 4 "E" waves + 4 "M" waves, one of each per SIMD.
//     E waves (VALU / memory): E(t) = dy tile (prefetched two steps ahead) + the gate inputs of tile t from LDS -> dZ2 tile t into ring
//                              slot t % (KT + 1), plus the LayerNorm-parameter and bias partials;
//     M waves (matrix cores) : F(t - 1) = dYg[t - 1] from the 4 partial tiles of the previous step;
//                              M(t) = weight-gradient MFMAs of tile t + transposed-conv MFMAs for output step t -> `red[t & 1]`;
//                              R(t + 2) = the gate inputs U2 = P + b, S2 = sigmoid(Q + b) of tile t + 2, RECOMPUTED from the G tiles the
//                              workgroup holds anyway (K = KT * 16: KT product steps per 16 output channels) -> `USt[t & 1]` (RECOMP).
//                              The forward then stores neither U2 nor S2 (2 x rows2 x c2 elements written once and read once per block).
//   iteration t:  barrier | E waves: E(t + 1)  ||  M waves: F(t - 1), M(t), R(t + 2)
// RECOMP costs KT more product steps per 16 output channels and step: + 50 % matrix work in this kernel.  Measured on MI355X
// (profiles/r3-02_*): with fp32 products the CUs that hold two workgroups become MFMA-bound (C2: 27.3 -> 34.0 us, 18.2 -> 21.6 us per
// launch, more than tc2_ln_fwd and the hooks gain), with bf16 products the step gains 5 % (C3: 0.790 -> 0.753 ms).  The host therefore
// recomputes for bf16 activations and reads the stored gate inputs for fp32 (STGCN_TC2_RECOMP=0/1 overrides).
// With one wave of each kind on a SIMD the E wave's memory / LDS latencies are covered by the M wave's instructions and vice versa, which a single
// wave walking E then M cannot do (phase stamps of the one-role version: 3.7 k cycles per step for 1.5 k cycles of MFMAs).  (Round 6, tools/ubench/overlap.hip:
// the fp32 MFMA and the VALU of a SIMD do NOT execute side by side -- the gain is latency hiding, the SIMD's time is the sum of both streams.)  The ring has a
// spare slot so that E(t + 1) never overwrites a tile M(t) still reads; ONE barrier per step.
// (Round 5, pass r5-02: the C2 instance sat at exactly 128 VGPRs -- two workgroups per CU, which the launch geometry counts on -- and an
//  unrelated edit moved it to 129: one workgroup per CU, two rounds, 26.0 + 18.0 -> 31.3 + 22.4 us.  The instances the stated configurations
//  run in fp32 now ASK for four waves per SIMD; the others keep the compiler's choice, they would spill.)
template <int C2, int KT, bool TRAINING, int ACT, bool RECOMP, typename ET>
__global__ __launch_bounds__(512, (C2 == 64 && KT == 3 && ACT == 0 && !RECOMP) ? 4 : 1) void tc2_bwd_kernel(Tc2BwdArgs a) {
    typedef Mma<ET> MM;
    const ET* const dy_ = et_ptr<ET>(a.dy);
    const ET* const ym_ = et_ptr<ET>(a.y ? a.y : a.dy);   // (no y: a valid address of the same shape, the value is not used)
    const bool mask_y = TRAINING && a.y != nullptr;      // uniform
    const ET* const U_ = et_ptr<ET>(a.U);
    const ET* const S_ = et_ptr<ET>(a.S);
    const ET* const G_ = et_ptr<ET>(a.G);
    ET* const dYg_ = et_ptr<ET>(a.dYg);
    constexpr int NC = 2 * C2, LDZ = NC + 4, NTW = NC / 64, QW = NC / 64, IT = C2 / 64, LDG = 20, RING = KT + 1, RED = 4 * 16 * LDG;
    extern __shared__ float stgcn_smem[];
    float* const Zt = stgcn_smem;                      // [KT + 1][16][LDZ]  ring of dZ2 tiles
    float* const GT = Zt + RING * 16 * LDZ;            // [T1][16 ch][LDG]  G tiles, transposed (GT[t][i][row])
    float* const red = GT + a.T1 * 16 * LDG;           // [2][4 waves][16 rows][LDG]  transposed-conv partials of the 4 M waves, double buffered
    float* const cs = red + 2 * RED;                   // [T2][4]: c1, c2, mean, rstd
    float* const USt = cs + 4 * a.T2;                  // [2][16][LDZ]  recomputed gate inputs [U | S] of tiles t, t + 1 (written by the M waves)
    const bool roleE = threadIdx.x < 256;              // wave-uniform
    const int tid = threadIdx.x & 255, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int N = a.N, T1 = a.T1, T2 = a.T2;
    const int r = tid >> 4, cq = tid & 15;             // row r, float4 column cq (+ 16 it) of the 16-row tile
    // Round 4: a workgroup walks a contiguous run of whole (window, node tile) items -- gridDim.x = min(items, two per CU) -- and keeps its
    // dW_eff2 / db_eff2 accumulators across them: on grids of several rounds (C3: 1344 items, the 8192-node graph: 8192) the stationary
    // weights are loaded once per workgroup instead of once per item and the partial-sum table shrinks from one block per item to one per
    // workgroup (8192-node graph: 201 MB -> 12.6 MB per block, read once more by the reduction).  At C2 (416 items) nothing changes.
    const long items = (long)a.B * a.node_tiles;
    const long it_lo = items * (long)blockIdx.x / (long)gridDim.x, it_hi = items * ((long)blockIdx.x + 1) / (long)gridDim.x;
    float* const part = a.part + (size_t)blockIdx.x * (KT * 16 * NC + NC);
    STGCN_PHASE(8, 0);

    if (roleE) {
        // =========================================== E waves ===========================================================
        struct Tile { Raw4<ET> dy[IT], u[IT], s[IT], y[IT]; };   // (raw: converted and masked where E consumes them, see tc1_bwd_kernel)
        f32x4 dbu[IT], dbq[IT];                            // bias partials: over all items of this workgroup
#pragma unroll
        for (int it = 0; it < IT; ++it) { dbu[it] = zero4(); dbq[it] = zero4(); }
        const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
        const uint64_t n4 = ((uint64_t)N * C2) >> 2;
        for (long item = it_lo; item < it_hi; ++item) {
        const int b = (int)(item / a.node_tiles), nt = (int)(item - (long)b * a.node_tiles), n0 = nt * 16;
        const bool rv = n0 + r < N;
        const int rc = rv ? n0 + r : N - 1;                // clamped row: rows beyond N read a valid address and are masked to zero
        auto fetch = [&](int t2, Tile& t) {
            const size_t e0 = (((size_t)b * T2 + (t2 < T2 ? t2 : T2 - 1)) * N + rc) * C2 + 4 * cq;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                t.dy[it] = ldraw4(dy_ + e0 + 64 * it);
                if constexpr (TRAINING) t.y[it] = ldraw4(ym_ + e0 + 64 * it);   // (unconditional: a branch around a load resets the wait counts)
                if constexpr (!RECOMP) {
                    t.u[it] = ldraw4(U_ + e0 + 64 * it);
                    t.s[it] = ldraw4(S_ + e0 + 64 * it);
                }
            }
        };
        Tile pA, pB;   // tiles 0, 2, .. / 1, 3, ..: requested two steps ahead of the E that consumes them
        fetch(0, pA);
        fetch(1, pB);
        f32x4 gam[IT], dgam[IT], dbet[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            gam[it] = ld4(a.gamma + (size_t)rc * C2 + 4 * (cq + 16 * it));
            dgam[it] = zero4(); dbet[it] = zero4();
        }
        // slab constants of this window's T2 slabs: 32 lanes per slab, 8 slabs at a time, 8 independent loads per lane in flight
        {
            const int grp = tid >> 5, l32 = tid & 31;
            for (int t0 = 0; t0 < T2; t0 += 8) {   // uniform trip count
                const int t = t0 + grp;
                const long slab = (long)b * T2 + (t < T2 ? t : 0);
                float x = 0.f, y = 0.f;
                if (a.slabconst) {
                    if (t < T2 && l32 == 0) {
                        const float2 c = a.slabconst[slab];
                        x = c.x; y = c.y;
                    }
                } else {
                    const float2* rs = a.rowstat + slab * N;
                    for (int i0 = 0; i0 < N; i0 += 256) {
                        float2 v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = i0 + l32 + 32 * j;
                            v[j] = (t < T2 && i < N) ? rs[i] : make_float2(0.f, 0.f);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            x += v[j].x;
                            y += v[j].y;
                        }
                    }
                }
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) {
                    x += __shfl_xor(x, m);
                    y += __shfl_xor(y, m);
                }
                if (t < T2 && l32 == 0) {
                    const float inv = a.slabconst ? 1.0f : 1.0f / ((float)N * (float)C2);
                    cs[4 * t] = x * inv;
                    cs[4 * t + 1] = y * inv;
                    cs[4 * t + 2] = a.mean[slab];
                    cs[4 * t + 3] = a.rstd[slab];
                }
            }
        }
        const uint64_t q0 = (((uint64_t)rc * C2) >> 2) + cq;
        // E(t): branch free (rows beyond N carry s = 0, dy = 0: every product vanishes)
        auto E = [&](int t, const Tile& tl) {
            float* const Zs = Zt + (t % RING) * 16 * LDZ;
            const float* const Us = USt + (t & 1) * 16 * LDZ + r * LDZ;
            const float c1 = cs[4 * t], c2 = cs[4 * t + 1], mean = cs[4 * t + 2], rstd = cs[4 * t + 3];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                f32x4 dy = rv ? cvt4(tl.dy[it]) : zero4();
                const int c4 = cq + 16 * it;
                f32x4 u, s;
                if constexpr (RECOMP) {
                    u = ld4(Us + 4 * c4);
                    s = rv ? ld4(Us + C2 + 4 * c4) : zero4();   // rows beyond N: s = 0 makes every product of the gate backward vanish
                } else {
                    u = cvt4(tl.u[it]);
                    s = rv ? cvt4(tl.s[it]) : zero4();   // rows beyond N: s = 0 makes every product of the gate backward vanish
                }
                if constexpr (TRAINING) {
                    if (mask_y) {   // (uniform) a dropped element of y is -0.0 (drop_encode)
                        const f32x4 yv = cvt4(tl.y[it]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float yi = yv[i];
                            dy[i] = drop_kept(yi) ? dy[i] * a.keep_scale : 0.f;
                        }
                    } else {
                        const f32x4 k = dropout_scale4(((uint64_t)b * T2 + t) * n4 + q0 + 16 * it, a.seed, off, a.thresh, a.keep_scale);
#pragma unroll
                        for (int i = 0; i < 4; ++i) dy[i] *= k[i];
                    }
                }
                f32x4 du, dq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xh = (gate_fwd(u[i], s[i], ACT) - mean) * rstd;   // ACT is a compile-time constant: no tanhf call / branch for GLU
                    const float gg = dy[i] * gam[it][i];
                    const float dh = rstd * (gg - c1 - xh * c2);
                    dgam[it][i] += dy[i] * xh;
                    dbet[it][i] += dy[i];
                    float du_, dq_;
                    gate_bwd(dh, u[i], s[i], ACT, du_, dq_);
                    du[i] = du_;
                    dq[i] = dq_;
                }
                dbu[it] += du;
                dbq[it] += dq;
                st4(Zs + r * LDZ + 4 * c4, du);
                st4(Zs + r * LDZ + C2 + 4 * c4, dq);
            }
        };
        __syncthreads();   // (A) cs complete (written by E waves), GT complete (written by M waves)
        if constexpr (RECOMP) __syncthreads();   // (A2) gate inputs of tile 0 recomputed
        STGCN_PHASE(8, 1);
        E(0, pA);
        fetch(2, pA);
        // step t1: gate / LayerNorm backward of tile t1 + 1 from the set that holds it, then that set's next request (unconditional: the tile
        // index is clamped -- a branch around the loads would reset the compiler's wait counts).  Unrolled by two over the named sets, first
        // step peeled (exact counts at the loop header), as in tc1_bwd_kernel.
        auto stepE = [&](int t1, Tile& ts) __attribute__((always_inline)) {
            __syncthreads();   // (B) tile t1 visible to the M waves
            if (t1 + 1 < T2) E(t1 + 1, ts);
            fetch(t1 + 3, ts);
        };
        int t1 = 0;
        stepE(t1, pB);
        for (++t1; t1 + 1 < T1; t1 += 2) {
            stepE(t1, pA);
            stepE(t1 + 1, pB);
        }
        if (t1 < T1) stepE(t1, pA);
        STGCN_PHASE(8, 4);
        __syncthreads();       // (C) last partial tiles visible
        if (rv) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const size_t o = (size_t)b * N * C2 + (size_t)(n0 + r) * C2 + 4 * (cq + 16 * it);
                st4_wt(a.dgam_part + o, dgam[it]);
                st4_wt(a.dbet_part + o, dbet[it]);
            }
        }
        }   // (next item: its prologue rewrites `cs`, whose last readers -- this role's E(t) -- are behind barrier (C))
        __syncthreads();       // (D) `red` free: it becomes the bias-reduction buffer
        // db_eff2[o] = sum over the 16 rows (threads with equal cq: lanes 16 apart, then the 4 waves through LDS)
        float* bred = red;     // [4 waves][NC] (NC <= 256: fits the 2 x 1280 floats of `red`)
#pragma unroll
        for (int it = 0; it < IT; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = dbu[it][i], y = dbq[it][i];
                x += __shfl_xor(x, 16); y += __shfl_xor(y, 16);
                x += __shfl_xor(x, 32); y += __shfl_xor(y, 32);
                if (g == 0) {
                    bred[w * NC + 4 * (l15 + 16 * it) + i] = x;
                    bred[w * NC + C2 + 4 * (l15 + 16 * it) + i] = y;
                }
            }
        }
        __syncthreads();       // (E)
        if (tid < NC) part[(size_t)KT * 16 * NC + tid] = (bred[tid] + bred[NC + tid]) + (bred[2 * NC + tid] + bred[3 * NC + tid]);
        STGCN_PHASE(8, 6);
    } else {
        // =========================================== M waves ===========================================================
        // stationary weights of the transposed conv: wave w contracts o in [w*NC/4, (w+1)*NC/4) of every tap
        typename MM::frag Wr[KT][QW];
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int q = 0; q < QW; ++q) Wr[k][q] = MM::cvt(ld4(a.Wd + (size_t)(k * 16 + l15) * NC + w * (NC / 4) + q * 16 + 4 * g));
        // forward weights of this wave's output channels (P tiles w + 4j, Q tiles C2/16 further) for the recomputation of the gate inputs
        constexpr int NTP = C2 / 64, MT = C2 / 16;
        typename MM::frag wP[NTP][KT], wQ[NTP][KT];
        f32x4 bp[NTP], bq[NTP];
        if constexpr (RECOMP) {
#pragma unroll
            for (int j = 0; j < NTP; ++j) {
#pragma unroll
                for (int kc = 0; kc < KT; ++kc) {
                    wP[j][kc] = MM::cvt(ld4(a.Wp + ((size_t)((w + 4 * j) * KT + kc) * 64 + lane) * 4));
                    wQ[j][kc] = MM::cvt(ld4(a.Wp + ((size_t)((w + 4 * j + MT) * KT + kc) * 64 + lane) * 4));
                }
                bp[j] = ld4(a.bias + 16 * (w + 4 * j) + 4 * g);
                bq[j] = ld4(a.bias + C2 + 16 * (w + 4 * j) + 4 * g);
            }
        }
        // R(t): Z^T[o][row] = W_eff2^T im2col(G)^T for tile t: A = the weights, B[k = ch 4g + s][n = row l15] from the transposed G tiles;
        // D leaves a lane with 4 consecutive channels of row l15, P and Q of one channel in the same lane
        auto R = [&](int t) __attribute__((always_inline)) {
            float* const Us = USt + (t & 1) * 16 * LDZ + l15 * LDZ;
            typename MM::frag fb[KT];
#pragma unroll
            for (int kc = 0; kc < KT; ++kc) fb[kc] = MM::cvt(gather4(GT + ((t + kc) * 16 + 4 * g) * LDG + l15, LDG));
#pragma unroll
            for (int j = 0; j < NTP; ++j) {
                f32x4 accP = zero4(), accQ = zero4();
#pragma unroll
                for (int kc = 0; kc < KT; ++kc) MM::mma_a2(wP[j][kc], wQ[j][kc], fb[kc], accP, accQ);
                f32x4 u, sg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[e] = accP[e] + bp[j][e];
                    sg[e] = sigmoid_f(accQ[e] + bq[j][e]);
                }
                st4(Us + 16 * (w + 4 * j) + 4 * g, u);
                st4(Us + C2 + 16 * (w + 4 * j) + 4 * g, sg);
            }
        };
        f32x4 accw[KT][NTW];                               // dW_eff2 of all items of this workgroup
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int j = 0; j < NTW; ++j) accw[k][j] = zero4();
        for (long item = it_lo; item < it_hi; ++item) {
        const int b = (int)(item / a.node_tiles), nt = (int)(item - (long)b * a.node_tiles), n0 = nt * 16;
        const bool rv = n0 + r < N;
        // all G tiles of this (window, node tile), transposed (the previous item's F(T1 - 1) reads `red` only: its GT value was taken before barrier (C))
        for (int idx = tid; idx < T1 * 64; idx += 256) {
            const int t = idx >> 6, rem = idx & 63, rr = rem >> 2, q = rem & 3;
            const f32x4 v = n0 + rr < N ? ldx4(G_ + (((size_t)b * T1 + t) * N + n0 + rr) * 16 + 4 * q) : zero4();
#pragma unroll
            for (int i = 0; i < 4; ++i) GT[(t * 16 + 4 * q + i) * LDG + rr] = v[i];
        }
        __syncthreads();   // (A)
        if constexpr (RECOMP) {
            R(0);
            __syncthreads();   // (A2)
            if (T2 > 1) R(1);
        }
        for (int t1 = 0; t1 < T1; ++t1) {
            __syncthreads();   // (B) dZ2 tile t1 and the partial tiles of step t1 - 1 are visible
            if (t1 > 0) {      // F(t1 - 1): dYg = relu'(G) * (sum of the 4 waves' partial tiles), thread (row r, channel cq)
                const int t = t1 - 1;
                const float* rd = red + (t & 1) * RED;
                float v = (rd[(0 * 16 + r) * LDG + cq] + rd[(1 * 16 + r) * LDG + cq]) + (rd[(2 * 16 + r) * LDG + cq] + rd[(3 * 16 + r) * LDG + cq]);
                if (!(GT[(t * 16 + cq) * LDG + r] > 0.f)) v = 0.f;
                if (rv) stx1(dYg_ + (((size_t)b * T1 + t) * N + n0 + r) * 16 + cq, v);
            }
            if (t1 < T2) {     // weight gradient of tile t1: A[m = i][k = row] = G[t1 + tap][row][i] (one 16-byte read), B[k = row][n = o] = dZ2
                const float* const Zs = Zt + (t1 % RING) * 16 * LDZ;
                f32x4 bz[NTW];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j) bz[j][s] = Zs[(4 * g + s) * LDZ + (w * NTW + j) * 16 + l15];
                }
                typename MM::frag fz[NTW];
#pragma unroll
                for (int j = 0; j < NTW; ++j) fz[j] = MM::cvt(bz[j]);
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const typename MM::frag af = MM::cvt(ld4(GT + ((t1 + k) * 16 + l15) * LDG + 4 * g));
#pragma unroll
                    for (int j = 0; j < NTW; j += 2) MM::mma_b2(af, fz[j], fz[j + 1], accw[k][j], accw[k][j + 1]);
                }
            }
            // transposed conv for output step t1: taps with 0 <= t1 - tap < T2; two independent MFMA chains
            f32x4 accd[2] = {zero4(), zero4()};
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int ts = t1 - k;
                if (ts >= 0 && ts < T2) {   // uniform
                    const float* zr = Zt + (ts % RING) * 16 * LDZ + l15 * LDZ + w * (NC / 4) + 4 * g;
#pragma unroll
                    for (int q = 0; q < QW; ++q) {
                        MM::mma_split(Wr[k][q], MM::cvt(ld4(zr + q * 16)), accd[0], accd[1]);
                    }
                }
            }
            st4(red + (t1 & 1) * RED + (w * 16 + l15) * LDG + 4 * g, accd[0] + accd[1]);   // D[m = i = 4g + r][n = row = l15]
            if constexpr (RECOMP) {
                if (t1 + 2 < T2) R(t1 + 2);   // slot (t1 & 1): tile t1's gate inputs were last read by E(t1), before barrier (B) of this step
            }
        }
        // (the ReLU mask of the last tile is read BEFORE barrier (C) -- behind it a wave that has finished F(T1 - 1) goes straight
        //  into the next item's staging, which rewrites all of GT while a slower wave of this role could still be reading it; `red` is safe,
        //  its next writer sits behind the next item's barriers (A) and (B))
        const bool g_last = GT[((T1 - 1) * 16 + cq) * LDG + r] > 0.f;
        __syncthreads();       // (C)
        {
            const int t = T1 - 1;
            const float* rd = red + (t & 1) * RED;
            float v = (rd[(0 * 16 + r) * LDG + cq] + rd[(1 * 16 + r) * LDG + cq]) + (rd[(2 * 16 + r) * LDG + cq] + rd[(3 * 16 + r) * LDG + cq]);
            if (!g_last) v = 0.f;
            if (rv) stx1(dYg_ + (((size_t)b * T1 + t) * N + n0 + r) * 16 + cq, v);
        }
        }
        STGCN_PHASE(8, 5);
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                st4_wt(part + (size_t)((w * NTW + j) * 16 + l15) * (KT * 16) + k * 16 + 4 * g, accw[k][j]);   // TRANSPOSED partial [NC][KT * 16]: a lane's 4 rows are 16 contiguous bytes
        __syncthreads();       // (D)
        __syncthreads();       // (E)
    }
}

// ================================================================================================
// K3: backward of  tmp_conv1 -> Align(c0 -> c1)  (layers.py:252, :223) on 16-node tiles of one window, walking the time axis:
//     per tile t1:  dA -> dH = dA Wa^T -> gate backward with the saved U1, S1 -> dZ1 tile (LDS ring only)
//     dW_eff1 += im2col(x)^T dZ1, db_eff1 += sum dZ1, dWa += H^T dA, dba += sum dA              (per-workgroup partials)
//     dx[t] = sum_tap dZ1[t - tap] W_eff1[tap]^T  (+ the LayerNorm-backward row partials of the layer that produced x, stgcn_ln_hook)
// replaces align_gate_bwd + tconv_bwd_weight.tc1 + tconv_bwd_data.tc1 (+ ln_bwd_rowstats of the previous block): dZ1, the largest
// tensor of the backward pass, never leaves the chip and x / U1 / S1 are read once.
// 12 waves: 4 E waves (tile production: dH as 4 small MFMAs, gate backward on the VALU, dx stores, hook epilogue) | 4 Mw waves (weight-gradient MFMAs, 24 accumulator tiles each)
// | 4 Md waves (transposed conv, the whole W_eff1 slice of their 16 input channels stationary in registers), one barrier per step.
//
// Work distribution: the kernel is MFMA-bound and the path only offers B * ceil(N/16) (window, node tile) items (416 at C2 for 256
// CUs), so whole items cannot be balanced (two per workgroup on 208 CUs measured 81 % of the chip).  The unit of work is therefore
// one OUTPUT STEP of one item, weighted by its MFMA count; the linear sequence (item, step) is cut into `gridDim.x` ranges of equal
// weight and every workgroup walks its range, keeping the weight-gradient accumulators in registers across items.  A range that
// starts inside an item first re-forms the Kt - 1 dZ1 tiles in front of it (work of the E waves only: no conv / weight-gradient
// MFMA is repeated; a tile's weight-gradient contribution belongs to the range that owns output step t1 = tile index).  The cut is a
// pure function of blockIdx: results are bitwise reproducible.
// Template: C0 = 64 (NC = 128), CIN in {16, 32, 64}, KT taps.
// ================================================================================================
struct Tc1BwdArgs {
    const float* dA;          // [B][T1][N][16]
    const float* U;           // [B][T1][N][C0]   saved gate inputs of tmp_conv1
    const float* S;
    const float* x;           // [B][T][N][CIN]
    const float* WaD;         // [C0][16] dense Align map (PK_ALIGN_DENSE)
    const float* Wd;          // [KT*CIN][NC] dense W_eff1 (PK_TCONV_DENSE)
    float* dx;                // [B][T][N][CIN]
    float* part;              // [wgs][KT*CIN*NC + NC + C0*16 + 16]  dW_eff1 (transposed: [NC][KT*CIN]) | db_eff1 | dWa | dba
    LnRowstatOut rs;          // hook: row partials of the LayerNorm in front of x (rs.rowstat == null: none)
    int B, T, T1, N, node_tiles;
};
inline size_t tc1_bwd_lds_bytes(int C0, int CIN, int Kt, bool x6 = false) {
    // x tiles: fp32, transposed [CIN][20]; or (X6) three ROW-MAJOR bf16 planes [16][CIN + 16] (the weight-gradient waves read them through the transposing LDS read)
    const size_t xt = x6 ? (size_t)(Kt + 1) * 3 * 16 * (CIN + 16) * sizeof(short) : (size_t)(Kt + 1) * CIN * 20 * sizeof(float);
    const size_t zt = x6 ? (size_t)(Kt + 1) * 3 * 16 * (2 * C0 + 8) * sizeof(short) : (size_t)(Kt + 1) * 16 * (2 * C0 + 4) * sizeof(float);   // dZ1 tiles: fp32, or three bf16 planes
    const size_t rest = ((size_t)(Kt + 1) * 16 * 16 + 2 * 16 * (C0 + 4) + 2 * 16 * (CIN + 4)) * sizeof(float);
    // X6: + the low plane of the transposed conv's stationary weights (4 waves x Kt * 2 C0 / 16 fragments x 64 lanes x 8 bytes; hi / mid stay in registers)
    return zt + xt + rest + (x6 ? (size_t)4 * Kt * (2 * C0 / 16) * 64 * 4 * sizeof(short) : 0);
}
inline int tc1_bwd_part_floats(int C0, int CIN, int Kt) { return Kt * CIN * 2 * C0 + 2 * C0 + C0 * 16 + 16; }

// MFMA weight of output step s of an item = MFMAs one SIMD issues for it (any proportional measure works: it only places the cuts)
__host__ __device__ inline int tc1_bwd_step_weight(int s, int T1, int KT, int CIN) {
    int w = 0;
    if (s < T1) w += KT * (CIN / 16) * 8 + 4;              // weight gradient of tile s: KT*MI m-tiles x 2 n-tiles x 4, + the Align gradient
    for (int k = 0; k < KT; ++k)
        if (s - k >= 0 && s - k < T1) w += 32;             // transposed conv: NC / 16 chunks x 4 per tap
    return w;
}

// One tap of the X6 transposed conv (Md waves of tc1_bwd), hand-scheduled: 8 fragments of 16 dZ1 columns, per fragment
//     ds_read_b128 (h | m) + ds_read_b64 l of the dZ1 tile, ds_read_b64 of the weight's low plane  ->  v[B .. B+11] = [bl bh bm bh' | ah' al]
//     (Ah'|Al)(Bl|Bh) -> ca ,  (Ah|Am)(Bh|Bm) -> cb ,  (Ah|Am)(Bm|Bh') -> cc        (the six products; three independent accumulators)
// The operands of the three instructions are OVERLAPPING windows of one register tuple, which the compiler cannot express (it copied ~13
// registers per fragment, fused plane reads into ds_read2_b64 and waited for every fragment's loads right after issuing them: the Md waves
// were the critical role of the kernel, profiles/r6-43_tc1_bwd_timing_only.txt).  Here: two v_mov_b64 per fragment, loads two fragments ahead
// in two register sets (v144 - v155, v156 - v167: clobbers), in-order LDS returns counted by lgkmcnt.  The trailing s_nop cover the
// MFMA-write -> VALU-read hazard the compiler cannot see.
#define STGCN_TC1BWD_MD_TAP_ASM \
    "ds_read_b128 v[146:149], %[zh] offset:0\n\t" \
    "ds_read_b64 v[144:145], %[zl] offset:0\n\t" \
    "ds_read_b64 v[154:155], %[wl] offset:0\n\t" \
    "ds_read_b128 v[158:161], %[zh] offset:64\n\t" \
    "ds_read_b64 v[156:157], %[zl] offset:32\n\t" \
    "ds_read_b64 v[166:167], %[wl] offset:512\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[150:151], v[146:147]\n\t" \
    "v_mov_b64 v[152:153], %[l0]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[152:155], v[144:147], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w0], v[146:149], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w0], v[148:151], %[cc]\n\t" \
    "ds_read_b128 v[146:149], %[zh] offset:128\n\t" \
    "ds_read_b64 v[144:145], %[zl] offset:64\n\t" \
    "ds_read_b64 v[154:155], %[wl] offset:1024\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[162:163], v[158:159]\n\t" \
    "v_mov_b64 v[164:165], %[l1]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[164:167], v[156:159], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w1], v[158:161], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w1], v[160:163], %[cc]\n\t" \
    "ds_read_b128 v[158:161], %[zh] offset:192\n\t" \
    "ds_read_b64 v[156:157], %[zl] offset:96\n\t" \
    "ds_read_b64 v[166:167], %[wl] offset:1536\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[150:151], v[146:147]\n\t" \
    "v_mov_b64 v[152:153], %[l2]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[152:155], v[144:147], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w2], v[146:149], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w2], v[148:151], %[cc]\n\t" \
    "ds_read_b128 v[146:149], %[zh] offset:256\n\t" \
    "ds_read_b64 v[144:145], %[zl] offset:128\n\t" \
    "ds_read_b64 v[154:155], %[wl] offset:2048\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[162:163], v[158:159]\n\t" \
    "v_mov_b64 v[164:165], %[l3]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[164:167], v[156:159], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w3], v[158:161], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w3], v[160:163], %[cc]\n\t" \
    "ds_read_b128 v[158:161], %[zh] offset:320\n\t" \
    "ds_read_b64 v[156:157], %[zl] offset:160\n\t" \
    "ds_read_b64 v[166:167], %[wl] offset:2560\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[150:151], v[146:147]\n\t" \
    "v_mov_b64 v[152:153], %[l4]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[152:155], v[144:147], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w4], v[146:149], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w4], v[148:151], %[cc]\n\t" \
    "ds_read_b128 v[146:149], %[zh] offset:384\n\t" \
    "ds_read_b64 v[144:145], %[zl] offset:192\n\t" \
    "ds_read_b64 v[154:155], %[wl] offset:3072\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[162:163], v[158:159]\n\t" \
    "v_mov_b64 v[164:165], %[l5]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[164:167], v[156:159], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w5], v[158:161], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w5], v[160:163], %[cc]\n\t" \
    "ds_read_b128 v[158:161], %[zh] offset:448\n\t" \
    "ds_read_b64 v[156:157], %[zl] offset:224\n\t" \
    "ds_read_b64 v[166:167], %[wl] offset:3584\n\t" \
    "s_waitcnt lgkmcnt(3)\n\t" \
    "v_mov_b64 v[150:151], v[146:147]\n\t" \
    "v_mov_b64 v[152:153], %[l6]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[152:155], v[144:147], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w6], v[146:149], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w6], v[148:151], %[cc]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_mov_b64 v[162:163], v[158:159]\n\t" \
    "v_mov_b64 v[164:165], %[l7]\n\t" \
    "s_nop 1\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[ca], v[164:167], v[156:159], %[ca]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cb], %[w7], v[158:161], %[cb]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[cc], %[w7], v[160:163], %[cc]\n\t" \
    "s_nop 7\n\t" \
    "s_nop 7\n\t"
// One output step of the X6 weight-gradient waves (Mw) of tc1_bwd at CIN = 64, KT = 3, hand-scheduled like the tap above.  Every operand comes out of
// a ROW-MAJOR bf16 tile through the transposing LDS read (ds_read_b64_tr_b16, ld_tr4): the two dZ1 fragments of the step as [l h m h'] tuples
// (v136 - v151), the 12 x fragments as [h m h' l] tuples in two register sets (v152 - v159 / v160 - v167), loads two fragments ahead; six MFMAs per
// x fragment with overlapping operand windows, no VALU instruction at all (the compiler's form: 24 two-byte gathers + packing per step and ~10
// register copies per fragment).  x planes: 16 * 80 shorts apart (2560 bytes), m-tiles 32 bytes apart.
#define STGCN_TC1BWD_MW_STEP_ASM \
    "ds_read_b64_tr_b16 v[136:137], %[zl] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[138:139], %[zh] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[140:141], %[zh] offset:8\n\t" \
    "ds_read_b64_tr_b16 v[142:143], %[zh] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[144:145], %[zl] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[146:147], %[zh] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[148:149], %[zh] offset:72\n\t" \
    "ds_read_b64_tr_b16 v[150:151], %[zh] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[152:153], %[x0] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[154:155], %[x0] offset:2560\n\t" \
    "ds_read_b64_tr_b16 v[156:157], %[x0] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[158:159], %[x0] offset:5120\n\t" \
    "ds_read_b64_tr_b16 v[160:161], %[x0] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[162:163], %[x0] offset:2592\n\t" \
    "ds_read_b64_tr_b16 v[164:165], %[x0] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[166:167], %[x0] offset:5152\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c0_0], v[156:159], v[136:139], %[c0_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c0_1], v[156:159], v[144:147], %[c0_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c0_0], v[152:155], v[138:141], %[c0_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c0_1], v[152:155], v[146:149], %[c0_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c0_0], v[152:155], v[140:143], %[c0_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c0_1], v[152:155], v[148:151], %[c0_1]\n\t" \
    "ds_read_b64_tr_b16 v[152:153], %[x0] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[154:155], %[x0] offset:2624\n\t" \
    "ds_read_b64_tr_b16 v[156:157], %[x0] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[158:159], %[x0] offset:5184\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c1_0], v[164:167], v[136:139], %[c1_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c1_1], v[164:167], v[144:147], %[c1_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c1_0], v[160:163], v[138:141], %[c1_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c1_1], v[160:163], v[146:149], %[c1_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c1_0], v[160:163], v[140:143], %[c1_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c1_1], v[160:163], v[148:151], %[c1_1]\n\t" \
    "ds_read_b64_tr_b16 v[160:161], %[x0] offset:96\n\t" \
    "ds_read_b64_tr_b16 v[162:163], %[x0] offset:2656\n\t" \
    "ds_read_b64_tr_b16 v[164:165], %[x0] offset:96\n\t" \
    "ds_read_b64_tr_b16 v[166:167], %[x0] offset:5216\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c2_0], v[156:159], v[136:139], %[c2_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c2_1], v[156:159], v[144:147], %[c2_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c2_0], v[152:155], v[138:141], %[c2_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c2_1], v[152:155], v[146:149], %[c2_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c2_0], v[152:155], v[140:143], %[c2_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c2_1], v[152:155], v[148:151], %[c2_1]\n\t" \
    "ds_read_b64_tr_b16 v[152:153], %[x1] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[154:155], %[x1] offset:2560\n\t" \
    "ds_read_b64_tr_b16 v[156:157], %[x1] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[158:159], %[x1] offset:5120\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c3_0], v[164:167], v[136:139], %[c3_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c3_1], v[164:167], v[144:147], %[c3_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c3_0], v[160:163], v[138:141], %[c3_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c3_1], v[160:163], v[146:149], %[c3_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c3_0], v[160:163], v[140:143], %[c3_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c3_1], v[160:163], v[148:151], %[c3_1]\n\t" \
    "ds_read_b64_tr_b16 v[160:161], %[x1] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[162:163], %[x1] offset:2592\n\t" \
    "ds_read_b64_tr_b16 v[164:165], %[x1] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[166:167], %[x1] offset:5152\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c4_0], v[156:159], v[136:139], %[c4_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c4_1], v[156:159], v[144:147], %[c4_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c4_0], v[152:155], v[138:141], %[c4_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c4_1], v[152:155], v[146:149], %[c4_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c4_0], v[152:155], v[140:143], %[c4_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c4_1], v[152:155], v[148:151], %[c4_1]\n\t" \
    "ds_read_b64_tr_b16 v[152:153], %[x1] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[154:155], %[x1] offset:2624\n\t" \
    "ds_read_b64_tr_b16 v[156:157], %[x1] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[158:159], %[x1] offset:5184\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c5_0], v[164:167], v[136:139], %[c5_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c5_1], v[164:167], v[144:147], %[c5_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c5_0], v[160:163], v[138:141], %[c5_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c5_1], v[160:163], v[146:149], %[c5_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c5_0], v[160:163], v[140:143], %[c5_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c5_1], v[160:163], v[148:151], %[c5_1]\n\t" \
    "ds_read_b64_tr_b16 v[160:161], %[x1] offset:96\n\t" \
    "ds_read_b64_tr_b16 v[162:163], %[x1] offset:2656\n\t" \
    "ds_read_b64_tr_b16 v[164:165], %[x1] offset:96\n\t" \
    "ds_read_b64_tr_b16 v[166:167], %[x1] offset:5216\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c6_0], v[156:159], v[136:139], %[c6_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c6_1], v[156:159], v[144:147], %[c6_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c6_0], v[152:155], v[138:141], %[c6_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c6_1], v[152:155], v[146:149], %[c6_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c6_0], v[152:155], v[140:143], %[c6_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c6_1], v[152:155], v[148:151], %[c6_1]\n\t" \
    "ds_read_b64_tr_b16 v[152:153], %[x2] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[154:155], %[x2] offset:2560\n\t" \
    "ds_read_b64_tr_b16 v[156:157], %[x2] offset:0\n\t" \
    "ds_read_b64_tr_b16 v[158:159], %[x2] offset:5120\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c7_0], v[164:167], v[136:139], %[c7_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c7_1], v[164:167], v[144:147], %[c7_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c7_0], v[160:163], v[138:141], %[c7_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c7_1], v[160:163], v[146:149], %[c7_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c7_0], v[160:163], v[140:143], %[c7_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c7_1], v[160:163], v[148:151], %[c7_1]\n\t" \
    "ds_read_b64_tr_b16 v[160:161], %[x2] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[162:163], %[x2] offset:2592\n\t" \
    "ds_read_b64_tr_b16 v[164:165], %[x2] offset:32\n\t" \
    "ds_read_b64_tr_b16 v[166:167], %[x2] offset:5152\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c8_0], v[156:159], v[136:139], %[c8_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c8_1], v[156:159], v[144:147], %[c8_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c8_0], v[152:155], v[138:141], %[c8_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c8_1], v[152:155], v[146:149], %[c8_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c8_0], v[152:155], v[140:143], %[c8_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c8_1], v[152:155], v[148:151], %[c8_1]\n\t" \
    "ds_read_b64_tr_b16 v[152:153], %[x2] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[154:155], %[x2] offset:2624\n\t" \
    "ds_read_b64_tr_b16 v[156:157], %[x2] offset:64\n\t" \
    "ds_read_b64_tr_b16 v[158:159], %[x2] offset:5184\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c9_0], v[164:167], v[136:139], %[c9_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c9_1], v[164:167], v[144:147], %[c9_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c9_0], v[160:163], v[138:141], %[c9_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c9_1], v[160:163], v[146:149], %[c9_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c9_0], v[160:163], v[140:143], %[c9_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c9_1], v[160:163], v[148:151], %[c9_1]\n\t" \
    "ds_read_b64_tr_b16 v[160:161], %[x2] offset:96\n\t" \
    "ds_read_b64_tr_b16 v[162:163], %[x2] offset:2656\n\t" \
    "ds_read_b64_tr_b16 v[164:165], %[x2] offset:96\n\t" \
    "ds_read_b64_tr_b16 v[166:167], %[x2] offset:5216\n\t" \
    "s_waitcnt lgkmcnt(4)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c10_0], v[156:159], v[136:139], %[c10_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c10_1], v[156:159], v[144:147], %[c10_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c10_0], v[152:155], v[138:141], %[c10_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c10_1], v[152:155], v[146:149], %[c10_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c10_0], v[152:155], v[140:143], %[c10_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c10_1], v[152:155], v[148:151], %[c10_1]\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c11_0], v[164:167], v[136:139], %[c11_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c11_1], v[164:167], v[144:147], %[c11_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c11_0], v[160:163], v[138:141], %[c11_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c11_1], v[160:163], v[146:149], %[c11_1]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c11_0], v[160:163], v[140:143], %[c11_0]\n\t" \
    "v_mfma_f32_16x16x32_bf16 %[c11_1], v[160:163], v[148:151], %[c11_1]\n\t" \
    "s_nop 7\n\t" \
    "s_nop 7\n\t"
// X6 (round 6, fp32 blocks): the weight-gradient and transposed-conv products (Mw and Md waves) as "bf16x6" -- fp32-accurate
// products on the bf16 matrix pipe (Frag3, stgcn_device.hip.h).  Both operands are tiles the E waves produce: x tiles (staged once, read
// by four waves for KT steps) and dZ1 tiles (formed once); E splits them into bf16 planes where it writes them (x: three row-major planes; dZ1: a
// (h | m) plane + an l plane).  The transposed conv (Md waves) holds the (h | m) halves of its 24 stationary weight fragments in registers (96) and
// their l plane in a wave-private LDS region (as triples they would be 144 registers).  Inner loops of both matrix roles: the hand-scheduled blocks above.
template <int C0, int CIN, int KT, int ACT, typename ET, bool X6 = false>
__device__ __forceinline__ void tc1_bwd_body(const Tc1BwdArgs& a) {
    static_assert(C0 == 64 && (CIN == 16 || CIN == 32 || CIN == 64), "shapes covered by the role split below");
    static_assert(!X6 || std::is_same<ET, float>::value, "bf16x6 is a product form of the fp32 blocks");
    typedef Mma<ET> MM;
    const ET* const dA_ = et_ptr<ET>(a.dA);
    const ET* const U_ = et_ptr<ET>(a.U);
    const ET* const S_ = et_ptr<ET>(a.S);
    const ET* const x_ = et_ptr<ET>(a.x);
    ET* const dx_ = et_ptr<ET>(a.dx);
    const ET* const hy_ = et_ptr<ET>(a.rs.y);
    constexpr int NC = 2 * C0, LDZ = NC + 4, RING = KT + 1, LDX = 20, LDH = C0 + 4, LDO = CIN + 4, MI = CIN / 16, QD = NC / 16;
    extern __shared__ float stgcn_smem[];
    float* const Zt = stgcn_smem;                      // [RING][16][LDZ]   dZ1 tiles
    // (X6) dZ1 tiles as a (h | m) plane of 16-byte groups + an l plane of 8-byte groups (row strides = 8 / 4 dwords mod 64: the b128 / b64 reads of
    // 16 rows are conflict-free); x tiles as three row-major planes [16][LDXR] (row stride = an odd multiple of 8 dwords at CIN = 64 / 32)
    constexpr int LDZHM = 2 * NC + 16, LDZL = NC + 8, ZLOFS = 16 * LDZHM, ZSLOT = 16 * (LDZHM + LDZL), LDXR = CIN + 16, XPL = 16 * LDXR;
    float* const XT = Zt + (X6 ? RING * ZSLOT / 2 : RING * 16 * LDZ);   // [RING][CIN][LDX]  x tiles, transposed (XT[t][ch][row]); X6: three bf16 planes of them instead (XTh)
    float* const dAe = XT + (X6 ? RING * 3 * XPL / 2 : RING * CIN * LDX);   // [RING][16][16]    dA tiles (read by the E waves for dH and by the Mw waves for dWa)
    float* const Ht = dAe + RING * 16 * 16;            // [2][16][LDH]      H = act(U) * S tiles (owned tiles only)
    float* const Xo = Ht + 2 * 16 * LDH;               // [2][16][LDO]      dx tiles
    short* const Zh = reinterpret_cast<short*>(Zt);                  // (X6) [RING]{[16][LDZHM], [16][LDZL]}  bf16 form of the dZ1 tiles (instead of the fp32 tiles)
    short* const XTh = reinterpret_cast<short*>(XT);                 // (X6) [RING][3][16][LDXR]  bf16 planes of the x tiles, row major
    short* const Wl = reinterpret_cast<short*>(Xo + 2 * 16 * LDO);   // (X6) [4 waves][KT * QD][64 lanes][4]  low plane of the Md waves' stationary weights
    const int role = threadIdx.x >> 8;                 // 0 = E, 1 = Mw, 2 = Md (wave-uniform)
    const int tid = threadIdx.x & 255, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int N = a.N, T = a.T, T1 = a.T1;
    const int r = tid >> 4, cq = tid & 15;             // E role, x / dx tiles: row r, float4 column cq
    const int er = l15, ecq = 4 * w + g;               // E role, dZ1 tiles: row er, float4 column ecq (the D layout of the dH product below)
    float* const part = a.part + (size_t)blockIdx.x * (KT * CIN * NC + NC + C0 * 16 + 16);
    STGCN_PHASE(10, 0);

    // ---- this workgroup's range of the (item, step) sequence: equal MFMA weight per workgroup (identical in every role) -------
    long Wi = 0;                                       // weight of one item
    for (int s = 0; s < T; ++s) Wi += tc1_bwd_step_weight(s, T1, KT, CIN);
    const long items = (long)a.B * a.node_tiles, Wtot = Wi * items;
    const long w_lo = Wtot * (long)blockIdx.x / (long)gridDim.x, w_hi = Wtot * ((long)blockIdx.x + 1) / (long)gridDim.x;
    auto unit_at = [&](long pos, long& item, int& step) {   // first unit (item, step) whose start position in the sequence is >= pos
        item = pos / Wi;
        const long off = pos - item * Wi;
        long acc = 0;
        step = 0;
        while (step < T && acc < off) {
            acc += tc1_bwd_step_weight(step, T1, KT, CIN);
            ++step;
        }
        if (step >= T) { ++item; step = 0; }
    };
    long item0, item1;
    int s0, s1;
    unit_at(w_lo, item0, s0);
    unit_at(w_hi, item1, s1);
    if (item1 > items) { item1 = items; s1 = 0; }

    STGCN_PHASE(10, 1);
    if (role == 0) {
        // =========================================== E waves ===========================================================
        // dH^T[ch][row] = Wa[ch][j] dA^T[j][row] on the matrix cores (4 MFMAs per wave and tile: K = 16): wave w owns channels 16w .. 16w+15,
        // A[m = ch][k] = Wa[16w + l15][4g + s] stationary in 4 registers, B[k][n = row] = dA[row = l15][4g + s] (one 16-byte LDS read; both
        // operands use the k order j = 4g + s).  D leaves a lane with channels 16w + 4g .. + 3 of row l15 = its U / S / dZ1 quad.
        const typename MM::frag wa = MM::cvt(ld4(a.WaD + (size_t)(16 * w + l15) * 16 + 4 * g));
        f32x4 dbu = zero4(), dbq = zero4(), dba = zero4();
        const uint64_t hoff = ln_rowstat_offset(a.rs);   // (once: inside hook_fetch it was a dependent load + s_waitcnt vmcnt(0) in every time step)
        struct Tile { Raw4<ET> u, s; };   // (raw: converted and masked where E consumes them)
        STGCN_ACC_DECL();
        for (long item = item0; item <= item1 && item < items; ++item) {
            const int sb = item == item0 ? s0 : 0, se = item == item1 ? s1 : T;
            if (sb >= se) continue;                    // (uniform over the workgroup: every role evaluates the same list)
            const int b = (int)(item / a.node_tiles), n0 = (int)(item - (long)b * a.node_tiles) * 16;
            const bool rv = n0 + r < N, erv = n0 + er < N;
            const int rc = rv ? n0 + r : N - 1, erc = erv ? n0 + er : N - 1;
            const int t_lo = sb - KT + 1 > 0 ? sb - KT + 1 : 0;            // first dZ1 tile this range needs
            const int t_hi = se < T1 ? se : T1;                            // tiles t_lo .. t_hi - 1
            auto fetch = [&](int t1, Tile& t) __attribute__((always_inline)) {
                const int tc = t1 < T1 ? t1 : T1 - 1;
                const size_t e0 = (((size_t)b * T1 + tc) * N + erc) * C0 + 4 * ecq;
                t.u = ldraw4(U_ + e0);
                t.s = ldraw4(S_ + e0);
            };
            // dA tile t -> registers of 64 threads (row rowq >> 2, quad rowq & 3) -> ring slot t % RING; owned tiles count towards dba
            // (the loads of the step loop are UNCONDITIONAL -- clamped addresses, the value masked afterwards: a branch around a load makes the
            //  compiler's wait counts start over at the join, and every later use then waits for everything in flight)
            auto get_dA = [&](int t, int rowq) __attribute__((always_inline)) {
                const int tc = t < T1 ? t : T1 - 1, nr = n0 + (rowq >> 2), nc = nr < N ? nr : N - 1;
                return ldraw4(dA_ + (((size_t)b * T1 + tc) * N + nc) * 16 + 4 * (rowq & 3));   // (masked by put_dA)
            };
            auto put_dA = [&](int t, int rowq, const Raw4<ET>& raw) __attribute__((always_inline)) {
                const f32x4 v = (t < T1 && n0 + (rowq >> 2) < N) ? cvt4(raw) : zero4();
                st4(dAe + (t % RING) * 256 + (rowq >> 2) * 16 + 4 * (rowq & 3), v);
                if (t >= sb && t < t_hi) dba += v;
            };
            auto get_x = [&](int xt) __attribute__((always_inline)) {
                const int xc = xt < T ? xt : T - 1, qc = cq < CIN / 4 ? cq : CIN / 4 - 1;
                return ldraw4(x_ + (((size_t)b * T + xc) * N + rc) * CIN + 4 * qc);   // (masked by put_x)
            };
            auto put_x = [&](int xt, const Raw4<ET>& raw) {
                const f32x4 v = (rv && xt < T) ? cvt4(raw) : zero4();
                if (cq < CIN / 4) {
                    if constexpr (X6) {   // three bf16 planes, ROW major: three 8-byte stores (transposed they were twelve 2-byte stores with 4-way bank
                                          // conflicts: 3.8 us of the launch, profiles/r6-50_*); the weight-gradient waves transpose in their LDS read (ld_tr4)
                        st_frag3(XTh + (size_t)(xt % RING) * 3 * XPL + r * LDXR + 4 * cq, XPL, split3(v));
                    } else {
                        float* d = XT + (size_t)(xt % RING) * CIN * LDX + (4 * cq) * LDX + r;
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e * LDX] = v[e];
                    }
                }
            };
            // E(t): dH = dA Wa^T, gate backward, dZ1 tile -> ring; owned tiles (t >= sb) also leave their H tile for the Align gradient and
            // count towards the bias partials
            auto E = [&](int t, const Tile& tl) __attribute__((always_inline)) {
                const f32x4 d4 = ld4(dAe + (t % RING) * 256 + er * 16 + 4 * g);
                const f32x4 dh = MM::mma(wa, MM::cvt(d4), zero4());
                const f32x4 tu = cvt4(tl.u), tsv = erv ? cvt4(tl.s) : zero4();   // rows beyond N: s = 0 makes every product of the gate backward vanish
                f32x4 du, dq, h;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float du_, dq_;
                    gate_bwd(dh[i], tu[i], tsv[i], ACT, du_, dq_);
                    du[i] = du_;
                    dq[i] = dq_;
                    h[i] = gate_fwd(tu[i], tsv[i], ACT);
                }
                if constexpr (X6) {   // the tile as three bf16 planes: split ONCE, where it is formed
                    short* const Zp = Zh + (size_t)(t % RING) * ZSLOT;
                    st_frag3_hml(Zp + er * LDZHM + 8 * ecq, Zp + ZLOFS + er * LDZL + 4 * ecq, split3(du));
                    st_frag3_hml(Zp + er * LDZHM + 8 * (C0 / 4 + ecq), Zp + ZLOFS + er * LDZL + C0 + 4 * ecq, split3(dq));
                } else {
                    float* const Zs = Zt + (t % RING) * 16 * LDZ + er * LDZ;
                    st4(Zs + 4 * ecq, du);
                    st4(Zs + C0 + 4 * ecq, dq);
                }
                if (t >= sb) {                         // uniform
                    dbu += du;
                    dbq += dq;
                    st4(Ht + (t & 1) * 16 * LDH + er * LDH + 4 * ecq, h);
                }
            };
            // hook operands of output step t (the previous block's OUTPUT y = this block's x, and its dropout mask): requested ONE STEP
            // AHEAD of the dx tile they meet -- forming them inside F put dependent memory round trips on the E waves' critical path of
            // every step.  With g = mask * dx * gamma and y = mask * (xhat * gamma + beta) the two row sums need no xhat:
            //     sum g = sum mask dx gamma ,   sum g xhat = sum_kept dx (y - keep_scale * beta)
            struct Hook { Raw4<ET> y; f32x4 k; };
            const bool hk = a.rs.rowstat != nullptr;           // uniform
            const f32x4 hgam = (hk && cq < CIN / 4) ? ld4(a.rs.gamma + (size_t)rc * CIN + 4 * cq) : zero4();
            f32x4 hbks = (hk && cq < CIN / 4) ? ld4(a.rs.beta + (size_t)rc * CIN + 4 * cq) : zero4();   // keep_scale * beta
            if (hk && a.rs.training) {
#pragma unroll
                for (int i = 0; i < 4; ++i) hbks[i] *= a.rs.keep_scale;
            }
            const ET* const hyp = hk ? hy_ : x_;               // (no hook: a valid address of the same shape, the value is never used)
            auto hook_fetch = [&](int t, Hook& h) __attribute__((always_inline)) {
                {
                    const long slab = (long)b * T + (t < T ? t : T - 1);
                    const size_t e = ((size_t)slab * N + rc) * CIN + 4 * (cq < CIN / 4 ? cq : CIN / 4 - 1);
                    h.y = ldraw4(hyp + e);
                    h.k[0] = 1.f; h.k[1] = 1.f; h.k[2] = 1.f; h.k[3] = 1.f;
                    if (hk && a.rs.training && !a.rs.mask_from_y) {
                        h.k = dropout_scale4((uint64_t)slab * (((uint64_t)N * CIN) >> 2) + (((uint64_t)rc * CIN + 4 * cq) >> 2), a.rs.seed, hoff,
                                             a.rs.thresh, a.rs.keep_scale);
                    }
                }
            };
            // finish output step t: dx tile from LDS -> global (16-byte rows) + hook row partials
            auto F = [&](int t, const Hook& h) __attribute__((always_inline)) {
                if (cq < CIN / 4) {
                    const f32x4 v = et_round4<ET>(ld4(Xo + (t & 1) * 16 * LDO + r * LDO + 4 * cq));   // (the hook below sees what the tensor holds)
                    if (rv) stx4_wt2(dx_ + (((size_t)b * T + t) * N + n0 + r) * CIN + 4 * cq, v);
                    if (hk) {   // uniform
                        float2 p = make_float2(0.f, 0.f);
                        if (rv) {
                            const f32x4 hy = cvt4(h.y);
                            f32x4 kk = h.k;
                            if (a.rs.training && a.rs.mask_from_y) {   // (uniform) dropped iff the block output is -0.0 (drop_encode): no Philox in the step
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float yi = hy[i];
                                    kk[i] = drop_kept(yi) ? a.rs.keep_scale : 0.f;
                                }
                            }
#pragma unroll
                            // (explicit fused multiply-adds: F is inlined twice -- in the step loop and behind it -- and the two copies must round alike, or the
                            //  row partials of a range's LAST step differ from the others' by an ulp and the step's result depends on where the ranges were cut,
                            //  i.e. on the batch size: seen as a 1e-5 drift between chained micro-batches after AdamW had normalised near-zero LayerNorm gradients)
                            for (int i = 0; i < 4; ++i) {
                                p.x = __builtin_fmaf(v[i] * kk[i], hgam[i], p.x);
                                if (kk[i] > 0.f) p.y = __builtin_fmaf(v[i], hy[i] - hbks[i], p.y);
                            }
                        }
#pragma unroll
                        for (int m = CIN / 8; m >= 1; m >>= 1) {
                            p.x += __shfl_xor(p.x, m);
                            p.y += __shfl_xor(p.y, m);
                        }
                        if (rv && cq == 0) a.rs.rowstat[((long)b * T + t) * N + n0 + r] = p;
                    }
                } else if (hk) {   // keep the shuffles of partially used waves convergent (CIN < 64)
                    float2 p = make_float2(0.f, 0.f);
#pragma unroll
                    for (int m = CIN / 8; m >= 1; m >>= 1) {
                        p.x += __shfl_xor(p.x, m);
                        p.y += __shfl_xor(p.y, m);
                    }
                }
            };
            // ---- prologue of the range: tiles t_lo .. min(sb, T1 - 1) and the x tiles of the first weight-gradient step ----------
            // every tile the range start needs (the Kt - 1 tiles in front of it, tile sb and tile sb + 1) is requested at once, BEFORE the
            // barrier that frees the ring: fetching them one by one put three dependent memory latencies in front of every range
            STGCN_ACC2_BEGIN();
            Tile pr[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) fetch(t_lo + k, pr[k]);
            Raw4<ET> xs[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) xs[k] = get_x(sb + k);
            // dA tiles t_lo .. t_lo + KT: one float4 per thread (wave k stages tile t_lo + k).  The ring is free: every E thread passed
            // the last two barriers of the previous range after its last read of it
            put_dA(t_lo + w, lane, get_dA(t_lo + w, lane));
            __syncthreads();   // (A0) the previous range is fully consumed (ring, H, dA, dx tiles free); the dA tiles visible
#pragma unroll
            for (int k = 0; k < KT; ++k) put_x(sb + k, xs[k]);
            // Two register sets of step operands, A for the steps sb, sb + 2, .. and B for sb + 1, sb + 3, ..: what step i consumes was
            // requested in step i - 2 (the loop is unrolled by two, the sets are named: a rotation through copies would wait for the younger
            // set at the copy).  One step of cover was enough for the fp32 steps (4.7 us); a bf16 step is 1 us and waited for its loads:
            // without them the C3 launch takes 66 of its 105 us (r3-33).
            struct Pre { Tile p; Raw4<ET> x, da; Hook h; };
            Pre A, B;
            fetch(t_lo + KT, A.p);     // tile min(sb, T1 - 1) + 1: the next one E will need (one of the batch when the range starts early)
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int t = t_lo + k;
                if (t <= sb && t < T1) E(t, pr[k]);                     // uniform
                else if (t == (sb < T1 ? sb : T1 - 1) + 1) A.p = pr[k];  // uniform
            }
            // (requests in the order of the step loop -- all of A, then all of B, within a set p, da, x, h -- so that the wait counts the
            //  compiler derives for the loop entry equal those of the back edge)
            A.da = get_dA(sb + 2, tid & 63);
            A.x = get_x(sb + KT);
            hook_fetch(sb, A.h);       // (step sb has no F: requested for the symmetry of the counts, never read)
            fetch(sb + 2, B.p);
            B.da = get_dA(sb + 3, tid & 63);
            B.x = get_x(sb + KT + 1);
            hook_fetch(sb, B.h);       // F(sb) runs in step sb + 1
            STGCN_ACC2_END();
            // step i: finish dx tile i - 1 (F), gate backward of tile i + 1 (E), stage dA tile i + 2 and x tile i + KT, request what step i + 2 needs
            auto step = [&](int i, Pre& s) __attribute__((always_inline)) {
                __syncthreads();   // (B) tile i (dZ1, H, x, dA) visible to the M waves; dx tile i - 1 visible to the E waves
                STGCN_ACC_BEGIN();
                if (i > sb) F(i - 1, s.h);
                if (i + 1 < t_hi) E(i + 1, s.p);
                if (tid < 64 && i + 2 > t_lo + KT) put_dA(i + 2, tid, s.da);   // slot (i + 2) % RING (tile i - 2 lived there: last read in step i - 2;
                                                                                 //  tiles up to t_lo + KT were staged by the prologue)
                put_x(i + KT, s.x);      // slot (i + KT) % RING = (i - 1) % RING: last read by step i - 1; tiles beyond T are zeros
                fetch(i + 3, s.p);
                s.da = get_dA(i + 4, tid & 63);
                s.x = get_x(i + KT + 2);
                hook_fetch(i + 1, s.h);
                STGCN_ACC_END();
            };
            // (the first step is peeled: the loop is then entered and re-entered in the same state -- step A just done -- and the wait counts
            //  the compiler derives at its header are the exact ones of the back edge instead of vmcnt(0))
            int i = sb;
            step(i, A);
            for (++i; i + 1 < se; i += 2) {
                step(i, B);
                step(i + 1, A);
            }
            if (i < se) step(i, B);
            __syncthreads();       // (C) last dx tile visible
            if ((se - sb) & 1) F(se - 1, B.h);
            else F(se - 1, A.h);
        }
        STGCN_PHASE(10, 4);
        STGCN_ACC_STORE(10, 8, tid == 0);
        STGCN_ACC2_STORE(10, 11, tid == 0);
        // ---- partials of the E role: db_eff1 (a wave owns its 16 channels: the 16 rows are the lanes of a group), dba (4 waves through LDS)
        __syncthreads();           // (D) every role is done with the LDS tiles: Zt becomes the reduction buffer
        float* bred = Zt;          // [NC] db_eff1, then at 4 * NC: [4 waves][16] dba
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = dbu[i], y = dbq[i];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                x += __shfl_xor(x, m);
                y += __shfl_xor(y, m);
            }
            if (l15 == 0) {
                bred[4 * ecq + i] = x;
                bred[C0 + 4 * ecq + i] = y;
            }
        }
        {   // dba[j]: a thread's partial belongs to quad (lane & 3) of some rows: lanes with equal quad are 4 apart, then 4 waves through LDS
            f32x4 v = dba;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = v[i];
                x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
                v[i] = x;
            }
            if (lane < 4) st4(bred + 4 * NC + w * 16 + 4 * lane, v);
        }
        __syncthreads();           // (E)
        if (tid < NC) part[(size_t)KT * CIN * NC + tid] = bred[tid];
        if (tid < 16) part[(size_t)KT * CIN * NC + NC + C0 * 16 + tid] = (bred[4 * NC + tid] + bred[4 * NC + 16 + tid]) + (bred[4 * NC + 32 + tid] + bred[4 * NC + 48 + tid]);
        STGCN_PHASE(10, 6);
    } else if (role == 1) {
        // =========================================== Mw waves: weight gradients ==========================================
        f32x4 accw[KT * MI][2], acca = zero4();
#pragma unroll
        for (int m = 0; m < KT * MI; ++m) {
            accw[m][0] = zero4();
            accw[m][1] = zero4();
        }
        STGCN_ACC_DECL();
        for (long item = item0; item <= item1 && item < items; ++item) {
            const int sb = item == item0 ? s0 : 0, se = item == item1 ? s1 : T;
            if (sb >= se) continue;
            __syncthreads();   // (A0)
            for (int i = sb; i < se; ++i) {
                __syncthreads();   // (B)
                STGCN_ACC_BEGIN();
                if (i < T1) {
                  if constexpr (X6 && CIN == 64 && KT == 3 && STGCN_ON_DEVICE && STGCN_MW_ASM) {
#if defined(__HIP_DEVICE_COMPILE__)
                    static_assert(!(X6 && CIN == 64) || (XPL * 2 == 2560 && LDZHM * 2 == 544), "immediate offsets of STGCN_TC1BWD_MW_STEP_ASM");
                    const int trow = 4 * g + (l15 >> 2), tq = l15 & 3;   // the lane's address in a transposing read: row, 8-byte group
                    const short* const Zp = Zh + (size_t)(i % RING) * ZSLOT;
#define STGCN_C2(f) [c##f##_0] "+v"(accw[f][0]), [c##f##_1] "+v"(accw[f][1])
                    asm volatile(STGCN_TC1BWD_MW_STEP_ASM
                                 : STGCN_C2(0), STGCN_C2(1), STGCN_C2(2), STGCN_C2(3), STGCN_C2(4), STGCN_C2(5), STGCN_C2(6), STGCN_C2(7), STGCN_C2(8), STGCN_C2(9),
                                   STGCN_C2(10), STGCN_C2(11)
                                 : [zh] "v"(lds_addr(Zp + trow * LDZHM + 8 * (8 * w + tq))), [zl] "v"(lds_addr(Zp + ZLOFS + trow * LDZL + 4 * (8 * w + tq))),
                                   [x0] "v"(lds_addr(XTh + (size_t)(i % RING) * 3 * XPL + trow * LDXR + 4 * tq)),
                                   [x1] "v"(lds_addr(XTh + (size_t)((i + 1) % RING) * 3 * XPL + trow * LDXR + 4 * tq)),
                                   [x2] "v"(lds_addr(XTh + (size_t)((i + 2) % RING) * 3 * XPL + trow * LDXR + 4 * tq))
                                 : "memory", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151",
                                   "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167");
#undef STGCN_C2
#endif
                  } else if constexpr (X6) {
                    // B[k = row 4g + s][n = o]: the lane's 4 rows of one column, gathered from each plane (2-byte reads)
                    const int o0 = (2 * w) * 16 + l15;   // column o of a row: element o & 3 of the h quad of group o >> 2 (m: + 4); columns o0 and o0 + 16
                    const short* const Zp = Zh + (size_t)(i % RING) * ZSLOT + (4 * g) * LDZHM + 8 * (o0 >> 2) + (o0 & 3);
                    const short* const Zl = Zh + (size_t)(i % RING) * ZSLOT + ZLOFS + (4 * g) * LDZL + o0;
                    Frag3 fz0, fz1;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        fz0.h[s] = Zp[s * LDZHM]; fz0.m[s] = Zp[s * LDZHM + 4]; fz0.l[s] = Zl[s * LDZL];
                        fz1.h[s] = Zp[s * LDZHM + 32]; fz1.m[s] = Zp[s * LDZHM + 36]; fz1.l[s] = Zl[s * LDZL + 16];
                    }
                    const bf16x8 z0lh = cat8(fz0.l, fz0.h), z0mm = cat8(fz0.m, fz0.m), z0hh = cat8(fz0.h, fz0.h);
                    const bf16x8 z1lh = cat8(fz1.l, fz1.h), z1mm = cat8(fz1.m, fz1.m), z1hh = cat8(fz1.h, fz1.h);
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const short* xt = XTh + (size_t)((i + k) % RING) * 3 * XPL;
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) {   // A[m = ch][k = row]
                            const Frag3 fa = {ld_tr4(xt + mi * 16, LDXR, g, l15), ld_tr4(xt + XPL + mi * 16, LDXR, g, l15), ld_tr4(xt + 2 * XPL + mi * 16, LDXR, g, l15)};
                            const bf16x8 ahl = cat8(fa.h, fa.l), ahm = cat8(fa.h, fa.m);
                            f32x4& c0 = accw[k * MI + mi][0];
                            f32x4& c1 = accw[k * MI + mi][1];
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahl, z0lh, c0, 0, 0, 0);
                            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahl, z1lh, c1, 0, 0, 0);
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, z0mm, c0, 0, 0, 0);
                            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, z1mm, c1, 0, 0, 0);
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, z0hh, c0, 0, 0, 0);
                            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm, z1hh, c1, 0, 0, 0);
                        }
                    }
                  } else {
                    const float* const Zs = Zt + (i % RING) * 16 * LDZ;
                    f32x4 bz[2];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        bz[0][s] = Zs[(4 * g + s) * LDZ + (2 * w) * 16 + l15];
                        bz[1][s] = Zs[(4 * g + s) * LDZ + (2 * w + 1) * 16 + l15];
                    }
                    const typename MM::frag fz0 = MM::cvt(bz[0]), fz1 = MM::cvt(bz[1]);
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float* xt = XT + (size_t)((i + k) % RING) * CIN * LDX + l15 * LDX + 4 * g;
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)   // A[m = ch][k = row]
                            MM::mma_b2(MM::cvt(ld4(xt + mi * 16 * LDX)), fz0, fz1, accw[k * MI + mi][0], accw[k * MI + mi][1]);
                    }
                  }
                    // dWa[i0 = 16w + ..][j] += H^T dA : A[m = ch][k = row] = Ht[row][16w + l15], B[k = row][n = j] = dA[row][j]
                    const float* hh = Ht + (i & 1) * 16 * LDH + (4 * g) * LDH + 16 * w + l15;
                    const float* dd = dAe + (i % RING) * 256 + (4 * g) * 16 + l15;
                    acca = MM::mma(MM::cvt(gather4(hh, LDH)), MM::cvt(gather4(dd, 16)), acca);
                }
                STGCN_ACC_END();
            }
            __syncthreads();       // (C)
        }
        STGCN_PHASE(10, 5);
        STGCN_ACC_STORE(10, 9, tid == 0);
#pragma unroll
        for (int m = 0; m < KT * MI; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                st4_wt(part + (size_t)((2 * w + j) * 16 + l15) * (KT * CIN) + m * 16 + 4 * g, accw[m][j]);   // TRANSPOSED partial [NC][KT * CIN] (16-byte write-through stores)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) part[(size_t)KT * CIN * NC + NC + (16 * w + 4 * g + rr) * 16 + l15] = acca[rr];
        __syncthreads();           // (D)
        __syncthreads();           // (E)
    } else {
        // =========================================== Md waves: transposed conv ===========================================
        // wave w < MI owns input channels 16w .. 16w+15: A[m = ci][k = o] = W_eff1[(tap, ci)][o], the whole K = KT * NC in registers
        typename MM::frag Wr[X6 ? 1 : KT][X6 ? 1 : QD];
        s16x8 Whm[X6 ? KT : 1][X6 ? QD : 1];   // (X6) hi | mid planes in registers (one 32-deep A operand as is), the low plane in this wave's LDS region
        short* const Wlw = Wl + (size_t)w * KT * QD * 256 + lane * 4;
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int q = 0; q < QD; ++q) {
                const f32x4 wv = w < MI ? ld4(a.Wd + (size_t)(k * CIN + 16 * w + l15) * NC + 16 * q + 4 * g) : zero4();
                if constexpr (X6) {
                    const Frag3 f = split3(wv);
                    Whm[k][q] = s16x8{f.h[0], f.h[1], f.h[2], f.h[3], f.m[0], f.m[1], f.m[2], f.m[3]};
                    *reinterpret_cast<s16x4*>(Wlw + (k * QD + q) * 256) = f.l;   // (wave-private, lane-linear: read back by the same lane)
                } else {
                    Wr[k][q] = MM::cvt(wv);
                }
            }
        STGCN_ACC_DECL();
        for (long item = item0; item <= item1 && item < items; ++item) {
            const int sb = item == item0 ? s0 : 0, se = item == item1 ? s1 : T;
            if (sb >= se) continue;
            __syncthreads();   // (A0)
            for (int i = sb; i < se; ++i) {
                __syncthreads();   // (B)
                STGCN_ACC_BEGIN();
                if (w < MI) {
                    f32x4 accd[2] = {zero4(), zero4()};
                    f32x4 accx = zero4();   // (X6) third accumulator: one per kind of product pair
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const int ts = i - k;
                        if (ts >= 0 && ts < T1) {   // uniform
                          if constexpr (X6) {
                            static_assert(!X6 || QD == 8, "the hand-scheduled tap covers 8 fragments");
                            const short* const zh = Zh + (size_t)(ts % RING) * ZSLOT + l15 * LDZHM + 8 * g;          // + 32 q: group 4 q + g of row l15
                            const short* const zl = Zh + (size_t)(ts % RING) * ZSLOT + ZLOFS + l15 * LDZL + 4 * g;   // + 16 q
#if defined(__HIP_DEVICE_COMPILE__)
#define STGCN_LO4(x) __builtin_shufflevector(x, x, 0, 1, 2, 3)
                            asm volatile(STGCN_TC1BWD_MD_TAP_ASM
                                         : [ca] "+v"(accd[0]), [cb] "+v"(accd[1]), [cc] "+v"(accx)
                                         : [zh] "v"(lds_addr(zh)), [zl] "v"(lds_addr(zl)), [wl] "v"(lds_addr(Wlw + k * QD * 256)),
                                           [w0] "v"(Whm[k][0]), [w1] "v"(Whm[k][1]), [w2] "v"(Whm[k][2]), [w3] "v"(Whm[k][3]),
                                           [w4] "v"(Whm[k][4]), [w5] "v"(Whm[k][5]), [w6] "v"(Whm[k][6]), [w7] "v"(Whm[k][7]),
                                           [l0] "v"(STGCN_LO4(Whm[k][0])), [l1] "v"(STGCN_LO4(Whm[k][1])), [l2] "v"(STGCN_LO4(Whm[k][2])), [l3] "v"(STGCN_LO4(Whm[k][3])),
                                           [l4] "v"(STGCN_LO4(Whm[k][4])), [l5] "v"(STGCN_LO4(Whm[k][5])), [l6] "v"(STGCN_LO4(Whm[k][6])), [l7] "v"(STGCN_LO4(Whm[k][7]))
                                         : "memory", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155",
                                           "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167");
#undef STGCN_LO4
#else
#pragma unroll
                            for (int q = 0; q < QD; ++q) {   // (host emulator: the same products into the same accumulators)
                                const s16x8 bhm = *reinterpret_cast<const s16x8*>(zh + 32 * q);
                                const s16x4 bl = *reinterpret_cast<const s16x4*>(zl + 16 * q);
                                const s16x4 wl = *reinterpret_cast<const s16x4*>(Wlw + (k * QD + q) * 256);
                                const s16x8 whm = Whm[k][q];
                                const s16x8 whl = {whm[0], whm[1], whm[2], whm[3], wl[0], wl[1], wl[2], wl[3]};
                                const s16x8 blh = {bl[0], bl[1], bl[2], bl[3], bhm[0], bhm[1], bhm[2], bhm[3]};
                                const s16x8 bmh = {bhm[4], bhm[5], bhm[6], bhm[7], bhm[0], bhm[1], bhm[2], bhm[3]};
                                accd[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, whl), __builtin_bit_cast(bf16x8, blh), accd[0], 0, 0, 0);
                                accd[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, whm), __builtin_bit_cast(bf16x8, bhm), accd[1], 0, 0, 0);
                                accx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, whm), __builtin_bit_cast(bf16x8, bmh), accx, 0, 0, 0);
                            }
#endif
                          } else {
                            const float* zr = Zt + (ts % RING) * 16 * LDZ + l15 * LDZ + 4 * g;
#pragma unroll
                            for (int q = 0; q < QD; ++q) {
                                MM::mma_split(Wr[k][q], MM::cvt(ld4(zr + 16 * q)), accd[0], accd[1]);   // B[k = o][n = row]
                            }
                          }
                        }
                    }
                    if constexpr (X6) accd[0] += accx;
                    st4(Xo + (i & 1) * 16 * LDO + l15 * LDO + 16 * w + 4 * g, accd[0] + accd[1]);   // D[m = ci = 16w + 4g + r][n = row]
                }
                STGCN_ACC_END();
            }
            __syncthreads();       // (C)
        }
        STGCN_ACC_STORE(10, 10, tid == 0);
        __syncthreads();           // (D)
        __syncthreads();           // (E)
    }
}
template <int C0, int CIN, int KT, int ACT, typename ET>
__global__ __launch_bounds__(768) void tc1_bwd_kernel(Tc1BwdArgs a) {
    tc1_bwd_body<C0, CIN, KT, ACT, ET>(a);
}
template <int C0, int CIN, int KT, int ACT>
__global__ __launch_bounds__(768) void tc1_bwd_x6_kernel(Tc1BwdArgs a) {
    tc1_bwd_body<C0, CIN, KT, ACT, float, true>(a);
}

// ================================================================================================
// F1 (time-stepping): tmp_conv1 + gate + Align(c0 -> c1) forward on 16-node tiles of one window (layers.py:252, :223).
//   Z^T[o][row] = W_eff1^T[o][K] im2col(x)^T[K][row] per output step: A = the packed weights (PK_TCONV_FWD fragments), the whole
//   K = Kt * c_in of a wave's two o-tiles (P and Q half of 16 channels) stationary in registers; B = x tiles from an LDS ring (every
//   input tile is read from memory once and serves Kt output steps).  D leaves a lane with P and Q of 4 channels of one row: bias,
//   sigmoid, gate and the 16-byte U1 / S1 stores run on registers, and the lane's h values are already in B-operand layout for the
//   Align product A^T[j][row] += Wa^T[j][i] h^T[i][row] over the wave's own 16 channels (4 MFMAs, no LDS round trip); the four waves'
//   partial A tiles are summed by the E waves one step later.
// 8 waves: 4 M waves (MFMA + gate epilogue) | 4 E waves (x tile loads -> ring, A = sum of partials + bias -> memory); one barrier / step.
// Work distribution as in tc1_bwd_kernel: the B * ceil(N/16) (window, node tile) items do not fill 256 CUs evenly (416 at C2), so the
// sequence of (item, output step) units -- all of equal MFMA weight here -- is cut into gridDim.x equal ranges; a range that starts
// inside an item only re-reads the Kt - 1 input tiles in front of it (no arithmetic is repeated).
// ================================================================================================
struct Tc1FwdArgs {
    const float* x;           // [B][T][N][CIN]
    const float* Wp;          // packed W_eff1 (PK_TCONV_FWD): KCH = Kt*CIN/16 chunks, NC = 2*C0 columns
    const float* bias;        // b_eff1 [2*C0]
    const float* WaD;         // [C0][16] dense Align map (PK_ALIGN_DENSE)
    const float* ba;          // [16] Align bias (PK_ALIGN_BIAS)
    float* U;                 // [B][T1][N][C0]
    float* S;
    float* A;                 // [B][T1][N][16]
    int B, T, T1, N, node_tiles;
    int chain_out;            // chained launch: counter chain_out + b * T1 + t counts the node tiles of A[b][t] written (-1: none)
};
inline size_t tc1_fwd_lds_bytes(int CIN, int Kt, bool x6 = false) {
    // x tiles: fp32 rows, or (X6) three bf16 planes per tile; then the partial Align tiles
    const size_t ring = x6 ? (size_t)(Kt + 1) * 3 * 16 * (CIN + 4) * sizeof(short) : (size_t)(Kt + 1) * 16 * (CIN + 8) * sizeof(float);
    return ring + (size_t)2 * 4 * 16 * 20 * sizeof(float);
}

// bid / nb: this workgroup's index among the nb workgroups of the role (the kernel's own grid, or the role's share of a chained launch);
// chain.words != null: every A tile is published on counter chain_out + b * T1 + t (node_tiles arrivals complete a slab)
// X6 (round 6, fp32 blocks only): the conv product as "bf16x6" -- fp32-accurate products on the bf16 matrix pipe (Frag3, stgcn_device.hip.h).
// The E waves split every x tile into three bf16 planes when they stage it (once per tile: it then serves KT taps of four M waves); the M
// waves split their stationary weights once; the product loop reads plane triples and issues 3 x v_mfma_f32_16x16x32_bf16 per 16-deep step
// instead of 4 x v_mfma_f32_16x16x4_f32, with no conversion in the loop.  The small Align product stays on fp32 MFMAs.
template <int C0, int CIN, int KT, int ACT, typename ET, bool X6 = false>
__device__ __forceinline__ void tc1_fwd_body(const Tc1FwdArgs& a, const int bid, const int nb, const ChainCtl& chain) {
    static_assert(C0 == 64 && (CIN == 16 || CIN == 32 || CIN == 64), "shapes covered by the role split below");
    static_assert(!X6 || std::is_same<ET, float>::value, "bf16x6 is a product form of the fp32 blocks");
    typedef Mma<ET> MM;
    const ET* const x_ = et_ptr<ET>(a.x);
    ET* const U_ = et_ptr<ET>(a.U);
    ET* const S_ = et_ptr<ET>(a.S);
    ET* const A_ = et_ptr<ET>(a.A);
    constexpr int RING = KT + 1, LDXS = CIN + 8, CC = CIN / 16, KCH = KT * CC, MT = C0 / 16, RED = 4 * 16 * 20;
    extern __shared__ float stgcn_smem[];
    constexpr int LDH = CIN + 4, PST = 16 * LDH, SLOT3 = 3 * PST;   // (X6) row stride / plane stride / ring-slot stride in shorts
    float* const Xs = stgcn_smem;                      // [RING][16][LDXS]  x tiles, row major
    short* const Xh = reinterpret_cast<short*>(stgcn_smem);   // (X6) [RING][3 planes][16][LDH] bf16
    float* const red = X6 ? reinterpret_cast<float*>(Xh + RING * SLOT3) : Xs + RING * 16 * LDXS;   // [2][4 waves][16 rows][20]  partial Align tiles, double buffered
    const bool roleM = threadIdx.x < 256;              // wave-uniform
    const int tid = threadIdx.x & 255, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int N = a.N, T = a.T, T1 = a.T1;
    // this workgroup's range of the (item, step) sequence (identical in both roles): units [u_lo, u_hi), item = b * node_tiles + tile
    const long items = (long)a.B * a.node_tiles, units = items * T1;
    const long u_lo = units * (long)bid / (long)nb, u_hi = units * ((long)bid + 1) / (long)nb;
    const long item0 = u_lo / T1, item1 = u_hi / T1;
    const int s0 = (int)(u_lo - item0 * T1), s1 = (int)(u_hi - item1 * T1);
    const bool chained = chain.words != nullptr && a.chain_out >= 0;
    STGCN_PHASE(11, 0);
#ifdef STGCN_PHASE_TIMING   // (diagnostic build: thread 0 = M wave 0; cycles of MFMA issue / epilogue / barrier wait / item transitions, tools/gpu_phases.py kid 11)
    long long pt_mma = 0, pt_epi = 0, pt_wait = 0, pt_a = 0, pt_b = 0, pt_c = 0, pt_steps = 0, pt_trans = 0;
#endif

    if (roleM) {
        // stationary weights: A[m = o][k] fragments of o-tiles w (P half) and w + MT (Q half)
        typename MM::frag wP[X6 ? 1 : KCH], wQ[X6 ? 1 : KCH];
        Frag3 wP3[X6 ? KCH : 1], wQ3[X6 ? KCH : 1];
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            const f32x4 vp = ld4(a.Wp + ((size_t)(w * KCH + kc) * 64 + lane) * 4), vq = ld4(a.Wp + ((size_t)((w + MT) * KCH + kc) * 64 + lane) * 4);
            if constexpr (X6) {
                wP3[kc] = split3(vp);
                wQ3[kc] = split3(vq);
            } else {
                wP[kc] = MM::cvt(vp);
                wQ[kc] = MM::cvt(vq);
            }
        }
        const int c = 16 * w + 4 * g;                  // this lane's 4 channels
        const f32x4 bp = ld4(a.bias + c), bq = ld4(a.bias + C0 + c);
        f32x4 waT_;                                    // A[m = j = l15][k = i = 16w + 4g + s] = Wa[i][j]
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) waT_[sI] = a.WaD[(size_t)(c + sI) * 16 + l15];
        const typename MM::frag waT = MM::cvt(waT_);
        for (long item = item0; item <= item1 && item < items; ++item) {
            const int sb = item == item0 ? s0 : 0, se = item == item1 ? s1 : T1;
            if (sb >= se) continue;                    // (uniform over the workgroup: both roles evaluate the same list)
            const int b = (int)(item / a.node_tiles), n0 = (int)(item - (long)b * a.node_tiles) * 16;
            const bool rowv = n0 + l15 < N;
            __syncthreads();   // (A) x tiles sb .. sb + KT - 1 of this item staged
#ifdef STGCN_PHASE_TIMING
            if (pt_c == 0) STGCN_PHASE(11, 1);       // end of the first prologue
            else pt_trans += clock64() - pt_c;       // item transitions: barrier (C) .. barrier (A)
            pt_c = clock64();
#endif
            for (int i = sb; i < se; ++i) {
                __syncthreads();   // (B) x tile i + KT - 1 visible; partial tiles of step i - 1 visible to the E waves
#ifdef STGCN_PHASE_TIMING
                pt_a = clock64();
                pt_wait += pt_a - pt_c;
                ++pt_steps;
#endif
                f32x4 accP = zero4(), accQ = zero4();
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    const int tap = kc / CC, cc = kc % CC;
                    // B[k = ci][n = row]
                    if constexpr (X6) mma3_a2(wP3[kc], wQ3[kc], ld_frag3(Xh + (size_t)((i + tap) % RING) * SLOT3 + l15 * LDH + cc * 16 + 4 * g, PST), accP, accQ);
                    else MM::mma_a2(wP[kc], wQ[kc], MM::cvt(ld4(Xs + (size_t)((i + tap) % RING) * 16 * LDXS + l15 * LDXS + cc * 16 + 4 * g)), accP, accQ);
                }
#ifdef STGCN_PHASE_TIMING
                __builtin_amdgcn_sched_barrier(0);
                pt_b = clock64();
                __builtin_amdgcn_sched_barrier(0);
                pt_mma += pt_b - pt_a;
#endif
                f32x4 u, sg, h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[e] = accP[e] + bp[e];
                    sg[e] = sigmoid_f(accQ[e] + bq[e]);
                    h[e] = gate_fwd(u[e], sg[e], ACT);
                }
                if (rowv) {
                    const size_t o = (((size_t)b * T1 + i) * N + n0 + l15) * C0 + c;
                    stx4_wt2(U_ + o, u);
                    stx4_wt2(S_ + o, sg);
                }
                // this wave's share of A^T[j][row]: its 16 channels of the K = C0 contraction
                const f32x4 pa = MM::mma(waT, MM::cvt(h), zero4());
                st4(red + (i & 1) * RED + (w * 16 + l15) * 20 + 4 * g, pa);   // D[m = j = 4g + r][n = row = l15]
#ifdef STGCN_PHASE_TIMING
                __builtin_amdgcn_sched_barrier(0);
                pt_c = clock64();
                pt_epi += pt_c - pt_b;
#endif
            }
            __syncthreads();       // (C) last partial tiles visible
#ifdef STGCN_PHASE_TIMING
            pt_wait += clock64() - pt_c;   // (the wait at (C) counts as barrier wait)
            pt_c = clock64();
#endif
        }
        STGCN_PHASE(11, 2);
#ifdef STGCN_PHASE_TIMING
        if (stgcn_phase_kid == 11 && threadIdx.x == 0 && blockIdx.x < 4096) {
            stgcn_phase_buf[blockIdx.x * 16 + 8] = pt_mma;
            stgcn_phase_buf[blockIdx.x * 16 + 9] = pt_epi;
            stgcn_phase_buf[blockIdx.x * 16 + 10] = pt_wait;
            stgcn_phase_buf[blockIdx.x * 16 + 11] = pt_steps;
            stgcn_phase_buf[blockIdx.x * 16 + 12] = pt_trans;
        }
#endif
    } else {
        // =========================================== E waves ===========================================================
        const int r = tid >> 4, cq = tid & 15;         // x tiles: row r, float4 column cq (< CIN / 4)
        const int fr = lane >> 2, fq = lane & 3;       // A tiles (first E wave only): row fr, channels 4 fq .. 4 fq + 3
        const f32x4 bj = ld4(a.ba + 4 * fq);
        for (long item = item0; item <= item1 && item < items; ++item) {
            const int sb = item == item0 ? s0 : 0, se = item == item1 ? s1 : T1;
            if (sb >= se) continue;
            const int b = (int)(item / a.node_tiles), n0 = (int)(item - (long)b * a.node_tiles) * 16;
            const bool rv = n0 + r < N;
            const int rc = rv ? n0 + r : N - 1;
            auto get_x = [&](int xt) __attribute__((always_inline)) {
                const int xc = xt < T ? xt : T - 1, qc = cq < CIN / 4 ? cq : CIN / 4 - 1;
                return ldraw4(x_ + (((size_t)b * T + xc) * N + rc) * CIN + 4 * qc);   // (raw: converted and masked by put_x, see tc1_bwd_kernel)
            };
            auto put_x = [&](int xt, const Raw4<ET>& raw) {
                const f32x4 v = (rv && xt < T) ? cvt4(raw) : zero4();
                if constexpr (X6) {
                    if (cq < CIN / 4) st_frag3(Xh + (size_t)(xt % RING) * SLOT3 + r * LDH + 4 * cq, PST, split3(v));   // split ONCE, where the tile is staged
                } else {
                    if (cq < CIN / 4) st4(Xs + (size_t)(xt % RING) * 16 * LDXS + r * LDXS + 4 * cq, v);
                }
            };
            // A[t] = sum of the 4 waves' partial tiles + bias: ONE wave, 16 bytes per lane, written through (round 3: 256 scalar stores).  In a
            // chained launch the wave then drains its stores and bumps the slab's arrival counter: the graph conv of slab (b, t) starts when
            // all node tiles have arrived, while this workgroup walks on.
            auto F = [&](int t) {
                if (tid >= 64) return;   // (wave-uniform)
                const float* rd = red + (t & 1) * RED + fr * 20 + 4 * fq;
                const f32x4 v = ((ld4(rd) + ld4(rd + 16 * 20)) + (ld4(rd + 2 * 16 * 20) + ld4(rd + 3 * 16 * 20))) + bj;
                if (n0 + fr < N) stx4_wt(A_ + (((size_t)b * T1 + t) * N + n0 + fr) * 16 + 4 * fq, v);
                if (chained) {
                    chain_drain_stores();
                    if (lane == 0) chain_publish(chain, a.chain_out + b * T1 + t);
                }
            };
            Raw4<ET> xs[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) xs[k] = get_x(sb + k);
#pragma unroll
            for (int k = 0; k < KT; ++k) put_x(sb + k, xs[k]);   // (the previous range's M steps are over: every role passed its barrier (C))
            Raw4<ET> xA = get_x(sb + KT), xB = get_x(sb + KT + 1);   // x tiles requested two steps ahead, in two named sets (see tc1_bwd_kernel)
            __syncthreads();   // (A)
            auto step = [&](int i, Raw4<ET>& xs) __attribute__((always_inline)) {
                __syncthreads();   // (B)
                if (i > sb) F(i - 1);
                put_x(i + KT, xs);           // slot (i + KT) % RING = (i - 1) % RING: last read by step i - 1
                xs = get_x(i + KT + 2);
            };
            int i = sb;
            step(i, xA);
            for (++i; i + 1 < se; i += 2) {
                step(i, xB);
                step(i + 1, xA);
            }
            if (i < se) step(i, xB);
            __syncthreads();       // (C)
            F(se - 1);
        }
    }
}
template <int C0, int CIN, int KT, int ACT, typename ET>
__global__ __launch_bounds__(512) void tc1_fwd_kernel(Tc1FwdArgs a) {
    tc1_fwd_body<C0, CIN, KT, ACT, ET>(a, (int)blockIdx.x, (int)gridDim.x, ChainCtl{nullptr, 0, 0u});
}
template <int C0, int CIN, int KT, int ACT>
__global__ __launch_bounds__(512) void tc1_fwd_x6_kernel(Tc1FwdArgs a) {
    tc1_fwd_body<C0, CIN, KT, ACT, float, true>(a, (int)blockIdx.x, (int)gridDim.x, ChainCtl{nullptr, 0, 0u});
}
// (Round 4, pass r4-06: the same body under __launch_bounds__(512, 4) -- 128 VGPRs, two workgroups REALLY sharing a CU; the plain fp32
//  CIN = 64 instance takes 133 VGPRs = 3 waves per SIMD, so "two per CU" had been two rounds of one -- measured 28.5 -> 30.5 us: the fp32
//  matrix pipe of the CU is the limit, a second chain only doubles the prologues.  The instance was removed again;
//  profiles/r4-06_tc1_fwd_two_per_cu.txt.)

// ================================================================================================
// F3+F4 fused: tmp_conv2 + GLU/GTU + LayerNorm([N, c2]) + Dropout of ONE (b, t2) slab per workgroup (layers.py:254-256).
// LayerNorm normalises over all N * c2 values of a slab, so the slab is the natural owner: the conv output never goes to
// memory before it is normalised (the stage-per-launch path wrote U2, S2 and row partials, then re-read U2, S2 in ln_norm_kernel).
//   Z^T[o][row] = W_eff2^T[o][K] im2col(G)^T[K][row]     A = the packed weights (PK_TCONV_FWD fragments are A fragments of the
//                                                        transposed product), stationary in registers; B = G rows from LDS
//   D leaves a lane with 4 consecutive channels of one row, and with wave w owning o-tiles w and w + c2/16 the P and Q halves of a
//   channel meet in the same lane: bias, sigmoid, gate, the two-pass slab statistics (values stay in registers), the affine map,
//   the Philox mask and the 16-byte U2 / S2 / y stores all run on registers -- no LDS round trip of the accumulators.
// grid = B * T2 workgroups of 256 * HV threads (4 * HV waves: wave = (pair p = wave & 3, group hf = wave >> 2); group hf takes the
// 16-row tiles hf, hf + HV, ...).  Template: C2 = 64 (p owns channels 16p..16p+15), KT taps, NTI row tiles per wave (N <= 16 * HV * NTI),
// HV = 2 or 4 tile groups: with 4 (16 waves, one workgroup per CU) four waves per SIMD interleave their MFMA runs and VALU epilogues.
// ================================================================================================
struct Tc2LnFwdArgs {
    const float* G;        // [B][T1][N][16]
    const float* Wp;       // packed W_eff2 (PK_TCONV_FWD): K = Kt*16 (KCH = Kt chunks), NC = 2*C2 columns
    const float* bias;     // b_eff2 [2*C2]
    const float* gamma;    // [N][C2]
    const float* beta;
    float* U;              // [B*T2*N][C2]  gate inputs, or null: not stored (tc2_bwd_kernel recomputes them)
    float* S;
    float* y;              // [B*T2*N][C2]
    float* mean;           // [B*T2]
    float* rstd;
    int T1, T2, N, NPR, act, training;   // NPR = roundup16(N)
    int chain_in;          // chained launch: counter chain_in + b * T1 + t counts the parts of G[b][t] written (chain_expect of them); -1: none
    unsigned chain_expect;
    float eps, keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
    ChainCtl peer;                 // PP > 1: ticket / finished / sticky words of the launch (header only, ncount = 0)
    unsigned long long* peer_slots;   // PP > 1: [slabs][PP] exchange words (peer_word), zero at launch start
};
constexpr int kLdG = 24;   // row stride of the staged G tiles: stride / 4 = 6 spreads the 16 lanes of a ds_read_b128 service group over all banks
constexpr int kLdGh = 20;  // (X6) row stride, in shorts, of a staged bf16 plane of G (16 channels + 4: 8-byte rows 40 bytes apart)
inline size_t tc2_ln_fwd_lds_bytes(int Kt, int N, bool x6 = false) {
    const size_t NPR = (size_t)(N + 15) / 16 * 16;
    return (x6 ? (size_t)Kt * 3 * NPR * kLdGh * sizeof(short) : (size_t)Kt * NPR * kLdG * sizeof(float)) + 64 * sizeof(float);
}

// bid = the (b, t2) slab of this workgroup; chained launch: the KT input slabs G[b][t2 + tap] are awaited on counters chain_in + b * T1 + t2 + tap
// PP > 1 (round 6): PP workgroups share a slab, each owns a contiguous range of its node tiles (the per-slab chain -- staging, tile passes,
// statistics, normalise -- is what a launch of few slabs costs, whatever the batch: 15 us per workgroup at 207 nodes, half the chip idle
// when block 1 offers 128 slabs).  Slab and part come from a start-order TICKET (peers hold consecutive tickets: a workgroup only ever
// waits for workgroups that have started or are next to start -- no residency assumption); the parts' (mean, M2) meet through one 64-bit
// word each (peer_word) and are merged in part order by every peer: bitwise the same statistics in all of them, run to run.
// X6 (round 6, fp32 blocks): the conv product as "bf16x6" (Frag3, stgcn_device.hip.h) -- the G rows are split into three bf16 planes when they
// are staged (once per slab; every row then serves the four channel-tile waves), the weights once per wave; on a SIMD whose time is the SUM
// of its fp32 MFMA and VALU cycles this takes the matrix part off the VALU lanes (a bf16 MFMA runs beside them).
template <int C2, int KT, int NTI, int HV, int PP, typename ET, bool X6 = false>
__device__ __forceinline__ void tc2_ln_fwd_body(const Tc2LnFwdArgs& a, const int bid, const ChainCtl& chain) {
    static_assert(C2 == 64, "wave pairing below assumes 4 channel tiles per half");
    static_assert(!X6 || std::is_same<ET, float>::value, "bf16x6 is a product form of the fp32 blocks");
    typedef Mma<ET> MM;
    ET* const U_ = et_ptr<ET>(a.U);
    ET* const S_ = et_ptr<ET>(a.S);
    ET* const y_ = et_ptr<ET>(a.y);
    constexpr int NC = 2 * C2, MT = C2 / 16;
    extern __shared__ float stgcn_smem[];
    float* const Gs = stgcn_smem;                          // [KT][NPR][kLdG]  (NPR = this workgroup's rows)
    short* const Gh = reinterpret_cast<short*>(stgcn_smem);   // (X6) [KT][3 planes][NPR][kLdGh] bf16
    float* const red = X6 ? reinterpret_cast<float*>(Gh + (size_t)KT * 3 * a.NPR * kLdGh)
                          : Gs + (size_t)KT * a.NPR * kLdG;   // [3 * 4 * HV] wave statistics | [2 * PP] peers' (mean, M2) | ticket word
    const int tid = threadIdx.x, wv = tid >> 6, p = wv & (MT - 1), hf = wv >> 2, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    STGCN_PHASE(9, 0);
    // stationary weights: A[m = o][k] fragments of o-tiles p (P half) and p + MT (Q half)  (requested before the ticket: one wait covers both)
    typename MM::frag wP[X6 ? 1 : KT], wQ[X6 ? 1 : KT];
    Frag3 wP3[X6 ? KT : 1], wQ3[X6 ? KT : 1];
#pragma unroll
    for (int kc = 0; kc < KT; ++kc) {
        const f32x4 vp = ld4(a.Wp + ((size_t)(p * KT + kc) * 64 + lane) * 4), vq = ld4(a.Wp + ((size_t)((p + MT) * KT + kc) * 64 + lane) * 4);
        if constexpr (X6) {
            wP3[kc] = split3(vp);
            wQ3[kc] = split3(vq);
        } else {
            wP[kc] = MM::cvt(vp);
            wQ[kc] = MM::cvt(vq);
        }
    }
    int vb = bid;
    if constexpr (PP > 1) vb = chain_enter_peer(a.peer, reinterpret_cast<unsigned*>(red + 3 * 4 * HV + 2 * PP));
    const long slab = PP > 1 ? vb / PP : vb;
    const int part = PP > 1 ? vb % PP : 0;
    const int b = (int)(slab / a.T2), t2 = (int)(slab - (long)b * a.T2), N = a.N;
    // this workgroup's node tiles [tile0, tile0 + ntiles) of the slab's NPR / 16 (balanced cut), staged as NPR = 16 * ntiles local rows
    const int alltiles = a.NPR >> 4, tile0 = PP > 1 ? part * alltiles / PP : 0, ntiles = PP > 1 ? (part + 1) * alltiles / PP - tile0 : alltiles;
    const int NPR = ntiles << 4, row0 = tile0 << 4;
    const bool cin = chain.words != nullptr && a.chain_in >= 0;

    // stage this workgroup's rows of the KT input slabs G[b][t2 + tap] (zero rows beyond N)
    const ET* Gb = et_ptr<ET>(a.G) + ((size_t)b * a.T1 + t2) * N * 16;
    if (cin) {   // (the weight loads above are in flight while the last of the KT slabs arrives)
#pragma unroll
        for (int tap = 0; tap < KT; ++tap) chain_wait(chain, a.chain_in + b * a.T1 + t2 + tap, a.chain_expect);
        for (int idx = tid; idx < KT * NPR * 4; idx += 256 * HV) {
            const int q = idx & 3, rr = (idx >> 2) % NPR, tap = (idx >> 2) / NPR, gr = row0 + rr;
            const int eo = (tap * N + (gr < N ? gr : N - 1)) * 16 + 4 * q;
            const f32x4 v = cvt4(ldraw4_sc1(Gb, (long)KT * N * 16, eo));
            if constexpr (X6) st_frag3(Gh + ((size_t)tap * 3 * NPR + rr) * kLdGh + 4 * q, NPR * kLdGh, split3(gr < N ? v : zero4()));
            else st4(Gs + ((size_t)tap * NPR + rr) * kLdG + 4 * q, gr < N ? v : zero4());
        }
    } else
    for (int idx = tid; idx < KT * NPR * 4; idx += 256 * HV) {
        const int q = idx & 3, rr = (idx >> 2) % NPR, tap = (idx >> 2) / NPR, gr = row0 + rr;
        const f32x4 v = gr < N ? ldx4(Gb + ((size_t)tap * N + gr) * 16 + 4 * q) : zero4();
        if constexpr (X6) st_frag3(Gh + ((size_t)tap * 3 * NPR + rr) * kLdGh + 4 * q, NPR * kLdGh, split3(v));   // split ONCE, where the row is staged
        else st4(Gs + ((size_t)tap * NPR + rr) * kLdG + 4 * q, v);
    }
    const int c = 16 * p + 4 * g;   // this lane's 4 channels
    const f32x4 bp = ld4(a.bias + c), bq = ld4(a.bias + C2 + c);
    __syncthreads();
    STGCN_PHASE(9, 1);

    unsigned kbits2 = 0;
    // LayerNorm parameters of this lane's elements: consumed after the statistics.  Requested before the tile passes (interleaved form) or
    // between the matrix phase and the VALU phase (phased form: the accumulators of all tiles are live during the matrix phase).
    // (Kept as an always-inline lambda: with the same loop written in place the register allocator of ROCm 7.2 spills 17 VGPRs in the
    //  16-wave variant -- 128 VGPRs + 72 bytes of scratch, 23 -> 35 us -- and with the lambda it settles at 116 VGPRs.)
    f32x4 ga[NTI], be[NTI];
    auto load_affine = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
            const int row = row0 + (hf + HV * j) * 16 + l15, rcl = row < N ? row : N - 1;
            ga[j] = ld4(a.gamma + (size_t)rcl * C2 + c);
            be[j] = ld4(a.beta + (size_t)rcl * C2 + c);
        }
    };
    // PHASED (round 6, the forms of up to 4 tiles per wave): ALL matrix products of the wave first, then the VALU work of all its tiles.
    // What the hardware does with it (tools/ubench/overlap.hip, profiles/r6-01_valu_mfma_overlap.txt): an fp32 MFMA and a VALU instruction
    // of the same SIMD never overlap -- not across waves either -- so a SIMD's time here is the SUM of its 13 tile passes' 24 MFMAs x 32
    // cycles (10 k) and their ~250 VALU instructions at ~3.5 cycles (14 k with four waves issuing), whatever the order: phase stamps of both
    // forms agree (profiles/r6-05_phases_tc2_ln_fwd.txt: matrix phase 9.1 k, VALU phase 7.6 k, wait for the SIMD's other waves 9.3 k cycles
    // for wave 0), and the launch times are equal within noise.  Kept because the phased form is what the PP > 1 instances were built and
    // tested on; the lever of this kernel is its VALU instruction count (the GTU / GLU branch below: - 0.3 us per launch).
    constexpr bool PHASED = NTI <= 4;
    if constexpr (!PHASED) load_affine();
    const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
    const uint64_t n4 = ((uint64_t)N * C2) >> 2;
    // Per row tile: MFMAs, then the gate (U = P + b, S = sigmoid(Q + b), h = act(U) * S, kept in hh) and the keep bits of the dropout mask.
    f32x4 hh[NTI];
    unsigned kbits = 0;
    float sum = 0.f, cnt_l = 0.f;
    f32x4 aP[PHASED ? NTI : 1], aQ[PHASED ? NTI : 1];
    auto tile_mma = [&](int nt, f32x4& accP, f32x4& accQ) __attribute__((always_inline)) {
        accP = zero4();
        accQ = zero4();
#pragma unroll
        for (int kc = 0; kc < KT; ++kc) {
            // B[k = 4g + s][n = row]
            if constexpr (X6) mma3_a2(wP3[kc], wQ3[kc], ld_frag3(Gh + ((size_t)kc * 3 * NPR + nt * 16 + l15) * kLdGh + 4 * g, NPR * kLdGh), accP, accQ);
            else MM::mma_a2(wP[kc], wQ[kc], MM::cvt(ld4(Gs + ((size_t)kc * NPR + nt * 16 + l15) * kLdG + 4 * g)), accP, accQ);
        }
    };
    auto tile_valu = [&](int j, int row, const f32x4& accP, const f32x4& accQ) __attribute__((always_inline)) {
        if (a.training) {
            const f32x4 k = dropout_scale4((uint64_t)slab * n4 + (((size_t)(row < N ? row : 0) * C2 + c) >> 2), a.seed, off, a.thresh, 1.0f);
            kbits |= ((k[0] > 0.f ? 1u : 0u) | (k[1] > 0.f ? 2u : 0u) | (k[2] > 0.f ? 4u : 0u) | (k[3] > 0.f ? 8u : 0u)) << (4 * (j & 7));
        }
        if (row < N) {
            f32x4 u, sg, h;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u[i] = accP[i] + bp[i];
                sg[i] = sigmoid_f(accQ[i] + bq[i]);
            }
            // (a real branch on the uniform `act`: written as a select, both gates were computed for every element -- 8 more
            //  transcendentals per lane and tile in the GLU blocks, on a SIMD whose VALU time adds to its fp32 MFMA time)
            if (a.act == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = u[i] * sg[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) h[i] = tanh_f(u[i]) * sg[i];
            }
            const size_t o = ((size_t)slab * N + row) * C2 + c;
            if (a.U) {   // (uniform) only when a consumer of the stored gate inputs exists: the stage tests, the unfused backward
                stx4_wt(U_ + o, u);
                stx4_wt(S_ + o, sg);
            }
            hh[j] = h;
            sum += (h[0] + h[1]) + (h[2] + h[3]);
            cnt_l += 4.f;
        }
    };
    if constexpr (PHASED) {
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
            aP[j] = zero4();
            aQ[j] = zero4();
            if (hf + HV * j < ntiles) tile_mma(hf + HV * j, aP[j], aQ[j]);   // uniform per wave
        }
        load_affine();
        STGCN_PHASE(9, 2);
#pragma unroll
        for (int j = 0; j < NTI; ++j) {
            hh[j] = zero4();
            if (hf + HV * j < ntiles) tile_valu(j, row0 + (hf + HV * j) * 16 + l15, aP[j], aQ[j]);
        }
    } else {
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
        const int nt = hf + HV * j, row = row0 + nt * 16 + l15;   // nt: tile of this workgroup's range (its LDS rows), row: row of the slab
        hh[j] = zero4();
        if (nt < ntiles) {   // uniform per wave
            tile_mma(nt, aP[0], aQ[0]);
            tile_valu(j, row, aP[0], aQ[0]);
        }
        if ((j & 7) == 7 && j + 1 < NTI) {   // (NTI = 14: the keep bits of the first 8 tiles move to the upper word)
            kbits2 = kbits;
            kbits = 0;
        }
    }
    }
    STGCN_PHASE(9, 3);
    // slab statistics with ONE barrier: per-wave (count, mean, M2) about the wave's own mean, merged exactly (Chan et al.)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sum += __shfl_xor(sum, m);
        cnt_l += __shfl_xor(cnt_l, m);
    }
    const float mean_w = cnt_l > 0.f ? sum / cnt_l : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
        const int row = row0 + (hf + HV * j) * 16 + l15;
        if (row < N && hf + HV * j < ntiles) {
#pragma unroll
            for (int i = 0; i < 4; ++i) m2 += (hh[j][i] - mean_w) * (hh[j][i] - mean_w);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) m2 += __shfl_xor(m2, m);
    if (lane == 0) {
        red[3 * wv] = cnt_l;
        red[3 * wv + 1] = mean_w;
        red[3 * wv + 2] = m2;
    }
    // (the barrier only publishes `red`: it must not wait for the U2 / S2 stores of the tile loop -- __syncthreads() drains vmcnt; and the
    //  merge is the closed form about the slab mean, two independent sums, instead of a 16-step chain of dependent divisions: the
    //  statistics phase was 5.6 us of a 23 us launch, r3-58)
    float nn = 0.f, mean = 0.f, M2 = 0.f;
    if constexpr (NTI == 6) {   // (the 6-tile bf16 form sits at its register limit: the chained merge, which holds fewer values)
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4 * HV; ++k) {
            const float nw = red[3 * k], mw = red[3 * k + 1], qw = red[3 * k + 2];
            if (nw > 0.f) {
                const float d = mw - mean, nt2 = nn + nw;
                mean += d * (nw / nt2);
                M2 += qw + d * d * (nn * nw / nt2);
                nn = nt2;
            }
        }
    } else {
        barrier_only();
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < 4 * HV; ++k) {
            nn += red[3 * k];
            sm += red[3 * k] * red[3 * k + 1];
        }
        mean = sm / nn;
#pragma unroll
        for (int k = 0; k < 4 * HV; ++k) {
            const float d = red[3 * k + 1] - mean;
            M2 += red[3 * k + 2] + red[3 * k] * d * d;
        }
    }
    unsigned fin = 0u;
    if constexpr (PP > 1) {
        // ---- the parts of the slab exchange their (mean, M2): thread 0 publishes this part's word, threads q < PP fetch part q's (this
        // part's own comes back from the registers: same bits), every thread merges the PP triples in part order (Chan et al.)
        float* const pr = red + 3 * 4 * HV;   // [PP][2]
        unsigned long long* const slots = a.peer_slots + (size_t)slab * PP;
        if (tid == 0 && !chain_withhold(a.peer, vb)) chain_st64(slots + part, peer_word(mean, M2));   // (test setting: the launch's first workgroup withholds its word)
        if (wv == 0) __builtin_amdgcn_wave_barrier();   // (no instruction: orders the publish before the polls for the compiler -- and for the emulator's fibers)
        if (tid < PP) {
            float pm = mean, pq = M2;
            if (tid != part) {
                const unsigned long long w = chain_poll_word(a.peer, slots + tid, (int)slab);
                pm = peer_word_mean(w);
                pq = w ? peer_word_m2(w) : __builtin_nanf("");   // a wait that gave up: NaN statistics -> NaN outputs of this part (and the sticky word)
            }
            pr[2 * tid] = pm;
            pr[2 * tid + 1] = pq;
        }
        barrier_only();
        // every poll of this workgroup is done: it counts as finished with the exchange words NOW (the reply is only looked at after the
        // stores below: the atomic's round trip stays off the critical path)
        if (tid == 0) fin = chain_add(a.peer.words + 1, 1u);
        nn = 0.f; mean = 0.f; M2 = 0.f;
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int r0 = (q * alltiles / PP) << 4, r1 = ((q + 1) * alltiles / PP) << 4;
            const float nw = (float)(((r1 < N ? r1 : N) - (r0 < N ? r0 : N)) * C2), mw = pr[2 * q], qw = pr[2 * q + 1];
            if (nw > 0.f) {
                const float d = mw - mean, nt2 = nn + nw;
                mean += d * (nw / nt2);
                M2 += qw + d * d * (nn * nw / nt2);
                nn = nt2;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(M2 / nn + a.eps);
    if (tid == 0 && part == 0) {
        a.mean[slab] = mean;
        a.rstd[slab] = rstd;
    }
    STGCN_PHASE(9, 4);
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
        const int row = row0 + (hf + HV * j) * 16 + l15;
        if (row < N && hf + HV * j < ntiles) {
            const size_t e = (size_t)row * C2 + c;
            const unsigned kb = ((NTI > 8 && j < 8) ? kbits2 : kbits) >> (4 * (j & 7));
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = (hh[j][i] - mean) * rstd * ga[j][i] + be[j][i];
                if (a.training) o[i] = drop_encode(o[i], a.keep_scale, ((kb >> i) & 1u) != 0u);   // dropped: -0.0, kept zero: +0.0
            }
            stx4_wt(y_ + (size_t)slab * N * C2 + e, o);
        }
    }
    STGCN_PHASE(9, 5);
    if constexpr (PP > 1) {
        // the launch's last workgroup to finish its exchange zeroes the words, the ticket and the count: the next launch finds them clean
        // (all threads of that workgroup take a word each: one thread's 2 * slabs dependent-issue stores were microseconds of kernel tail)
        float* const lastw = red + 3 * 4 * HV + 2 * PP + 1;
        if (tid == 0) *lastw = fin == a.peer.total - 1u ? 1.f : 0.f;
        barrier_only();
        if (*lastw != 0.f) {
            for (int i = tid; i < a.peer.ncount; i += 256 * HV) chain_st64(a.peer_slots + i, 0ull);
            if (tid == 0) {
                chain_st(a.peer.words, 0u);
                chain_st(a.peer.words + 1, 0u);
            }
        }
    }
}
template <int C2, int KT, int NTI, int HV, int PP, typename ET>
__global__ __launch_bounds__(256 * HV) void tc2_ln_fwd_kernel(Tc2LnFwdArgs a) {
    tc2_ln_fwd_body<C2, KT, NTI, HV, PP, ET>(a, (int)blockIdx.x, ChainCtl{nullptr, 0, 0u});
}
template <int C2, int KT, int NTI, int HV, int PP>
__global__ __launch_bounds__(256 * HV) void tc2_ln_fwd_x6_kernel(Tc2LnFwdArgs a) {
    tc2_ln_fwd_body<C2, KT, NTI, HV, PP, float, true>(a, (int)blockIdx.x, ChainCtl{nullptr, 0, 0u});
}

// ================================================================================================
// Chained forward of one ST block (stgcn_device.hip.h "Chained launches"): tmp_conv1 + Align (role 1, n1 workgroups of 512 threads walking the
// time axis) -> graph conv (role 2, n2 = slabs x parts workgroups of gc_threads threads, slab (b, t) as soon as its node tiles have
// arrived) [-> tmp_conv2 + LayerNorm + dropout (role 3, one workgroup of 1024 threads per output slab, as soon as its KT input slabs have
// arrived)] in ONE launch.  Roles and items come from the ticket; waves beyond a role's width leave at once (a finished wave does not take
// part in s_barrier); every workgroup reserves the widest role's threads and LDS.
// ================================================================================================
#ifdef STGCN_EXPERIMENTS
template <int CIN, int KT, int NTI, bool WITH_TC2, typename ET>
__global__ __launch_bounds__(WITH_TC2 ? 1024 : 512) void stblock_fwd_chain_kernel(Tc1FwdArgs a1, GconvFwdArgs a2, Tc2LnFwdArgs a3, ChainCtl chain, int n1, int n2,
                                                                                   int gc_threads, int slot) {
    extern __shared__ float stgcn_smem[];
    const int vb = chain_enter(chain, reinterpret_cast<unsigned*>(stgcn_smem) + slot);
    if (vb < n1) {
        if (threadIdx.x >= 512) return;
        tc1_fwd_body<64, CIN, KT, 0, ET>(a1, vb, n1, chain);
    } else if (vb < n1 + n2) {
        if ((int)threadIdx.x >= gc_threads) return;
        gconv_fwd_body<1, 16, ET, 1>(a2, vb - n1, gc_threads, chain);
    } else {
        if constexpr (WITH_TC2) tc2_ln_fwd_body<64, KT, NTI, 4, 1, ET>(a3, vb - n1 - n2, chain);
    }
    chain_exit(chain);
}
#endif

}  // namespace stgcn
