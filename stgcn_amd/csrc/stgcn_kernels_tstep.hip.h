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
#include "stgcn_device.hip.h"

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
    const float* dy;          // [B][T2][N][C2]
    const float* U;           // [B][T2][N][C2]  saved gate inputs of tmp_conv2
    const float* S;
    const float* gamma;       // [N][C2]
    const float* mean;        // [B*T2]
    const float* rstd;
    const float2* rowstat;    // [B*T2*N]  (sum g, sum g*xhat) per row
    const float2* slabconst;  // [B*T2] (c1, c2) or null: rebuilt from rowstat by every workgroup
    const float* G;           // [B][T1][N][16]  tmp_conv2 input (relu output, also the relu mask)
    const float* Wd;          // [Kt*16][2*C2]   dense W_eff of tmp_conv2 (PK_TCONV_DENSE)
    float* dYg;               // [B][T1][N][16]
    float* dZ;                // optional [B*T2*N][2*C2]  (stage tests only)
    float* part;              // [wgs][Kt*16*NC + NC]  dW_eff2 | db_eff2 partials
    float* dgam_part;         // [B][N*C2]
    float* dbet_part;
    int B, T1, T2, N, act, training, node_tiles;
    float keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
};

constexpr int kTsMaxT = 32;   // time steps of G kept in LDS (host falls back to the unfused kernels beyond)
inline size_t tc2_bwd_lds_bytes(int C2, int Kt, int T1, int T2) {
    return ((size_t)Kt * 16 * (2 * C2 + 4) + (size_t)T1 * 16 * 20 + 4 * 16 * 20 + 4 * (size_t)T2) * sizeof(float);
}

template <int C2, int KT>
__global__ __launch_bounds__(256) void tc2_bwd_kernel(Tc2BwdArgs a) {
    constexpr int NC = 2 * C2, LDZ = NC + 4, NTW = NC / 64, QW = NC / 64, IT = C2 / 64, LDG = 20;
    extern __shared__ float stgcn_smem[];
    float* const Zt = stgcn_smem;                      // [KT][16][LDZ]  ring of dZ2 tiles
    float* const GT = Zt + KT * 16 * LDZ;              // [T1][16 ch][LDG]  G tiles, transposed (GT[t][i][row])
    float* const red = GT + a.T1 * 16 * LDG;           // [4 waves][16 rows][LDG]
    float* const cs = red + 4 * 16 * LDG;              // [T2][4]: c1, c2, mean, rstd
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int b = (int)blockIdx.x / a.node_tiles, nt = (int)blockIdx.x - b * a.node_tiles, n0 = nt * 16;
    const int N = a.N, T1 = a.T1, T2 = a.T2;
    const int r = tid >> 4, cq = tid & 15;             // elementwise phase: row r, float4 columns cq + 16*it
    const bool rv = n0 + r < N;

    // ---- slab constants of this window's T2 slabs (wave w: slabs w, w + 4, ..) -------------------------------------
    for (int t = w; t < T2; t += 4) {
        const long slab = (long)b * T2 + t;
        float x = 0.f, y = 0.f;
        if (a.slabconst) {
            const float2 c = a.slabconst[slab];
            x = c.x; y = c.y;
        } else {
            const float2* rs = a.rowstat + slab * N;
            for (int i = lane; i < N; i += 64) {
                const float2 v = rs[i];
                x += v.x;
                y += v.y;
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                x += __shfl_xor(x, m);
                y += __shfl_xor(y, m);
            }
            const float inv = 1.0f / ((float)N * (float)C2);
            x *= inv; y *= inv;
        }
        if (lane == 0) {
            cs[4 * t] = x;
            cs[4 * t + 1] = y;
            cs[4 * t + 2] = a.mean[slab];
            cs[4 * t + 3] = a.rstd[slab];
        }
    }
    // ---- all G tiles of this (window, node tile), transposed ------------------------------------------------------------
    for (int idx = tid; idx < T1 * 64; idx += 256) {
        const int t = idx >> 6, rem = idx & 63, rr = rem >> 2, q = rem & 3;
        const f32x4 v = n0 + rr < N ? ld4(a.G + (((size_t)b * T1 + t) * N + n0 + rr) * 16 + 4 * q) : zero4();
#pragma unroll
        for (int i = 0; i < 4; ++i) GT[(t * 16 + 4 * q + i) * LDG + rr] = v[i];
    }
    // ---- stationary weights of the transposed conv: wave w contracts o in [w*NC/4, (w+1)*NC/4) of every tap ----------------
    f32x4 Wr[KT][QW];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int q = 0; q < QW; ++q) Wr[k][q] = ld4(a.Wd + (size_t)(k * 16 + l15) * NC + w * (NC / 4) + q * 16 + 4 * g);
    f32x4 gam[IT], dgam[IT], dbet[IT], dbu[IT], dbq[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        gam[it] = rv ? ld4(a.gamma + (size_t)(n0 + r) * C2 + 4 * (cq + 16 * it)) : zero4();
        dgam[it] = zero4(); dbet[it] = zero4(); dbu[it] = zero4(); dbq[it] = zero4();
    }
    f32x4 accw[KT][NTW];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int j = 0; j < NTW; ++j) accw[k][j] = zero4();
    const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
    const int n4 = (N * C2) >> 2;

    // one-step software prefetch of the streamed tiles
    f32x4 dyn[IT], un[IT], sn[IT];
    auto fetch = [&](int t2) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const size_t e = (((size_t)b * T2 + t2) * N + n0 + r) * C2 + 4 * (cq + 16 * it);
            dyn[it] = rv ? ld4(a.dy + e) : zero4();
            un[it] = rv ? ld4(a.U + e) : zero4();
            sn[it] = rv ? ld4(a.S + e) : zero4();
        }
    };
    if (T2 > 0) fetch(0);
    __syncthreads();

    for (int t1 = 0; t1 < T1; ++t1) {
        const bool step = t1 < T2;                      // uniform: a new dZ2 tile this step
        float* const Zs = Zt + (t1 % KT) * 16 * LDZ;
        if (step) {
            const float c1 = cs[4 * t1], c2 = cs[4 * t1 + 1], mean = cs[4 * t1 + 2], rstd = cs[4 * t1 + 3];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                f32x4 dy = dyn[it];
                const f32x4 u = un[it], s = sn[it];
                const int c4 = cq + 16 * it;
                if (a.training) {
                    const f32x4 k = dropout_scale4(((uint64_t)b * T2 + t1) * n4 + (((uint64_t)(n0 + r) * C2) >> 2) + c4, a.seed, off, a.thresh,
                                                   a.keep_scale);
#pragma unroll
                    for (int i = 0; i < 4; ++i) dy[i] *= k[i];
                }
                f32x4 du = zero4(), dq = zero4();
                if (rv) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xh = (gate_fwd(u[i], s[i], a.act) - mean) * rstd;
                        const float gg = dy[i] * gam[it][i];
                        const float dh = rstd * (gg - c1 - xh * c2);
                        dgam[it][i] += dy[i] * xh;
                        dbet[it][i] += dy[i];
                        float du_, dq_;
                        gate_bwd(dh, u[i], s[i], a.act, du_, dq_);
                        du[i] = du_;
                        dq[i] = dq_;
                    }
                    dbu[it] += du;
                    dbq[it] += dq;
                    if (a.dZ) {
                        float* z = a.dZ + (((size_t)b * T2 + t1) * N + n0 + r) * NC + 4 * c4;
                        st4(z, du);
                        st4(z + C2, dq);
                    }
                }
                st4(Zs + r * LDZ + 4 * c4, du);
                st4(Zs + r * LDZ + C2 + 4 * c4, dq);
            }
        }
        __syncthreads();   // dZ2 tile of this step visible; `red` of the previous step consumed
        if (t1 + 1 < T2) fetch(t1 + 1);
        if (step) {
            // weight gradient: A[m = i][k = row] = G[t1 + tap][row][i] (transposed tiles: one 16-byte read), B[k = row][n = o] = dZ2
            f32x4 bz[NTW];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) bz[j][s] = Zs[(4 * g + s) * LDZ + (w * NTW + j) * 16 + l15];
            }
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const f32x4 af = ld4(GT + ((t1 + k) * 16 + l15) * LDG + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) accw[k][j] = mfma4(af[s], bz[j][s], accw[k][j]);
            }
        }
        // transposed conv for output step t1: taps with 0 <= t1 - tap < T2
        f32x4 accd = zero4();
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int ts = t1 - k;
            if (ts >= 0 && ts < T2) {   // uniform
                const float* zr = Zt + (ts % KT) * 16 * LDZ + l15 * LDZ + w * (NC / 4) + 4 * g;
#pragma unroll
                for (int q = 0; q < QW; ++q) {
                    const f32x4 z = ld4(zr + q * 16);
#pragma unroll
                    for (int s = 0; s < 4; ++s) accd = mfma4(Wr[k][q][s], z[s], accd);
                }
            }
        }
        st4(red + (w * 16 + l15) * LDG + 4 * g, accd);   // D[m = i = 4g + r][n = row = l15]
        __syncthreads();
        {
            const int i = cq;   // thread (row r, channel i)
            float v = (red[(0 * 16 + r) * LDG + i] + red[(1 * 16 + r) * LDG + i]) + (red[(2 * 16 + r) * LDG + i] + red[(3 * 16 + r) * LDG + i]);
            if (!(GT[(t1 * 16 + i) * LDG + r] > 0.f)) v = 0.f;
            if (rv) a.dYg[(((size_t)b * T1 + t1) * N + n0 + r) * 16 + i] = v;
        }
    }

    // ---- per-workgroup partials ---------------------------------------------------------------------------------------------
    float* part = a.part + (size_t)blockIdx.x * (KT * 16 * NC + NC);
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) part[(size_t)(k * 16 + 4 * g + rr) * NC + (w * NTW + j) * 16 + l15] = accw[k][j][rr];
    if (rv) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const size_t o = (size_t)b * N * C2 + (size_t)(n0 + r) * C2 + 4 * (cq + 16 * it);
            st4(a.dgam_part + o, dgam[it]);
            st4(a.dbet_part + o, dbet[it]);
        }
    }
    // db_eff2[o] = sum over the 16 rows (threads with equal cq: lanes 16 apart, then the 4 waves through LDS)
    __syncthreads();   // `red` reads of the last step done
    float* bred = red;   // [4 waves][NC] (NC <= 256: fits the 1280 floats of `red`)
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
    __syncthreads();
    if (tid < NC) part[(size_t)KT * 16 * NC + tid] = (bred[tid] + bred[NC + tid]) + (bred[2 * NC + tid] + bred[3 * NC + tid]);
}

}  // namespace stgcn
