// Tiled graph convolution for graphs that do not fit the slab-resident kernels (N > 512 nodes, e.g. the 8192-node
// dense stress graph of BASELINE.json configs[4]; reference model/layers.py:143-172 ChebGraphConv, :194-206 GraphConv,
// :222-231 GraphConvLayer).  The slab kernels keep a whole (b, t) slab of X and every polynomial T_k(L) of the operator
// in LDS / L2; at N = 8192 one polynomial is 256 MB and forming it costs N^3, so this path applies the reference's own
// recursion to the activations instead (layers.py:153-161):
//     X_1 = L X_0 ;  X_k = 2 L X_{k-1} - X_{k-2}                          one GEMM launch per term (gso_gemm_kernel)
//     G   = relu(sum_k X_k W_k + b + X_0)                                  row pass (gconv_rows_fwd_kernel)
// and in backward the transposed sum dA = sum_k T_k(L^T) g_k, g_k = dY W_k^T (+ dY for k = 0, the residual), by the
// Clenshaw recurrence, which is again one GEMM with an axpy epilogue per term (launcher: launch_gconv_bwd_tiled):
//     b_K = g_K ;  b_k = g_k + 2 L^T b_{k+1} - b_{k+2} ;  dA = g_0 + L^T b_1 - b_2
// The parameter gradients dW_k = X_k^T dY, db = 1^T dY and the g_k come from one row pass (gconv_rows_bwd_kernel).
//
// Layouts: activations [slabs][N][16] fp32 as everywhere else (a GEMM column is one channel of one slab); the operator
// is a dense row-major [NP][NP] matrix, NP = roundup(N, 128), zero padded (stgcn_gso_prepare, tiled mode).
#pragma once
#include "stgcn_device.hip.h"

namespace stgcn {

constexpr int kGtBM = 128;                  // nodes (output rows) per workgroup tile
constexpr int kGtSL = 8;                    // slabs per workgroup tile: 8 x 16 channels = 128 GEMM columns
constexpr int kGtBK = 32;                   // contraction (source node) chunk staged per pipeline step
constexpr int kGtLDM = kGtBK + 4;           // LDS row stride of the operator tile   [128][36]
constexpr int kGtLDX = kGtSL * 16 + 4;      // LDS row stride of the activation tile [32][132]
constexpr int kGtLdsFloats = 2 * (kGtBM * kGtLDM + kGtBK * kGtLDX);   // double buffered: 70.7 KB, two workgroups per CU

// out[s][n][c] = alpha * sum_m M[n][m] X[s][m][c] + b1 * Z1[s][n][c] + b2 * Z2[s][n][c]        n < N, s < slabs
// (Z1 / Z2 nullable; out may alias Z1 or Z2: every element is read and written by the same lane, X must not alias out)
struct GsoGemmArgs {
    const float* M;      // [NP][NP] dense, zero padded
    const float* X;      // [slabs][N][16]
    const float* Z1;
    const float* Z2;
    float* out;
    float alpha, b1, b2;
    int N, NP, row_tiles, col_tiles;   // row_tiles = ceil(N / 128), col_tiles = ceil(slabs / 8)
    long slabs;
};

// Workgroup = 4 waves in a 2 x 2 arrangement, wave (wm, wn) owns 64 nodes x 4 slabs = 4 x 4 MFMA tiles (64 accumulator
// VGPRs, 16 independent MFMA chains).  A operand = operator rows (lane: node l15, 4 consecutive source nodes = one 16-B
// LDS read, the "16-chunk" k permutation of stgcn_device.hip.h), B operand = activations (lane: source node 4g+s, channel
// l15), so D leaves lane (g, l15) with channel l15 of nodes 4g..4g+3: 64-B store segments, the 16 x 16 tile is 1 KiB
// contiguous.  Per 32-node chunk a wave issues 128 MFMAs (4096 cycles) against 8 x 16-B global loads and 40 LDS reads per
// lane: the kernel is MFMA-bound as long as the next chunk (prefetched into registers during the MFMAs) arrives in time.
// Tiles that share operator rows run on one XCD (xcd_item): per XCD the resident workgroups stream a few operator row
// tiles and all column tiles in lockstep through its L2.
__global__ __launch_bounds__(256) void gso_gemm_kernel(GsoGemmArgs a) {
    extern __shared__ float stgcn_smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wm = w & 1, wn = w >> 1;
    const int item = xcd_item((int)blockIdx.x, a.row_tiles * a.col_tiles);
    const int rt = item / a.col_tiles, ct = item - rt * a.col_tiles;
    const int n0 = rt * kGtBM, N = a.N, NP = a.NP;
    const long slab0 = (long)ct * kGtSL;
    float* const Ms = stgcn_smem;                         // [2][128][LDM]
    float* const Xs = stgcn_smem + 2 * kGtBM * kGtLDM;    // [2][32][LDX]

    f32x4 pm[4], px[4];
    auto fetch = [&](int kb) {
        const int k0 = kb * kGtBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;                 // 8 float4 per operator row
            pm[i] = ld4(a.M + (size_t)(n0 + row) * NP + k0 + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, c4 = f & 3, ml = (f >> 2) & 31, sl = f >> 7;   // 128 threads read 2 KiB of one slab
            const int m = k0 + ml;
            const long slab = slab0 + sl;
            px[i] = (m < N && slab < a.slabs) ? ld4(a.X + ((size_t)slab * N + m) * 16 + c4 * 4) : zero4();
        }
    };
    auto stage = [&](int buf) {
        float* ms = Ms + buf * kGtBM * kGtLDM;
        float* xs = Xs + buf * kGtBK * kGtLDX;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            st4(ms + row * kGtLDM + c4 * 4, pm[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, c4 = f & 3, ml = (f >> 2) & 31, sl = f >> 7;
            st4(xs + ml * kGtLDX + sl * 16 + c4 * 4, px[i]);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = zero4();

    const int nkb = (N + kGtBK - 1) / kGtBK;   // operator columns >= N are zero padding
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) fetch(kb + 1);
        const float* ms = Ms + buf * kGtBM * kGtLDM + (wm * 64 + l15) * kGtLDM + 4 * g;
        const float* xs = Xs + buf * kGtBK * kGtLDX + 4 * g * kGtLDX + wn * 64 + l15;
#pragma unroll
        for (int kc = 0; kc < kGtBK / 16; ++kc) {
            f32x4 af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = ld4(ms + mt * 16 * kGtLDM + kc * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float bf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[nt] = xs[(kc * 16 + s) * kGtLDX + nt * 16];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma4(af[mt][s], bf[nt], acc[mt][nt]);
            }
        }
        if (kb + 1 < nkb) stage(buf ^ 1);   // the other buffer was last read before the barrier that ended step kb - 1
        __syncthreads();
    }

    // epilogue: acc[mt][nt][r] = (M X)[node n0 + wm*64 + mt*16 + 4g + r][slab slab0 + wn*4 + nt][channel l15]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const long slab = slab0 + wn * 4 + nt;
        if (slab >= a.slabs) continue;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 64 + mt * 16 + 4 * g + r;
                if (n < N) {
                    const size_t o = ((size_t)slab * N + n) * 16 + l15;
                    float v = a.alpha * acc[mt][nt][r];
                    if (a.Z1) v += a.b1 * a.Z1[o];
                    if (a.Z2) v += a.b2 * a.Z2[o];
                    a.out[o] = v;
                }
            }
        }
    }
}

// ================================================================================================
// Row pass of the tiled forward: G = relu(sum_k X_k W_k + b + X_0) (layers.py:165-168 / :198-199, :229, :253).
// One wave per 16-row tile (grid-stride); A operand = 16 rows x 16 channels of X_k straight from HBM (one 16-B load
// per lane, 1 KiB per wave), B operand = W_k in registers.  HBM-bound: (terms + 1) x 64 B read, 64 B written per row.
// ================================================================================================
constexpr int kGcMaxTerms = 8;
struct GcRowsFwdArgs {
    const float* X0;     // [rows][16]
    const float* Xk;     // X_k = Xk + (k - 1) * kstride, k = 1 .. terms - 1
    const float* W;      // cheb: [terms][16][16] ; kipf: [16][16] (applies to X_1, no X_0 term)
    const float* bias;   // [16] or null
    float* G;            // [rows][16]
    long rows, kstride;
    int terms, kipf;
};

__global__ __launch_bounds__(256) void gconv_rows_fwd_kernel(GcRowsFwdArgs a) {
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    f32x4 wf[kGcMaxTerms];   // B[kk = c = 4g + s][col = j = l15] = W_k[c][j]
#pragma unroll
    for (int k = 0; k < kGcMaxTerms; ++k) {
        wf[k] = zero4();
        if (k < a.terms && !(a.kipf && k == 0)) {
            const float* Wk = a.W + (a.kipf ? 0 : (size_t)k * 256);
#pragma unroll
            for (int s = 0; s < 4; ++s) wf[k][s] = Wk[(4 * g + s) * 16 + l15];
        }
    }
    const float bb = a.bias ? a.bias[l15] : 0.f;
    const long tiles = (a.rows + 15) >> 4;
    for (long tile = (long)blockIdx.x * 4 + w; tile < tiles; tile += (long)gridDim.x * 4) {
        const long row0 = tile << 4, row = row0 + l15;
        const bool in = row < a.rows;
        f32x4 y = zero4();
#pragma unroll
        for (int k = 0; k < kGcMaxTerms; ++k) {
            if (k < a.terms && !(a.kipf && k == 0)) {
                const float* Xs = k == 0 ? a.X0 : a.Xk + (size_t)(k - 1) * a.kstride;
                const f32x4 xa = in ? ld4(Xs + (size_t)row * 16 + 4 * g) : zero4();   // A[row = l15][c = 4g + s]
#pragma unroll
                for (int s = 0; s < 4; ++s) y = mfma4(xa[s], wf[k][s], y);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long rr = row0 + 4 * g + r;   // D[row = 4g + r][j = l15]
            if (rr < a.rows) a.G[(size_t)rr * 16 + l15] = fmaxf(y[r] + bb + a.X0[(size_t)rr * 16 + l15], 0.f);
        }
    }
}

// ================================================================================================
// Row pass of the tiled backward: per 16-row tile
//     g_k = dY W_k^T  (k = 0 .. terms-1; + dY on k = 0: the residual of layers.py:229)   -> Gk[k]
//     dW_k += X_k^T dY ,  db += 1^T dY                                                      (registers)
// Workgroup b owns the contiguous tile range [b * tiles_per_wg, ...); its four waves take alternate tiles and are summed
// through LDS in a fixed order, so the partials (one [(terms + 1) * 256] block per workgroup, same layout as the slab
// kernel's) and the reduced gradients are bitwise reproducible.
// ================================================================================================
struct GcRowsBwdArgs {
    const float* dY;     // [rows][16]
    const float* X0;
    const float* Xk;     // X_k = Xk + (k - 1) * kstride
    const float* W;
    float* Gk;           // g_k = Gk + k * gstride
    float* part;         // [gridDim.x][(terms + 1) * 256]
    long rows, kstride, gstride;
    int terms, kipf, tiles_per_wg;
};

__global__ __launch_bounds__(256) void gconv_rows_bwd_kernel(GcRowsBwdArgs a) {
    extern __shared__ float stgcn_smem[];   // [4][(terms + 1) * 256]
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int terms = a.terms, PS = (terms + 1) * 256;
    f32x4 wt[kGcMaxTerms];   // B[kk = j = 4g + s][col = i = l15] = W_k[i][j]
#pragma unroll
    for (int k = 0; k < kGcMaxTerms; ++k)
        wt[k] = (k < terms && !(a.kipf && k == 0)) ? ld4(a.W + (a.kipf ? 0 : (size_t)k * 256) + l15 * 16 + 4 * g) : zero4();
    f32x4 dw[kGcMaxTerms], db = zero4();
#pragma unroll
    for (int k = 0; k < kGcMaxTerms; ++k) dw[k] = zero4();

    const long tiles = (a.rows + 15) >> 4;
    const long t0 = (long)blockIdx.x * a.tiles_per_wg;
    long t1 = t0 + a.tiles_per_wg;
    if (t1 > tiles) t1 = tiles;
    for (long tile = t0 + w; tile < t1; tile += 4) {
        const long row0 = tile << 4, row = row0 + l15;
        const f32x4 ya = row < a.rows ? ld4(a.dY + (size_t)row * 16 + 4 * g) : zero4();   // A[row = l15][j = 4g + s]
        float yb[4];                                                                       // B[kk = row 4g + s][col = j = l15]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long rr = row0 + 4 * g + s;
            yb[s] = rr < a.rows ? a.dY[(size_t)rr * 16 + l15] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) db = mfma4(1.0f, yb[s], db);
#pragma unroll
        for (int k = 0; k < kGcMaxTerms; ++k) {
            if (k < terms) {
                const float* Xs = k == 0 ? a.X0 : a.Xk + (size_t)(k - 1) * a.kstride;
                f32x4 gk = zero4();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const long rr = row0 + 4 * g + s;
                    const float xa = rr < a.rows ? Xs[(size_t)rr * 16 + l15] : 0.f;   // A[i = l15][kk = row 4g + s]
                    dw[k] = mfma4(xa, yb[s], dw[k]);
                    gk = mfma4(ya[s], wt[k][s], gk);
                }
                float* Go = a.Gk + (size_t)k * a.gstride;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long rr = row0 + 4 * g + r;   // D[row = 4g + r][i = l15]
                    if (rr < a.rows) Go[(size_t)rr * 16 + l15] = gk[r] + (k == 0 ? yb[r] : 0.f);
                }
            }
        }
    }
    // D[i = 4g + r][j = l15] -> slot k, element i * 16 + j
    float* mine = stgcn_smem + w * PS;
#pragma unroll
    for (int k = 0; k < kGcMaxTerms; ++k) {
        if (k < terms) {
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[k * 256 + (4 * g + r) * 16 + l15] = dw[k][r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[terms * 256 + (4 * g + r) * 16 + l15] = db[r];
    __syncthreads();
    float* part = a.part + (size_t)blockIdx.x * PS;
    for (int e = tid; e < PS; e += 256) part[e] = (stgcn_smem[e] + stgcn_smem[PS + e]) + (stgcn_smem[2 * PS + e] + stgcn_smem[3 * PS + e]);
}

// dense zero-padded transposed copy of the operator: D[h][i] = L[i][h]
__global__ __launch_bounds__(256) void gso_dense_t_kernel(const float* L, int N, int NP, float* D) {
    const long e = (long)blockIdx.x * kThreads + (long)threadIdx.x;
    if (e >= (long)NP * NP) return;
    const int h = (int)(e / NP), i = (int)(e - (long)h * NP);
    D[e] = (h < N && i < N) ? L[(size_t)i * N + h] : 0.f;
}

}  // namespace stgcn
