// Kernels of the fused output head ('TNFF' OutputBlock, reference model/layers.py:260-284):
//   temporal conv (Ko taps) + GLU  -> tconv_fwd_kernel (stgcn_kernels_fwd.hip.h)
//   LayerNorm([N, c0])             -> ln_fwd_kernel
//   fc1 + ReLU + dropout + fc2     -> fc_fwd_kernel            (layers.py:279-282)
// and their backward: fc_bwd_kernel here, LayerNorm / conv backward from stgcn_kernels_bwd.hip.h.
#pragma once
#include "stgcn_kernels_fwd.hip.h"

namespace stgcn {

// ================================================================================================
// fc1 (c0 -> c1) + ReLU + inverted dropout + fc2 (c1 -> 1) on TR-row tiles of the LayerNorm output.
//   hd  = dropout(relu(yln @ W1^T + b1))      saved for backward
//   out = hd @ w2 + b2                        one value per (b, n) row
// wave w owns output-channel tiles w + 4j (NT = c1 / 64); c1 must be 128 (32 float4 columns per row, one
// per lane of a half-wave, so the fc2 dot product is a half-wave shuffle reduction).
// With ln.U set the head's LayerNorm (layers.py:278) runs in the tile staging instead of in its own launch: the rows of a tile belong
// to one or two (b, t) slabs, whose statistics every workgroup rebuilds from the conv epilogue's row partials (N float2 per slab);
// yln is still written (the fc1 weight gradient reads it), mean / rstd by the workgroup that holds the slab's first row.
// ================================================================================================
struct FcFwdArgs {
    TapSrc ts;            // yln [rows][c0] (taps = 1)
    LnFwdArgs ln;         // fused LayerNorm: U, S, gamma, beta, rowstat, y (= yln), mean, rstd, N, C, act, eps; ln.U == null: yln is an input
    const float* W1p;     // packed PK_LIN_FWD: K = c0, cols = c1
    const float* b1;      // [c1] or null
    const float* w2;      // [c1]
    const float* b2;      // [1] or null
    float* hd;            // [rows][c1]
    float* out;           // [rows]
    int KCH, c1, training;
    float keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
};

template <int WM, typename ET>
__global__ __launch_bounds__(256) void fc_fwd_kernel(FcFwdArgs a) {
    constexpr int TR = WM * 16, NT = 2;
    extern __shared__ float stgcn_smem[];
    float* At = stgcn_smem + kTileHdr;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    const long row0 = (long)xcd_item(blockIdx.x, gridDim.x) * TR;
    const int KP = a.KCH * 16;   // = c0 <= 128
    // every load of the tile is requested up front: the fc1 weight fragments of the whole K (<= 8 chunks), then the rows
    PreW<NT, 8> w;
    pre_load_weights<NT, 8>(w, a.W1p, a.KCH, wave, 4);
    if (a.ln.U) {
        const int N = a.ln.N, C = a.ln.C, c4n0 = C >> 2, lda0 = KP + 4;
        const long rows = a.ts.rows, rend = row0 + TR < rows ? row0 + TR : rows;
        for (long slab = row0 / N; slab * N < rend; ++slab) {   // (uniform: every thread walks the same slabs)
            float mean, rstd;
            slab_stats_from_rows(a.ln.rowstat + (size_t)slab * N, N, C, a.ln.eps, stgcn_smem, mean, rstd);
            if (slab * N >= row0 && threadIdx.x == 0) {
                a.ln.mean[slab] = mean;
                a.ln.rstd[slab] = rstd;
            }
            for (int idx = threadIdx.x; idx < TR * c4n0; idx += kThreads) {
                const int row = idx / c4n0, c4 = idx - row * c4n0;
                const long R = row0 + row;
                if (R >= rows || R / N != slab) continue;
                const int n = (int)(R - slab * N);
                const f32x4 u = ldx4(et_ptr<ET>(a.ln.U) + (size_t)R * C + 4 * c4), sg = ldx4(et_ptr<ET>(a.ln.S) + (size_t)R * C + 4 * c4);
                const f32x4 ga = ld4(a.ln.gamma + (size_t)n * C + 4 * c4), be = ld4(a.ln.beta + (size_t)n * C + 4 * c4);
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (gate_fwd(u[i], sg[i], a.ln.act) - mean) * rstd * ga[i] + be[i];
                st4(At + row * lda0 + 4 * c4, o);
                stx4_wt(et_ptr<ET>(a.ln.y) + (size_t)R * C + 4 * c4, o);
            }
        }
        for (int idx = threadIdx.x; idx < TR * c4n0; idx += kThreads) {   // rows past the end of the last tile
            const int row = idx / c4n0, c4 = idx - row * c4n0;
            if (row0 + row >= rows) st4(At + row * lda0 + 4 * c4, zero4());
        }
    } else {
        stage_tile_fwd<TR, 4, kThreads, ET>(a.ts, row0, KP, At, KP + 4);
    }
    __syncthreads();
    f32x4 acc[WM][NT];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4();
    pre_mma<WM, NT, 8, ET>(acc, At, KP + 4, 0, a.KCH, w);
    __syncthreads();
    const int c1 = a.c1, ldz = c1 + 4;
    float* Zt = At;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = (wave + 4 * j) * 16 + l15;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Zt[(i * 16 + 4 * g + r) * ldz + col] = acc[i][j][r];
    }
    __syncthreads();
    const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
    const int c4n = c1 >> 2, c4sh = pow2_shift(c4n);   // 32
    const float b2 = a.b2 ? a.b2[0] : 0.f;
    for (int idx = threadIdx.x; idx < TR * c4n; idx += kThreads) {
        const int row = fast_div(idx, c4n, c4sh), c4 = idx - row * c4n;
        const long R = row0 + row;
        f32x4 h = ld4(Zt + row * ldz + 4 * c4);
        const f32x4 w2 = ld4(a.w2 + 4 * c4);
        if (a.b1) {
            const f32x4 b1 = ld4(a.b1 + 4 * c4);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] += b1[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = fmaxf(h[i], 0.f);
        if (a.training) {
            const f32x4 k = dropout_scale4((uint64_t)R * c4n + c4, a.seed, off, a.thresh, a.keep_scale);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] *= k[i];
        }
        float p = h[0] * w2[0] + h[1] * w2[1] + h[2] * w2[2] + h[3] * w2[3];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) p += __shfl_xor(p, m);   // the 32 lanes of a half-wave hold one row
        if (R < a.ts.rows) {
            stx4_wt(et_ptr<ET>(a.hd) + (size_t)R * c1 + 4 * c4, h);
            if (c4 == 0) a.out[R] = p + b2;
        }
    }
}

// ================================================================================================
// Backward of fc2 / dropout / ReLU / fc1 on TR-row tiles (grid-stride, so that partials stay few):
//   dh1 = dout[row] * w2[c] * (hd != 0 ? keep_scale : 0)            (relu' and the dropout mask in one test)
//   dyln = dh1 @ W1                                                   -> [rows][c0]
//   partials: dw2[c] = sum_rows dout * hd ; db2 = sum_rows dout       (dW1 / db1 come from tconv_bwd_weight_kernel)
// ================================================================================================
struct FcBwdArgs {
    const float* dout;    // [rows]
    const float* hd;      // [rows][c1]
    const float* w2;      // [c1]
    const float* W1d;     // packed PK_LIN_BWD: K = c1, cols = c0
    float* dh1;           // [rows][c1]
    float* dyln;          // [rows][c0]
    float* part;          // [wgs][c1 + 2]: dw2 | db2 | sum (pred - target)^2 / n
    long rows;
    int c0, c1, KCH;      // KCH = c1 / 16
    float grad_scale;     // keep_scale when dropout was active, else 1
    // fused MSE loss (pred != null): dout[R] = 2 (pred[R] - target[R]) / n * loss_scale is formed here instead of read from `dout`
    const float* pred;
    const float* target;
    const long* target_index;   // nullable: target += *target_index * target_index_stride
    long target_index_stride;
    float loss_scale;
    LnRowstatOut rs;      // row partials of the head's LayerNorm backward (dyln is its output gradient); rs.rowstat == null: off
};

template <int WM, int NT, typename ET>
__global__ __launch_bounds__(256) void fc_bwd_kernel(FcBwdArgs a) {
    constexpr int TR = WM * 16;
    extern __shared__ float stgcn_smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int c1 = a.c1, c0 = a.c0, lda = c1 + 4, c4n = c1 >> 2;
    float* At = stgcn_smem;                // [TR][lda]
    float* red = stgcn_smem + TR * lda;    // [8][c1] + [8] + [8]
    const long tiles = (a.rows + TR - 1) / TR;
    const int c4 = tid % c4n, rsub = tid / c4n;            // c4n == 32 -> rsub in 0..7, fixed channel group per thread
    f32x4 dw2 = zero4();
    float db2 = 0.f, lsum = 0.f;
    const f32x4 w2 = ld4(a.w2 + 4 * c4);
    const float* tgt = a.target;
    if (a.pred && a.target_index) tgt += *a.target_index * a.target_index_stride;
    const float inv_n = 1.0f / (float)a.rows;
    PreW<NT, 8> w;   // fc1 weight fragments of the whole K = c1 (8 chunks), requested before the first tile is touched
    pre_load_weights<NT, 8>(w, a.W1d, a.KCH, wave, 4);
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long row0 = t * TR;
        __syncthreads();
        for (int row = rsub; row < TR; row += kThreads / c4n) {
            const long R = row0 + row;
            f32x4 d = zero4();
            if (R < a.rows) {
                float go;
                if (a.pred) {   // uniform
                    const float df = a.pred[R] - tgt[R];
                    go = 2.0f * df * inv_n * a.loss_scale;
                    if (c4 == 0) lsum += df * df;
                } else {
                    go = a.dout[R];
                }
                const f32x4 h = ldx4(et_ptr<ET>(a.hd) + (size_t)R * c1 + 4 * c4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i] = h[i] != 0.f ? go * w2[i] * a.grad_scale : 0.f;
                    dw2[i] += go * h[i];
                }
                if (c4 == 0) db2 += go;
                stx4_wt(et_ptr<ET>(a.dh1) + (size_t)R * c1 + 4 * c4, d);
            }
            st4(At + row * lda + 4 * c4, d);
        }
        __syncthreads();
        f32x4 acc[WM][NT];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = zero4();
        pre_mma<WM, NT, 8, ET>(acc, At, lda, 0, a.KCH, w);
        if (a.rs.rowstat && c0 == c1) {
            // dyln through the LDS tile: 16-byte row-major stores, and the LayerNorm-backward row partials while the row is on chip
            __syncthreads();   // every wave is done reading At
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (wave + 4 * j) * 16 + l15;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) At[(i * 16 + 4 * g + r) * lda + col] = acc[i][j][r];
            }
            __syncthreads();
            const uint64_t hoff = ln_rowstat_offset(a.rs);
            for (int row = rsub; row < TR; row += kThreads / c4n) {   // c4n lanes (half a wave for c0 = 128) hold one row
                const long R = row0 + row;
                const bool rin = R < a.rows;
                const f32x4 v = et_round4<ET>(ld4(At + row * lda + 4 * c4));   // (the row partials below see what the tensor holds)
                if (rin) stx4_wt(et_ptr<ET>(a.dyln) + (size_t)R * c0 + 4 * c4, v);
                const long slab = rin ? R / a.rs.N : 0;
                const int node = rin ? (int)(R - slab * a.rs.N) : 0;
                float2 p = rin ? ln_rowstat4<ET>(a.rs, hoff, v, slab, node, 4 * c4) : make_float2(0.f, 0.f);
                for (int m = c4n >> 1; m >= 1; m >>= 1) {
                    p.x += __shfl_xor(p.x, m);
                    p.y += __shfl_xor(p.y, m);
                }
                if (rin && c4 == 0) a.rs.rowstat[R] = p;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = (wave + 4 * j) * 16 + l15;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long R = row0 + i * 16 + 4 * g + r;
                        if (R < a.rows && col < c0) stx1(et_ptr<ET>(a.dyln) + (size_t)R * c0 + col, acc[i][j][r]);
                    }
            }
        }
    }
    __syncthreads();
    st4(red + rsub * c1 + 4 * c4, dw2);
    if (c4 == 0) {
        red[8 * c1 + rsub] = db2;
        red[8 * c1 + 8 + rsub] = lsum;
    }
    __syncthreads();
    float* part = a.part + (size_t)blockIdx.x * (c1 + 2);
    if (tid < c1) {
        float s = 0.f;
        for (int k = 0; k < kThreads / c4n; ++k) s += red[k * c1 + tid];
        part[tid] = s;
    }
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < kThreads / c4n; ++k) s += red[8 * c1 + k];
        part[c1] = s;
        float l = 0.f;
        for (int k = 0; k < kThreads / c4n; ++k) l += red[8 * c1 + 8 + k];
        part[c1 + 1] = l * inv_n;
    }
}

// ================================================================================================
// Multi-tensor AdamW (torch.optim.AdamW as configured at main.py:148: amsgrad False, maximize False):
//     p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// One launch updates every live parameter tensor (those with a gradient); the pointer table travels in the
// kernel arguments.  t and lr may come from device memory so that a captured hipGraph stays correct.
// ================================================================================================
struct AdamwTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
};
constexpr int kAdamwMaxTensors = 48;
constexpr int kAdamwChunk = 512;    // small chunks: ~500 workgroups for the 245 K parameters (the kernel is pure latency)
struct AdamwArgs {
    AdamwTensor t[kAdamwMaxTensors];
    int start[kAdamwMaxTensors + 1];
    int count;
    float lr, b1, b2, eps, wd;
    float lb1, lb2;       // log(beta1), log(beta2) computed in double on the host
    long step;
    const long* step_dev;
    const float* lr_dev;
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamwArgs a) {
    int jb = 0;
    while (jb + 1 < a.count && (int)blockIdx.x >= a.start[jb + 1]) ++jb;
    const AdamwTensor& T = a.t[jb];
    const long base = ((long)blockIdx.x - a.start[jb]) * kAdamwChunk;
    // bias corrections 1 - beta^t: -expm1f(t * log(beta)) keeps full relative accuracy for small t (a double-precision
    // pow() here is a multi-microsecond serial latency chain in every thread of a kernel that moves only 1 MB)
    const float t = (float)(a.step_dev ? *a.step_dev : a.step);
    const float lr = a.lr_dev ? *a.lr_dev : a.lr;
    const float bc1 = -expm1f(t * a.lb1);
    const float rs2 = rsqrtf(-expm1f(t * a.lb2));
    const float decay = 1.0f - lr * a.wd, step_size = lr / bc1;
#pragma unroll
    for (int k = 0; k < kAdamwChunk / kThreads; ++k) {
        const long e = base + k * kThreads + threadIdx.x;
        if (e < T.n) {
            const float g = T.g[e];
            const float m = a.b1 * T.m[e] + (1.0f - a.b1) * g;
            const float v = a.b2 * T.v[e] + (1.0f - a.b2) * g * g;
            T.m[e] = m;
            T.v[e] = v;
            T.p[e] = T.p[e] * decay - step_size * (m / (sqrtf(v) * rs2 + a.eps));
        }
    }
}

// ================================================================================================
// nn.MSELoss() (main.py:136, mean over all B*N elements) and its gradient in ONE launch:
//     loss = mean((pred - y)^2) ;  dpred = 2 (pred - y) * grad_scale / n
// The reference's loss + l.backward() head is 5 ATen launches (mse, mean, ones_like fill, mse_backward, scale) for 6624
// elements; every launch of this path costs 5-8 us whatever its size.  One workgroup of 1024 threads, fixed summation
// order (bitwise reproducible).
// ================================================================================================
__global__ __launch_bounds__(1024) void mse_loss_grad_kernel(const float* pred, const float* y, long n, float gscale, float* loss,
                                                              float* dpred, const long* y_idx_dev, long y_idx_stride) {
    if (y_idx_dev) y += *y_idx_dev * y_idx_stride;   // labels straight from the resident series (device-side windowing)
    extern __shared__ float stgcn_smem[];   // [16] wave sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float k = 2.0f * gscale / (float)n;
    float acc = 0.f;
    for (long e = tid; e < n; e += 1024) {
        const float d = pred[e] - y[e];
        acc += d * d;
        dpred[e] = k * d;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) stgcn_smem[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += stgcn_smem[w];
        loss[0] = s / (float)n;
    }
}

}  // namespace stgcn
