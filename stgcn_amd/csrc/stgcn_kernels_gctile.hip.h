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
constexpr int kGtBK = 32;                   // contraction (source node) chunk staged per pipeline step
constexpr int kGtLDM = kGtBK + 4;           // LDS row stride of the operator tile   [128][36]
// The column extent of a workgroup tile is a template parameter: NTW = slab tiles per wave (3, 4 or 5), a workgroup covers
// 2 * NTW slabs = 96 / 128 / 160 GEMM columns.  The launcher picks the NTW whose grid wastes least of its last round of
// resident workgroups (C5: 160 slabs -> NTW 5 = 16 x 64 = 1024 workgroups = exactly two rounds on 512 slots, where
// NTW 4 needs 1280 = 2.5 rounds; 96 slabs -> NTW 3 = 1024 as well).
inline int gt_ldx(int ntw) { return 2 * ntw * 16 + 4; }                                        // LDS row stride of the activation tile
inline int gt_lds_floats(int ntw) { return 2 * (kGtBM * kGtLDM + kGtBK * gt_ldx(ntw)); }      // double buffered: 62.5 / 70.7 / 78.8 KB

// out[s][n][c] = alpha * sum_m M[n][m] X[s][m][c] + b1 * Z1[s][n][c] + b2 * Z2[s][n][c]        n < N, s < slabs
// (Z1 / Z2 nullable; out may alias Z1 or Z2: every element is read and written by the same lane, X must not alias out)
struct GsoGemmArgs {
    const float* M;      // [NP][NP] dense, zero padded
    const float* X;      // [slabs][N][16]
    const float* Z1;
    const float* Z2;
    float* out;
    float alpha, b1, b2;
    int N, NP, row_tiles, col_tiles;   // row_tiles = ceil(N / 128), col_tiles = ceil(slabs / (2 * NTW))
    long slabs;
};

// Workgroup = 4 waves in a 2 x 2 arrangement, wave (wm, wn) owns 64 nodes x NTW slabs = 4 x NTW MFMA tiles (16 NTW
// accumulator VGPRs, 4 NTW independent MFMA chains).  A operand = operator rows (lane: node l15, 4 consecutive source nodes =
// one 16-B LDS read, the "16-chunk" k permutation of stgcn_device.hip.h), B operand = activations (lane: source node 4g+s,
// channel l15), so D leaves lane (g, l15) with channel l15 of nodes 4g..4g+3: 64-B store segments, the 16 x 16 tile is 1 KiB
// contiguous.  Per 32-node chunk a wave issues 32 NTW MFMAs (1024 NTW cycles) against 4 + NTW 16-B global loads and
// 8 + 8 NTW LDS reads per lane: the kernel is MFMA-bound as long as the next chunk (prefetched into registers during the
// MFMAs) arrives in time.  Tiles that share operator rows run on one XCD (xcd_item): per XCD the resident workgroups stream
// a few operator row tiles and all column tiles in lockstep through its L2.
template <int NTW>
__global__ __launch_bounds__(256) void gso_gemm_kernel(GsoGemmArgs a) {
    extern __shared__ float stgcn_smem[];
    constexpr int SL = 2 * NTW, LDX = SL * 16 + 4;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wm = w & 1, wn = w >> 1;
    const int item = xcd_item((int)blockIdx.x, a.row_tiles * a.col_tiles);
    const int rt = item / a.col_tiles, ct = item - rt * a.col_tiles;
    const int n0 = rt * kGtBM, N = a.N, NP = a.NP;
    const long slab0 = (long)ct * SL;
    float* const Ms = stgcn_smem;                         // [2][128][LDM]
    float* const Xs = stgcn_smem + 2 * kGtBM * kGtLDM;    // [2][32][LDX]

    f32x4 pm[4], px[NTW];
    auto fetch = [&](int kb) {
        const int k0 = kb * kGtBK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;                 // 8 float4 per operator row
            pm[i] = ld4(a.M + (size_t)(n0 + row) * NP + k0 + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {   // 32 nodes x SL slabs x 4 float4 = 128 SL = 256 NTW float4
            const int f = tid + 256 * i, c4 = f & 3, ml = (f >> 2) & 31, sl = f >> 7;   // 128 threads read 2 KiB of one slab
            const int m = k0 + ml;
            const long slab = slab0 + sl;
            px[i] = (m < N && slab < a.slabs) ? ld4(a.X + ((size_t)slab * N + m) * 16 + c4 * 4) : zero4();
        }
    };
    auto stage = [&](int buf) {
        float* ms = Ms + buf * kGtBM * kGtLDM;
        float* xs = Xs + buf * kGtBK * LDX;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + 256 * i, row = f >> 3, c4 = f & 7;
            st4(ms + row * kGtLDM + c4 * 4, pm[i]);
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int f = tid + 256 * i, c4 = f & 3, ml = (f >> 2) & 31, sl = f >> 7;
            st4(xs + ml * LDX + sl * 16 + c4 * 4, px[i]);
        }
    };

    f32x4 acc[4][NTW];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = zero4();

    const int nkb = (N + kGtBK - 1) / kGtBK;   // operator columns >= N are zero padding
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) fetch(kb + 1);
        const float* ms = Ms + buf * kGtBM * kGtLDM + (wm * 64 + l15) * kGtLDM + 4 * g;
        const float* xs = Xs + buf * kGtBK * LDX + 4 * g * LDX + wn * NTW * 16 + l15;
#pragma unroll
        for (int kc = 0; kc < kGtBK / 16; ++kc) {
            f32x4 af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = ld4(ms + mt * 16 * kGtLDM + kc * 16);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float bf[NTW];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bf[nt] = xs[(kc * 16 + s) * LDX + nt * 16];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = mfma4(af[mt][s], bf[nt], acc[mt][nt]);
            }
        }
        if (kb + 1 < nkb) stage(buf ^ 1);   // the other buffer was last read before the barrier that ended step kb - 1
        __syncthreads();
    }

    // epilogue: acc[mt][nt][r] = (M X)[node n0 + wm*64 + mt*16 + 4g + r][slab slab0 + wn*NTW + nt][channel l15]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const long slab = slab0 + wn * NTW + nt;
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

template <typename ET>
__global__ __launch_bounds__(256) void gconv_rows_fwd_kernel(GcRowsFwdArgs a) {
    typedef Mma<ET> MM;
    const ET* const X0_ = et_ptr<ET>(a.X0);
    const ET* const Xk_ = et_ptr<ET>(a.Xk);
    ET* const G_ = et_ptr<ET>(a.G);
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
    extern __shared__ float stgcn_smem[];          // [4 waves][16][20]
    float* const To = stgcn_smem + w * (16 * 20);
    const long tiles = (a.rows + 15) >> 4;
    for (long tile = (long)blockIdx.x * 4 + w; tile < tiles; tile += (long)gridDim.x * 4) {
        const long row0 = tile << 4, row = row0 + l15;
        const bool in = row < a.rows;
        // (all requests of the tile first -- the terms' A fragments raw, the residual values -- then the MFMAs: a load -> unpack -> MFMA chain
        //  per term waited for every load in turn)
        const long rowc = in ? row : a.rows - 1;
        Raw4<ET> xr[kGcMaxTerms];
#pragma unroll
        for (int k = 0; k < kGcMaxTerms; ++k) {
            if (k < a.terms && !(a.kipf && k == 0)) {
                const ET* Xs = k == 0 ? X0_ : Xk_ + (size_t)(k - 1) * a.kstride;
                xr[k] = ldraw4(Xs + (size_t)rowc * 16 + 4 * g);   // A[row = l15][c = 4g + s]
            }
        }
        Raw4<ET> x0r = xr[0];
        if (a.kipf) x0r = ldraw4(X0_ + (size_t)rowc * 16 + 4 * g);   // (the residual; term 0 of the Kipf conv is not a product)
        f32x4 y = zero4();
#pragma unroll
        for (int k = 0; k < kGcMaxTerms; ++k) {
            if (k < a.terms && !(a.kipf && k == 0)) y = MM::mma(MM::cvt(in ? cvt4(xr[k]) : zero4()), MM::cvt(wf[k]), y);
        }
        // D[row = 4g + r][j = l15] -> row major through the wave's LDS tile: one 4-element store per lane, the residual is the A fragment of term 0
#pragma unroll
        for (int r = 0; r < 4; ++r) To[(4 * g + r) * 20 + l15] = y[r] + bb;
        wave_lds_sync();
        const f32x4 o = ld4(To + l15 * 20 + 4 * g) + cvt4(x0r);
        if (in) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(o[i], 0.f);
            stx4(G_ + (size_t)row * 16 + 4 * g, v);
        }
        wave_lds_sync();   // (the next tile overwrites it)
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

constexpr int kGcRowsTile = 3 * 16 * 20;   // floats of a wave's three 16 x 16 transposition tiles (row stride 20) in gconv_rows_bwd_kernel
inline size_t gc_rows_bwd_lds_bytes(int terms) {   // the tiles alias the [4][(terms + 1) * 256] reduction buffer of the kernel's end
    const size_t red = (size_t)4 * (terms + 1) * 256, tl = (size_t)4 * kGcRowsTile;
    return (red > tl ? red : tl) * sizeof(float);
}
template <typename ET>
__global__ __launch_bounds__(256) void gconv_rows_bwd_kernel(GcRowsBwdArgs a) {
    typedef Mma<ET> MM;
    const ET* const dY_ = et_ptr<ET>(a.dY);
    const ET* const X0_ = et_ptr<ET>(a.X0);
    const ET* const Xk_ = et_ptr<ET>(a.Xk);
    ET* const Gk_ = et_ptr<ET>(a.Gk);
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
    // Every tensor of a tile moves as ONE 4-element access per lane in the fragment layout that is also row major (row l15, channels
    // 4g .. 4g + 3: a wave instruction covers the 16 x 16 tile contiguously); the transposed operands (B = dY with the rows as K, A = X_k^T)
    // and the G_k tiles (D leaves a lane with one channel of 4 rows) go through wave-private LDS tiles.  Gathered / scattered straight from
    // memory it was 4 + 4 terms scalar loads and 4 terms scalar stores per tile and lane, each term's round trip waited for in turn.
    constexpr int LDT = 20;
    float* const Ty = stgcn_smem + w * kGcRowsTile, * const Tx = Ty + 16 * LDT, * const To = Tx + 16 * LDT;
    for (long tile = t0 + w; tile < t1; tile += 4) {
        const long row0 = tile << 4, row = row0 + l15;
        const bool in = row < a.rows;
        const size_t ro = (size_t)(in ? row : a.rows - 1) * 16 + 4 * g;
        const Raw4<ET> yr = ldraw4(dY_ + ro);
        Raw4<ET> xr[kGcMaxTerms];
#pragma unroll
        for (int k = 0; k < kGcMaxTerms; ++k)
            if (k < terms) xr[k] = ldraw4((k == 0 ? X0_ : Xk_ + (size_t)(k - 1) * a.kstride) + ro);
        const f32x4 ya = in ? cvt4(yr) : zero4();                                          // A[row = l15][j = 4g + s]
        st4(Ty + l15 * LDT + 4 * g, ya);
        wave_lds_sync();
        const f32x4 yb = gather4(Ty + (4 * g) * LDT + l15, LDT);                          // B[kk = row 4g + s][col = j = l15]
        const f32x4 ones = {1.f, 1.f, 1.f, 1.f};
        const typename MM::frag fya = MM::cvt(ya), fyb = MM::cvt(yb);
        db = MM::mma(MM::cvt(ones), fyb, db);
#pragma unroll
        for (int k = 0; k < kGcMaxTerms; ++k) {
            if (k < terms) {
                st4(Tx + l15 * LDT + 4 * g, in ? cvt4(xr[k]) : zero4());
                wave_lds_sync();
                const f32x4 xa = gather4(Tx + (4 * g) * LDT + l15, LDT);                  // A[i = l15][kk = row 4g + s]
                f32x4 gk = zero4();
                MM::mma_2x(MM::cvt(xa), fyb, dw[k], fya, MM::cvt(wt[k]), gk);
#pragma unroll
                for (int r = 0; r < 4; ++r) To[(4 * g + r) * LDT + l15] = gk[r];          // D[row = 4g + r][i = l15]
                wave_lds_sync();
                f32x4 o = ld4(To + l15 * LDT + 4 * g);
                if (k == 0) o += ya;
                if (in) stx4(Gk_ + (size_t)k * a.gstride + (size_t)row * 16 + 4 * g, o);
                wave_lds_sync();   // (the next term overwrites both tiles)
            }
        }
    }
    __syncthreads();   // (the tiles above live in the reduction buffer below)
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

// ================================================================================================
// bf16 operator products (opt-in: stgcn_set_gc_precision; BASELINE.json configs[4] is a bf16 config).
// The fp32-input MFMA runs at 1/16 of the bf16 rate, and at N = 8192 the operator GEMMs are 99 % of the step's FLOPs, so
// the tiled path can form them on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulation):
//     precision 2 ("bf16")   : operands rounded to bf16                      -- ~3 significant digits per product
//     precision 1 ("bf16x3") : operands split x = hi + lo (two bf16), M X ~= Mh Xh + Mh Xl + Ml Xh (the dropped Ml Xl term
//                              and the split residuals are ~2^-17 relative): fp32-class results at 3 bf16 MFMAs per product,
//                              still 16/3 of the fp32 MFMA rate
// Both operands are k-contiguous 16-bit planes so that a fragment (8 consecutive k of one row) is one 16-B LDS read:
//     operator          Mh, Ml : [NP][NP]                      (stgcn_gso_prepare, second matrix slot of gso_pad / gso_t_pad)
//     activations  "operand form" Xh, Xl : [CP][NP], row = GEMM column = slab * 16 + channel, CP = roundup128(slabs * 16)
// gc_pack_operand_kernel converts [slabs][N][16] fp32 into operand form once per chain; every GEMM of the recursion writes
// its fp32 result AND the operand form of that result for the next term from its epilogue (D leaves a lane with 4
// consecutive nodes of one column = one 8-B store per plane), so no other transposition pass exists.
// The k order inside the 32-deep MFMA is irrelevant for correctness: lane (row l15, group g) of A and lane (column l15,
// group g) of B hold the same 8 k values.
// ================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// LDS image of a staged plane: 128 rows of BK bf16 (BK = 64: 128-B rows, BK = 32: 64-B rows), NO padding; the 16-B chunk c of
// row r sits at chunk position c ^ gb_swz<BK>(r).  A ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} (+32 for the upper half) over 64 banks (MI355X_MICROARCH.md, LDS): a fragment read takes rows
// l15 = 0-3,12-15 at chunk g and rows 4-11 at chunk g+1 in ONE group, which collides 2-way for every padded row stride of the
// form 16 B x odd (r56: a third of the LDS-active cycles were conflicts).  BK = 64: with (r >> 1) & 7 the 8 even and the 8 odd
// rows of a group each cover the 8 chunk positions of their 128-B half of the bank row exactly once.  BK = 32 (four rows per
// 256-B bank row): the four rows of a group that share r mod 4 are r = q, q+12 at chunk g and q+4, q+8 at chunk g+1, and
// (0, 3, 2, 1)[r >> 2] sends them to four different positions.  The lane-linear image a global -> LDS copy writes (8 or 16 whole
// rows per wave instruction) is conflict-free by construction.
template <int BK> __device__ __forceinline__ int gb_swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); }
constexpr int kGbDefaultBK = 32;   // r61: equal on the 1280-workgroup launches, 7-16 % faster on the 768-workgroup ones
inline int gb_lds_floats(int split, int bk) { return 2 * (split ? 4 : 2) * 128 * (bk / 2); }   // BK 64: 64 / 128 KB, BK 32: 32 / 64 KB

__device__ __forceinline__ unsigned bf16_rne(float x) {   // round-to-nearest-even, finite inputs
    const unsigned u = __builtin_bit_cast(unsigned, x);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_to_f32(unsigned h) { return __builtin_bit_cast(float, h << 16); }
// 4 consecutive values -> 4 bf16 (hi) and 4 bf16 of the remainders (lo), packed little-endian
__device__ __forceinline__ void bf16_split4(f32x4 v, u32x2& hi, u32x2& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = bf16_rne(v[i]);
        l[i] = bf16_rne(v[i] - bf16_to_f32(h[i]));
    }
    hi[0] = h[0] | (h[1] << 16); hi[1] = h[2] | (h[3] << 16);
    lo[0] = l[0] | (l[1] << 16); lo[1] = l[2] | (l[3] << 16);
}

// dense zero-padded bf16 planes of the operator (T = 0) or of its transpose (T = 1): hi[h][i], lo[h][i]
// (LD >= NP: leading dimension of the planes in bf16 elements, see gc_plane_ld)
__global__ __launch_bounds__(256) void gso_bf16_kernel(const float* L, int N, int NP, int LD, int T, unsigned short* hi, unsigned short* lo) {
    const long e = (long)blockIdx.x * kThreads + (long)threadIdx.x;
    if (e >= (long)NP * NP) return;
    const int h = (int)(e / NP), i = (int)(e - (long)h * NP);
    const float v = (h < N && i < N) ? (T ? L[(size_t)i * N + h] : L[(size_t)h * N + i]) : 0.f;
    const unsigned hh = bf16_rne(v);
    const size_t o = (size_t)h * LD + i;
    hi[o] = (unsigned short)hh;
    lo[o] = (unsigned short)bf16_rne(v - bf16_to_f32(hh));
}

// X [slabs][N][16] fp32 -> operand form (hi, lo) [CP][NP]; grid = (ceil(NP / 256), slabs), 256 nodes of one slab per workgroup
template <typename ET>
__global__ __launch_bounds__(256) void gc_pack_operand_kernel(const float* X, int N, int NP, int LD, float* Oh, float* Ol) {
    extern __shared__ float stgcn_smem[];   // [256][17]
    const int tid = threadIdx.x, m0 = (int)blockIdx.x * 256;
    const long slab = blockIdx.y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, node = f >> 2, c4 = f & 3, m = m0 + node;
        const f32x4 v = m < N ? ldx4(et_ptr<ET>(X) + ((size_t)slab * N + m) * 16 + c4 * 4) : zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j) stgcn_smem[node * 17 + c4 * 4 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, c = e >> 6, q = e & 63, m = m0 + 4 * q;   // a wave writes 512 contiguous bytes of one column
        if (m < NP) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = stgcn_smem[(4 * q + j) * 17 + c];
            u32x2 hi, lo;
            bf16_split4(v, hi, lo);
            const size_t o = (((size_t)slab * 16 + c) * LD + m) >> 1;   // float units (2 bf16 each)
            *reinterpret_cast<u32x2*>(Oh + o) = hi;
            *reinterpret_cast<u32x2*>(Ol + o) = lo;
        }
    }
}

struct GsoGemmBfArgs {
    const float* Mh;     // operator planes, [NP][NP] bf16 viewed as [NP][NP/2] floats
    const float* Ml;
    const float* Xh;     // operand form of the input, [CP][NP] bf16
    const float* Xl;
    float* Oh;           // operand form of the result for the next term (nullable)
    float* Ol;
    const float* Z1;     // fp32 epilogue exactly as gso_gemm_kernel
    const float* Z2;
    float* out;
    float alpha, b1, b2;
    int N, NP, LD, row_tiles, col_tiles;   // LD: leading dimension of all 16-bit planes (bf16 elements, gc_plane_ld)
    long slabs;
};

// Same 128 x 128 workgroup tile and 2 x 2 waves of 64 x 64 as gso_gemm_kernel; per BK-deep step a wave issues BK/2 (bf16) or
// 3 BK/2 (bf16x3) MFMAs of 16 cycles.  BK = 32 halves the LDS footprint (32 / 64 KB): three (bf16) / two (bf16x3) workgroups
// per CU instead of two / one, i.e. more independent waves to fill the copy / barrier / fragment-read gaps of each other.
template <int SPLIT, int BK, typename ET>
__global__ __launch_bounds__(256) void gso_gemm_bf16_kernel(GsoGemmBfArgs a) {
    extern __shared__ float stgcn_smem[];
    constexpr int NPL = SPLIT ? 2 : 1;              // planes per operand
    constexpr int LD = BK / 2;                      // floats per staged row
    constexpr int PLANE = 128 * LD;                 // floats per staged plane (128 rows)
    constexpr int BUF = 2 * NPL * PLANE;            // floats per pipeline buffer: A planes then B planes
    constexpr int CPR = BK / 8;                     // 16-B chunks per row (8 / 4)
    constexpr int RPI = 64 / CPR;                   // rows one wave instruction of the global -> LDS copy fills (8 / 16)
    constexpr int NI = 128 / (4 * RPI);             // copy instructions per plane and wave (4 / 2)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wm = w & 1, wn = w >> 1;
    const int item = xcd_item((int)blockIdx.x, a.row_tiles * a.col_tiles);
    const int rt = item / a.col_tiles, ct = item - rt * a.col_tiles;
    const int n0 = rt * 128, c0 = ct * 128, N = a.N, NPH = a.LD >> 1;   // NPH: floats per 16-bit row

    // Staging: direct global -> LDS copies (glds16), no VGPR round trip and no ds_write pass.  One instruction of a wave fills
    // RPI rows = 1 KiB of the plane (lane-linear); the swizzle of the LDS image is applied to the per-lane SOURCE chunk: the
    // lane whose slot is (row, position p) fetches chunk p ^ gb_swz(row) of that row.  Wave w, instruction i covers rows
    // (w + 4 i) * RPI .. + RPI - 1 of every plane.
    const int srow = lane / CPR, spos = lane % CPR;   // this lane's slot inside a block of RPI rows
    auto issue = [&](int kb, int buf) {
        const int k0h = kb * (BK / 2);   // float units
        float* base = stgcn_smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int rb = (w + 4 * i) * RPI, row = rb + srow;
            const int ch = (spos ^ gb_swz<BK>(row)) << 2;
            const size_t oa = (size_t)(n0 + row) * NPH + k0h + ch, ob = (size_t)(c0 + row) * NPH + k0h + ch;
            glds16(a.Mh + oa, base + rb * LD);
            glds16(a.Xh + ob, base + NPL * PLANE + rb * LD);
            if (SPLIT) {
                glds16(a.Ml + oa, base + PLANE + rb * LD);
                glds16(a.Xl + ob, base + (NPL + 1) * PLANE + rb * LD);
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = zero4();

    const int nkb = (N + BK - 1) / BK;   // k >= N: zero operator columns, zero operand padding
    issue(0, 0);
    __syncthreads();   // (drains vmcnt: the copies of chunk 0 have landed)
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        if (kb + 1 < nkb) issue(kb + 1, buf ^ 1);   // the other buffer was last read before the barrier that ended step kb - 1
        const float* As = stgcn_smem + buf * BUF + (wm * 64 + l15) * LD;
        const float* Bs = stgcn_smem + buf * BUF + NPL * PLANE + (wn * 64 + l15) * LD;
        const int sw = gb_swz<BK>(l15);   // tile rows start at multiples of 16: gb_swz(row) == gb_swz(l15)
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 ah[4], al[4], bh[4], bl[4];
            const int co = ((ks * 4 + g) ^ sw) << 2;   // swizzled position of chunk ks*4 + g (8 bf16) in the row
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ah[t] = __builtin_bit_cast(bf16x8, ld4(As + t * 16 * LD + co));
                bh[t] = __builtin_bit_cast(bf16x8, ld4(Bs + t * 16 * LD + co));
                if (SPLIT) {
                    al[t] = __builtin_bit_cast(bf16x8, ld4(As + PLANE + t * 16 * LD + co));
                    bl[t] = __builtin_bit_cast(bf16x8, ld4(Bs + PLANE + t * 16 * LD + co));
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    if (SPLIT) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bl[nt], acc[mt][nt], 0, 0, 0);
                    }
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                }
        }
        __syncthreads();   // all waves done with `buf`; the copies into the other buffer have landed
    }

    // acc[mt][nt][r] = (M X)[node n0 + wm*64 + mt*16 + 4g + r][column c0 + wn*64 + nt*16 + l15]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = c0 + wn * 64 + nt * 16 + l15;
        const long slab = col >> 4;
        const int ch = col & 15;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int nb = n0 + wm * 64 + mt * 16 + 4 * g;
            f32x4 v = zero4();
            if (slab < a.slabs) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nb + r < N) {
                        const size_t o = ((size_t)slab * N + nb + r) * 16 + ch;
                        float t = a.alpha * acc[mt][nt][r];
                        if (a.Z1) t += a.b1 * ldx1(et_ptr<ET>(a.Z1) + o);
                        if (a.Z2) t += a.b2 * ldx1(et_ptr<ET>(a.Z2) + o);
                        stx1(et_ptr<ET>(a.out) + o, t);
                        v[r] = t;
                    }
                }
            }
            if (a.Oh) {   // operand form of the result (zeros in the node / column padding)
                u32x2 hi, lo;
                bf16_split4(v, hi, lo);
                const size_t o = ((size_t)col * a.LD + nb) >> 1;
                *reinterpret_cast<u32x2*>(a.Oh + o) = hi;
                *reinterpret_cast<u32x2*>(a.Ol + o) = lo;
            }
        }
    }
}


// ================================================================================================
// bf16 operator product on 256 x (32 * NT) workgroup tiles (round 3; the 128 x 128 kernel above moves 4 MB of operand panels per
// workgroup through L2 for 128 x 128 x K outputs and reads each LDS byte for 32 FLOP: PMC r56 put it at 2.7 GB of L2 -> fabric reads
// per launch against 0.18 GB of unique operands, with the LDS array as busy as the matrix pipes).
//   * 8 waves = 4 (rows) x 2 (columns); a wave owns 64 rows x 16 * NT columns = 4 x NT accumulator tiles: per 32-deep step it reads
//     4 + NT fragments (16 B per lane each) for 4 * NT MFMAs -- NT = 10: 40 MFMAs of 17.5 cycles per 14 KB of LDS reads, i.e. the LDS
//     array is busy 62 % of the matrix time instead of ~100 %;
//   * ONE workgroup per CU and a grid of (N / 256) x (columns / (32 * NT)) tiles that is a whole number of rounds: the host picks NT so
//     that the C5 launches are exactly 256 tiles (2560 columns: NT = 10, 1536 columns: NT = 6) -- no tail round;
//   * tiles are handed out row-major through xcd_item: the 32 workgroups of an XCD hold 4 row panels x all column panels, so every
//     operator row panel is fetched into that XCD's L2 once per launch and every column panel 8 times in total (once per XCD):
//     (134 + 8 x 42) MB = 0.47 GB of fabric reads at the C5 size;
//   * staging as above: direct global -> LDS copies (global_load_lds_dwordx4), XOR-swizzled unpadded image, double buffered.
// SPLIT-free (one MFMA per product): bf16 activations, or fp32 activations with stgcn_set_gc_precision("bf16").
// Requires NP % 256 == 0; the operand form is allocated gc_operand_slack rows beyond CP so that the last column tile may read (and write
// zeros) past the 128-row granularity of CP.
// ================================================================================================
constexpr int kGbBigBM = 256;
#ifndef STGCN_GEMM_DBG
#define STGCN_GEMM_DBG 0   // timing experiments only (wrong results): 3 = no MFMAs (the copy + fragment-read stream alone)
#endif
// pipeline buffers: 32-deep steps: 4 (the copies of step kb + 3 are issued while step kb computes); 64-deep steps: 2 (whole 128-B lines per
// staged row, the copies of step kb + 1 are issued at the start of step kb)
constexpr int kGbEpiLd = 20;   // row stride (floats) of a wave's 64 x 16 transposition tile in the epilogue of the big kernel
constexpr int gb_big_stages(int bk) { return bk == 64 ? 2 : 4; }
inline size_t gb_big_lds_bytes(int nt, int bk) { return (size_t)gb_big_stages(bk) * (kGbBigBM + 32 * nt) * (bk / 2) * sizeof(float); }
// wait until at most N of this wave's vector-memory operations (here: global -> LDS copies) are outstanding, WITHOUT draining the rest
// (__syncthreads() waits for vmcnt(0): with one workgroup per CU that puts a whole HBM round trip into every pipeline step)
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(N >= 0 && N <= 15, "vmcnt immediates used by the pipelined GEMM");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
template <int NT, typename ET, int BK>
__global__ __launch_bounds__(512) void gso_gemm_bf16_big_kernel(GsoGemmBfArgs a) {
    extern __shared__ float stgcn_smem[];
    constexpr int LD = BK / 2, BM = kGbBigBM, BN = 32 * NT, KS = BK / 32;
    constexpr int PA = BM * LD, BUF = PA + BN * LD;   // floats: A plane, whole pipeline buffer
    constexpr int CPR = BK / 8, RPI = 64 / CPR;       // 16-B chunks per staged row; rows one wave instruction of the global -> LDS copy fills
    constexpr int NIA = BM / RPI, NI = NIA + BN / RPI;
    constexpr int CNT = (NI + 7) / 8;                 // copy instructions per wave and stage (the same for every wave: slots past NI repeat the last block)
    constexpr int ST = gb_big_stages(BK), D = ST - 1; // prefetch distance in pipeline steps
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int wm = w & 3, wn = w >> 2;
    const int item = xcd_item((int)blockIdx.x, a.row_tiles * a.col_tiles);
    const int rt = item / a.col_tiles, ct = item - rt * a.col_tiles;
    const int n0 = rt * BM, c0 = ct * BN, N = a.N, NPH = a.LD >> 1;   // NPH: floats per 16-bit row
    const int srow = lane / CPR, spos = lane % CPR;   // this lane's slot inside a block of RPI rows
    // swizzled source chunk of the slot: blocks start at multiples of RPI rows; gb_swz<32> has period 16 = RPI, gb_swz<64> period 16 = 2 RPI
    // (the second variant serves the odd blocks)
    const int sch0 = (spos ^ gb_swz<BK>(srow)) << 2, sch1 = (spos ^ gb_swz<BK>(srow + RPI)) << 2;
    auto issue = [&](int kb, int buf, int j0, int j1) __attribute__((always_inline)) {   // copy instructions j0 .. j1 - 1 of stage kb
        const int k0h = kb * (BK / 2);   // float units
        float* base = stgcn_smem + buf * BUF;
#pragma unroll
        for (int j = j0; j < j1; ++j) {   // (wave-uniform)
            const int i = w + 8 * j < NI ? w + 8 * j : NI - 1;   // (a repeated block is copied twice: same bytes to the same place)
            const int sch = (i & 1) ? sch1 : sch0;                // (NIA is even: the parity of a block within its plane is the parity of i)
            if (i < NIA) glds16(a.Mh + (size_t)(n0 + i * RPI + srow) * NPH + k0h + sch, base + i * RPI * LD);
            else glds16(a.Xh + (size_t)(c0 + (i - NIA) * RPI + srow) * NPH + k0h + sch, base + PA + (i - NIA) * RPI * LD);
        }
    };
    f32x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero4();

    const int nkb = (N + BK - 1) / BK;   // k >= N: zero operator columns, zero operand padding
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nkb) issue(s, s, 0, CNT);
    if (nkb >= D) wait_vmcnt_le<(D - 1) * CNT>(); else wait_vmcnt_le<0>();   // stage 0 has landed (this wave's copies; the barrier covers the others')
    barrier_only();
    const int sw = gb_swz<BK>(l15);   // (tile rows start at multiples of 16: gb_swz(row) == gb_swz(l15))
    // The copy instructions of a stage are spread over the MFMA groups of the step (a burst of them fills the vector-memory queue and the
    // issuing wave's MFMAs wait behind it).  32-deep steps: both groups (the stage is needed two steps later); 64-deep steps: the first three
    // of the four groups (the stage is read right after this step's closing barrier: the last group is its landing time).
    constexpr int CG = BK == 64 ? 3 : 2, CQ = (CNT + CG - 1) / CG;
    auto step = [&](int kb, auto issue_tag) __attribute__((always_inline)) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        constexpr bool MM = STGCN_GEMM_DBG != 3;
        constexpr int NH = (NT + 1) / 2, R = NT - NH;
        const int buf = kb % ST;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int co = ((ks * 4 + g) ^ sw) << 2;   // swizzled position of this lane group's 8 k values in a staged row
            const float* As = stgcn_smem + buf * BUF + (wm * 64 + l15) * LD + co;
            const float* Bs = stgcn_smem + buf * BUF + PA + (wn * 16 * NT + l15) * LD + co;
            // fragment reads in two groups so that the second group's LDS reads are in flight during the first group's MFMAs (all the
            // reads first and one lgkmcnt(0) in front of 4 * NT MFMAs left the LDS array and the matrix pipes taking turns)
            bf16x8 ah[4], bh[NT];
#pragma unroll
            for (int t = 0; t < 4; ++t) ah[t] = __builtin_bit_cast(bf16x8, ld4(As + t * 16 * LD));
#pragma unroll
            for (int t = 0; t < NH; ++t) bh[t] = __builtin_bit_cast(bf16x8, ld4(Bs + t * 16 * LD));
            __builtin_amdgcn_sched_barrier(0);
            // first half of the columns; between its MFMAs the reads of the second half and this group's share of the copies of step
            // kb + D (their buffer was last read in step kb - 1: every wave has passed that step's closing barrier)
            constexpr int G0 = 0;
            const int ga = 2 * ks, gb = 2 * ks + 1;   // the two MFMA groups of this slice
            const int ca0 = ga * CQ < CNT ? ga * CQ : CNT, ca1 = (ga + 1) * CQ < CNT ? (ga + 1) * CQ : CNT;
            const int cb0 = gb * CQ < CNT ? gb * CQ : CNT, cb1 = (gb + 1) * CQ < CNT ? (gb + 1) * CQ : CNT;
            const int na = ISSUE && ga < CG ? ca1 - ca0 : 0, nb = ISSUE && gb < CG ? cb1 - cb0 : 0;
#pragma unroll
            for (int t = NH; t < NT; ++t) bh[t] = __builtin_bit_cast(bf16x8, ld4(Bs + t * 16 * LD));
            if (na > 0) issue(kb + D, (kb + D) % ST, ca0, ca1);
#pragma unroll
            for (int nt = 0; nt < NH; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (MM) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    else asm volatile("" ::"v"(ah[mt]), "v"(bh[nt]));
                }
#pragma unroll
            for (int t = 0; t < R; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, G0);   // 4 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, G0);   // 1 LDS read
            }
            {   // (a second pipeline over the same MFMAs: mixed in one pipeline the solver clumped the reads and the copies)
                constexpr int MPA = (4 * NH) / CQ > 0 ? (4 * NH) / CQ : 1;
#pragma unroll
                for (int j = 0; j < CQ; ++j) {
                    if (j < na) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MPA, 2);   // MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 2);     // one global -> LDS copy
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // second half of the columns with its share of the copies (a copy instruction issued among bare MFMAs costs ~60 cycles of
            // issue, 100 - 185 in front of the LDS reads: MI355X_MICROARCH.md)
            if (nb > 0) issue(kb + D, (kb + D) % ST, cb0, cb1);
#pragma unroll
            for (int nt = NH; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    if (MM) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[nt], acc[mt][nt], 0, 0, 0);
                    else asm volatile("" ::"v"(ah[mt]), "v"(bh[nt]));
                }
            {
                constexpr int MPC = (4 * R) / CQ > 0 ? (4 * R) / CQ : 1;
#pragma unroll
                for (int j = 0; j < CQ; ++j) {
                    if (j < nb) {
                        __builtin_amdgcn_sched_group_barrier(0x008, MPC, 1);   // MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);     // one global -> LDS copy
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // stage kb + 1 must have landed before the next step reads it: while D stages are in flight behind it, allow (D - 1) of them to
        // stay outstanding; in the tail (nothing issued this step) drain
        if constexpr (ISSUE) wait_vmcnt_le<(D - 1) * CNT>(); else wait_vmcnt_le<0>();
        barrier_only();   // all waves done reading `buf`, and every wave's share of stage kb + 1 has landed
    };
    int kb = 0;
    for (; kb + D < nkb; ++kb) step(kb, std::true_type());    // steady state: one stage issued per step
    for (; kb < nkb; ++kb) step(kb, std::false_type());       // tail: nothing left to issue, drain
    if (STGCN_GEMM_DBG == 8) {   // timing experiment: no epilogue (one store keeps the accumulators alive)
        f32x4 v = zero4();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) v += acc[mt][nt];
        if (v[0] + v[1] + v[2] + v[3] == 12345.678f) stx1(et_ptr<ET>(a.out), v[0]);
        return;
    }
    // acc[mt][nt][r] = (M X)[node n0 + wm*64 + mt*16 + 4g + r][column c0 + wn*16*NT + nt*16 + l15]: a lane holds ONE channel of 4 nodes,
    // the slab layout [slab][node][16] wants 4 channels of one node per lane.  Written straight from the fragments (160 2-byte stores per
    // lane, 320 2-byte loads of the addends) the epilogue took 66 us of a 283 us launch and 137 us with two addends (r3-19): the 64 x 16
    // piece of every column group goes through a wave-private LDS tile (the pipeline buffers are free: every wave has passed the last
    // step's barrier) and leaves as 8 / 16-byte accesses, 16 nodes x 32 / 64 B contiguous per wave instruction.
    float* T = stgcn_smem + w * (64 * kGbEpiLd);
    const int pn = lane >> 1, pc = (lane & 1) * 8;   // transposed slot: node 32 i + pn, channels pc .. pc + 7
    const bool addends = a.Z1 || a.Z2;
    // one column group through the tile; H1 / H2: addends present (wave-uniform, resolved once per kernel: the loads of a group are issued
    // together and nothing in the loop branches on them)
    auto slab_pass = [&](int nt, long slab, auto h1, auto h2) __attribute__((always_inline)) {
        constexpr bool H1 = decltype(h1)::value, H2 = decltype(h2)::value;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(mt * 16 + 4 * g + r) * kGbEpiLd + l15] = a.alpha * acc[mt][nt][r];
        wave_lds_sync();
        f32x4 t[2][2], z1[2][2], z2[2][2];
        size_t o[2];
        bool ok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int node = n0 + wm * 64 + 32 * i + pn;
            ok[i] = node < N;
            o[i] = ((size_t)slab * N + (ok[i] ? node : N - 1)) * 16 + pc;   // (clamped: the loads below are unconditional)
            if (H1) ldx8(et_ptr<ET>(a.Z1) + o[i], z1[i][0], z1[i][1]);
            if (H2) ldx8(et_ptr<ET>(a.Z2) + o[i], z2[i][0], z2[i][1]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* tp = T + (32 * i + pn) * kGbEpiLd + pc;
            t[i][0] = ld4(tp);
            t[i][1] = ld4(tp + 4);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (H1) t[i][h] += a.b1 * z1[i][h];
                if (H2) t[i][h] += a.b2 * z2[i][h];
            }
            if (ok[i]) stx8(et_ptr<ET>(a.out) + o[i], t[i][0], t[i][1]);
            if ((H1 || H2) && a.Oh) {   // the operand form below needs the sums in the fragment layout
                st4(tp, t[i][0]);
                st4(tp + 4, t[i][1]);
            }
        }
        wave_lds_sync();
    };
    auto epilogue = [&](auto h1, auto h2) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int colb = c0 + wn * 16 * NT + nt * 16;   // (wave-uniform: one slab per column group)
        const long slab = colb >> 4;
        const bool live = slab < a.slabs;
        if (live) slab_pass(nt, slab, h1, h2);
        if (a.Oh) {   // operand form of the result (zeros in the node / column padding); the low plane only where a split product reads it
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int nb = n0 + wm * 64 + mt * 16 + 4 * g;
                f32x4 v = zero4();
                if (live) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < N) v[r] = addends ? T[(mt * 16 + 4 * g + r) * kGbEpiLd + l15] : a.alpha * acc[mt][nt][r];
                }
                u32x2 hi, lo;
                bf16_split4(v, hi, lo);
                const size_t o = ((size_t)(colb + l15) * a.LD + nb) >> 1;
                *reinterpret_cast<u32x2*>(a.Oh + o) = hi;
                if (a.Ol) *reinterpret_cast<u32x2*>(a.Ol + o) = lo;
            }
        }
        wave_lds_sync();   // (the next column group overwrites the tile)
    }
    };
    if (a.Z1 && a.Z2) epilogue(std::true_type(), std::true_type());
    else if (a.Z1) epilogue(std::true_type(), std::false_type());
    else if (a.Z2) epilogue(std::false_type(), std::true_type());
    else epilogue(std::false_type(), std::false_type());
}

}  // namespace stgcn
