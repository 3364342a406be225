// Backward kernels of the fused ST-Conv block (gfx950).  Math: SURVEY.md section 8a rows a2/a4/a6
// (checked against autograd in oracle/stblock_stages.py).  Data flow: DESIGN.md section 3.
//
//   dy --[ln_bwd_stats]--> c1,c2 --[ln_gate_bwd]--> dZ2, partial dgamma/dbeta
//   dZ2 --[tconv_bwd_data (relu mask)]--> dYg --[gconv_bwd]--> dA, partial dW_gc/db_gc
//   dA --[align_gate_bwd]--> dZ1, partial dWa/dba --[tconv_bwd_data]--> dx
//   (G, dZ2), (x, dZ1) --[tconv_bwd_weight]--> partial dW_eff/db_eff
//   partials --[reduce_kernel]--> parameter gradients in the reference's layouts
#pragma once
#include <stdlib.h>
#include <string.h>
#include "stgcn_device.hip.h"
#include "stgcn_kernels_fwd.hip.h"
#include "stgcn_kernels_gctile.hip.h"
#include "stgcn_kernels_tstep.hip.h"
#include "stgcn_kernels_thin.hip.h"

namespace stgcn {

// ---- which fused kernels replace the stage-per-launch path (host side; plan, launchers and the gradient flush must agree) ----
// STGCN_FUSE=<bit mask> (read once): 1 = tc2_bwd_kernel (LayerNorm/dropout/gate backward + tmp_conv2 weight gradient + transposed conv in
// one launch).  Default: everything on; 0 reproduces the round-1 launch sequence (A/B runs, stage tests).
//                                  2 = tc2_ln_fwd_kernel (tmp_conv2 + gate + LayerNorm + dropout of one slab per workgroup).
//                                  4 = LayerNorm-backward row partials in the epilogue of the kernel that produces dy (stgcn_ln_hook).
//                                  8 = tc1_bwd_kernel (Align + gate backward + tmp_conv1 weight gradient + transposed conv in one launch).
//                                 16 = tc1_fwd_kernel (time-stepping tmp_conv1 + gate + Align forward, weights stationary).
enum FuseBit { FUSE_TC2_BWD = 1, FUSE_TC2_LN_FWD = 2, FUSE_ROWSTATS = 4, FUSE_TC1_BWD = 8, FUSE_TC1_FWD = 16, FUSE_HEAD_LN_FWD = 32, FUSE_HEAD_LN_BWD = 64 };
inline int fuse_mask() {
    static const int m = getenv("STGCN_FUSE") ? atoi(getenv("STGCN_FUSE")) : 0x7fffffff;
    return m;
}
// stage tests: also write the on-chip intermediates of the fused kernels (dZ2) to their round-1 workspace slots
inline int g_debug_stages = 0;
inline bool tc2_ln_fwd_fused_ok(int c1, int c2, int Kt, int N) {
    return (fuse_mask() & FUSE_TC2_LN_FWD) && c1 == 16 && c2 == 64 && Kt >= 2 && Kt <= 4 && N <= 448 && tc2_ln_fwd_lds_bytes(Kt, N) <= 150 * 1024;
}
// shapes tc1_bwd_kernel covers (whether a call uses it also depends on need_dx: the kernel always forms the input gradient)
// test knob (stgcn_set_tc1_bwd_wgs): pretend the device has this many CUs (0 = ask the runtime).  The launch heuristics that size a
// grid by the CU count -- the (item, step) ranges of the two tc1 kernels, the wave count of tc2_ln_fwd -- then take their small-device
// branches on the emulator too: ranges cut inside items, the 8-wave variant of tc2_ln_fwd
inline int g_tc1_bwd_wgs = 0;
inline int device_cus() {
    static int cus = 0;
    if (g_tc1_bwd_wgs > 0) return g_tc1_bwd_wgs;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}
// workgroups per slab of tc2_ln_fwd_kernel (round 6): 1 when the slabs alone fill the device; 2 / 4 when PP * slabs workgroups still fit the
// compute units one each (the peers exchange their LayerNorm statistics through L2).  stgcn_set_tc2ln_peers forces 1 / 2 / 4 (0 = this rule).
constexpr int kTc2LnMaxPeers = 4;
inline int g_tc2ln_peers = 0;
inline int tc2_ln_peers(int N, long slabs2) {
    const int tiles = (N + 15) / 16;
    int pp = 1;
    if (g_tc2ln_peers > 0) pp = g_tc2ln_peers;
    else if (4 * slabs2 <= device_cus()) pp = 4;
    else if (2 * slabs2 <= device_cus()) pp = 2;
    if (N > 384) pp = 1;                       // (the peer instances hold at most 24 node tiles per slab)
    while (pp > 1 && tiles < 2 * pp) pp >>= 1;   // at least two node tiles per workgroup
    return pp;
}
// tc2_bwd_kernel recomputes the gate inputs of tmp_conv2 (and the forward does not store them) for bf16 activations, reads the stored
// ones for fp32 (see the kernel's header comment for the measurement); STGCN_TC2_RECOMP=0/1 forces one
inline bool tc2_recompute(int dtype_bf16) {
    static const int force = STGCN_EXP_ENV("STGCN_TC2_RECOMP") ? atoi(STGCN_EXP_ENV("STGCN_TC2_RECOMP")) : -1;
    return force >= 0 ? force != 0 : dtype_bf16 != 0;
}
// fewest output steps a range of the two tc1 time-stepping kernels is cut down to when a batch offers fewer (window, node tile) items than
// the device has compute units (measured at C2 shapes, profiles/r6-08_tc1_small_batch_ranges.txt: bs 4 tc1_bwd 36.0 -> 21.6 us, tc1_fwd
// 17.6 -> 11.2 us with 2; 1 and 3 are slower -- more partial blocks for the reduction / longer chains)
inline int tc1_min_steps() { return 2; }
inline bool tc1_ts_shape(int c_in, int c0, int c1, int Kt) { return c0 == 64 && c1 == 16 && Kt == 3 && (c_in == 16 || c_in == 32 || c_in == 64); }
inline bool tc1_bwd_shape_ok(int c_in, int c0, int c1, int Kt) {
    return (fuse_mask() & FUSE_TC1_BWD) && tc1_ts_shape(c_in, c0, c1, Kt) && tc1_bwd_lds_bytes(c0, c_in, Kt) <= 150 * 1024;
}
inline bool tc1_fwd_shape_ok(int c_in, int c0, int c1, int Kt) { return (fuse_mask() & FUSE_TC1_FWD) && tc1_ts_shape(c_in, c0, c1, Kt); }
inline bool tc2_bwd_fused_ok(int c1, int c2, int Kt, int T1, int T2) {
    return (fuse_mask() & FUSE_TC2_BWD) && c1 == 16 && ((c2 == 64 && Kt >= 2 && Kt <= 4) || (c2 == 128 && Kt == 3)) && T1 <= kTsMaxT &&
           tc2_bwd_lds_bytes(c2, Kt, T1, T2) <= 80 * 1024;
}

// ---- which graph-conv implementation a block uses (host side; plan, launchers and stgcn_gso_prepare must agree) -------
// Slab-resident kernels (gconv_fwd_kernel / gconv_bwd_kernel): up to 512 nodes and as many terms as the backward's LDS
// footprint allows; everything else runs the tiled GEMM path of stgcn_kernels_gctile.hip.h.  The node threshold is a
// runtime knob (stgcn_set_gc_tiled_min_nodes) so that both paths can be compared on the same graph.
inline int g_gc_tiled_min_n = 513;
// arithmetic of the operator products on the tiled path: 0 = fp32 MFMA (exact fp32, default), 1 = bf16x3 (split operands,
// fp32-class), 2 = bf16 (stgcn_set_gc_precision; see stgcn_kernels_gctile.hip.h)
inline int g_gc_precision = 0;
// operator products of the slab-resident graph conv (N <= 512): 0 exact fp32 MFMAs, 1 bf16x3 (stgcn_kernels_gcslab16.hip.h)
inline int g_slab_gc_precision = 0;
// matrix products of the BACKWARD kernels of fp32 blocks: 0 exact fp32 MFMAs (default), 1 "bf16x3" (Mma<f32x>: split operands, three bf16
// MFMAs per product, ~2^-16 relative -- inside the 1e-3 gradient bar, outside "exact fp32"; stgcn_set_bwd_precision)
inline int g_bwd_precision = (STGCN_EXP_ENV("STGCN_BWD_PRECISION") && !strcmp(STGCN_EXP_ENV("STGCN_BWD_PRECISION"), "bf16x3")) ? 1 : 0;
inline long gc_operand_cols(long slabs) { return (slabs * 16 + 127) / 128 * 128; }   // CP: rows of the bf16 operand form
// rows ALLOCATED per operand plane: the wide column tiles of gso_gemm_bf16_big_kernel (up to 320 columns) may run past CP
inline long gc_operand_alloc(long slabs) { return gc_operand_cols(slabs) + 384; }
// Leading dimension (bf16 elements) of every 16-bit plane (operator hi / lo, operand form).  NP itself is a power-of-two
// multiple of 128 for the sizes that matter (8192 nodes: 16 KiB rows), so the 128 rows of a tile would all start in the same
// L2 channel; the pad (stgcn_set_gc_ld_pad, multiple of 8 elements) staggers them.
inline int g_gc_ld_pad = 0;
inline int gc_plane_ld(int NP) { return NP + g_gc_ld_pad; }
inline bool gc_is_tiled(int N, int terms) {
    if (N >= g_gc_tiled_min_n) return true;
    const long NP = (N + 15) / 16 * 16;
    return ((long)terms * 16 * (NP + 4) + NP * 20) * (long)sizeof(float) > 160 * 1024;
}
inline int gc_padded_nodes(int N, int terms) { return gc_is_tiled(N, terms) ? (N + kGtBM - 1) / kGtBM * kGtBM : (N + 15) / 16 * 16; }

// ================================================================================================
// Geometry of the backward launches and of the partial-sum arena (host + plan use the same numbers)
// ================================================================================================
struct WgradGeom {
    int M, Mtiles, MTW, mchunks, Mpad, NC, rows_per_chunk, chunks;
    long off;   // arena offset: [chunks][Mpad*NC] then [chunks][NC] bias partials
    long floats;
};
// LayerNorm backward on big slabs (N * C / 4 >= 64 * 256 float4 columns, e.g. the 8192-node graph): every workgroup of
// ln_gate_bwd_kernel would rebuild the slab constants c1, c2 from N row partials (colgroups x slabs x N x 8 B: 4 GB at
// C5), so a tiny kernel forms them once per slab (ln_slab_consts_kernel) and the workgroups read two floats per slab.
constexpr int kLnBigColgroups = 64;
struct BwdGeom {
    int ln_spg, ln_sg;       // slabs per group / groups for the LayerNorm parameter partials
    int al_wgs;              // workgroups of align_gate_bwd (grid-stride over 64-row tiles)
    WgradGeom w1, w2;
    long off_ln_g, off_ln_b, off_gc, off_al, total;
    int gc_stride;           // floats per slab in the graph-conv partials: (terms + 1) * 256
    int gc_count;            // partial blocks of the graph conv: slabs (slab kernels) or workgroups of the tiled row pass
    int gc_tiles_per_wg;     // tiled row pass: 16-row tiles per workgroup
    int al_stride;           // floats per workgroup in the align partials: c0*c1 + c1 (+ 16*2*c0 + 2*c0 on the thin path)
    int thin;                // first layer handled by the thin kernels (Kt*c_in <= 4, c0 == 64, c1 == 16)
    int k1;                  // tmp_conv2 / LayerNorm backward fused into tc2_bwd_kernel (no dZ2, no w2 partials; ln_sg = B)
    int node_tiles;          // ceil(N / 16)
    int k1_wgs, k1_stride;   // workgroups (min(B * node_tiles, 2 per CU)) and floats per workgroup (Kt*16*NC2 + NC2) of its dW_eff2 | db_eff2 partials
    long off_k1;
    int k3;                  // tmp_conv1 / Align backward fused into tc1_bwd_kernel (needs need_dx): no dZ1, no w1 / align partials
    int k3_wb, k3_wgs, k3_stride;   // windows per workgroup, workgroups (node_tiles * ceil(B / wb)), floats per workgroup
    long off_k3;
};

inline WgradGeom wgrad_geom(long rows, int K, int NC, long off, int target_wgs = 256) {
    WgradGeom g;
    g.M = K;
    g.Mtiles = (K + 15) / 16;
    g.MTW = g.Mtiles <= 3 ? g.Mtiles : 4;   // m-tiles per workgroup (1, 2, 3 or 4)
    g.mchunks = (g.Mtiles + g.MTW - 1) / g.MTW;
    g.Mpad = g.mchunks * g.MTW * 16;
    g.NC = NC;
    // aim at ~target_wgs workgroups in total (256: one per CU; fewer partials to write and re-read), >= 64 rows per chunk
    long target = target_wgs / g.mchunks;
    if (target < 1) target = 1;
    long rpc = (rows + target - 1) / target;
    rpc = (rpc + 63) / 64 * 64;   // whole 64-row reduction steps
    if (rpc < 64) rpc = 64;
    g.rows_per_chunk = (int)rpc;
    g.chunks = (int)((rows + rpc - 1) / rpc);
    g.off = off;
    g.floats = (long)g.chunks * ((long)g.Mpad * NC + NC);
    return g;
}

// The thin first layer (K = Kt * c_in <= 4) as wave-per-tile kernels (round 5, stgcn_kernels_thin.hip.h); STGCN_THIN=0 selects the row-tile
// kernels of rounds 1 - 4 (A/B runs; the stage tests run both).  Read per call: the tests switch it.
inline bool thin_wave_tiles() {
    const char* e = getenv("STGCN_THIN");
    return !(e && e[0] == '0');
}
inline BwdGeom bwd_geom(int B, int T, int N, int c_in, int c0, int c1, int c2, int Kt, int terms, int need_dx) {
    BwdGeom g;
    const int T1 = T - Kt + 1, T2 = T1 - Kt + 1;
    const long rows1 = (long)B * T1 * N, rows2 = (long)B * T2 * N, slabs1 = (long)B * T1, slabs2 = (long)B * T2;
    const long n = (long)N * c2, n4 = n / 4;
    const int colgroups = (int)((n4 + kThreads - 1) / kThreads);
    int sg = (512 + colgroups - 1) / colgroups;   // ~512 workgroups
    if (colgroups >= kLnBigColgroups) sg = (4096 + colgroups - 1) / colgroups;   // big slabs (N*C >= 64 K): ~4096 workgroups, short slab walks
    if (sg > slabs2) sg = (int)slabs2;
    if (sg < 1) sg = 1;
    g.ln_spg = (int)((slabs2 + sg - 1) / sg);
    g.ln_sg = (int)((slabs2 + g.ln_spg - 1) / g.ln_spg);
    g.k1 = tc2_bwd_fused_ok(c1, c2, Kt, T1, T2) ? 1 : 0;
    g.node_tiles = (N + 15) / 16;
    {   // tc2_bwd_kernel: whole (window, node tile) items per workgroup, at most two workgroups per CU (its residency: 8 waves of ~100 VGPRs,
        // 57 KB of LDS) -- beyond that a workgroup walks several items and keeps ONE partial block (C2: 416 items -> unchanged)
        const long items = (long)B * g.node_tiles, cap = 2L * device_cus();
        g.k1_wgs = (int)(items < cap ? items : cap);
    }
    g.k1_stride = Kt * 16 * 2 * c2 + 2 * c2;
    if (g.k1) {   // one LayerNorm-parameter partial per window
        g.ln_spg = T2;
        g.ln_sg = B;
    }
    const long tiles1 = (rows1 + kTileRows - 1) / kTileRows;
    g.thin = (Kt * c_in <= 4 && c0 == 64 && c1 == 16) ? 1 : 0;   // thin_tc1_bwd_kernel keeps K <= 4 rows of W_eff in registers
    // grid-stride workgroups of align_gate_bwd (23.5 KB of LDS each: several per CU for latency hiding).  The thin
    // first-layer kernel carries a 13 KB partial per workgroup, so fewer, longer workgroups win there
    // (measured; 512 = 2 resident workgroups per CU at 62 KB of LDS -- a 513th would wait for a second round).
    // (round 5: the wave-per-tile thin kernel holds 3 workgroups per CU; STGCN_THIN_WGS overrides the cap for sweeps)
    static const int thin_cap_env = STGCN_EXP_ENV("STGCN_THIN_WGS") ? atoi(STGCN_EXP_ENV("STGCN_THIN_WGS")) : 0;
    // (pass r5-04: 256 / 384 / 512 / 768 workgroups are within noise at C2 and C3; at the 1.3 M rows of the 8192-node graph 768 -- three per
    //  CU, its residency -- take 74 us against 81)
    const int al_cap = g.thin ? (thin_cap_env > 0 ? thin_cap_env : (rows1 >= (1L << 18) && thin_wave_tiles() ? 768 : 512)) : 1024;   // (768 = three per CU: the wave-per-tile form's residency; the row-tile form of STGCN_THIN=0 holds two)
    g.al_wgs = (int)(tiles1 < al_cap ? tiles1 : al_cap);
    long o = 0;
    auto take = [&](long f) { long at = o; o += (f + 63) / 64 * 64; return at; };
    g.off_ln_g = take((long)g.ln_sg * n);
    g.off_ln_b = take((long)g.ln_sg * n);
    g.gc_stride = (terms + 1) * 256;
    g.gc_count = (int)slabs1;
    g.gc_tiles_per_wg = 0;
    if (gc_is_tiled(N, terms)) {   // ~1024 workgroups, whole groups of 4 tiles (one per wave)
        const long tiles16 = (rows1 + 15) / 16;
        long per = (tiles16 + 1023) / 1024;
        per = (per + 3) / 4 * 4;
        g.gc_tiles_per_wg = (int)per;
        g.gc_count = (int)((tiles16 + per - 1) / per);
    }
    g.off_gc = take((long)g.gc_count * g.gc_stride);
    g.al_stride = c0 * c1 + c1 + (g.thin ? 16 * 2 * c0 + 2 * c0 : 0);
    g.off_al = take((long)g.al_wgs * g.al_stride);
    g.w1 = wgrad_geom(rows1, Kt * c_in, 2 * c0, 0);
    g.w1.off = take(g.w1.floats);
    g.w2 = wgrad_geom(rows2, Kt * c1, 2 * c2, 0);
    g.w2.off = take(g.k1 ? 0 : g.w2.floats);
    g.off_k1 = take(g.k1 ? (long)g.k1_wgs * g.k1_stride : 0);
    g.k3 = (need_dx && !g.thin && tc1_bwd_shape_ok(c_in, c0, c1, Kt)) ? 1 : 0;
    g.k3_wb = 0;
    {   // one workgroup per CU walking an equal-weight range of the (window, node tile, output step) sequence
        const long items = (long)B * g.node_tiles;
#ifdef STGCN_EXPERIMENTS
        static const int per_cu = getenv("STGCN_TC1_BWD_PER_CU") ? atoi(getenv("STGCN_TC1_BWD_PER_CU")) : 1;   // (tuning: 2 = two interleaved chains per CU, twice the partials)
#else
        constexpr int per_cu = 1;
#endif
        long wgs = (long)device_cus() * (per_cu > 0 ? per_cu : 1);
        // (round 6) small batches: fewer items than compute units -- the ranges are then cut INSIDE items (at least kTc1MinSteps output steps
        // each) instead of leaving one workgroup to walk a whole item alone: the launch is one workgroup's chain, whatever the batch
        const long by_steps = items * (long)T / tc1_min_steps();
        const long most = items > by_steps ? items : by_steps;
        if (wgs > most) wgs = most;
        g.k3_wgs = (int)wgs;
    }
    g.k3_stride = tc1_bwd_part_floats(c0, c_in, Kt);
    g.off_k3 = take(g.k3 ? (long)g.k3_wgs * g.k3_stride : 0);
    g.total = o;
    return g;
}
inline int64_t bwd_partial_floats(int B, int T, int N, int c_in, int c0, int c1, int c2, int Kt, int terms, int need_dx) {
    return bwd_geom(B, T, N, c_in, c0, c1, c2, Kt, terms, need_dx).total;
}

// ================================================================================================
// B1a: per-ROW partial sums of g = dy_m * gamma and g * xhat (LayerNorm backward).
//      Fully parallel streaming kernel (one float4 per thread; the C/4 lanes of a row reduce by shuffles).
// ================================================================================================
struct LnBwdArgs {
    const float* dy;     // [slabs][n]
    const float* y;      // ln_bwd_rowstats_kernel only: [slabs][n] the LayerNorm's OUTPUT (after dropout), or null: form xhat from U, S below
    const float* beta;   // [n] (with y)
    const float* U;
    const float* S;
    const float* gamma;
    const float* mean;
    const float* rstd;
    float2* rowstat;     // [slabs*N]  (sum g, sum g*xhat) per row
    float2* slabconst;   // [slabs] (c1, c2) formed by ln_slab_consts_kernel, or null: ln_gate_bwd_kernel rebuilds them itself
    float* dZ;           // [slabs*N][2*C]
    float* dgam_part;    // [sg][n]
    float* dbet_part;
    int n, N, C, act, training, spg;
    long slabs;
    float keep_scale;
    uint32_t thresh;
    uint64_t seed, offset;
    const uint64_t* offset_dev;
};

template <typename ET>
__global__ __launch_bounds__(256) void ln_bwd_rowstats_kernel(LnBwdArgs a) {
    const int n4 = a.n >> 2, c4n = a.C >> 2;
    const long total = a.slabs * (long)n4;
    const long e = (long)blockIdx.x * kThreads + threadIdx.x;
    const bool valid = e < total;
    const long slab = valid ? e / n4 : 0;
    const int q = valid ? (int)(e - slab * n4) : 0;
    float s1 = 0.f, s2 = 0.f;
    if (valid) {
        const size_t base = (size_t)slab * a.n + 4 * (size_t)q;
        f32x4 dy = ldx4(et_ptr<ET>(a.dy) + base);
        const f32x4 ga = ld4(a.gamma + 4 * q);
        f32x4 k = {1.f, 1.f, 1.f, 1.f};
        if (a.training) {
            const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
            k = dropout_scale4((uint64_t)slab * n4 + q, a.seed, off, a.thresh, a.keep_scale);
        }
        if (a.y) {   // (uniform) from the LayerNorm output: sum g = sum mask dy gamma ; sum g xhat = sum_kept dy (y - keep_scale * beta)
            const f32x4 y = ldx4(et_ptr<ET>(a.y) + base), be = ld4(a.beta + 4 * q);
            const float ks = a.training ? a.keep_scale : 1.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s1 += dy[i] * k[i] * ga[i];
                if (k[i] > 0.f) s2 += dy[i] * (y[i] - ks * be[i]);
            }
        } else {
            const f32x4 u = ldx4(et_ptr<ET>(a.U) + base), s = ldx4(et_ptr<ET>(a.S) + base);
            const float mean = a.mean[slab], rstd = a.rstd[slab];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (gate_fwd(u[i], s[i], a.act) - mean) * rstd;
                const float gg = dy[i] * k[i] * ga[i];
                s1 += gg;
                s2 += gg * xh;
            }
        }
    }
    for (int m = c4n >> 1; m >= 1; m >>= 1) {   // the c4n lanes of one row are contiguous and aligned
        s1 += __shfl_xor(s1, m);
        s2 += __shfl_xor(s2, m);
    }
    if (valid && (q & (c4n - 1)) == 0) a.rowstat[slab * a.N + fast_div(q, c4n, pow2_shift(c4n))] = make_float2(s1, s2);
}

// c1 = mean(g), c2 = mean(g * xhat) of one slab from its row partials (thread-strided partial sums, block_sum2, like
// ln_gate_bwd_kernel's own rebuild); grid = slabs.  Launched only for big slabs (kLnBigColgroups).
__global__ __launch_bounds__(256) void ln_slab_consts_kernel(LnBwdArgs a) {
    extern __shared__ float stgcn_smem[];
    const long slab = blockIdx.x;
    float x = 0.f, y = 0.f;
    const float2* rs = a.rowstat + slab * a.N;
    int r0 = 0;
    if ((reinterpret_cast<uintptr_t>(rs) & 15) == 0) {   // two rows per 16-byte load, four loads in flight (as slab_stats_from_rows)
        const f32x4* rs4 = reinterpret_cast<const f32x4*>(rs);
        const int N2 = a.N >> 1;
        f32x4 t0 = zero4(), t1 = zero4();
        int q = threadIdx.x;
        for (; q + 3 * kThreads < N2; q += 4 * kThreads) {
            const f32x4 v0 = rs4[q], v1 = rs4[q + kThreads], v2 = rs4[q + 2 * kThreads], v3 = rs4[q + 3 * kThreads];
            t0 += v0 + v1;
            t1 += v2 + v3;
        }
        for (; q < N2; q += kThreads) t0 += rs4[q];
        t0 += t1;
        x = t0[0] + t0[2];
        y = t0[1] + t0[3];
        r0 = N2 * 2;
    }
    for (int r = r0 + threadIdx.x; r < a.N; r += kThreads) {
        const float2 v = rs[r];
        x += v.x;
        y += v.y;
    }
    block_sum2(x, y, stgcn_smem);
    if (threadIdx.x == 0) a.slabconst[slab] = make_float2(x / (float)a.n, y / (float)a.n);
}

// ================================================================================================
// B1b: dH = rstd * (g - c1 - xhat * c2), gate backward -> dZ = [dU | dQ]; partial dgamma / dbeta.
// c1 = mean(g), c2 = mean(g * xhat) of each slab are rebuilt from the row partials by the workgroup.
// A thread owns one float4 column of the [N*C] slab and walks `spg` consecutive slabs.
// grid = (ceil(n/4 / 256), sg); dynamic LDS: 8 + 2*spg floats
// ================================================================================================
template <typename ET>
__global__ __launch_bounds__(256) void ln_gate_bwd_kernel(LnBwdArgs a) {
    const ET* const dy_ = et_ptr<ET>(a.dy);
    const ET* const U_ = et_ptr<ET>(a.U);
    const ET* const S_ = et_ptr<ET>(a.S);
    ET* const dZ_ = et_ptr<ET>(a.dZ);
    extern __shared__ float stgcn_smem[];
    float* cs = stgcn_smem + 8;
    const int n4 = a.n >> 2;
    const int q = (int)blockIdx.x * kThreads + (int)threadIdx.x;
    const bool valid = q < n4;
    const int sg = blockIdx.y;
    long s0 = (long)sg * a.spg, s1 = s0 + a.spg;
    if (s1 > a.slabs) s1 = a.slabs;
    if (a.slabconst) {
        for (long i = threadIdx.x; i < s1 - s0; i += kThreads) {
            const float2 c = a.slabconst[s0 + i];
            cs[2 * i] = c.x;
            cs[2 * i + 1] = c.y;
        }
    } else {
        for (long slab = s0; slab < s1; ++slab) {
            float x = 0.f, y = 0.f;
            const float2* rs = a.rowstat + slab * a.N;
            for (int r = threadIdx.x; r < a.N; r += kThreads) {
                const float2 v = rs[r];
                x += v.x;
                y += v.y;
            }
            block_sum2(x, y, stgcn_smem);
            if (threadIdx.x == 0) {
                cs[2 * (slab - s0)] = x / (float)a.n;
                cs[2 * (slab - s0) + 1] = y / (float)a.n;
            }
        }
    }
    __syncthreads();
    if (!valid) return;
    const int c4n = a.C >> 2;
    const int node = fast_div(q, c4n, pow2_shift(c4n)), c4 = q - node * c4n;
    const f32x4 ga = ld4(a.gamma + 4 * q);
    const uint64_t off = a.offset + (a.offset_dev ? *a.offset_dev : 0);
    f32x4 dg = zero4(), db = zero4();
    const int N = a.N;
    // one-slab software prefetch: the loads of slab+1 are issued before the stores of slab
    f32x4 dy_n = zero4(), u_n = zero4(), s_n = zero4();
    if (s0 < s1) {
        const size_t base = (size_t)s0 * a.n + 4 * (size_t)q;
        dy_n = ldx4(dy_ + base); u_n = ldx4(U_ + base); s_n = ldx4(S_ + base);
    }
    for (long slab = s0; slab < s1; ++slab) {
        f32x4 dy = dy_n;
        const f32x4 u = u_n, s = s_n;
        if (slab + 1 < s1) {
            const size_t nb = (size_t)(slab + 1) * a.n + 4 * (size_t)q;
            dy_n = ldx4(dy_ + nb); u_n = ldx4(U_ + nb); s_n = ldx4(S_ + nb);
        }
        const float mean = a.mean[slab], rstd = a.rstd[slab], c1 = cs[2 * (slab - s0)], c2 = cs[2 * (slab - s0) + 1];
        if (a.training) {
            const f32x4 k = dropout_scale4((uint64_t)slab * n4 + q, a.seed, off, a.thresh, a.keep_scale);
#pragma unroll
            for (int i = 0; i < 4; ++i) dy[i] *= k[i];
        }
        f32x4 du, dq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (gate_fwd(u[i], s[i], a.act) - mean) * rstd;
            const float gg = dy[i] * ga[i];
            const float dh = rstd * (gg - c1 - xh * c2);
            dg[i] += dy[i] * xh;
            db[i] += dy[i];
            float du_, dq_;
            gate_bwd(dh, u[i], s[i], a.act, du_, dq_);
            du[i] = du_;
            dq[i] = dq_;
        }
        ET* z = dZ_ + ((size_t)slab * N + node) * (2 * a.C) + 4 * c4;
        stx4_wt(z, du);
        stx4_wt(z + a.C, dq);
    }
    st4_wt(a.dgam_part + (size_t)sg * a.n + 4 * (size_t)q, dg);
    st4_wt(a.dbet_part + (size_t)sg * a.n + 4 * (size_t)q, db);
}

// ================================================================================================
// B2/B5: transposed temporal convolution  dX[b,t,n,i] = sum_k sum_o dZ[b,t-k,n,o] W_eff[k*Cin+i][o]
// as a row-tile GEMM over the implicit matrix [rows = (b,t,n)] x [K = Kt*NC], weights packed by
// PK_TCONV_BWD.  Optional relu mask (dX *= (G > 0)) for the gradient entering the graph conv.
// LAYOUT 0: >= 4 n-tiles  (wave w: all 4 m-tiles, n-tiles w + 4j)
// LAYOUT 1: 1 n-tile      (wave w: m-tile w)
// LAYOUT 2: 2 n-tiles     (wave w: m-tiles 2*(w>>1) + {0,1}, n-tile w & 1)
// ================================================================================================
struct TconvBwdDataArgs {
    TapSrc ts;            // dZ viewed through Kt taps, dir = -1
    const float* Wp;      // packed, K = Kt*NC (KCH chunks), cols = roundup16(Cin)
    int KCH, Cin;
    const float* Gmask;   // [rows][Cin] or null
    float* dX;            // [rows][Cin]
};

// WAVES = 8 (LAYOUT 0 only): the second half of the workgroup owns m-tiles 2..3 of every n-tile column, i.e. WM = 2.
// Register budgets that keep the grids of the C2 shapes resident in one round: the narrow-output layout (LAYOUT 1) is held
// to 5 waves per SIMD (<= 96 registers; at 4 its 1035-tile launch would leave 11 workgroups to a second round), the 4-wave
// variants to 4 (1024 workgroup slots).
template <int WM, int NT, int LAYOUT, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, LAYOUT == 1 ? 5 : (WAVES == 4 ? 4 : 1)) void tconv_bwd_data_kernel(TconvBwdDataArgs a) {
    constexpr int THREADS = WAVES * 64;
    extern __shared__ float stgcn_smem[];
    int* rowbase = reinterpret_cast<int*>(stgcn_smem);
    int* rowt = rowbase + 64;
    float* At = stgcn_smem + kTileHdr;
    const int wv = threadIdx.x >> 6, wave = wv & 3, lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
    const long row0 = (long)xcd_item(blockIdx.x, gridDim.x) * kTileRows;
    const int mt0 = LAYOUT == 0 ? (wv >> 2) * WM : (LAYOUT == 1 ? wave : 2 * (wave >> 1));
    const int nt0 = LAYOUT == 0 ? wave : (LAYOUT == 1 ? 0 : (wave & 1));

    STGCN_PHASE(2, 0);
    stagger_start();
    tile_rowinfo(a.ts, row0, rowbase, rowt);
    __syncthreads();
    STGCN_PHASE(2, 1);
    // taps whose source time step t - tap is out of range for EVERY row of this tile contribute exact zeros:
    // skip their K segments (the head has T1 = 1: 3 of its 4 taps are empty for any given output step)
    if (wv == 0) {
        unsigned m = 0;
        const int t = rowt[lane];   // 64 rows == 64 lanes
        for (int tap = 0; tap < a.ts.taps && tap < 32; ++tap) {
            const int tt = t - tap;
            if (tt >= 0 && tt < a.ts.Tsrc) m |= 1u << tap;
        }
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) m |= __shfl_xor(m, x);
        if (lane == 0) rowbase[128] = (int)m;   // scratch word 0 of the tile header
    }
    f32x4 acc[WM][NT], acc2[WM][NT];   // two accumulator sets (even / odd chunks) for MFMA ILP
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[i][j] = zero4();
            acc2[i][j] = zero4();
        }
    __syncthreads();
    const unsigned tapmask = (unsigned)rowbase[128];
    const int KP = a.KCH * 16;
#if STGCN_PIPE_BWD
    {
        auto next_seg = [&](int from) {   // next segment with an in-range tap for some row of the tile; KP if none
            for (int k0 = from; k0 < KP; k0 += kSegMax) {
                const int ks = (KP - k0) < kSegMax ? (KP - k0) : kSegMax;
                for (int tap = k0 / a.ts.C; tap <= (k0 + ks - 1) / a.ts.C; ++tap)
                    if (tap >= 32 || ((tapmask >> tap) & 1u)) return k0;
            }
            return KP;
        };
        TileRegs<kTileRows, THREADS> regs;
        int k0 = next_seg(0);
        if (k0 < KP) {
            const int ks0 = (KP - k0) < kSegMax ? (KP - k0) : kSegMax;
            tile_prefetch_segment<kTileRows, THREADS>(a.ts, rowbase, rowt, k0, ks0, regs);
            tile_commit_segment<kTileRows, THREADS>(ks0, regs, At, ks0 + 4);
        }
        while (k0 < KP) {   // uniform over the workgroup
            const int kseg = (KP - k0) < kSegMax ? (KP - k0) : kSegMax;
            const int kn = next_seg(k0 + kSegMax), ksegn = (KP - kn) < kSegMax ? (KP - kn) : kSegMax;
            SegWeights<NT> w;
            seg_load_weights<NT>(w, kseg >> 4, a.Wp, k0 >> 4, a.KCH, nt0, 4);
            if (kn < KP) tile_prefetch_segment<kTileRows, THREADS>(a.ts, rowbase, rowt, kn, ksegn, regs);
            __syncthreads();
            seg_mma_w<WM, NT>(acc, At, kseg + 4, mt0, kseg >> 4, w);
            if (kn < KP) {
                __syncthreads();
                tile_commit_segment<kTileRows, THREADS>(ksegn, regs, At, ksegn + 4);
            }
            k0 = kn;
        }
    }
#else
    bool first = true;
    for (int k0 = 0; k0 < KP; k0 += kSegMax) {
        const int kseg = (KP - k0) < kSegMax ? (KP - k0) : kSegMax;
        const int tap_lo = k0 / a.ts.C, tap_hi = (k0 + kseg - 1) / a.ts.C;
        bool any = false;
        for (int tap = tap_lo; tap <= tap_hi; ++tap) any = any || tap >= 32 || ((tapmask >> tap) & 1u);
        if (!any) continue;   // uniform over the workgroup
        if (!first) __syncthreads();
        first = false;
        tile_load_segment<kTileRows, THREADS>(a.ts, rowbase, rowt, k0, kseg, At, kseg + 4);
        __syncthreads();
        STGCN_PHASE(2, 2 + 2 * ((k0 / kSegMax) % 6));
        const int half = (kseg >> 4) >> 1, rest = (kseg >> 4) - half;
        seg_mma<WM, NT>(acc, At, kseg + 4, mt0, rest, a.Wp, k0 >> 4, a.KCH, nt0, 4);
        if (half > 0) seg_mma<WM, NT>(acc2, At + rest * 16, kseg + 4, mt0, half, a.Wp, (k0 >> 4) + rest, a.KCH, nt0, 4);
        STGCN_PHASE(2, 3 + 2 * ((k0 / kSegMax) % 6));
    }
#endif
    STGCN_PHASE(2, 14);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = (nt0 + 4 * j) * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long R = row0 + (mt0 + i) * 16 + 4 * g + r;
                if (R < a.ts.rows && col < a.Cin) {
                    float v = acc[i][j][r] + acc2[i][j][r];
                    const size_t o = (size_t)R * a.Cin + col;
                    if (a.Gmask && !(a.Gmask[o] > 0.f)) v = 0.f;
                    a.dX[o] = v;
                }
            }
        }
    STGCN_PHASE(2, 15);
}

// ================================================================================================
// B3: graph-conv backward on one (b, t) slab, dY = masked upstream gradient:
//     dW_k = X_k^T dY, db = 1^T dY                       (partials per slab)
//     G_k = dY W_k^T ;  dA = G_0 + sum_{k>=1} T_k(L)^T G_k + dY      (the + dY is the residual of layers.py:229)
// Same fragment scheme as gconv_fwd_kernel with the transposed polynomials LTp (independent terms, no recursion).
// ================================================================================================
struct GconvBwdArgs {
    const float* dY;     // [slabs][N][16]
    const float* X0;     // [slabs][N][16]   (A)
    const float* Xk;     // [terms-1][slabs][N][16]
    const float* LTp;    // fragment-packed T_1^T .. T_{terms-1}^T (stgcn_gso_prepare), NP*NP floats each
    const float* W;
    float* dA;           // [slabs][N][16]
    float* part;         // [slabs][(terms+1)*256]
    int N, NP, Ks, kipf;
    int parts;           // workgroups per slab (see GconvFwdArgs); every part stages the slab and forms all G_k
    long slabs;
    // tiled path only (launch_gconv_bwd_tiled; the slab kernel ignores them)
    float* Gk;           // [terms][slabs][N][16] Clenshaw buffers g_k / b_k
    float* XT;           // two bf16 operand-form buffers (plan: ws_XT), used when g_gc_precision > 0
    int tiles_per_wg;    // BwdGeom::gc_tiles_per_wg
    int wgs;             // BwdGeom::gc_count
};

// Operator products of the slab-resident backward for one pair of terms: acc1[q] += T_k0^T x G_k0, acc2[q] += T_{k0+1}^T x G_{k0+1} on
// the NQ node tiles of this wave (fragment offsets fo[q]), K = the KCH node chunks.  Same loop discipline as gconv_fwd_kernel: a ring of RG
// fragment chunks in registers with static indices, no branch in the body (TWO, NQ compile-time), the refill of a slot right behind the
// MFMAs that read it -- the p <- n <- load rotation this replaces waited for every load it had just issued.
template <typename MM, int MAXQ, int NQ, bool TWO, int RG>
__device__ __forceinline__ void gc_bwd_products(const float* T1, const float* T2, const float* G1, const float* G2, int LDX, int KCH, int wave, int WAVES,
                                                int lane, f32x4 (&acc1)[MAXQ], f32x4 (&acc2)[MAXQ]) {
    const int g = lane >> 4, l15 = lane & 15;
    f32x4 r1[RG][NQ], r2[RG][NQ];
    int fo[NQ];   // (floats; < 2^31: at most 7 terms of 512 x 512)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        fo[q] = ((wave + WAVES * q) * KCH * 64 + lane) * 4;
#pragma unroll
        for (int d = 0; d < RG; ++d) {
            const int dc = d < KCH ? d : KCH - 1;
            r1[d][q] = ld4(T1 + fo[q] + 256 * dc);
            if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * dc);
        }
    }
    auto chunk = [&](int kc, int d, auto load_tag) __attribute__((always_inline)) {
        const typename MM::frag af1 = MM::cvt(ld4(G1 + l15 * LDX + kc * 16 + 4 * g));
        const typename MM::frag af2 = MM::cvt(TWO ? ld4(G2 + l15 * LDX + kc * 16 + 4 * g) : zero4());
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (TWO) MM::mma_2x(af1, MM::cvt(r1[d][q]), acc1[q], af2, MM::cvt(r2[d][q]), acc2[q]);
            else acc1[q] = MM::mma(af1, MM::cvt(r1[d][q]), acc1[q]);
            if (decltype(load_tag)::value) {
                r1[d][q] = ld4(T1 + fo[q] + 256 * (kc + RG));
                if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * (kc + RG));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int kc0 = 0;
    for (; kc0 + 2 * RG <= KCH; kc0 += RG) {
#pragma unroll
        for (int d = 0; d < RG; ++d) chunk(kc0 + d, d, std::true_type());
    }
    for (; kc0 < KCH; kc0 += RG) {
#pragma unroll
        for (int d = 0; d < RG; ++d) {
            if (kc0 + d < KCH) {
                if (kc0 + d + RG < KCH) chunk(kc0 + d, d, std::true_type());
                else chunk(kc0 + d, d, std::false_type());
            }
        }
    }
}
// The same products for bf16 activations from the operator's bf16 fragment PLANE on v_mfma_f32_16x16x32_bf16 (see gconv_fwd_body B16P): G1 / G2 are
// bf16 [16][LDB] tiles in LDS, T1 / T2 the hi planes of the transposed polynomials, K = KC32 chunks of 32 nodes.
template <int MAXQ, int NQ, bool TWO, int RG>
__device__ __forceinline__ void gc_bwd_products_b16p(const float* T1, const float* T2, const unsigned short* G1, const unsigned short* G2, int LDB, int KC32,
                                                     int wave, int WAVES, int lane, f32x4 (&acc1)[MAXQ], f32x4 (&acc2)[MAXQ]) {
    const int g = lane >> 4, l15 = lane & 15;
    f32x4 r1[RG][NQ], r2[RG][NQ];
    int fo[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        fo[q] = ((wave + WAVES * q) * KC32 * 64 + lane) * 4;
#pragma unroll
        for (int d = 0; d < RG; ++d) {
            const int dc = d < KC32 ? d : KC32 - 1;
            r1[d][q] = ld4(T1 + fo[q] + 256 * dc);
            if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * dc);
        }
    }
    auto chunk = [&](int kc, int d, auto load_tag) __attribute__((always_inline)) {
        const bf16x8 af1 = __builtin_bit_cast(bf16x8, ld4(reinterpret_cast<const float*>(G1 + l15 * LDB + kc * 32 + 8 * g)));
        const bf16x8 af2 = __builtin_bit_cast(bf16x8, TWO ? ld4(reinterpret_cast<const float*>(G2 + l15 * LDB + kc * 32 + 8 * g)) : zero4());
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc1[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, __builtin_bit_cast(bf16x8, r1[d][q]), acc1[q], 0, 0, 0);
            if (TWO) acc2[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2, __builtin_bit_cast(bf16x8, r2[d][q]), acc2[q], 0, 0, 0);
            if (decltype(load_tag)::value) {
                r1[d][q] = ld4(T1 + fo[q] + 256 * (kc + RG));
                if (TWO) r2[d][q] = ld4(T2 + fo[q] + 256 * (kc + RG));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    int kc0 = 0;
    for (; kc0 + 2 * RG <= KC32; kc0 += RG) {
#pragma unroll
        for (int d = 0; d < RG; ++d) chunk(kc0 + d, d, std::true_type());
    }
    for (; kc0 < KC32; kc0 += RG) {
#pragma unroll
        for (int d = 0; d < RG; ++d) {
            if (kc0 + d < KC32) {
                if (kc0 + d + RG < KC32) chunk(kc0 + d, d, std::true_type());
                else chunk(kc0 + d, d, std::false_type());
            }
        }
    }
}
template <int MAXQ, int RG>
__device__ __forceinline__ void gc_bwd_products_b16p_nq(int nq, bool two, const float* T1, const float* T2, const unsigned short* G1, const unsigned short* G2,
                                                        int LDB, int KC32, int wave, int WAVES, int lane, f32x4 (&acc1)[MAXQ], f32x4 (&acc2)[MAXQ]) {
#define STGCN_GCB(NQV) \
    if (nq == NQV) { \
        if (two) gc_bwd_products_b16p<MAXQ, NQV, true, RG>(T1, T2, G1, G2, LDB, KC32, wave, WAVES, lane, acc1, acc2); \
        else gc_bwd_products_b16p<MAXQ, NQV, false, RG>(T1, T2, G1, G2, LDB, KC32, wave, WAVES, lane, acc1, acc2); \
    }
    STGCN_GCB(1)
    if constexpr (MAXQ >= 2) { STGCN_GCB(2) }
    if constexpr (MAXQ >= 3) { STGCN_GCB(3) }
#undef STGCN_GCB
}
template <typename MM, int MAXQ, int RG>
__device__ __forceinline__ void gc_bwd_products_nq(int nq, bool two, const float* T1, const float* T2, const float* G1, const float* G2, int LDX, int KCH,
                                                   int wave, int WAVES, int lane, f32x4 (&acc1)[MAXQ], f32x4 (&acc2)[MAXQ]) {
#define STGCN_GCB(NQV) \
    if (nq == NQV) { \
        if (two) gc_bwd_products<MM, MAXQ, NQV, true, RG>(T1, T2, G1, G2, LDX, KCH, wave, WAVES, lane, acc1, acc2); \
        else gc_bwd_products<MM, MAXQ, NQV, false, RG>(T1, T2, G1, G2, LDX, KCH, wave, WAVES, lane, acc1, acc2); \
    }
    STGCN_GCB(1)
    if constexpr (MAXQ >= 2) { STGCN_GCB(2) }
    if constexpr (MAXQ >= 3) { STGCN_GCB(3) }
    if constexpr (MAXQ >= 4) { STGCN_GCB(4) }
#undef STGCN_GCB
}

template <int MAXQ, int MAXW, typename ET>   // wave count = blockDim.x / 64 <= MAXW (see gconv_fwd_kernel)
__global__ __launch_bounds__(MAXW * 64) void gconv_bwd_kernel(GconvBwdArgs a) {
    typedef Mma<ET> MM;
    extern __shared__ float stgcn_smem[];
    const int THREADS = blockDim.x, NW = THREADS >> 6, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int P = a.parts, prt = (int)(blockIdx.x % (unsigned)P);
    const long slab = blockIdx.x / (unsigned)P;
    const int wave = prt + P * __builtin_amdgcn_readfirstlane(w), WAVES = P * NW;   // owned node tiles: wave + WAVES * q (scalar: uniform branches)
    const int N = a.N, NP = a.NP, LDX = NP + 4, HT = NP >> 4, KCH = NP >> 4, LDY = 20, Ks = a.Ks;
    float* const GT0 = stgcn_smem;                 // GT(k) = GT0 + k*16*LDX, transposed [c][node]
    float* const dYs = stgcn_smem + Ks * 16 * LDX; // [NP][LDY] row major

    // ---- stage dY (row major) and all X_k (transposed) -----------------------------------------
    const ET* dYsl = et_ptr<ET>(a.dY) + (size_t)slab * N * 16;
    // (the 1 + Ks requests of two trips in flight per thread, raw, addresses clamped: load -> LDS store pairs waited for every load in turn)
    for (int idx0 = tid; idx0 < NP * 4; idx0 += 2 * THREADS) {
        constexpr int KM = 8;   // terms (check_desc: Ks <= 8)
        Raw4<ET> ry[2], rx[2][KM];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = idx0 + u * THREADS, n = idx >> 2, c4 = idx & 3;
            const size_t o = (size_t)(n < N ? n : N - 1) * 16 + c4 * 4;
            ry[u] = ldraw4(dYsl + o);
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (k < Ks) {   // (uniform)
                    const ET* Xsl = (k == 0 ? et_ptr<ET>(a.X0) : et_ptr<ET>(a.Xk) + (size_t)(k - 1) * a.slabs * N * 16) + (size_t)slab * N * 16;
                    rx[u][k] = ldraw4(Xsl + o);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = idx0 + u * THREADS, n = idx >> 2, c4 = idx & 3;
            if (idx < NP * 4) {
                st4(dYs + n * LDY + c4 * 4, n < N ? cvt4(ry[u]) : zero4());
#pragma unroll
                for (int k = 0; k < KM; ++k) {
                    if (k < Ks) {
                        const f32x4 v = n < N ? cvt4(rx[u][k]) : zero4();
#pragma unroll
                        for (int i = 0; i < 4; ++i) GT0[(k * 16 + c4 * 4 + i) * LDX + n] = v[i];
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- parameter-gradient partials: job kk < Ks -> dW_kk = X_kk^T dY ; kk == Ks -> db via A = 1 ----
    float* part = a.part + (size_t)slab * (Ks + 1) * 256;
    for (int kk = wave; kk <= Ks; kk += WAVES) {
        f32x4 c0 = zero4(), c1 = zero4();
        for (int kc = 0; kc < KCH; ++kc) {
            f32x4 af;
            if (kk < Ks) af = ld4(GT0 + (kk * 16 + l15) * LDX + kc * 16 + 4 * g);
            else { af[0] = 1.f; af[1] = 1.f; af[2] = 1.f; af[3] = 1.f; }
            const float* yb = dYs + (kc * 16 + 4 * g) * LDY + l15;
            MM::mma_split(MM::cvt(af), MM::cvt(gather4(yb, LDY)), c0, c1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[kk * 256 + (4 * g + r) * 16 + l15] = c0[r] + c1[r];
    }
    __syncthreads();   // X_k no longer needed: GT buffers become G_k

    // ---- G_k = dY W_k^T on ALL node tiles (they are the K dimension of the products below; with parts > 1 every part of
    //      the slab forms them: 4 MFMAs per tile and term) ---------------------------------------------------------
    for (int k = 0; k < Ks; ++k) {
        f32x4 wf = zero4();   // B[kk = j][col = i] = W_k[i = l15][j = 4g + s]
        if (!(a.kipf && k == 0)) wf = ld4(a.W + (a.kipf ? 0 : (size_t)k * 256) + l15 * 16 + 4 * g);
        for (int ht = w; ht < HT; ht += NW) {
            const f32x4 af = ld4(dYs + (ht * 16 + l15) * LDY + 4 * g);   // A[h = l15][j = 4g + s]
            const f32x4 d = MM::mma(MM::cvt(af), MM::cvt(wf), zero4());
            st4(GT0 + (k * 16 + l15) * LDX + ht * 16 + 4 * g, d);      // D[h = 4g + r][i = l15]
        }
    }

    __syncthreads();   // every G_k complete

    // ---- dA = G_0 + sum_{k >= 1} T_k^T G_k + dY on the owned node tiles: the terms are independent products with the
    //      precomputed polynomials (two terms per pass over the k chunks, separate accumulators) ---------------------
    const size_t MSZ = (size_t)NP * NP;
    f32x4 acc1[MAXQ], acc2[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        acc1[q] = zero4();
        acc2[q] = zero4();
    }
    int nq = 0;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) nq += wave + WAVES * q < HT ? 1 : 0;
    for (int k0 = 1; k0 < Ks; k0 += 2) {
        const float* T1 = a.LTp + (size_t)(k0 - 1) * MSZ;
        const float* G1 = GT0 + k0 * 16 * LDX;
        gc_bwd_products_nq<MM, MAXQ, (MAXQ == 2 ? 4 : gc_ring(MAXQ))>(nq, k0 + 1 < Ks, T1, T1 + MSZ, G1, G1 + 16 * LDX, LDX, KCH, wave, WAVES, lane, acc1, acc2);
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int ht = wave + WAVES * q;
        const int h = ht * 16 + l15;
        if (ht < HT && h < N) {
            const f32x4 y = ld4(dYs + h * LDY + 4 * g);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = GT0[(4 * g + r) * LDX + h] + (acc1[q][r] + acc2[q][r]) + y[r];
            stx4_wt(et_ptr<ET>(a.dA) + ((size_t)slab * N + h) * 16 + 4 * g, o);
        }
    }
}

// ================================================================================================
// B3 (round 3): the same backward with the slab split over `parts` workgroups and the parameter-gradient jobs on their own waves.
// The one-workgroup-per-slab kernel above walks four serial phases (stage dY and Ks terms of X, dW jobs on 4 of its 13 waves,
// G_k on all tiles, the operator products) on 192 - 320 workgroups: at the C2 size its duration does not depend on the batch size
// (19 us from bs 4 to bs 32 for the second block, profiles/r2-40_batch_sweep.log).  Here
//   * waves 0 .. nwa-1 of part p own the node tiles p + parts * (w + nwa * q): they stage dY (all rows: G_k is the K dimension of the
//     operator products), form G_k = dY W_k^T for k >= 1 on all tiles into LDS ([c][node], the A operand) and G_0^T for their own tiles
//     in registers (W_0 as the A operand, dY rows as B: the D layout of the operator products), then run the products and write dA;
//   * the remaining waves ("job waves") form the slab's parameter gradients dW_k = X_k^T dY (k = part + parts * j) and db straight from
//     global memory -- no LDS, no staging of X_k -- concurrently with the operator products of the other waves.
// LDS holds dY and Ks - 1 terms only (44 KB at 207 nodes, Ks = 3: three workgroups per CU instead of two).
// grid = slabs * parts, block = (nwa + njw) * 64 with njw = ceil((Ks + 1) / parts).
// ================================================================================================
// B16P (bf16 activations): G_k tiles as bf16 [c][node] planes in LDS (half the LDS of that part) and the operator products from the
// transposed polynomials' bf16 fragment planes on 32-deep MFMAs (gc_bwd_products_b16p).
inline size_t gconv_bwd2_lds_bytes(int NP, int N, int Ks, bool b16p) {
    const size_t gk = b16p ? (size_t)(Ks - 1) * 16 * (gc_np32(N) + 8) * 2 : (size_t)(Ks - 1) * 16 * (NP + 4) * 4;
    return gk + ((size_t)NP * 20 + (size_t)(Ks + 1) * 16 * 20) * sizeof(float);   // + dY rows + the job waves' transposition tiles
}
template <int MAXQ, typename ET, bool B16P = false>
__global__ __launch_bounds__(768) void gconv_bwd2_kernel(GconvBwdArgs a, int nwa) {
    static_assert(!B16P || sizeof(ET) == 2, "the bf16-plane products are the bf16-activation form");
    typedef Mma<ET> MM;
    extern __shared__ float stgcn_smem[];
    const int THREADS = blockDim.x, tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int P = a.parts, prt = (int)(blockIdx.x % (unsigned)P);
    const long slab = blockIdx.x / (unsigned)P;
    const int N = a.N, NP = a.NP, LDX = NP + 4, HT = NP >> 4, KCH = NP >> 4, LDY = 20, Ks = a.Ks;
    const int NP32 = gc_np32(N), KC32 = NP32 >> 5, LDB = NP32 + 8;                 // (B16P)
    float* const GTk = stgcn_smem;                         // GT(k) = GTk + (k - 1)*16*LDX for k >= 1, transposed [c][node]
    unsigned short* const GHk = reinterpret_cast<unsigned short*>(stgcn_smem);   // (B16P) the same tiles as bf16 [c][LDB]
    float* const dYs = stgcn_smem + (B16P ? (Ks - 1) * 16 * LDB / 2 : (Ks - 1) * 16 * LDX);   // [NP][LDY] row major
    const ET* const dYsl = et_ptr<ET>(a.dY) + (size_t)slab * N * 16;

    // ---- every wave: stage dY (row major) ------------------------------------------------------------------------------------
    for (int idx = tid; idx < NP * 4; idx += THREADS) {
        const int n = idx >> 2, c4 = idx & 3;
        st4(dYs + n * LDY + c4 * 4, n < N ? ldx4(dYsl + (size_t)n * 16 + c4 * 4) : zero4());
    }
    __syncthreads();   // (1)

    if (w >= nwa) {
        // =========================================== job waves: parameter-gradient partials ======================================
        __syncthreads();   // (2) (nothing to wait for: keeps the barrier count of the workgroup)
        float* part = a.part + (size_t)slab * (Ks + 1) * 256;
        for (int kk = prt + P * (w - nwa); kk <= Ks; kk += P * ((THREADS >> 6) - nwa)) {
            // job kk < Ks: dW_kk = X_kk^T dY ; kk == Ks: db via A = 1.  A[m = c][k = node]: a lane needs ONE channel of 4 consecutive nodes --
            // 4 scalar loads 16 elements apart, and requested one chunk ahead with ldx1 (which converts at the load) that was no request
            // ahead at all: every chunk waited for 4 dependent round trips (C3: the job waves made this kernel 99 us against 55 for the
            // one-workgroup-per-slab form, r3-48).  Now a chunk = the 16 x 16 tile as ONE 4-element load per lane (node lane >> 2, channels
            // 4 (lane & 3) ..), raw, in a ring of 4 chunks with static indices; it is transposed through a wave-private LDS tile when its
            // turn comes.
            f32x4 c0 = zero4(), c1 = zero4();
            if (kk < Ks) {
                const ET* const Xsl = (kk == 0 ? et_ptr<ET>(a.X0) : et_ptr<ET>(a.Xk) + (size_t)(kk - 1) * a.slabs * N * 16) + (size_t)slab * N * 16;
                float* const TJ = dYs + NP * LDY + (w - nwa) * (16 * 20);   // [16 nodes][20]
                constexpr int RG = 4;
                const int jn = lane >> 2, jc = (lane & 3) * 4;
                auto req = [&](int kc) __attribute__((always_inline)) {
                    const int n = kc * 16 + jn;
                    return ldraw4(Xsl + (size_t)(n < N ? n : N - 1) * 16 + jc);
                };
                Raw4<ET> xr[RG];
#pragma unroll
                for (int d = 0; d < RG; ++d) xr[d] = req(d < KCH ? d : KCH - 1);
                auto chunk = [&](int kc, int d, auto load_tag) __attribute__((always_inline)) {
                    st4(TJ + jn * 20 + jc, kc * 16 + jn < N ? cvt4(xr[d]) : zero4());
                    if (decltype(load_tag)::value) xr[d] = req(kc + RG);
                    wave_lds_sync();
                    const f32x4 af = gather4(TJ + (4 * g) * 20 + l15, 20);   // X[node 4g + s][c = l15]
                    MM::mma_split(MM::cvt(af), MM::cvt(gather4(dYs + (kc * 16 + 4 * g) * LDY + l15, LDY)), c0, c1);
                    wave_lds_sync();   // (the next chunk overwrites the tile)
                };
                int kc0 = 0;
                for (; kc0 + 2 * RG <= KCH; kc0 += RG) {
#pragma unroll
                    for (int d = 0; d < RG; ++d) chunk(kc0 + d, d, std::true_type());
                }
                for (; kc0 < KCH; kc0 += RG) {
#pragma unroll
                    for (int d = 0; d < RG; ++d) {
                        if (kc0 + d < KCH) {
                            if (kc0 + d + RG < KCH) chunk(kc0 + d, d, std::true_type());
                            else chunk(kc0 + d, d, std::false_type());
                        }
                    }
                }
            } else {
                const f32x4 one = {1.f, 1.f, 1.f, 1.f};
                for (int kc = 0; kc < KCH; ++kc) MM::mma_split(MM::cvt(one), MM::cvt(gather4(dYs + (kc * 16 + 4 * g) * LDY + l15, LDY)), c0, c1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) part[kk * 256 + (4 * g + r) * 16 + l15] = c0[r] + c1[r];
        }
        return;
    }

    // =============================================== tile waves ==============================================================
    const int wave = prt + P * __builtin_amdgcn_readfirstlane(w), WAVES = P * nwa;   // owned node tiles: wave + WAVES * q (scalar: uniform branches)
    // G_k = dY W_k^T, k >= 1, on ALL node tiles (the K dimension of the products below)
    for (int k = 1; k < Ks; ++k) {
        const typename MM::frag wf = MM::cvt(ld4(a.W + (a.kipf ? 0 : (size_t)k * 256) + l15 * 16 + 4 * g));   // B[kk = j][col = i] = W_k[i = l15][j = 4g + s]
        for (int ht = w; ht < HT; ht += nwa) {
            const f32x4 af = ld4(dYs + (ht * 16 + l15) * LDY + 4 * g);   // A[h = l15][j = 4g + s]
            const f32x4 gk = MM::mma(MM::cvt(af), wf, zero4());           // D[h = 4g + r][i = l15]
            if constexpr (B16P) *reinterpret_cast<u32x2_t*>(GHk + ((k - 1) * 16 + l15) * LDB + ht * 16 + 4 * g) = pack_bf16x4(gk);
            else st4(GTk + ((k - 1) * 16 + l15) * LDX + ht * 16 + 4 * g, gk);
        }
        if constexpr (B16P) {   // the plane's padding columns NP .. LDB - 1 meet zero operator columns, but must not hold NaN bit patterns
            for (int idx = tid; idx < 16 * (LDB - NP); idx += nwa * 64) GHk[((k - 1) * 16 + idx / (LDB - NP)) * LDB + NP + idx % (LDB - NP)] = 0;
        }
    }
    // G_0^T of the owned tiles in the D layout of the operator products: A[m = i = l15][k = j] = W_0[i][j], B[k = j][n = h = l15] = dY[h][j]
    f32x4 g0[MAXQ];
    {
        const typename MM::frag w0 = MM::cvt(a.kipf ? zero4() : ld4(a.W + l15 * 16 + 4 * g));
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int ht = wave + WAVES * q;
            g0[q] = ht < HT ? MM::mma(w0, MM::cvt(ld4(dYs + (ht * 16 + l15) * LDY + 4 * g)), zero4()) : zero4();
        }
    }
    __syncthreads();   // (2) every G_k complete

    const size_t MSZ = (size_t)NP * NP;
    f32x4 acc1[MAXQ], acc2[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        acc1[q] = zero4();
        acc2[q] = zero4();
    }
    int nq = 0;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) nq += wave + WAVES * q < HT ? 1 : 0;
    for (int k0 = 1; k0 < Ks; k0 += 2) {
        if constexpr (B16P) {
            const size_t PSZ = gc_plane_floats(NP, N);
            const float* T1 = a.LTp + (size_t)(Ks - 1) * MSZ + (size_t)(k0 - 1) * 2 * PSZ;   // hi plane of T_k0^T (each term: hi, lo)
            const unsigned short* G1 = GHk + (k0 - 1) * 16 * LDB;
            gc_bwd_products_b16p_nq<MAXQ, (MAXQ == 1 ? gc_ring(1) : MAXQ == 2 ? 2 : 1)>(nq, k0 + 1 < Ks, T1, T1 + 2 * PSZ, G1, G1 + 16 * LDB, LDB, KC32, wave, WAVES,
                                                                                          lane, acc1, acc2);
        } else {
        const float* T1 = a.LTp + (size_t)(k0 - 1) * MSZ;
        const float* G1 = GTk + (k0 - 1) * 16 * LDX;
        gc_bwd_products_nq<MM, MAXQ, (MAXQ == 1 ? gc_ring(1) : MAXQ == 2 ? 2 : 1)>(nq, k0 + 1 < Ks, T1, T1 + MSZ, G1, G1 + 16 * LDX, LDX, KCH, wave, WAVES, lane, acc1, acc2);
        }
    }
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int ht = wave + WAVES * q;
        const int h = ht * 16 + l15;
        if (ht < HT && h < N) {
            const f32x4 y = ld4(dYs + h * LDY + 4 * g);
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = g0[q][r] + (acc1[q][r] + acc2[q][r]) + y[r];
            stx4_wt(et_ptr<ET>(a.dA) + ((size_t)slab * N + h) * 16 + 4 * g, o);
        }
    }
}

// ================================================================================================
// B4: backward of the Align(c0 -> c1) 1x1 map and of the first gate:
//     dH = dA Wa^T ; dZ1 = gate_bwd(dH, U1, S1) ; partial dWa[i][j] = sum H[.,i] dA[.,j], dba = sum dA
// grid-stride over 64-row tiles; wave w owns column tiles w + 4j of H (NTA = c0 / 64).
// ================================================================================================
struct AlignBwdArgs {
    const float* dA;     // [rows][c1]
    const float* U;      // [rows][c0]  (null: recompute U, S from x -- first block, Kt*c_in <= 16)
    const float* S;
    TapSrc ts;           // x viewed through Kt taps (recompute path)
    const float* Wd;     // dense W_eff [KPd][2*c0] (recompute path)
    const float* bias;   // b_eff [2*c0]
    int KPd;
    const float* WaT;    // packed: K = c1 (KCH chunks), cols = c0
    float* dZ;           // [rows][2*c0]
    float* part;         // [wgs][c0*c1 + c1]
    long rows;
    int c0, c1, KCH, act;
};

template <int NTA>
__global__ __launch_bounds__(256) void align_gate_bwd_kernel(AlignBwdArgs a) {
    extern __shared__ float stgcn_smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int c0 = a.c0, c1 = a.c1, LDA = c1 + 4, LDH = c0 + 4;
    float* dAt = stgcn_smem;                     // [64][LDA]
    float* Ht = stgcn_smem + 64 * LDA;           // [64][LDH]  dH, then H in place
    float* red = Ht + 64 * LDH;                  // [16][c1] for the dba reduction
    const long tiles = (a.rows + kTileRows - 1) / kTileRows;
    f32x4 wacc[NTA];                             // partial dWa tile (rows i = coltile*16 + 4g + r, col j = l15); c1 == 16
#pragma unroll
    for (int j = 0; j < NTA; ++j) wacc[j] = zero4();
    float bsum = 0.f;                            // thread (rg = tid >> 4, jj = tid & 15): column jj, rows rg, rg+16, ..
    const int c4n = c0 >> 2;
    STGCN_PHASE(6, 0);
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long row0 = t * kTileRows;
        const bool ph = t == blockIdx.x;
        __syncthreads();   // previous tile fully consumed
        if (ph) STGCN_PHASE(6, 1);
        for (int idx = tid; idx < kTileRows * (c1 >> 2); idx += kThreads) {
            const int r = idx / (c1 >> 2), q = idx - r * (c1 >> 2);
            st4(dAt + r * LDA + 4 * q, row0 + r < a.rows ? ld4(a.dA + (size_t)(row0 + r) * c1 + 4 * q) : zero4());
        }
        __syncthreads();
        {
            const int rg = tid >> 4, jj = tid & 15;
            if (jj < c1) bsum += dAt[rg * LDA + jj] + dAt[(rg + 16) * LDA + jj] + dAt[(rg + 32) * LDA + jj] + dAt[(rg + 48) * LDA + jj];
        }
        // dH = dA @ Wa^T  (wave w: column tiles w + 4j), D layout -> LDS
        f32x4 acc[4][NTA];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NTA; ++j) acc[i][j] = zero4();
        seg_mma<4, NTA>(acc, dAt, LDA, 0, a.KCH, a.WaT, 0, a.KCH, wave, 4);
#pragma unroll
        for (int j = 0; j < NTA; ++j) {
            const int col = (wave + 4 * j) * 16 + l15;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ht[(i * 16 + 4 * g + r) * LDH + col] = acc[i][j][r];
        }
        __syncthreads();
        if (ph) STGCN_PHASE(6, 2);
        // row-major pass with 16-byte global accesses: gate backward -> dZ, H back into the tile
        for (int idx = tid; idx < kTileRows * c4n; idx += kThreads) {
            const int row = fast_div(idx, c4n, pow2_shift(c4n)), c4 = idx - row * c4n;
            const long R = row0 + row;
            f32x4 h = zero4();
            if (R < a.rows) {
                const f32x4 dh = ld4(Ht + row * LDH + 4 * c4);
                f32x4 u, s;
                if (a.U) {
                    u = ld4(a.U + (size_t)R * c0 + 4 * c4);
                    s = ld4(a.S + (size_t)R * c0 + 4 * c4);
                } else {   // cheap conv (K <= 16): Z = im2col(x) @ W_eff + b_eff recomputed instead of stored
                    const unsigned per_b = (unsigned)(a.ts.Tdst * a.ts.N), Ru = (unsigned)R, b = Ru / per_b, rem = Ru - b * per_b;
                    const float* xr = tap_base(a.ts) + ((size_t)b * tap_bstride(a.ts) + rem) * a.ts.C;
                    u = ld4(a.bias + 4 * c4);
                    f32x4 qv = ld4(a.bias + c0 + 4 * c4);
                    const int K = a.ts.taps * a.ts.C;
                    for (int kidx = 0; kidx < K; ++kidx) {
                        const int tap = kidx / a.ts.C, ch = kidx - tap * a.ts.C;
                        const float xv = xr[(size_t)tap * a.ts.N * a.ts.C + ch];
                        const f32x4 wp = ld4(a.Wd + (size_t)kidx * 2 * c0 + 4 * c4), wq = ld4(a.Wd + (size_t)kidx * 2 * c0 + c0 + 4 * c4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            u[i] = fmaf(xv, wp[i], u[i]);
                            qv[i] = fmaf(xv, wq[i], qv[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) s[i] = sigmoid_f(qv[i]);
                }
                f32x4 du, dq;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float du_, dq_;
                    gate_bwd(dh[i], u[i], s[i], a.act, du_, dq_);
                    du[i] = du_;
                    dq[i] = dq_;
                    h[i] = gate_fwd(u[i], s[i], a.act);
                }
                st4_wt(a.dZ + (size_t)R * 2 * c0 + 4 * c4, du);
                st4_wt(a.dZ + (size_t)R * 2 * c0 + c0 + 4 * c4, dq);
            }
            st4(Ht + row * LDH + 4 * c4, h);
        }
        __syncthreads();
        if (ph) STGCN_PHASE(6, 3);
        // dWa[i][j] += sum_rows H[row][i] dA[row][j] : A[row_op = i (l15)][kk = row 4g+s], B[kk = row][col = j (l15)]
#pragma unroll
        for (int j = 0; j < NTA; ++j) {
            const int col = (wave + 4 * j) * 16 + l15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* hh = Ht + (i * 16 + 4 * g) * LDH + col;
                const float* bb = dAt + (i * 16 + 4 * g) * LDA + l15;
#pragma unroll
                for (int s = 0; s < 4; ++s) wacc[j] = mfma4(hh[s * LDH], bb[s * LDA], wacc[j]);
            }
        }
    }
    STGCN_PHASE(6, 4);
    float* part = a.part + (size_t)blockIdx.x * (c0 * c1 + c1);
#pragma unroll
    for (int j = 0; j < NTA; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((wave + 4 * j) * 16 + 4 * g + r) * c1 + l15] = wacc[j][r];
    __syncthreads();
    red[(tid >> 4) * c1 + (tid & 15)] = bsum;
    __syncthreads();
    if (tid < c1) {
        float s = 0.f;
        for (int rg = 0; rg < 16; ++rg) s += red[rg * c1 + tid];
        part[c0 * c1 + tid] = s;
    }
}

// ================================================================================================
// B4-thin: backward of the thin first layer (K = Kt*c_in <= 16), everything in one kernel per 64-row tile:
//   recompute U, S, H from x;  dH = dA Wa^T;  gate backward -> dZ tile (kept in LDS; written to HBM only if an input
//   gradient is needed);  partial dWa, dba (as align_gate_bwd_kernel);  partial dW_eff = im2col(x)^T dZ by MFMA
//   (M = 16 padded taps x N = 2*c0 x K = 64 rows) and db_eff = column sums of dZ.
// Replaces align_gate_bwd + tconv_bwd_weight for that layer and removes the dZ1 round trip through HBM.
// ================================================================================================
struct ThinBwdArgs {
    const float* dA;     // [rows][16]
    TapSrc ts;           // x through Kt taps
    const float* Wd;     // dense W_eff [16][2*c0]
    const float* bias;   // b_eff [2*c0]
    const float* WaT;    // packed PK_ALIGN_BWD: K = 16, cols = c0
    float* dZ;           // [rows][2*c0] or null
    float* part;         // [wgs][c0*16 + 16 + 16*2*c0 + 2*c0] : dWa | dba | dW_eff (16 rows) | db_eff
    long rows;
    int c0, act;
};

template <typename ET>
__global__ __launch_bounds__(256) void thin_tc1_bwd_kernel(ThinBwdArgs a) {
    typedef Mma<ET> MM;
    extern __shared__ float stgcn_smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    const int c0 = a.c0, NC = 2 * c0, LDA = 20, LDH = c0 + 4, LDZ = NC + 4, LDX = 20;
    float* dAt = stgcn_smem;                     // [64][LDA]
    float* Ht = dAt + 64 * LDA;                  // [64][LDH]  dH, then H
    float* Zt = Ht + 64 * LDH;                   // [64][LDZ]  dZ = [dU | dQ]
    float* xt = Zt + 64 * LDZ;                   // [64][LDX]  im2col rows of x (16 padded taps)
    float* red = xt + 64 * LDX;                  // [256]
    const long tiles = (a.rows + kTileRows - 1) / kTileRows;
    const int K = a.ts.taps * a.ts.C, c4n = c0 >> 2;
    const unsigned per_b = (unsigned)(a.ts.Tdst * a.ts.N);
    f32x4 wacc = zero4();                        // dWa tile rows wave*16.. (c0 == 64: one column tile of H per wave)
    f32x4 gacc[2] = {zero4(), zero4()};          // dW_eff tiles: n-tiles 2*wave, 2*wave+1 (NC == 128)
    float bsum = 0.f, zsum = 0.f;                // dba column (tid&15), db_eff column (tid % NC)
    const int zcol = tid % NC, zpart = tid / NC, zrows = 64 / (kThreads / NC);
    // per-thread constants of the row pass: the channel quad c4 = tid % c4n is the same in every iteration (256 % c4n == 0), so
    // the dense W_eff rows (K <= 4 taps x 2 halves) and the bias of that quad live in registers for the whole kernel
    constexpr int kThinK = 4;
    const int c4 = tid & (c4n - 1);
    f32x4 wpr[kThinK], wqr[kThinK];
#pragma unroll
    for (int k = 0; k < kThinK; ++k) {
        // (bf16: the forward formed these gate inputs on the matrix cores from ROUNDED weights; the recomputation uses the same numbers)
        wpr[k] = et_round4<ET>(k < K ? ld4(a.Wd + (size_t)k * NC + 4 * c4) : zero4());
        wqr[k] = et_round4<ET>(k < K ? ld4(a.Wd + (size_t)k * NC + c0 + 4 * c4) : zero4());
    }
    const f32x4 bu = ld4(a.bias + 4 * c4), bqv = ld4(a.bias + c0 + 4 * c4);
    const ET* const xsrc = tap_base<ET>(a.ts);
    const ET* const dA_ = et_ptr<ET>(a.dA);
    ET* const dZ_ = et_ptr<ET>(a.dZ);
    const size_t xbs = (size_t)tap_bstride(a.ts);
    // dA rows (one 16-byte load per thread) and the K valid taps of x (one scalar load for the first 64*K threads) of a tile are requested
    // together, raw, ONE TILE AHEAD (unconditional: clamped rows, masked where they are stored); the Align weights of the wave's column
    // tile are stationary.  Loaded where they are used, every tile waited for its dA / x round trip and then for the weight fragment's.
    const int ra = tid >> 2, qa = tid & 3;
    const int rx = tid < kTileRows * K ? tid / K : kTileRows - 1, kx = tid < kTileRows * K ? tid - rx * K : 0;
    auto request = [&](long t, Raw4<ET>& da, Raw1<ET>& xv) __attribute__((always_inline)) {
        const long row0 = (t < tiles ? t : tiles - 1) * kTileRows;
        const long Ra = row0 + ra < a.rows ? row0 + ra : a.rows - 1, Rx = row0 + rx < a.rows ? row0 + rx : a.rows - 1;
        da = ldraw4(dA_ + (size_t)Ra * 16 + 4 * qa);
        const unsigned Ru = (unsigned)Rx, b = Ru / per_b, rem = Ru - b * per_b;
        const int tap = kx / a.ts.C, ch = kx - tap * a.ts.C;
        xv = ldraw1(xsrc + ((size_t)b * xbs + rem + (size_t)tap * a.ts.N) * a.ts.C + ch);
    };
    PreW<1, 1> waw;
    pre_load_weights<1, 1>(waw, a.WaT, 1, wave, 4);
    Raw4<ET> da_n;
    Raw1<ET> xv_n;
    request(blockIdx.x, da_n, xv_n);
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const long row0 = t * kTileRows;
        __syncthreads();
        for (int idx = tid; idx < kTileRows * (16 - K); idx += kThreads) {
            const int r = idx / (16 - K), k = K + idx - r * (16 - K);
            xt[r * LDX + k] = 0.f;
        }
        st4(dAt + ra * LDA + 4 * qa, row0 + ra < a.rows ? cvt4(da_n) : zero4());
        if (tid < kTileRows * K) xt[rx * LDX + kx] = row0 + rx < a.rows ? cvt1(xv_n) : 0.f;
        request(t + gridDim.x, da_n, xv_n);   // (past the last tile: a valid address, never used)
        __syncthreads();
        {
            const int rg = tid >> 4, jj = tid & 15;
            bsum += dAt[rg * LDA + jj] + dAt[(rg + 16) * LDA + jj] + dAt[(rg + 32) * LDA + jj] + dAt[(rg + 48) * LDA + jj];
        }
        // dH = dA @ Wa^T (K = 16: one chunk; wave w: column tile w, 4 m-tiles) -> LDS
        f32x4 acc[4][1];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = zero4();
        pre_mma<4, 1, 1, ET>(acc, dAt, LDA, 0, 1, waw);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ht[(i * 16 + 4 * g + r) * LDH + wave * 16 + l15] = acc[i][0][r];
        __syncthreads();
        // row-major pass: recompute gate inputs, gate backward, dZ / H tiles
        for (int idx = tid; idx < kTileRows * c4n; idx += kThreads) {
            const int row = idx / c4n;            // (c4 = idx % c4n == tid % c4n)
            const long R = row0 + row;
            f32x4 h = zero4(), du = zero4(), dq = zero4();
            if (R < a.rows) {
                const f32x4 dh = ld4(Ht + row * LDH + 4 * c4);
                f32x4 u = bu, qv = bqv;
#pragma unroll
                for (int k = 0; k < kThinK; ++k) {
                    if (k < K) {
                        const float xv = xt[row * LDX + k];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            u[i] = fmaf(xv, wpr[k][i], u[i]);
                            qv[i] = fmaf(xv, wqr[k][i], qv[i]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sg = sigmoid_f(qv[i]);
                    float du_, dq_;
                    gate_bwd(dh[i], u[i], sg, a.act, du_, dq_);
                    du[i] = du_;
                    dq[i] = dq_;
                    h[i] = gate_fwd(u[i], sg, a.act);
                }
                if (a.dZ) {
                    stx4_wt(dZ_ + (size_t)R * NC + 4 * c4, du);
                    stx4_wt(dZ_ + (size_t)R * NC + c0 + 4 * c4, dq);
                }
            }
            st4(Ht + row * LDH + 4 * c4, h);
            st4(Zt + row * LDZ + 4 * c4, du);
            st4(Zt + row * LDZ + c0 + 4 * c4, dq);
        }
        __syncthreads();
        for (int r = 0; r < zrows; ++r) zsum += Zt[(zpart * zrows + r) * LDZ + zcol];
        // dWa[i][j] += H^T dA ; dW_eff[k][o] += xcol^T dZ   (K dimension = the 64 rows of the tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rbase = i * 16 + 4 * g;
            if constexpr (std::is_same<ET, float>::value) {
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const int row = rbase + sx;
                    wacc = mfma4(Ht[row * LDH + wave * 16 + l15], dAt[row * LDA + l15], wacc);
                    const float xa = xt[row * LDX + l15];
                    gacc[0] = mfma4(xa, Zt[row * LDZ + (2 * wave) * 16 + l15], gacc[0]);
                    gacc[1] = mfma4(xa, Zt[row * LDZ + (2 * wave + 1) * 16 + l15], gacc[1]);
                }
            } else {
                wacc = MM::mma(MM::cvt(gather4(Ht + rbase * LDH + wave * 16 + l15, LDH)), MM::cvt(gather4(dAt + rbase * LDA + l15, LDA)), wacc);
                MM::mma_b2(MM::cvt(gather4(xt + rbase * LDX + l15, LDX)), MM::cvt(gather4(Zt + rbase * LDZ + (2 * wave) * 16 + l15, LDZ)),
                           MM::cvt(gather4(Zt + rbase * LDZ + (2 * wave + 1) * 16 + l15, LDZ)), gacc[0], gacc[1]);
            }
        }
    }
    const int pstride = c0 * 16 + 16 + 16 * NC + NC;
    float* part = a.part + (size_t)blockIdx.x * pstride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        part[(wave * 16 + 4 * g + r) * 16 + l15] = wacc[r];                                    // dWa[i][j]
        part[c0 * 16 + 16 + (4 * g + r) * NC + (2 * wave) * 16 + l15] = gacc[0][r];            // dW_eff[k][o]
        part[c0 * 16 + 16 + (4 * g + r) * NC + (2 * wave + 1) * 16 + l15] = gacc[1][r];
    }
    __syncthreads();
    red[tid] = bsum;
    __syncthreads();
    if (tid < 16) {
        float sacc = 0.f;
        for (int rg = 0; rg < 16; ++rg) sacc += red[rg * 16 + tid];
        part[c0 * 16 + tid] = sacc;
    }
    __syncthreads();
    red[tid] = zsum;
    __syncthreads();
    if (tid < NC) {
        float sacc = 0.f;
        for (int pz = 0; pz < kThreads / NC; ++pz) sacc += red[pz * NC + tid];
        part[c0 * 16 + 16 + 16 * NC + tid] = sacc;
    }
}

// the wave-per-tile form of round 5 (stgcn_kernels_thin.hip.h): same arguments, same partial layout
template <typename ET, int ACT>
__global__ __launch_bounds__(256, 3) void thin_tc1_bwd2_kernel(ThinBwdArgs a) {   // (3 workgroups per CU = 3 waves per SIMD: 170 registers)
    thin_tc1_bwd2_body<ET, ACT>(a);
}

// ================================================================================================
// B6: weight gradient of a temporal convolution
//     dW_eff[k*Cin + i][o] = sum_rows x[row + k*N][i] dZ[row][o]      db_eff[o] = sum_rows dZ[row][o]
// grid = (row chunks, m chunks).  The reduction over a chunk's rows runs in SR-row steps through LDS with the next
// step's global loads in flight (register prefetch).  The problem only offers ~one workgroup per CU (more row chunks
// would mean more partials), and one wave per SIMD cannot hide the load latency of a step behind its 3 K cycles of
// MFMAs, so a workgroup carries GROUPS independent wave groups (4 waves each, own LDS tiles, own prefetch): group q
// reduces steps q, q+GROUPS, ...; the groups' accumulators are summed through LDS in a fixed order at the end.
// Within a group wave w owns n-tiles w*NTW .. w*NTW+NTW-1 (NTW = NC/64) of all MTW m-tiles of this m-chunk.
// Output: per-chunk partials.
// ================================================================================================
#ifndef STGCN_WGRAD_GROUPS
#define STGCN_WGRAD_GROUPS 2
#endif
constexpr int kWgradGroups = STGCN_WGRAD_GROUPS;
struct TconvBwdWeightArgs {
    TapSrc ts;           // x viewed through Kt taps (dir = +1): implicit [rows][K = Kt*Cin]
    const float* dZ;     // [rows][NC]
    float* part;         // [chunks][Mpad*NC] ++ [chunks][NC]
    int NC, Mpad, rows_per_chunk, chunks;
};
// rows reduced per barrier interval: 64, or 32 where GROUPS tile sets of 64 rows would not fit the LDS
constexpr int wgrad_step_rows(int MTW, int NTW) {
    return (size_t)kWgradGroups * 64 * ((MTW * 16 + 4) + (NTW * 64 + 4)) * sizeof(float) <= 156 * 1024 ? 64 : 32;
}
constexpr size_t wgrad_lds_bytes(int MTW, int NTW) {
    return (size_t)kWgradGroups * wgrad_step_rows(MTW, NTW) * ((MTW * 16 + 4) + (NTW * 64 + 4)) * sizeof(float);
}

// VEC: the input channel count is a multiple of 4 (16-byte im2col loads); otherwise scalar loads (narrow first layer).
// The body is a device function of (flat workgroup index, workgroup count, m chunks) so that two independent weight gradients can
// share ONE launch (wgrad_pair_kernel below: an empty launch costs 4.6 us inside the replayed step).
template <int MTW, int NTW, bool VEC, typename ET>
__device__ __forceinline__ void tconv_bwd_weight_body(const TconvBwdWeightArgs& a, int flat_wg, int n_wgs, int mchunks) {
    extern __shared__ float stgcn_smem[];
    constexpr int GROUPS = kWgradGroups;
    constexpr int SR = wgrad_step_rows(MTW, NTW);
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;   // wave group, thread within the group
    const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
    constexpr int MC = MTW * 16, LDC = MC + 4;
    constexpr int NC = NTW * 64, LDZ = NC + 4;
    float* ct = stgcn_smem + grp * (SR * LDC + SR * LDZ);   // [SR][LDC]
    float* zt = ct + SR * LDC;                               // [SR][LDZ]
    // the m-chunks of one row chunk re-read the same dZ rows: keep them adjacent on one XCD (xcd_item)
    const int item = xcd_item(flat_wg, n_wgs);
    const int chunk = item / mchunks, mchunk = item % mchunks, m0 = mchunk * MC;
    const long crow0 = (long)chunk * a.rows_per_chunk;
    long crow1 = crow0 + a.rows_per_chunk;
    if (crow1 > a.ts.rows) crow1 = a.ts.rows;
    const int nsteps = (int)((crow1 - crow0 + SR - 1) / SR);
    const int nit = (nsteps + GROUPS - 1) / GROUPS;   // barrier intervals (the same for every group)
    const int K = a.ts.taps * a.ts.C;
    const unsigned per_b = (unsigned)(a.ts.Tdst * a.ts.N);   // rows < 2^31 (checked on the host): 32-bit divisions only
    const int csh = pow2_shift(a.ts.C);
    const ET* const xsrc = tap_base<ET>(a.ts);
    const ET* const dZ_ = et_ptr<ET>(a.dZ);
    const size_t xbs = (size_t)tap_bstride(a.ts);

    // staging registers (next step's tiles are fetched while the current step's MFMAs run)
    constexpr int NCR = (SR * (MC / 4) + kThreads - 1) / kThreads;   // float4 of the im2col tile per thread (vector path)
    constexpr int NCS = (SR * MC + kThreads - 1) / kThreads;         // scalars per thread (narrow-input path)
    constexpr int NZ = SR * NTW * 16 / kThreads;                      // SR * NC / 4 / 256 with NC = 64 * NTW
    Raw4<ET> creg[VEC ? NCR : 1], zreg[NZ];   // (raw: unpacked by store_regs -- an unpack right behind the load would wait for it there)
    float cs[VEC ? 1 : NCS];
    auto load_regs = [&](int step) {
        const long r0 = crow0 + (long)step * SR;
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NCR; ++i) {
                const int idx = tid + i * kThreads;
                const int r = idx / (MC / 4), q = idx - r * (MC / 4);
                Raw4<ET> v;
                v.v = {};
                if (r < SR) {
                    const long R = r0 + r;
                    const int kidx = m0 + 4 * q;
                    if (R < crow1 && kidx < K) {
                        const unsigned Ru = (unsigned)R, b = Ru / per_b, rem = Ru - b * per_b;
                        const int tap = fast_div(kidx, a.ts.C, csh), ch = kidx - tap * a.ts.C;
                        v = ldraw4(xsrc + ((size_t)b * xbs + rem + (size_t)tap * a.ts.N) * a.ts.C + ch);
                    }
                }
                creg[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCS; ++i) {
                const int idx = tid + i * kThreads;
                const int r = idx / MC, q = idx - r * MC;
                float v = 0.f;
                if (r < SR) {
                    const long R = r0 + r;
                    const int kidx = m0 + q;
                    if (R < crow1 && kidx < K) {
                        const unsigned Ru = (unsigned)R, b = Ru / per_b, rem = Ru - b * per_b;
                        const int tap = fast_div(kidx, a.ts.C, csh), ch = kidx - tap * a.ts.C;
                        v = ldx1(xsrc + ((size_t)b * xbs + rem + (size_t)tap * a.ts.N) * a.ts.C + ch);
                    }
                }
                cs[i] = v;
            }
        }
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const int idx = tid + z * kThreads;
            const int r = idx / (NC / 4), q = idx - r * (NC / 4);
            const long R = r0 + r;
            zreg[z].v = {};
            if (R < crow1) zreg[z] = ldraw4(dZ_ + (size_t)R * NC + 4 * q);
        }
    };
    auto store_regs = [&]() {
        if constexpr (VEC) {
#pragma unroll
            for (int i = 0; i < NCR; ++i) {
                const int idx = tid + i * kThreads;
                const int r = idx / (MC / 4), q = idx - r * (MC / 4);
                if (r < SR) st4(ct + r * LDC + 4 * q, cvt4(creg[i]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCS; ++i) {
                const int idx = tid + i * kThreads;
                const int r = idx / MC, q = idx - r * MC;
                if (r < SR) ct[r * LDC + q] = cs[i];
            }
        }
#pragma unroll
        for (int z = 0; z < NZ; ++z) {
            const int idx = tid + z * kThreads;
            const int r = idx / (NC / 4), q = idx - r * (NC / 4);
            st4(zt + r * LDZ + 4 * q, cvt4(zreg[z]));
        }
    };

    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = zero4();
    // bias partial: NC columns, tpc = 256 / NC threads per column (NC = 128 -> 2, NC = 256 -> 1)
    constexpr int tpc = kThreads / NC, rpt = SR / tpc;
    const int bcol = tid % NC, bpart = tid / NC;
    float bsum = 0.f;

    STGCN_PHASE(3, 0);
    if (grp < nsteps) load_regs(grp);
    STGCN_PHASE(3, 1);
    for (int it = 0; it < nit; ++it) {
        const int step = it * GROUPS + grp;
        const bool active = step < nsteps;   // uniform per wave group
        if (it > 0) __syncthreads();   // previous step's tiles consumed
        if (active) store_regs();
        __syncthreads();
        if (!active) continue;
        if (step + GROUPS < nsteps) load_regs(step + GROUPS);
        if (mchunk == 0) {
#pragma unroll 4
            for (int r = 0; r < rpt; ++r) bsum += zt[(bpart * rpt + r) * LDZ + bcol];
        }
        // A[m = l15][kk = g] = ct[row][m], B[kk = g][o = l15] = zt[row][o] with row = 16*k16 + 4g + s: lanes with
        // g = 0..3 read rows 4 apart, which the LDC/LDZ = 4 (mod 8) padding spreads over distinct banks
#pragma unroll 1
        for (int k16 = 0; k16 < SR / 16; ++k16) {
            if constexpr (std::is_same<ET, float>::value) {
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const int row = k16 * 16 + 4 * g + sx;
                    float av[MTW], bv[NTW];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) av[i] = ct[row * LDC + i * 16 + l15];
#pragma unroll
                    for (int j = 0; j < NTW; ++j) bv[j] = zt[row * LDZ + (wave * NTW + j) * 16 + l15];
#pragma unroll
                    for (int i = 0; i < MTW; ++i)
#pragma unroll
                        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma4(av[i], bv[j], acc[i][j]);
                }
            } else {
                const int row = k16 * 16 + 4 * g;
                typename Mma<ET>::frag fa[MTW], fb[NTW];
#pragma unroll
                for (int i = 0; i < MTW; ++i) fa[i] = Mma<ET>::cvt(gather4(ct + row * LDC + i * 16 + l15, LDC));
#pragma unroll
                for (int j = 0; j < NTW; ++j) fb[j] = Mma<ET>::cvt(gather4(zt + row * LDZ + (wave * NTW + j) * 16 + l15, LDZ));
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[i][j] = Mma<ET>::mma(fa[i], fb[j], acc[i][j]);
            }
        }
    }
    STGCN_PHASE(3, 2);
    // ---- combine the wave groups (fixed order: group 0 + group 1 + ...) ----------------------------------------
    if (GROUPS > 1) {
        float* xb = stgcn_smem;   // [MTW*NTW][256] float4 + [256] floats; fits the tile area for every instantiation
        for (int q = 1; q < GROUPS; ++q) {
            __syncthreads();   // tiles (or the previous exchange) consumed
            if (grp == q) {
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) st4(xb + ((i * NTW + j) * kThreads + tid) * 4, acc[i][j]);
                xb[MTW * NTW * kThreads * 4 + tid] = bsum;
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j) acc[i][j] += ld4(xb + ((i * NTW + j) * kThreads + tid) * 4);
                bsum += xb[MTW * NTW * kThreads * 4 + tid];
            }
        }
    }
    float* part = a.part + (size_t)chunk * a.Mpad * NC;
    if (grp == 0) {
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
                st4_wt(part + (size_t)((wave * NTW + j) * 16 + l15) * a.Mpad + m0 + i * 16 + 4 * g, acc[i][j]);   // TRANSPOSED partial [NC][Mpad]: a lane's 4 rows are 16 contiguous bytes
    }
    if (mchunk == 0) {   // uniform per workgroup
        float* bp = a.part + (size_t)a.chunks * a.Mpad * NC + (size_t)chunk * NC;
        if (tpc == 1) {
            if (grp == 0) bp[bcol] = bsum;
        } else {
            __syncthreads();
            if (grp == 0) stgcn_smem[tid] = bsum;
            __syncthreads();
            if (grp == 0 && tid < NC) {
                float s = 0.f;
                for (int p = 0; p < tpc; ++p) s += stgcn_smem[p * NC + tid];
                bp[tid] = s;
            }
        }
    }
    STGCN_PHASE(3, 3);
}
template <int MTW, int NTW, bool VEC, typename ET>
__global__ __launch_bounds__(256 * kWgradGroups) void tconv_bwd_weight_kernel(TconvBwdWeightArgs a) {
    tconv_bwd_weight_body<MTW, NTW, VEC, ET>(a, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y), (int)gridDim.y);
}
// two weight gradients of one backward call in one launch: workgroups [0, n1) run the first, [n1, n1 + n2) the second
template <int MTW1, int NTW1, int MTW2, int NTW2, typename ET>
__global__ __launch_bounds__(256 * kWgradGroups) void wgrad_pair_kernel(TconvBwdWeightArgs a1, int n1, int mc1, TconvBwdWeightArgs a2, int n2, int mc2) {
    if ((int)blockIdx.x < n1) tconv_bwd_weight_body<MTW1, NTW1, true, ET>(a1, (int)blockIdx.x, n1, mc1);     // (uniform per workgroup)
    else tconv_bwd_weight_body<MTW2, NTW2, true, ET>(a2, (int)blockIdx.x - n1, n2, mc2);
}

// ================================================================================================
// Final deterministic reduction of the partials into gradients laid out like the reference's parameters.
// A job enumerates a 3-D index (d0, d1, d2), d2 fastest and contiguous in the SOURCE (coalesced reads):
//     dst[d0*t0 + d1*t1 + d2*t2] = sum_p src[p*pstride + d0*s0 + d1*s1 + d2*s2]
// A workgroup owns 256/slices consecutive element GROUPS (a group = 4 consecutive d2 when the job's strides allow 16-byte
// loads, else 1 element); its slices (8, or 32 for jobs with >= 128 partials: the walk over p is a serial latency chain)
// take the partials p = slice, slice+slices, .. and are combined through LDS in a fixed order (bitwise reproducible).
// ================================================================================================
struct ReduceJob {
    const float* src;
    float* dst;
    float* p;     // optional fused AdamW (stgcn_grad_flush): parameter, exp_avg, exp_avg_sq laid out like dst; null = reduce only
    float* m;
    float* v;
    long pstride;
    int P;
    int vec;      // 1: n2, pstride, s0, s1 are multiples of 4 and src is 16-byte aligned
    int dvec;     // 1: vec and the 4 elements of a group are 16 contiguous, aligned bytes in dst / p / m / v too (t2 == 1: the flat jobs -- biases,
                  //    LayerNorm parameters, which are most of the elements on big graphs): the write and the AdamW state move as 16-byte accesses
    int slices;   // 4, 8 or 32
    int n0, n1, n2;
    int s0, s1, s2;
    int t0, t1, t2;
};
constexpr int kMaxReduceJobs = 32;   // 2 ST blocks x 10-12 + the head's 8-10 in one launch (kernel arguments <= 4 KiB)
struct ReduceArgs {
    ReduceJob job[kMaxReduceJobs];
    int start[kMaxReduceJobs + 1];
    int njobs;
    // fused AdamW (only read for jobs with p != null): same arithmetic as adamw_kernel
    float lr, b1, b2, eps, wd, lb1, lb2;
    long step;
    const long* step_dev;
    const float* lr_dev;
};
// host: classify the job and return its workgroup count
// element count from which a table takes the big-table forms below (STGCN_REDUCE_BIG: the emulator tests force them on small models)
inline long reduce_big_threshold() {
    static const long t = getenv("STGCN_REDUCE_BIG") ? atol(getenv("STGCN_REDUCE_BIG")) : 65536;
    return t;
}
inline int reduce_job_setup(ReduceJob& j) {
    j.vec = (j.n2 % 4 == 0 && j.s2 == 1 && j.pstride % 4 == 0 && j.s0 % 4 == 0 && j.s1 % 4 == 0 &&
             (reinterpret_cast<uintptr_t>(j.src) & 15) == 0) ? 1 : 0;
    const long n = (long)j.n0 * j.n1 * j.n2;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    j.dvec = (n >= reduce_big_threshold() && j.vec && j.t2 == 1 && j.t0 % 4 == 0 && j.t1 % 4 == 0 && al16(j.dst) && (!j.p || (al16(j.p) && al16(j.m) && al16(j.v)))) ? 1 : 0;
    // big tables with few partials (the LayerNorm parameters of an 8192-node graph: 524 288 elements x 16 windows): 4 slices -- twice the
    // elements per workgroup, 4 loads in flight per thread, a quarter of the threads in the state update instead of an eighth
    j.slices = j.P >= 128 ? 32 : (n >= reduce_big_threshold() && j.P <= 32) ? 4 : 8;
    const long per = (long)(kThreads / j.slices) * (j.vec ? 4 : 1);
    return (int)((n + per - 1) / per);
}

__global__ __launch_bounds__(256) void reduce_kernel(ReduceArgs a) {
    extern __shared__ float stgcn_smem[];   // [slices][32] float4
    int jb = 0;
    while (jb + 1 < a.njobs && (int)blockIdx.x >= a.start[jb + 1]) ++jb;
    const ReduceJob& j = a.job[jb];
    const int kReduceSlices = j.slices, kReduceElems = kThreads / kReduceSlices;
    const int el = threadIdx.x % kReduceElems, sl = threadIdx.x / kReduceElems;
    const int W = j.vec ? 4 : 1;
    const long e = (((long)blockIdx.x - a.start[jb]) * kReduceElems + el) * W;   // first element of this thread's group
    const long n = (long)j.n0 * j.n1 * j.n2;
    f32x4 acc = zero4();
    long doff = 0;
    // the optimizer state of this group, requested BEFORE the walk over the partials (the thread that combines the slices updates it: its
    // loads would otherwise be a second dependent memory round trip at the end of every workgroup's life)
    const int nw = j.vec ? 4 : 1;
    f32x4 om = zero4(), ov = zero4(), op = zero4();
    float ts = 0.f, lr = 0.f;
    if (e < n) {
        const int d2 = (int)(e % j.n2), d1 = (int)((e / j.n2) % j.n1), d0 = (int)(e / ((long)j.n1 * j.n2));
        const float* s = j.src + (long)d0 * j.s0 + (long)d1 * j.s1 + (long)d2 * j.s2;
        doff = (long)d0 * j.t0 + (long)d1 * j.t1 + (long)d2 * j.t2;
        if (sl == 0 && j.p) {
            ts = (float)(a.step_dev ? *a.step_dev : a.step);
            lr = a.lr_dev ? *a.lr_dev : a.lr;
            if (j.dvec) {
                om = ld4(j.m + doff);
                ov = ld4(j.v + doff);
                op = ld4(j.p + doff);
            } else
            for (int i = 0; i < nw; ++i) {
                const long o = doff + (long)i * j.t2;
                om[i] = j.m[o];
                ov[i] = j.v[o];
                op[i] = j.p[o];
            }
        }
        int p = sl;
        if (j.vec) {
            f32x4 a0 = zero4(), a1 = zero4(), a2 = zero4(), a3 = zero4();   // 4 x 16 B in flight per thread
            for (; p + 3 * kReduceSlices < j.P; p += 4 * kReduceSlices) {
                a0 += ld4(s + (size_t)p * j.pstride);
                a1 += ld4(s + (size_t)(p + kReduceSlices) * j.pstride);
                a2 += ld4(s + (size_t)(p + 2 * kReduceSlices) * j.pstride);
                a3 += ld4(s + (size_t)(p + 3 * kReduceSlices) * j.pstride);
            }
            for (; p < j.P; p += kReduceSlices) a0 += ld4(s + (size_t)p * j.pstride);
            acc = (a0 + a1) + (a2 + a3);
        } else {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (; p + 3 * kReduceSlices < j.P; p += 4 * kReduceSlices) {
                a0 += s[(size_t)p * j.pstride];
                a1 += s[(size_t)(p + kReduceSlices) * j.pstride];
                a2 += s[(size_t)(p + 2 * kReduceSlices) * j.pstride];
                a3 += s[(size_t)(p + 3 * kReduceSlices) * j.pstride];
            }
            for (; p < j.P; p += kReduceSlices) a0 += s[(size_t)p * j.pstride];
            acc[0] = (a0 + a1) + (a2 + a3);
        }
    }
    st4(stgcn_smem + (sl * kReduceElems + el) * 4, acc);
    __syncthreads();
    if (sl == 0 && e < n) {
        f32x4 t = zero4();
        for (int k = 0; k < kReduceSlices; ++k) t += ld4(stgcn_smem + (k * kReduceElems + el) * 4);
        if (j.dvec) st4(j.dst + doff, t);
        else
            for (int i = 0; i < nw; ++i) j.dst[doff + (long)i * j.t2] = t[i];
        if (j.p) {   // AdamW on the freshly reduced gradient (torch.optim.AdamW, see adamw_kernel)
            const float bc1 = -expm1f(ts * a.lb1);
            const float rs2 = rsqrtf(-expm1f(ts * a.lb2));
            const float decay = 1.0f - lr * a.wd, step_size = lr / bc1;
            f32x4 nm, nv, np_;
            for (int i = 0; i < nw; ++i) {
                const float g = t[i];
                nm[i] = a.b1 * om[i] + (1.0f - a.b1) * g;
                nv[i] = a.b2 * ov[i] + (1.0f - a.b2) * g * g;
                np_[i] = op[i] * decay - step_size * (nm[i] / (sqrtf(nv[i]) * rs2 + a.eps));
            }
            if (j.dvec) {
                st4(j.m + doff, nm);
                st4(j.v + doff, nv);
                st4(j.p + doff, np_);
            } else
                for (int i = 0; i < nw; ++i) {
                    const long o = doff + (long)i * j.t2;
                    j.m[o] = nm[i];
                    j.v[o] = nv[i];
                    j.p[o] = np_[i];
                }
        }
    }
}

}  // namespace stgcn
