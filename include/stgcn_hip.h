/* stgcn_hip.h -- C ABI of the MI355X-native STGCN ST-Conv-block training path (libstgcn_hip.so).
 *
 * The reference (hazdzz/STGCN) has no FFI layer: its hot path sits behind Python nn.Modules whose
 * arithmetic is delegated to ATen.  Each entry point below replaces the ATen call sequence of one
 * reference method; the Python mirror of the reference's module API (stgcn_amd/layers.py, models.py)
 * binds these symbols through ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 *   stgcn_stblock_forward   <- model/layers.py:250-258  STConvBlock.forward
 *                              (= layers.py:87-120 TemporalConvLayer.forward x2, :14-23 Align.forward,
 *                                 :222-231 GraphConvLayer.forward, :143-172 ChebGraphConv.forward or
 *                                 :194-206 GraphConv.forward, nn.LayerNorm :246/255, nn.Dropout :248/256)
 *   stgcn_stblock_backward  <- what autograd derives from the above when main.py:168 calls l.backward()
 *   stgcn_gso_prepare       <- main.py:101-103 (dense fp32 GSO upload); pads/transposes it once
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless stated; the library never allocates or frees.
 *   - activations are channels-last: a logical (B, C, T, N) tensor is stored (B, T, N, C) contiguous
 *     (the memory layout the reference itself produces after its first graph conv).
 *   - parameters are passed in the reference's native state_dict layouts.
 *   - `stream` is a hipStream_t; all work is enqueued on it, nothing synchronises.
 *   - every function returns STGCN_OK or an error code; stgcn_last_error() gives a message.
 */
#ifndef STGCN_HIP_H
#define STGCN_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision: bumped whenever an entry point changes its argument list or a struct its layout (round 3 added `y` to the
 * backward entry points and `stored_US2` to the plan: 1 -> 2 in effect, never recorded; round 4: stgcn_set_gemm_big_nt, the
 * chained-launch control words in `ws`: 3, then 4; round 5: stgcn_set_chain_spin_ticks, stgcn_outblock_chain_status: 5; round 6: stgcn_set_tc2ln_peers, stgcn_stblock_chain_status, stgcn_prepack_park / _flush, the exchange words of tmp_conv2 + LayerNorm in `ws`: 6).  stgcn_version() returns the value the LIBRARY was built with; a binding built
 * against another header must refuse to run (stgcn_amd/_lib.py does).                                                  */
#define STGCN_ABI_VERSION 6

#define STGCN_OK 0
#define STGCN_ERR_UNSUPPORTED 1 /* shape outside what the kernels cover (message says which) */
#define STGCN_ERR_INVALID 2     /* null pointer / bad enum / inconsistent sizes */
#define STGCN_ERR_LAUNCH 3      /* HIP reported a launch error */

#define STGCN_ACT_GLU 0 /* (P + R) * sigmoid(Q)       layers.py:105 */
#define STGCN_ACT_GTU 1 /* tanh(P + R) * sigmoid(Q)   layers.py:109 */
#define STGCN_GC_CHEB 0 /* cheb_graph_conv            layers.py:143-172 */
#define STGCN_GC_KIPF 1 /* graph_conv                 layers.py:194-206 */

/* Storage / arithmetic type of the ACTIVATIONS of a call (desc.dtype).  Parameters, their gradients, LayerNorm statistics, the
 * partial-sum arena, optimizer state and the head's prediction are fp32 in both modes.
 *   STGCN_DTYPE_F32  : fp32 tensors, exact fp32 matrix products (BASELINE.json configs[0], [1], [3])
 *   STGCN_DTYPE_BF16 : x, y, dy, dx and every tensor of `saved` / `ws` that holds activations or activation gradients are bf16
 *                      (2 bytes per element, round-to-nearest-even; the plan still counts 4-byte units), matrix products run on the
 *                      bf16 matrix cores with fp32 accumulation, everything else stays fp32 (configs[2], [4]).  The rounding points are
 *                      restated in oracle/stblock_stages.py (QuantBf16).  Pointers keep their `float*` type in this header: they are
 *                      opaque device addresses.                                                                                     */
#define STGCN_DTYPE_F32 0
#define STGCN_DTYPE_BF16 1

/* One STConvBlock(Kt, Ks, n_vertex, last_block_channel, channels, act_func, graph_conv_type, gso,
 * bias, droprate) (layers.py:241) applied to a batch. */
typedef struct stgcn_stblock_desc {
    int32_t B, T, N;          /* batch, input time steps, vertices                                   */
    int32_t c_in;             /* last_block_channel                                                  */
    int32_t c0, c1, c2;       /* channels[0..2]                                                      */
    int32_t Kt, Ks;           /* temporal / Chebyshev kernel sizes (Ks ignored for STGCN_GC_KIPF)    */
    int32_t act;              /* STGCN_ACT_*                                                         */
    int32_t graph_conv;       /* STGCN_GC_*                                                          */
    int32_t training;         /* 1: dropout active (nn.Module.train()), 0: eval                      */
    float droprate;           /* p of nn.Dropout                                                     */
    float ln_eps;             /* 1e-12 in the reference (layers.py:246)                              */
    int32_t need_dx;          /* backward: also produce the input gradient                           */
    int32_t reserved;         /* free-form tag (e.g. block index); only used to label the built-in kernel timer   */
    int32_t prepacked;        /* 1: stgcn_prepack already rewrote this call's weights into ws (forward skips its pack launch) */
    int32_t defer_reduce;     /* backward: 1 = leave the per-workgroup gradient partials in ws; stgcn_grad_flush reduces them     */
    /* Device-side windowing (script/dataloader.py:32-47 builds a (num, 1, n_his, N) tensor that repeats every time step n_his
     * times; here the first block can read its windows straight from the resident (time, N) series): window b of `x` starts
     * x_bstride ROWS (of c_in floats) after window b-1 (0 = dense, T*N; N = windows one time step apart), and the whole input
     * is shifted by *x_index_dev * x_index_stride floats (nullable: the batch position of a captured training step).
     * Only for blocks whose input needs no gradient (need_dx = 0).                                                          */
    int64_t x_bstride;
    const int64_t* x_index_dev;
    int64_t x_index_stride;
    int32_t dy_rowstats_ready; /* backward: 1 = the kernel that produced `dy` already wrote this block's LayerNorm-backward row partials
                                  (stgcn_ln_hook handed to that producer's backward call); the block skips its own pass over dy          */
    int32_t dtype;            /* STGCN_DTYPE_*                                                                               */
} stgcn_stblock_desc;

/* Parameter pointers, keyed like the reference state_dict under "st_blocks.<l>." :
 *   tc1_w  tmp_conv1.causal_conv.weight (2*c0, c_in, Kt, 1)   tc1_b  .bias (2*c0)
 *   tc1_aw tmp_conv1.align.align_conv.weight (c0, c_in, 1, 1) tc1_ab .bias (c0)   [read iff c_in > c0]
 *   al_w   graph_conv.align.align_conv.weight (c1, c0, 1, 1)  al_b   .bias (c1)   [read iff c0 > c1]
 *   gc_w   graph_conv.cheb_graph_conv.weight (Ks, c1, c1) | graph_conv.graph_conv.weight (c1, c1)
 *   gc_b   ....bias (c1) or NULL (enable_bias False)
 *   tc2_w  tmp_conv2.causal_conv.weight (2*c2, c1, Kt, 1)     tc2_b  .bias (2*c2)
 *   tc2_aw tmp_conv2.align.align_conv.weight (c2, c1, 1, 1)   tc2_ab .bias (c2)   [read iff c1 > c2]
 *   ln_w   tc2_ln.weight (N, c2)                              ln_b   tc2_ln.bias (N, c2)           */
typedef struct stgcn_stblock_params {
    const float *tc1_w, *tc1_b, *tc1_aw, *tc1_ab;
    const float *al_w, *al_b;
    const float *gc_w, *gc_b;
    const float *tc2_w, *tc2_b, *tc2_aw, *tc2_ab;
    const float *ln_w, *ln_b;
} stgcn_stblock_params;

/* Gradient outputs, same keys/layouts; a NULL entry is skipped.  Entries whose parameter is unused by
 * the forward (align convs with c_in <= c_out) are never written: the reference leaves .grad None.  */
typedef struct stgcn_stblock_grads {
    float *tc1_w, *tc1_b, *tc1_aw, *tc1_ab;
    float *al_w, *al_b;
    float *gc_w, *gc_b;
    float *tc2_w, *tc2_b, *tc2_aw, *tc2_ab;
    float *ln_w, *ln_b;
} stgcn_stblock_grads;

/* Buffer sizes and the layout of the `saved` / `ws` buffers (all in floats).  `saved` carries the
 * activations kept for backward, `ws` holds packed weights (written by forward, re-used by backward of
 * the same step) and backward temporaries.  Offsets are exposed so tests can check every stage.      */
typedef struct stgcn_stblock_plan {
    int64_t T1, T2, rows1, rows2, NP; /* T1 = T-Kt+1, T2 = T1-Kt+1, rowsX = B*TX*N, NP = roundup16(N)  */
    int64_t y_floats;                 /* B*T2*N*c2                                                     */
    int64_t saved_floats, ws_floats;
    /* saved */
    int64_t sv_U1, sv_S1;             /* [rows1][c0] gate inputs of tmp_conv1 (U = P + R, S = sigmoid Q) */
    int64_t sv_A;                     /* [rows1][c1] aligned graph-conv input X0                       */
    int64_t sv_Xk;                    /* [Ks-1][rows1][c1] Chebyshev terms X1..                        */
    int64_t sv_G;                     /* [rows1][c1] relu(graph conv + residual)                       */
    int64_t sv_U2, sv_S2;             /* [rows2][c2] gate inputs of tmp_conv2 -- only stored when stored_US2 is set (else empty)        */
    int64_t sv_mean, sv_rstd;         /* [B*T2]                                                        */
    int64_t sv_rowstat;               /* [rows2][2] per-row LayerNorm partials (mean, M2) of tmp_conv2 output */
    /* ws: packed weights */
    int64_t ws_W1p, ws_W1d, ws_b1, ws_Wap, ws_WaT, ws_ba, ws_W2p, ws_W2d, ws_b2;
    int64_t ws_W1dense;               /* [KP1][2*c0] W_eff row major (cheap first conv: gate inputs recomputed in backward) */
    int64_t recompute_tc1;            /* 1: U1/S1 are not stored (Kt*c_in <= 16)                          */
    int64_t ws_WaDense;               /* [c0][c1] dense Align map (thin first-layer kernels)               */
    int64_t thin_tc1;                 /* 1: first layer runs the thin (thread-per-row) kernels             */
    /* ws: backward temporaries */
    int64_t ws_rowstat_b;             /* [rows2][2] LayerNorm backward row partials (sum g, sum g*xhat)    */
    int64_t ws_dZ2;                   /* [rows2][2*c2]                                                 */
    int64_t ws_dYg;                   /* [rows1][c1]  d(relu out) masked                               */
    int64_t ws_dA;                    /* [rows1][c1]                                                   */
    int64_t ws_dZ1;                   /* [rows1][2*c0]                                                 */
    int64_t ws_part;                  /* partial-sum arena for the parameter gradients                 */
    int64_t part_floats;
    int64_t tiled_gc;                 /* 1: the graph conv runs the tiled GEMM path (N > 512 nodes, or more terms than the
                                         slab-resident backward holds in LDS); NP = roundup128(N) then                    */
    int64_t ws_Gk;                    /* tiled_gc: [terms][rows1][c1] Clenshaw buffers of the graph-conv backward         */
    int64_t ws_XT;                    /* tiled_gc: two bf16 operand-form buffers (hi, lo planes of [CP][NP]) for the bf16 /
                                         bf16x3 operator products (stgcn_set_gc_precision)                                */
    /* fused time-stepping kernels (stgcn_kernels_tstep.hip.h) */
    int64_t fused_tc2_bwd;            /* 1: LayerNorm/dropout/gate backward + tmp_conv2 weight gradient + transposed conv run as ONE
                                         launch per block; dZ2 stays on chip (ws_dZ2 is only written under stgcn_set_debug_stages)   */
    int64_t ws_W2dense;               /* [KP2][2*c2] W_eff of tmp_conv2, row major (stationary A operand of that kernel)             */
    int64_t fused_tc1_bwd;            /* 1: Align + gate backward + tmp_conv1 weight gradient + transposed conv run as ONE launch (needs
                                         need_dx); dZ1 stays on chip (ws_dZ1 is only written under stgcn_set_debug_stages)           */
    int64_t stored_US2;               /* 1: the forward writes U2 / S2 into `saved` (LayerNorm as a separate pass, the stage-per-launch
                                         backward, or stgcn_set_debug_stages); 0: tc2_bwd_kernel recomputes them from G               */
    int64_t ws_chain;                 /* control words of the chained launches (uint32: 4 header words -- ticket, finished workgroups, sticky
                                         error, spare -- then the per-slab arrival counters), chain_words of them.  Zeroed by the weight-pack
                                         launch that opens every forward / training step and re-armed by the last workgroup of each chained
                                         launch; word 2 != 0 after a call means a bounded wait gave up (the results of that call are void).  */
    int64_t chain_words;
} stgcn_stblock_plan;

int stgcn_version(void);
const char* stgcn_backend(void);    /* "hip-gfx950" for the product library                           */
const char* stgcn_last_error(void); /* thread-local message of the last failing call                  */

int stgcn_stblock_plan_query(const stgcn_stblock_desc* desc, stgcn_stblock_plan* plan);

/* gso: dense (N, N) row-major.  terms = operator terms of the block's graph conv including the identity (ChebGraphConv:
 * Ks, layers.py:147-161; GraphConv: 2).  gso_pad / gso_t_pad: (terms-1)*NP*NP floats each (at least 1), NP = roundup16(N):
 * the Chebyshev polynomials T_1(gso) .. T_{terms-1}(gso) (T_1 = gso, T_k = 2 gso T_{k-1} - T_{k-2}: the recursion the
 * reference applies to the activations, applied once to the constant operator) and their transposes, zero padded and
 * stored in MFMA fragment order (layout: stgcn_kernels_fwd.hip.h, gso_frag_kernel).  scratch: 3*NP*NP floats.        */
int stgcn_gso_prepare(const float* gso, int32_t N, int32_t terms, float* gso_pad, float* gso_t_pad, float* scratch, void* stream);

/* Buffer sizes of stgcn_gso_prepare for a graph of N nodes and `terms` operator terms: gso_pad / gso_t_pad hold `mats`
 * matrices of NP x NP floats each, scratch `scratch_mats` (0: scratch may be NULL).  Slab-resident graph conv (N <= 512):
 * NP = roundup16(N), mats = max(terms-1, 1) fragment-ordered polynomials.  Tiled graph conv (*tiled = 1; N > 512, the
 * 8192-node configs[4] of BASELINE.json): NP = roundup128(N) -- the dense zero-padded operator (gso_pad) and its
 * transpose (gso_t_pad), row major fp32, followed by the same matrix as two bf16 planes (hi, lo = bf16(x - hi)) whose rows
 * are NP + stgcn_set_gc_ld_pad elements apart (mats = 1 + ceil((NP + pad) / NP)); the
 * Chebyshev recursion of layers.py:153-161 then runs on the activations, one GEMM launch per term
 * (stgcn_kernels_gctile.hip.h).                                                                                         */
int stgcn_gso_layout(int32_t N, int32_t terms, int64_t* NP, int64_t* mats, int64_t* scratch_mats, int32_t* tiled);

/* Test knob: 1 = the fused kernels also write the intermediates they normally keep on chip (dZ2 -> ws_dZ2) so that stage tests
 * can compare them with the oracle.  Returns the previous value; other values only query.                                  */
int stgcn_set_debug_stages(int32_t on);

/* Test knob: pretend the device has n compute units (0 = ask the runtime).  The launches sized by the CU count -- the equal-weight
 * (window, node tile, step) ranges of tc1_fwd_kernel / tc1_bwd_kernel, the wave count of tc2_ln_fwd_kernel -- then take their
 * small-device branches (ranges cut inside items, 8-wave workgroups), which the emulator tests exercise that way.  Changes the
 * partial-sum arena of stgcn_stblock_plan_query.  Returns the previous value; n < 0 only queries.                              */
int stgcn_set_tc1_bwd_wgs(int32_t n);

/* When several workgroups share a (b, t) slab of the fused tmp_conv2 + LayerNorm + dropout forward (stgcn_set_tc2ln_peers; the default on
 * launches that would otherwise leave compute units idle), the parts of a slab wait for each other's LayerNorm statistics inside the
 * launch.  A wait is bounded (stgcn_set_chain_spin_ticks); a part whose wait ran out writes NaN outputs for its rows and sets a sticky
 * word in `ws` (1 + the index of the (b, t) slab), which stays until the next weight pack of this module re-arms the control words.
 * This call SYNCHRONISES `stream` and returns that word (0: every wait of the last forward completed).                              */
int stgcn_stblock_chain_status(const stgcn_stblock_desc* desc, const float* ws, uint32_t* sticky, void* stream);

/* Tuning / test knob: workgroups per (b, t) slab of the fused tmp_conv2 + LayerNorm + dropout forward (model/layers.py:254-256).
 * 0 (default) = by the device: 1 when the slabs alone fill the compute units, 2 or 4 when a launch would otherwise leave them idle
 * (the parts of a slab exchange their LayerNorm statistics inside the launch); 1 / 2 / 4 force a form (graphs beyond 384 nodes always
 * run 1).  Returns the previous value; any other n only queries.                                                                 */
int stgcn_set_tc2ln_peers(int32_t n);

/* Tuning / test knob: graphs with at least n nodes use the tiled graph conv (default 513).  Returns the previous value;
 * n < 1 only queries.  Operators prepared under one setting must be used under the same setting.                        */
int stgcn_set_gc_tiled_min_nodes(int32_t n);

/* Arithmetic of the operator products (L X) on the tiled graph-conv path: 0 = fp32 MFMA (exact fp32 products, default: the
 * 1e-4 parity bar of the fp32 configs), 1 = "bf16x3" (operands split into two bf16 each, three bf16 MFMAs per product,
 * fp32 accumulation: fp32-class results at 16/3 of the fp32 MFMA rate), 2 = bf16 operands, fp32 accumulation
 * (BASELINE.json configs[4] is quoted in bf16).  Everything else of the block stays fp32.  Returns the previous mode;
 * a mode outside 0..2 only queries.                                                                                     */
int stgcn_set_gc_precision(int32_t mode);

/* Arithmetic of the operator products T_k(L) X on the slab-resident graph-conv path (graphs up to 512 nodes): 0 = exact fp32 MFMAs,
 * 1 = "bf16x3" (as above: three bf16 MFMAs per product, ~2^-17 relative per product; the 16 x 16 weight contractions, residual, bias
 * and ReLU stay fp32).  The operator's bf16 fragment planes are always prepared (stgcn_gso_prepare), so the mode can change between
 * calls.  Returns the previous mode; a mode outside 0..1 only queries.                                                       */
int stgcn_set_slab_gc_precision(int32_t mode);

/* Matrix products of the BACKWARD kernels of fp32 blocks (dtype STGCN_DTYPE_F32; bf16 blocks are unaffected): 0 = exact fp32 MFMAs
 * (default), 1 = "bf16x3": every operand is split into two bf16 where it enters the matrix cores and a product is formed from three bf16
 * MFMAs with fp32 accumulation (~2^-16 relative per product: inside the 1e-3 gradient bar of the fp32 configurations, but not fp32
 * arithmetic -- an opt-in, reported as such by bench.py; the forward always keeps exact fp32 products).  Returns the previous mode; a mode
 * outside 0..1 only queries.                                                                                                          */
int stgcn_set_bwd_precision(int32_t mode);

/* Test / tuning knob: the column extent of the 256 x (32 * NT) tiles of the big bf16 operator GEMM (gso_gemm_bf16_big_kernel<NT>, taken
 * when the padded node count is a multiple of 256): nt in {4, 5, 6, 8, 10} forces that instance for every launch, 0 restores the
 * grid-rounds heuristic (BASELINE.json configs[4] at bs 16 picks NT = 10 for block 0's 2560 columns and NT = 6 for block 1's 1536;
 * every small test shape would pick NT = 4, so the tests force each instance).  Returns the previous value; other values only query. */
int stgcn_set_gemm_big_nt(int32_t nt);

/* Tuning knob: extra bf16 elements (multiple of 8) between consecutive rows of every 16-bit plane of the tiled graph conv
 * (operator hi / lo planes, activation operand form), so that the rows of a tile do not all start in the same L2 channel
 * when NP is a power of two.  Changes the sizes stgcn_gso_layout / stgcn_stblock_plan_query report: set it before preparing
 * operators and workspaces.  Returns the previous value; an invalid value only queries.                                 */
int stgcn_set_gc_ld_pad(int32_t pad);

/* y: (B, T2, N, c2).  seed/offset select the dropout stream (Philox4x32-10, counter = element/4, the
 * offset is the high 64 counter bits).  offset_dev (nullable) points to a DEVICE uint64 added to `offset` when
 * the kernel runs: a step counter the caller bumps on the stream, so that a captured hipGraph draws a fresh
 * mask on every replay.                                                                                  */
int stgcn_stblock_forward(const stgcn_stblock_desc* desc, const stgcn_stblock_params* params, const float* x,
                          const float* gso_pad, float* y, float* saved, float* ws, uint64_t seed, uint64_t offset,
                          const uint64_t* offset_dev, void* stream);

/* dy: (B, T2, N, c2); y: the output the matching forward wrote; dx: (B, T, N, c_in) or NULL.  `saved`/`ws` must be the buffers the
 * matching forward call filled; seed/offset must be the forward's.
 * BIT-LEVEL CONTRACT ON y (training mode): y must be the forward's output buffer BIT FOR BIT -- the dropout mask travels in the sign of
 * zero.  The forward stores a DROPPED element as -0.0 and a kept element whose LayerNorm output is an exact zero as +0.0, and the backward
 * (and the stgcn_ln_hook epilogues of whichever operator consumes y) reads "dropped iff y is -0.0" instead of regenerating the Philox mask.
 * A numerically equal copy whose -0.0 was canonicalised (y + 0.0, a recomputation, a serialisation round trip through text) makes every
 * dropped element count as kept, without an error.  (bf16 outputs: a kept value below 2^-134 in magnitude rounds to -0.0 and counts as
 * dropped -- its gradient contribution is below 2^-134 as well.)  Callers that cannot guarantee the bits run the library with
 * STGCN_HOOK_MASK=philox in the environment: the mask is then regenerated from seed / offset (about 100 VALU instructions per 4 elements
 * in the consumers' time steps).  y is also read for the LayerNorm-backward row partials when desc.dy_rowstats_ready == 0.            */
int stgcn_stblock_backward(const stgcn_stblock_desc* desc, const stgcn_stblock_params* params, const float* x,
                           const float* gso_t_pad, const float* dy, const float* y, const float* saved, float* ws,
                           const stgcn_stblock_grads* grads, float* dx, uint64_t seed, uint64_t offset,
                           const uint64_t* offset_dev, void* stream);

/* ---- LayerNorm-backward row partials in the PRODUCER of dy.  The backward of `Dropout(LayerNorm(h))` (layers.py:255-256) needs the
 *      per-slab means of g = mask * dy * gamma and g * xhat before it can touch a single element; the per-row sums they are built
 *      from can be formed by whichever kernel produces dy (the next module's input gradient) while the row is still on chip.
 *      With g = mask * dy * gamma and y = mask * (xhat * gamma + beta) (the block's output) the sums need nothing of the forward but y:
 *          sum g = sum mask dy gamma ,   sum g xhat = sum_kept dy (y - beta / (1 - p))
 *      and y is the consumer's own input.  stgcn_stblock_ln_hook describes the LayerNorm of a block's forward call (same desc / params /
 *      y / ws / seed / offset as that call); hand it to the backward call of the module that consumed the block's output
 *      (..._backward_hook below) and set desc.dy_rowstats_ready = 1 in the block's own backward call.  A NULL hook reproduces
 *      stgcn_*_backward.                                                                                                        */
typedef struct stgcn_ln_hook {
    float* rowstat;            /* [B*T2*N][2] destination (inside the block's ws)                                     */
    const float* y;            /* [B*T2*N][c2] the block's output (dtype below), BIT FOR BIT as the forward wrote it: in training
                                  mode the dropout mask is read off it (dropped iff -0.0; see stgcn_stblock_backward)       */
    const float *gamma, *beta; /* tc2_ln.weight / .bias (N, c2)                                                       */
    int32_t N, C, reserved, training;
    float droprate;
    int32_t dtype;             /* STGCN_DTYPE_* of U / S: must equal the dtype of the call the hook is handed to      */
    uint64_t seed, offset;
    const uint64_t* offset_dev;
} stgcn_ln_hook;
int stgcn_stblock_ln_hook(const stgcn_stblock_desc* desc, const stgcn_stblock_params* params, const float* y, float* ws, uint64_t seed,
                          uint64_t offset, const uint64_t* offset_dev, stgcn_ln_hook* hook);
int stgcn_stblock_backward_hook(const stgcn_stblock_desc* desc, const stgcn_stblock_params* params, const float* x,
                                const float* gso_t_pad, const float* dy, const float* y, const float* saved, float* ws,
                                const stgcn_stblock_grads* grads, float* dx, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_dev, const stgcn_ln_hook* dx_hook, void* stream);

/* out[e] = 0 or 1/(1-p): the keep-scale the forward applies to element e of y (n multiple of 4).     */
int stgcn_dropout_mask(float* out, int64_t n, float droprate, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                       void* stream);

/* ---- Output head: OutputBlock(Ko, last_block_channel, channels, end_channel, n_vertex, act_func, bias, droprate)
 *      (layers.py:260-284): tmp_conv1 (Ko taps, c_in -> c0, gated) -> LayerNorm([N, c0]) -> fc1 (c0 -> c1) -> ReLU
 *      -> Dropout -> fc2 (c1 -> end_channel).  stgcn_outblock_forward replaces OutputBlock.forward (layers.py:276-284),
 *      stgcn_outblock_backward what autograd derives from it.  Supported: c0 in {64,128}, c1 = 128, end_channel = 1. */
typedef struct stgcn_outblock_desc {
    int32_t B, T, N;          /* input (B, c_in, T, N) logical; T >= Ko (T == Ko in the reference models)        */
    int32_t c_in, c0, c1, c_end;
    int32_t Ko;
    int32_t act;              /* STGCN_ACT_*                                                                    */
    int32_t training;
    float droprate;
    float ln_eps;
    int32_t need_dx;
    int32_t dtype;            /* STGCN_DTYPE_* (x, dx, saved / ws activations; `out`, `dout`, pred / target stay fp32)   */
    int32_t prepacked;        /* as in stgcn_stblock_desc */
    int32_t defer_reduce;     /* as in stgcn_stblock_desc */
} stgcn_outblock_desc;

/* state_dict keys under "output.": tc_w tmp_conv1.causal_conv.weight (2*c0, c_in, Ko, 1), tc_b .bias,
 * tc_aw/tc_ab tmp_conv1.align.align_conv.{weight,bias} [read iff c_in > c0], ln_w/ln_b tc1_ln.{weight,bias} (N, c0),
 * fc1_w fc1.weight (c1, c0), fc1_b fc1.bias or NULL, fc2_w fc2.weight (1, c1), fc2_b fc2.bias or NULL              */
typedef struct stgcn_outblock_params {
    const float *tc_w, *tc_b, *tc_aw, *tc_ab, *ln_w, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} stgcn_outblock_params;
typedef struct stgcn_outblock_grads {
    float *tc_w, *tc_b, *tc_aw, *tc_ab, *ln_w, *ln_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    float* loss;    /* stgcn_outblock_backward_loss only (else NULL): the MSE value, written by the gradient reduction of the call (or by
                       stgcn_grad_flush when the reduction is deferred)                                                                  */
} stgcn_outblock_grads;

/* nn.MSELoss()(pred, target) fused into the head's backward (main.py:136/167-168): the seed gradient d loss / d pred =
 * 2 (pred - target) / n * grad_scale is formed inside the fc backward kernel instead of being read, and the loss value comes out of the
 * call's gradient reduction -- no separate loss launch.  pred = the (B, T1, N) output the forward of the same call chain wrote
 * (n = B*T1*N values, end_channel = 1); target: n floats, read at target + *target_index_dev * target_index_stride when an index is
 * given (device-side windowing, as in stgcn_mse_loss_grad).                                                                            */
typedef struct stgcn_head_loss {
    const float* pred;
    const float* target;
    const int64_t* target_index_dev;   /* nullable */
    int64_t target_index_stride;
    float grad_scale;                  /* e.g. 1 / world, or the tail-batch weight */
    int32_t reserved;
} stgcn_head_loss;

typedef struct stgcn_outblock_plan {
    int64_t T1, rows, rows_in;        /* T1 = T-Ko+1, rows = B*T1*N, rows_in = B*T*N                             */
    int64_t out_floats;               /* rows * c_end                                                           */
    int64_t saved_floats, ws_floats;
    int64_t sv_U, sv_S;               /* [rows][c0]                                                             */
    int64_t sv_mean, sv_rstd;         /* [B*T1]                                                                 */
    int64_t sv_yln;                   /* [rows][c0] LayerNorm output                                            */
    int64_t sv_hd;                    /* [rows][c1] dropout(relu(fc1))                                          */
    int64_t sv_rowstat;               /* [rows][2]                                                              */
    int64_t ws_Wp, ws_Wd, ws_b, ws_W1p, ws_W1d;
    int64_t ws_rowstat_b, ws_dh1, ws_dyln, ws_dZ, ws_part, part_floats;
    int64_t ws_chain;                 /* control words of the one-launch forward (uint32: 4 header words, then one arrival counter per
                                         LayerNorm slab), chain_words of them.  Zeroed by the weight pack (stgcn_prepack or the forward's
                                         own) and re-armed by the launch that used them.                                              */
    int64_t chain_words;
} stgcn_outblock_plan;

int stgcn_outblock_plan_query(const stgcn_outblock_desc* desc, stgcn_outblock_plan* plan);
/* out: (B, T1, N) (= logical (B, 1, T1, N)); dropout element index = row * c1 / 4 + c / 4                        */
int stgcn_outblock_forward(const stgcn_outblock_desc* desc, const stgcn_outblock_params* params, const float* x, float* out,
                           float* saved, float* ws, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream);
/* The one-launch forward (head_fwd_kernel) lets the row tiles of a window wait for each other inside the launch; a wait is bounded
 * (stgcn_set_chain_spin_ticks) and a tile whose wait ran out writes NaN predictions and sets a sticky word in `ws` (1 + the index of the
 * window counter it waited on), which stays until the next weight pack of this module re-arms the control words.  This call SYNCHRONISES
 * `stream` and returns that word (0: every wait of the last forward completed).  For tests and for callers that want more than the NaN loss. */
int stgcn_outblock_chain_status(const stgcn_outblock_desc* desc, const float* ws, uint32_t* sticky, void* stream);
/* Bound of one in-launch wait in ticks of the device's 100 MHz wall clock (default 200 000 000 = 2 s); ticks == 0 only queries.  Returns
 * the previous value.  Process-global (an atomic word read at launch time): it applies to every later forward on any stream or thread.
 * ticks < 0 is a TEST setting and must never be left in force: the bound is |ticks| and the first tile / workgroup of every launch that
 * exchanges statistics withholds its arrival, so that its peers' waits run out for certain (the NaN / sticky-word path on the device) --
 * every forward made under it produces NaN for those rows until the previous value is restored.
 * The sticky words are zeroed by the weight pack that opens a forward (stgcn_prepack or the forward's own): a status read reports the
 * LAST forward of that module only if it happens before the module's next forward.                                                      */
int64_t stgcn_set_chain_spin_ticks(int64_t ticks);
int stgcn_outblock_backward(const stgcn_outblock_desc* desc, const stgcn_outblock_params* params, const float* x,
                            const float* dout, const float* saved, float* ws, const stgcn_outblock_grads* grads, float* dx,
                            void* stream);
/* as above; dx_hook (nullable): LayerNorm of the module that produced x -- its backward row partials are written while dx is formed */
int stgcn_outblock_backward_hook(const stgcn_outblock_desc* desc, const stgcn_outblock_params* params, const float* x,
                                 const float* dout, const float* saved, float* ws, const stgcn_outblock_grads* grads, float* dx,
                                 const stgcn_ln_hook* dx_hook, void* stream);
/* as stgcn_outblock_backward_hook with the MSE loss fused in place of `dout` (grads->loss receives the loss value) */
int stgcn_outblock_backward_loss(const stgcn_outblock_desc* desc, const stgcn_outblock_params* params, const float* x,
                                 const stgcn_head_loss* loss, const float* saved, float* ws, const stgcn_outblock_grads* grads, float* dx,
                                 const stgcn_ln_hook* dx_hook, void* stream);

/* ---- Whole-model weight pack: the per-call pack launches of all ST blocks and of the head in ONE launch at the start of a
 *      training / inference step (the parameters only change in optimizer.step(), main.py:169).  Every forward whose desc
 *      has prepacked = 1 then skips its own pack; `ws` must be the buffers later handed to those forward / backward calls.
 *      head_desc may be NULL (no fused head).  Packs the backward-data operands regardless of need_dx.
 *      counters (nullable, <= 4): device-side int64 step counters advanced by this first launch of the step,
 *      *ptr = (*ptr + inc) % mod (mod 0: no wrap) -- the dropout stream position, the optimizer's step count (main.py:169) and the
 *      like ride on the pack launch instead of costing one tiny launch each inside a captured step.
 *      stgcn_prepack_park (round 6; same arguments): for callers whose NEXT call into the library is the forward of the model's first block
 *      (stgcn_amd/models.py).  It PARKS the job list (per host thread) instead of launching it: the next entry point on that thread launches
 *      it first -- except the forward of a block whose first layer is the thin one (Kt * c_in <= 4: STGCN's first block) on the same stream
 *      with prepacked = 1, which sends pack and layer out as ONE launch.  The parked list holds the pointers of the call: parameters and
 *      workspaces must stay alive until that next entry point (they are the modules' own buffers).  stgcn_prepack itself launches at once. */
typedef struct stgcn_prepack_block {
    const stgcn_stblock_desc* desc;
    const stgcn_stblock_params* params;
    float* ws;
} stgcn_prepack_block;
typedef struct stgcn_step_counter {
    int64_t* ptr;
    int64_t inc, mod;
} stgcn_step_counter;
int stgcn_prepack(int32_t n_blocks, const stgcn_prepack_block* blocks, const stgcn_outblock_desc* head_desc,
                  const stgcn_outblock_params* head_params, float* head_ws, int32_t n_counters, const stgcn_step_counter* counters,
                  void* stream);
int stgcn_prepack_park(int32_t n_blocks, const stgcn_prepack_block* blocks, const stgcn_outblock_desc* head_desc,
                  const stgcn_outblock_params* head_params, float* head_ws, int32_t n_counters, const stgcn_step_counter* counters,
                  void* stream);
/* Launches a job list that stgcn_prepack_park left parked on the calling thread (no-op otherwise): for callers whose first-block forward may not
 * happen after all (an exception between the two calls) -- the parked list holds pointers that must not outlive their tensors.            */
int stgcn_prepack_flush(void);

/* ---- Optimizer step: torch.optim.AdamW(lr, weight_decay) as main.py:148 configures it (betas (0.9, 0.999), eps 1e-8,
 *      amsgrad False), applied by optimizer.step() at main.py:169.  `tensors` is a HOST array of `count` entries (device
 *      pointers inside); only parameters that received a gradient are listed (the reference skips grad-None tensors).
 *      step is the 1-based update count; step_dev / lr_dev (nullable DEVICE scalars) override step / lr at run time so a
 *      captured hipGraph keeps counting and follows the LR schedule.                                                      */
typedef struct stgcn_adamw_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} stgcn_adamw_tensor;
int stgcn_adamw_step(const stgcn_adamw_tensor* tensors, int32_t count, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int64_t step, const int64_t* step_dev, const float* lr_dev, void* stream);

/* ---- Whole-model gradient flush: the final reductions of every backward call of a step whose desc had defer_reduce = 1
 *      (same desc / grads / ws as those calls) in ONE launch, optionally with optimizer.step() (main.py:169) applied to each
 *      gradient element as it is produced (opt/opt_count/hyper as for stgcn_adamw_step; every opt entry's `grad` must be one
 *      of the gradient buffers of this flush; opt == NULL: reduce only, e.g. before a data-parallel all-reduce).          */
typedef struct stgcn_flush_block {
    const stgcn_stblock_desc* desc;
    const stgcn_stblock_grads* grads;
    float* ws;
} stgcn_flush_block;
typedef struct stgcn_adamw_hyper {
    float lr, beta1, beta2, eps, weight_decay;
    int64_t step;
    const int64_t* step_dev;
    const float* lr_dev;
} stgcn_adamw_hyper;
int stgcn_grad_flush(int32_t n_blocks, const stgcn_flush_block* blocks, const stgcn_outblock_desc* head_desc,
                     const stgcn_outblock_grads* head_grads, float* head_ws, const stgcn_adamw_tensor* opt, int32_t opt_count,
                     const stgcn_adamw_hyper* hyper, void* stream);

/* ---- Loss: nn.MSELoss() as main.py:136 builds it (mean over all n = B*N elements) together with the gradient that
 *      l.backward() (main.py:168) feeds into the model output, in one launch:
 *          loss[0] = mean((pred - target)^2) ;  dpred[i] = 2 (pred[i] - target[i]) * grad_scale / n
 *      grad_scale = 1 for the reference's loop (1/k for a minibatch processed as k micro-batches).  target is read at
 *      target + *target_index_dev * target_index_stride floats (nullable: labels taken from the resident series).        */
int stgcn_mse_loss_grad(const float* pred, const float* target, int64_t n, float grad_scale, float* loss, float* dpred,
                        const int64_t* target_index_dev, int64_t target_index_stride, void* stream);

/* Built-in kernel timer (no reference counterpart; feeds bench.py's roofline object).  While enabled,
 * every kernel launch is bracketed by a hipEvent pair on the launch stream.  collect() synchronises on the
 * recorded events, writes {"<kernel label>@<tag>": {"calls": n, "total_ms": t}, ...} as JSON and resets.      */
int stgcn_profile_enable(int on);
int stgcn_profile_collect(char* json_buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* STGCN_HIP_H */
