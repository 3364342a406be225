// C-ABI of libstgcn_hip.so: argument checking, buffer planning and kernel launches.
// Declarations and the reference call sites each entry replaces: include/stgcn_hip.h.
#include "../../include/stgcn_hip.h"

#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include "stgcn_kernels_bwd.hip.h"
#include "stgcn_kernels_fwd.hip.h"
#include "stgcn_kernels_gctile.hip.h"
#include "stgcn_kernels_gcslab16.hip.h"
#include "stgcn_kernels_head.hip.h"

using namespace stgcn;

namespace {
thread_local char g_err[512] = "";
// storage / arithmetic type of the activations of the call being served (set by every entry point from its descriptor)
thread_local bool g_bf16 = false;
// Default OFF: measured on MI355X (profiles/r4-02_chain_ab.txt) the chained forward of C2's second block is SLOWER than three launches
// (tmp_conv1 + graph conv: 67.4 us against 28.6 + 14.7; all three stages: 75.7 against 61.7): a launch has ONE register and LDS budget, the
// widest role's (tc1_fwd: 133 VGPRs, 3 waves per SIMD), and a workgroup is only dispatched when the whole block -- surplus waves included --
// fits, so a graph-conv workgroup (73 VGPRs, ~5 per CU on its own) cannot start beside a tmp_conv1 workgroup at all and runs one per CU
// afterwards.  The protocol itself is sound (0 wrong words in 600 launches of tools/ubench/chain_probe.hip, all GPU tests green with it).
constexpr int kChainDefault = 0;   // (see fwd_chain_mode)
std::atomic<long long> g_chain_spin_ticks{kChainSpinTicks};   // bound of one in-launch wait (stgcn_set_chain_spin_ticks; read at launch time by any thread)
int g_gemm_big_nt = 0;   // stgcn_set_gemm_big_nt: forced column extent of the big bf16 operator GEMM's tiles (0 = heuristic)

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Where the backward takes the dropout mask of a block's LayerNorm from (tc2_bwd_kernel, the stgcn_ln_hook epilogues): read off the block
// output y -- the forward stores a dropped element as -0.0 and a kept exact zero as +0.0 (drop_encode, stgcn_device.hip.h), so the test is
// exact (round 4's "kept iff y != 0" lost a kept zero: ADVICE r4) -- or regenerated (Philox: ~100 VALU instructions per 4 elements in the
// consumers' time steps; measured on C2 fp32, pass r5-01: tc2_bwd 29.1 + 21.1 -> 28.3 + 20.2 us, tc1_bwd 61.7 -> 60.1, step 0.3756 -> 0.3679 ms).
// STGCN_HOOK_MASK=philox / y forces one (tests run both forms of both types).
inline bool hook_mask_from_y() {   // (the default is the same for fp32 and bf16 activations since round 5)
    const char* e = getenv("STGCN_HOOK_MASK");
    if (e && e[0] == 'p') return false;
    return true;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline int64_t rup(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- optional per-kernel timing: hipEvent pairs around every launch (off by default) ---------------
struct ProfSlot { const char* name; int tag; hipEvent_t e0, e1; };
int g_prof_tag = 0;   // caller-supplied tag (desc->reserved: e.g. the block index) appended to the kernel label
constexpr int kProfMax = 8192;
ProfSlot* g_prof = nullptr;
int g_prof_n = 0, g_prof_on = 0, g_prof_cap = 0;

inline int prof_begin(const char* name, hipStream_t st) {
    if (!g_prof_on) return -1;
    if (g_prof_n >= g_prof_cap) {
        if (g_prof_cap >= kProfMax) return -1;
        if (!g_prof) g_prof = (ProfSlot*)calloc(kProfMax, sizeof(ProfSlot));
        const int grow = g_prof_cap + 256 > kProfMax ? kProfMax : g_prof_cap + 256;
        for (int i = g_prof_cap; i < grow; ++i) {
            (void)hipEventCreate(&g_prof[i].e0);
            (void)hipEventCreate(&g_prof[i].e1);
        }
        g_prof_cap = grow;
    }
    const int i = g_prof_n++;
    g_prof[i].name = name;
    g_prof[i].tag = g_prof_tag;
    (void)hipEventRecord(g_prof[i].e0, st);
    return i;
}
inline void prof_end(int i, hipStream_t st) {
    if (i >= 0) (void)hipEventRecord(g_prof[i].e1, st);
}

// STGCN_LAUNCH_LOG=<file>: one line "label@tag <kernel> <workgroups> <threads>" per launch, so that an external profile
// (rocprofv3 counters are keyed by kernel symbol + grid) can be joined with the library's labels (tools/pmc_traffic.py)
inline void launch_log(const char* label, const char* kernel, dim3 grid, dim3 block) {
#ifdef STGCN_EMU_RACE
    emu::g.kname = kernel;   // (LDS race-check build of the CPU emulator: names the launch in its reports)
#endif
    static FILE* f = getenv("STGCN_LAUNCH_LOG") ? fopen(getenv("STGCN_LAUNCH_LOG"), "w") : nullptr;
    if (f) {
        fprintf(f, "%s@%d\t%s\t%u\t%u\n", label, g_prof_tag, kernel, grid.x * grid.y * grid.z, block.x * block.y * block.z);
        fflush(f);
    }
}

// every kernel launch of the library goes through this macro
#define STGCN_LAUNCH(label, st, kernel, grid, block, lds, ...)                                    \
    do {                                                                                          \
        launch_log(label, #kernel, grid, block);                                                  \
        const int pi_ = prof_begin(label, st);                                                    \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                            \
        prof_end(pi_, st);                                                                        \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(STGCN_ERR_LAUNCH, "%s: %s", label, hipGetErrorString(e_)); \
    } while (0)

// launch a kernel whose template arguments mention ET, instantiated for the activation type of the call (g_bf16)
#define STGCN_LAUNCH_ET(label, st, kernel, grid, block, lds, ...)                                 \
    do {                                                                                          \
        if (g_bf16) {                                                                             \
            using ET = bf16;                                                                      \
            STGCN_LAUNCH(label, st, kernel, grid, block, lds, __VA_ARGS__);                       \
        } else {                                                                                  \
            using ET = float;                                                                     \
            STGCN_LAUNCH(label, st, kernel, grid, block, lds, __VA_ARGS__);                       \
        }                                                                                         \
    } while (0)
// the same for BACKWARD kernels: fp32 blocks take ET = f32x (bf16x3 products) when stgcn_set_bwd_precision(1) is in force
#define STGCN_LAUNCH_ETB(label, st, kernel, grid, block, lds, ...)                                \
    do {                                                                                          \
        if (g_bf16) {                                                                             \
            using ET = bf16;                                                                      \
            STGCN_LAUNCH(label, st, kernel, grid, block, lds, __VA_ARGS__);                       \
        } else if (g_bwd_precision == 1) {                                                        \
            using ET = f32x;                                                                      \
            STGCN_LAUNCH(label, st, kernel, grid, block, lds, __VA_ARGS__);                       \
        } else {                                                                                  \
            using ET = float;                                                                     \
            STGCN_LAUNCH(label, st, kernel, grid, block, lds, __VA_ARGS__);                       \
        }                                                                                         \
    } while (0)
#define STGCN_ETB_VALUE(expr) (g_bf16 ? [&] { using ET = bf16; return (expr); }() : g_bwd_precision == 1 ? [&] { using ET = f32x; return (expr); }() : [&] { using ET = float; return (expr); }())
// value of an expression that mentions ET (e.g. wg_capacity of a kernel instantiation)
#define STGCN_ET_VALUE(expr) (g_bf16 ? [&] { using ET = bf16; return (expr); }() : [&] { using ET = float; return (expr); }())
// stage-per-launch kernels of round 1 that have no bf16 variant (the bf16 configurations run the fused paths)
#define STGCN_F32_ONLY(what)                                                                      \
    do {                                                                                          \
        if (g_bf16) return fail(STGCN_ERR_UNSUPPORTED, "%s: no bf16 variant (bf16 blocks need the fused time-stepping kernels)", what); \
    } while (0)

// ---- side stream ----------------------------------------------------------------------------------------------------
// The weight-gradient kernels of a backward call depend only on dZ, not on the data-gradient chain that follows it, and
// both are latency-bound launches of ~1 workgroup per CU, so they could run beside each other.  side_fork(st) returns a
// stream that starts after everything enqueued on st so far; side_join(st) makes st wait for it (always called before the
// entry point returns, so callers and hipGraph capture see one stream).  OFF by default: on MI355X the cross-queue
// dependencies of the 6 fork/join pairs per step cost more than the overlap returns at the C2 size (0.650 vs 0.577 ms per
// step under hipGraph replay, profiles/r19_r33_experiments.md); STGCN_SIDE_STREAM=1 enables it for larger problems.
struct SideStream { hipStream_t s; hipEvent_t fork, join; int state; };   // state: 0 not initialised, 1 on, -1 off
SideStream g_side = {nullptr, nullptr, nullptr, 0};
hipStream_t side_fork(hipStream_t st) {
    if (g_side.state == 0) {
#ifdef STGCN_EXPERIMENTS   // (round 5: measured slower on this stack -- cross-queue dependencies, r19-r33 / r3-37 -- and out of the product build)
        const char* e = getenv("STGCN_SIDE_STREAM");
#else
        const char* e = nullptr;
#endif
        g_side.state = -1;
        if (e && atoi(e) == 1 && hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming) == hipSuccess)
            g_side.state = 1;
    }
    if (g_side.state < 0) return st;
    if (hipEventRecord(g_side.fork, st) != hipSuccess || hipStreamWaitEvent(g_side.s, g_side.fork, 0) != hipSuccess) return st;
    return g_side.s;
}
void side_join(hipStream_t st, hipStream_t sd) {
    if (sd == st) return;
    if (hipEventRecord(g_side.join, sd) == hipSuccess) (void)hipStreamWaitEvent(st, g_side.join, 0);
}

// A second, deferred form for the head's weight-gradient launch in a fused training step (desc.defer_reduce): nothing before the step's
// stgcn_grad_flush reads its partials, so it may run beside the WHOLE backward of the ST blocks -- one fork / join pair per step
// instead of six.  defer_fork(st) returns the stream to launch on (st itself when off); the join happens in stgcn_grad_flush (or in the next
// defer_fork).  STGCN_SIDE_WGRAD=0/1 forces (default: see defer_enabled).
struct DeferStream { hipStream_t s; hipEvent_t fork, join; int state; int pending; };
DeferStream g_defer = {nullptr, nullptr, nullptr, 0, 0};
void defer_join(hipStream_t st) {
    if (!g_defer.pending) return;
    g_defer.pending = 0;
    if (hipEventRecord(g_defer.join, g_defer.s) == hipSuccess) (void)hipStreamWaitEvent(st, g_defer.join, 0);
}
hipStream_t defer_fork(hipStream_t st) {
    if (g_defer.state == 0) {
#ifdef STGCN_EXPERIMENTS
        const char* e = getenv("STGCN_SIDE_WGRAD");
#else
        const char* e = nullptr;
#endif
        g_defer.state = -1;
        if (e && atoi(e) == 1 && hipStreamCreateWithFlags(&g_defer.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&g_defer.fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&g_defer.join, hipEventDisableTiming) == hipSuccess)
            g_defer.state = 1;
    }
    if (g_defer.state < 0) return st;
    defer_join(st);   // (a launch of an earlier step that nobody flushed)
    if (hipEventRecord(g_defer.fork, st) != hipSuccess || hipStreamWaitEvent(g_defer.s, g_defer.fork, 0) != hipSuccess) return st;
    g_defer.pending = 1;
    return g_defer.s;
}

#define STGCN_CHECK_LAUNCH(name)                                                                  \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(STGCN_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

int check_desc(const stgcn_stblock_desc* d) {
    if (!d) return fail(STGCN_ERR_INVALID, "desc is NULL");
    if (d->B < 1 || d->T < 1 || d->N < 1 || d->c_in < 1) return fail(STGCN_ERR_INVALID, "B/T/N/c_in must be positive");
    if (d->Kt < 1) return fail(STGCN_ERR_INVALID, "Kt must be >= 1");
    if (d->graph_conv == STGCN_GC_CHEB && d->Ks < 1)
        return fail(STGCN_ERR_INVALID, "ERROR: the graph convolution kernel size Ks has to be a positive integer, but received %d.", d->Ks);
    if (d->act != STGCN_ACT_GLU && d->act != STGCN_ACT_GTU) return fail(STGCN_ERR_INVALID, "act must be glu or gtu");
    if (d->graph_conv != STGCN_GC_CHEB && d->graph_conv != STGCN_GC_KIPF) return fail(STGCN_ERR_INVALID, "bad graph_conv enum");
    if (d->T - 2 * (d->Kt - 1) < 1) return fail(STGCN_ERR_INVALID, "T=%d too short for two Kt=%d temporal convs", d->T, d->Kt);
    if (d->droprate < 0.f || d->droprate >= 1.f) return fail(STGCN_ERR_INVALID, "droprate must be in [0, 1)");
    if (!(d->c0 == 64 || d->c0 == 128) || !(d->c2 == 64 || d->c2 == 128))
        return fail(STGCN_ERR_UNSUPPORTED, "temporal-conv output channels must be 64 or 128 (got c0=%d c2=%d)", d->c0, d->c2);
    if (d->c1 != 16) return fail(STGCN_ERR_UNSUPPORTED, "graph-conv channels c1 must be 16 (got %d)", d->c1);
    if (d->N > 32768) return fail(STGCN_ERR_UNSUPPORTED, "N=%d > 32768 nodes", d->N);
    if ((int64_t)d->B * d->T * d->N >= (1ll << 31) / 256) return fail(STGCN_ERR_UNSUPPORTED, "B*T*N too large for 32-bit row indexing");
    if ((int64_t)d->B * (d->T - d->Kt + 1) > 65535)   // per-slab launches index the (b, t) slab with blockIdx.y
        return fail(STGCN_ERR_UNSUPPORTED, "B*(T-Kt+1) = %lld (b, t) slabs exceed the 65535 rows of a launch grid: split the batch",
                    (long long)d->B * (d->T - d->Kt + 1));
    if (d->graph_conv == STGCN_GC_CHEB && d->Ks > 8) return fail(STGCN_ERR_UNSUPPORTED, "Ks=%d > 8", d->Ks);
    if ((d->c_in & 3) != 0 && d->Kt * d->c_in > 16)
        return fail(STGCN_ERR_UNSUPPORTED, "c_in=%d: input channels must be a multiple of 4 unless Kt*c_in <= 16", d->c_in);
    if (d->dtype != STGCN_DTYPE_F32 && d->dtype != STGCN_DTYPE_BF16) return fail(STGCN_ERR_INVALID, "dtype must be STGCN_DTYPE_F32 or STGCN_DTYPE_BF16");
    if (d->x_bstride < 0 || ((d->x_bstride != 0 || d->x_index_dev) && d->need_dx))
        return fail(STGCN_ERR_INVALID, "strided / indexed input windows (x_bstride, x_index_dev) need need_dx = 0 and x_bstride >= 0");
    return STGCN_OK;
}

// workgroups (4 waves) of the forward: two 16-row tiles per wave, at most four workgroups per CU
inline int thin_fwd_wgs(int64_t rows) {
    static const int tpw_env = STGCN_EXP_ENV("STGCN_THIN_FWD_TPW") ? atoi(STGCN_EXP_ENV("STGCN_THIN_FWD_TPW")) : 2;   // tiles per wave (sweep knob, experiments build)
    const int tpw = tpw_env < 1 ? 1 : tpw_env;   // (0 or a non-numeric value used to divide by zero)
    const int64_t tiles = (rows + 15) / 16, want = (tiles + 4 * tpw - 1) / (4 * tpw), cap = 4L * device_cus();
    return (int)(want < 1 ? 1 : want < cap ? want : cap);
}

inline int terms(const stgcn_stblock_desc* d) { return d->graph_conv == STGCN_GC_KIPF ? 2 : d->Ks; }

struct Derived {
    int T1, T2, NP, KP1, KP2, NC1, NC2, CP_in, CP1, terms, tiled;
    int64_t rows0, rows1, rows2, slabs1, slabs2;
};
Derived derive(const stgcn_stblock_desc* d) {
    Derived v;
    v.T1 = d->T - d->Kt + 1;
    v.T2 = v.T1 - d->Kt + 1;
    v.terms = terms(d);
    v.tiled = gc_is_tiled(d->N, v.terms) ? 1 : 0;
    v.NP = gc_padded_nodes(d->N, v.terms);
    v.KP1 = (int)rup((int64_t)d->Kt * d->c_in, 16);
    v.KP2 = (int)rup((int64_t)d->Kt * d->c1, 16);
    v.NC1 = 2 * d->c0;
    v.NC2 = 2 * d->c2;
    v.CP_in = (int)rup(d->c_in, 16);
    v.CP1 = (int)rup(d->c1, 16);
    v.rows0 = (int64_t)d->B * d->T * d->N;
    v.rows1 = (int64_t)d->B * v.T1 * d->N;
    v.rows2 = (int64_t)d->B * v.T2 * d->N;
    v.slabs1 = (int64_t)d->B * v.T1;
    v.slabs2 = (int64_t)d->B * v.T2;
    return v;
}

uint32_t drop_thresh(float p) {
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    if (t < 0.0) t = 0.0;
    return (uint32_t)t;
}
// ---- residency: how many workgroups of a kernel the whole device holds at once ----------------------------------
// The problem sizes of this path give every launch only a few workgroups per CU and all of them start together, so a
// launch costs (rounds of resident workgroups) x (latency chain of one workgroup): 2070 tiles on 2048 slots run as
// long as 4096 would.  Launchers therefore pick the tile size / kernel variant whose grid fits ONE round.
template <typename K>
int wg_capacity(K kernel, int threads, size_t lds) {
    struct Entry { const void* k; int threads; size_t lds; int cap; };
    static Entry cache[64];
    static int n = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < n; ++i)
        if (cache[i].k == key && cache[i].threads == threads && cache[i].lds == lds) return cache[i].cap;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    int nb = 0;
    // a block the kernel cannot be launched with at all (more threads than its __launch_bounds__, more LDS than a CU has) gets capacity 0:
    // the launch heuristics then never pick that geometry (a failed query used to read as "one workgroup per CU")
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds) != hipSuccess || nb <= 0) {
        (void)hipGetLastError();
        nb = 0;
    }
    const int cap = nb * cus;
    if (n < 64) cache[n++] = Entry{key, threads, lds, cap};
    return cap;
}
inline int rounds_of(long wgs, int cap) { return cap <= 0 ? (1 << 30) : (int)((wgs + cap - 1) / cap); }

// pack jobs: appended to one PackArgs so that several modules can share a launch (stgcn_prepack)
struct PackList {
    PackArgs pa;
    int nj;
    PackList() : nj(0) { memset(&pa, 0, sizeof(pa)); }
    bool add(int kind, int n, float* dst, const float* w, const float* b, const float* aw, const float* ab, int Cin, int Cout, int Kt,
             int KCH) {
        if (nj >= kMaxPackJobs) return false;
        PackJob& j = pa.job[nj];
        j.kind = kind; j.n = n; j.dst = dst; j.w = w; j.b = b; j.aw = aw; j.ab = ab;
        j.Cin = Cin; j.Cout = Cout; j.Kt = Kt; j.KCH = KCH; j.gated = 1;
        pa.start[nj + 1] = pa.start[nj] + cdiv(n, kThreads);
        ++nj;
        return true;
    }
};
bool pack_fill_block(PackList& L, const stgcn_stblock_desc* d, const stgcn_stblock_params* P, const stgcn_stblock_plan& pl, float* ws,
                     bool all) {
    const Derived v = derive(d);
    bool ok = true;
    ok &= L.add(PK_TCONV_FWD, v.NC1 * v.KP1, ws + pl.ws_W1p, P->tc1_w, P->tc1_b, P->tc1_aw, P->tc1_ab, d->c_in, d->c0, d->Kt, v.KP1 / 16);
    if (d->need_dx || all)
        ok &= L.add(PK_TCONV_BWD, d->Kt * v.NC1 * v.CP_in, ws + pl.ws_W1d, P->tc1_w, P->tc1_b, P->tc1_aw, P->tc1_ab, d->c_in, d->c0, d->Kt,
                    d->Kt * v.NC1 / 16);
    ok &= L.add(PK_TCONV_BIAS, v.NC1, ws + pl.ws_b1, P->tc1_w, P->tc1_b, P->tc1_aw, P->tc1_ab, d->c_in, d->c0, d->Kt, 0);
    ok &= L.add(PK_ALIGN_FWD, d->c0 * d->c1, ws + pl.ws_Wap, P->al_w, P->al_b, nullptr, nullptr, d->c0, d->c1, 1, d->c0 / 16);
    ok &= L.add(PK_ALIGN_BWD, v.CP1 * d->c0, ws + pl.ws_WaT, P->al_w, P->al_b, nullptr, nullptr, d->c0, d->c1, 1, v.CP1 / 16);
    ok &= L.add(PK_ALIGN_BIAS, d->c1, ws + pl.ws_ba, P->al_w, P->al_b, nullptr, nullptr, d->c0, d->c1, 1, 0);
    ok &= L.add(PK_TCONV_FWD, v.NC2 * v.KP2, ws + pl.ws_W2p, P->tc2_w, P->tc2_b, P->tc2_aw, P->tc2_ab, d->c1, d->c2, d->Kt, v.KP2 / 16);
    ok &= L.add(PK_TCONV_BWD, d->Kt * v.NC2 * v.CP1, ws + pl.ws_W2d, P->tc2_w, P->tc2_b, P->tc2_aw, P->tc2_ab, d->c1, d->c2, d->Kt,
                d->Kt * v.NC2 / 16);
    ok &= L.add(PK_TCONV_BIAS, v.NC2, ws + pl.ws_b2, P->tc2_w, P->tc2_b, P->tc2_aw, P->tc2_ab, d->c1, d->c2, d->Kt, 0);
    const bool k3s = (tc1_bwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt) || tc1_fwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt)) && (d->Kt * d->c_in > 4);
    if (pl.recompute_tc1 || k3s)
        ok &= L.add(PK_TCONV_DENSE, v.KP1 * v.NC1, ws + pl.ws_W1dense, P->tc1_w, P->tc1_b, P->tc1_aw, P->tc1_ab, d->c_in, d->c0, d->Kt, 0);
    if (k3s && !pl.thin_tc1)
        ok &= L.add(PK_ALIGN_DENSE, d->c0 * d->c1, ws + pl.ws_WaDense, P->al_w, P->al_b, nullptr, nullptr, d->c0, d->c1, 1, 0);
    if (pl.fused_tc2_bwd)
        ok &= L.add(PK_TCONV_DENSE, v.KP2 * v.NC2, ws + pl.ws_W2dense, P->tc2_w, P->tc2_b, P->tc2_aw, P->tc2_ab, d->c1, d->c2, d->Kt, 0);
    ok &= L.add(PK_ZERO, (int)pl.chain_words, ws + pl.ws_chain, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0);   // control words of the chained launches
    return ok;
}
int launch_pack_list(const char* label, PackList& L, hipStream_t st) {
    if (L.nj == 0) return STGCN_OK;
    L.pa.njobs = L.nj;
    STGCN_LAUNCH(label, st, pack_kernel, dim3(L.pa.start[L.nj]), dim3(kThreads), 0, L.pa);
    return STGCN_OK;
}
int launch_pack(const stgcn_stblock_desc* d, const stgcn_stblock_params* P, const stgcn_stblock_plan& pl, float* ws,
                       hipStream_t st) {
    PackList L;
    if (!pack_fill_block(L, d, P, pl, ws, false)) return fail(STGCN_ERR_INVALID, "pack job table overflow");
    return launch_pack_list("pack", L, st);
}

// ---- the model's weight pack rides on the first launch that follows it (round 6) ------------------------------------------------------
// stgcn_prepack does not launch its (last) job list at once: it parks it here, and the next entry point decides.  If that is the forward of a
// block whose first layer is the thin one (K = Kt * c_in <= 4: STGCN's first block), pack and layer go out as ONE launch
// (pack_thin_fwd_kernel: 6.4 + 8.0 us and a kernel boundary -> ~9 us at C2); every other entry point launches the parked list first, as
// stgcn_prepack used to (flush_pending_pack: called at the top of every entry that reads a workspace).  Per host thread.
struct PendingPack {
    bool valid = false;
    PackList L;
    hipStream_t st = nullptr;
};
thread_local PendingPack g_pending_pack;
// "bf16x6" product form of the fp32 blocks (Frag3, stgcn_device.hip.h): fp32-accurate products on the bf16 matrix pipe, operands split once at
// their producer -- the default of tc1_fwd / tc2_ln_fwd / tc1_bwd since round 6 (measured error against the fp64 stage oracle equal to the
// fp32-MFMA path's, profiles/r6-29_x6_errors.txt).  STGCN_MFMA_X6=0 selects v_mfma_f32_16x16x4_f32 for every product (read per call: bench.py
// reports that step as config.secondary_fp32_mfma, the stage tests run both).
inline bool mfma_x6() {
    const char* e = getenv("STGCN_MFMA_X6");
    return !(e && e[0] == '0');
}
inline bool pack_fusion_on() {
    static const int off = STGCN_EXP_ENV("STGCN_PACK_FUSE") ? atoi(STGCN_EXP_ENV("STGCN_PACK_FUSE")) == 0 : 0;   // (A/B knob, experiments build)
    return !off;
}
int flush_pending_pack() {
    if (!g_pending_pack.valid) return STGCN_OK;
    g_pending_pack.valid = false;
    const int tag = g_prof_tag;
    g_prof_tag = 0;
    const int rc = launch_pack_list("prepack", g_pending_pack.L, g_pending_pack.st);
    g_prof_tag = tag;
    return rc;
}
#define STGCN_FLUSH_PENDING_PACK()                     \
    do {                                               \
        const int rc_pp_ = flush_pending_pack();       \
        if (rc_pp_) return rc_pp_;                     \
    } while (0)

int launch_ln_fwd(const char* label, LnFwdArgs ln, int64_t slabs, hipStream_t st) {
    const int n4 = ln.n / 4;
    ln.per = 1024;                                   // float4 columns per workgroup (4 per thread)
    const dim3 grid(cdiv(n4, ln.per), (unsigned)slabs), blk(kThreads);
    ln.stats_ready = 0;
    static const int min_chunks = getenv("STGCN_LN_STATS_MIN_CHUNKS") ? atoi(getenv("STGCN_LN_STATS_MIN_CHUNKS")) : 8;   // (test knob)
    if ((int)grid.x >= min_chunks) {   // many chunks per slab: the statistics once per slab instead of once per chunk
        STGCN_LAUNCH("ln_slab_stats", st, ln_slab_stats_kernel, dim3((unsigned)slabs), blk, 64, ln);
        ln.stats_ready = 1;
    }
    STGCN_LAUNCH_ET(label, st, (ln_norm_kernel<ET>), grid, blk, 64, ln);
    return STGCN_OK;
}

template <int NT>
int launch_tconv_fwd_nt(const char* label, const TconvFwdArgs& a, hipStream_t st) {
    // candidates: 16 / 32 / 48-row tiles with 4 waves, 64-row tiles with 8 waves.  Smallest tile (most workgroups) whose
    // grid is resident in one round; otherwise the fewest rounds.  STGCN_TCONV_TR=<rows> forces a tile (tuning knob).
    static const int force_tr = getenv("STGCN_TCONV_TR") ? atoi(getenv("STGCN_TCONV_TR")) : 0;
    const int trs[4] = {16, 32, 48, 64};
    int best = -1, best_rounds = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        const int tr = trs[i];
        const size_t lds = (size_t)tile_lds_floats(2 * a.Cout, tr) * sizeof(float);
        int cap;
        switch (tr) {
            case 16: cap = STGCN_ET_VALUE(wg_capacity(tconv_fwd_kernel<NT, 1, 4, ET>, 256, lds)); break;
            case 32: cap = STGCN_ET_VALUE(wg_capacity(tconv_fwd_kernel<NT, 2, 4, ET>, 256, lds)); break;
            case 48: cap = STGCN_ET_VALUE(wg_capacity(tconv_fwd_kernel<NT, 3, 4, ET>, 256, lds)); break;
            default: cap = STGCN_ET_VALUE(wg_capacity(tconv_fwd_kernel<NT, 4, 8, ET>, 512, lds)); break;
        }
        const int r = rounds_of(cdiv(a.ts.rows, tr), cap);
        if (force_tr ? tr == force_tr : r < best_rounds) {
            best = tr;
            best_rounds = r;
            if (force_tr) break;
        }
    }
    if (best < 0) best = 32;
    const dim3 grid(cdiv(a.ts.rows, best));
    const size_t lds = (size_t)tile_lds_floats(2 * a.Cout, best) * sizeof(float);
    switch (best) {
        case 16: STGCN_LAUNCH_ET(label, st, (tconv_fwd_kernel<NT, 1, 4, ET>), grid, dim3(256), lds, a); break;
        case 32: STGCN_LAUNCH_ET(label, st, (tconv_fwd_kernel<NT, 2, 4, ET>), grid, dim3(256), lds, a); break;
        case 48: STGCN_LAUNCH_ET(label, st, (tconv_fwd_kernel<NT, 3, 4, ET>), grid, dim3(256), lds, a); break;
        default: STGCN_LAUNCH_ET(label, st, (tconv_fwd_kernel<NT, 4, 8, ET>), grid, dim3(512), lds, a); break;
    }
    return STGCN_OK;
}
#ifdef STGCN_EXPERIMENTS
// v3 kernels (time-complete tiles: one workgroup = 16 nodes x all time steps of one window, weights of the whole K in registers)
template <int WAVES, int NT, int KCW>
int launch_tconv_fwd3(const char* label, const TconvFwdArgs& a, hipStream_t st) {
    constexpr int MG = 2;
    const int node_tiles = (a.ts.N + 15) / 16;
    const long B = a.ts.rows / ((long)a.ts.Tdst * a.ts.N);
    const size_t lds = tconv3_lds_bytes(a.ts.Tsrc, a.ts.C, 16 * WAVES * NT, MG);
    STGCN_LAUNCH(label, st, (tconv_fwd3_kernel<WAVES, NT, KCW, MG>), dim3((unsigned)(B * node_tiles)), dim3(WAVES * 64), lds, a, node_tiles);
    return STGCN_OK;
}
#endif
// v4: 32-row x 256-column tiles with streamed, double-buffered weight rounds (the output head; see tconv_fwd4_kernel)
inline bool tconv4_ok(const TconvFwdArgs& a) {
    static const int off = getenv("STGCN_TCONV4") ? atoi(getenv("STGCN_TCONV4")) == 0 : 0;   // A/B knob
    return !off && a.Cout == 128 && (a.ts.C & 3) == 0 && a.KCH * 16 == a.ts.taps * a.ts.C && !a.Wap && !a.H &&
           (size_t)tconv2_lds_floats(a.KCH * 16, 256, 32) * sizeof(float) <= 64 * 1024;
}
// Rows of a head tile.  The C2 head has B * N = 6624 rows: 207 tiles of 32 rows leave 49 CUs idle and give every other CU ONE workgroup.
// Measured (profiles/r3-05_head_tile_ab.txt): 16-row tiles take the two fc kernels from 18.5 / 16.9 to 15.7 / 13.8 us and the conv from
// 18.6 to 20.4 us (its 256 KB weight stream per workgroup does not shrink with the tile) -- so the fc kernels default to 16 rows, the
// conv / transposed conv to 32.  STGCN_HEAD_FC_TILE / STGCN_HEAD_TILE = 16 | 32 override.
inline int head_tile_rows() {
    static const int t = STGCN_EXP_ENV("STGCN_HEAD_TILE") ? atoi(STGCN_EXP_ENV("STGCN_HEAD_TILE")) : 32;
    return t == 16 ? 16 : 32;
}
inline int head_fc_tile_rows() {
    const char* e = getenv("STGCN_HEAD_FC_TILE");   // (read per call: the tests switch it)
    return (e && atoi(e) == 32) ? 32 : 16;
}
template <bool PLAIN>
int launch_tconv_fwd4(const char* label, const Tconv4Args& aa, hipStream_t st) {
    constexpr int KC = 4;
    const int TMr = head_tile_rows() / 16;
    const size_t lds = (size_t)(tconv2_lds_floats(aa.f.KCH * 16, 256, 16 * TMr) + 16) * sizeof(float);   // + 16: reduction words of the fused staging
    if constexpr (PLAIN) {   // the head's transposed conv: a backward kernel
        if (TMr == 1) STGCN_LAUNCH_ETB(label, st, (tconv_fwd4_kernel<1, KC, PLAIN, ET>), dim3(cdiv(aa.f.ts.rows, 16)), dim3(512), lds, aa);
        else STGCN_LAUNCH_ETB(label, st, (tconv_fwd4_kernel<2, KC, PLAIN, ET>), dim3(cdiv(aa.f.ts.rows, 32)), dim3(512), lds, aa);
        return STGCN_OK;
    }
    if (TMr == 1) STGCN_LAUNCH_ET(label, st, (tconv_fwd4_kernel<1, KC, PLAIN, ET>), dim3(cdiv(aa.f.ts.rows, 16)), dim3(512), lds, aa);
    else STGCN_LAUNCH_ET(label, st, (tconv_fwd4_kernel<2, KC, PLAIN, ET>), dim3(cdiv(aa.f.ts.rows, 32)), dim3(512), lds, aa);
    return STGCN_OK;
}
int launch_tconv_fwd(const char* label, const TconvFwdArgs& a, hipStream_t st) {
    if (tconv4_ok(a)) {
        Tconv4Args aa;
        memset(&aa, 0, sizeof(aa));
        aa.f = a;
        return launch_tconv_fwd4<false>(label, aa, st);
    }
#ifdef STGCN_EXPERIMENTS
    static const int ver = getenv("STGCN_TCONV_V") ? atoi(getenv("STGCN_TCONV_V")) : 1;   // 1: row tiles (fastest at C2); 3: time-complete tiles (opt-in)
    if (ver == 3 && (a.ts.C & 15) == 0 && a.KCH * 16 == a.ts.taps * a.ts.C && a.c1 == 16 * (a.Wap ? 1 : a.c1 / 16) &&
        tconv3_lds_bytes(a.ts.Tsrc, a.ts.C, 2 * a.Cout, 2) <= 64 * 1024) {
        if (a.Cout == 64 && a.KCH <= 3) return launch_tconv_fwd3<4, 2, 3>(label, a, st);
        if (a.Cout == 64 && a.KCH <= 12) return launch_tconv_fwd3<4, 2, 12>(label, a, st);
        if (a.Cout == 128 && a.KCH <= 4) return launch_tconv_fwd3<8, 2, 4>(label, a, st);
        if (a.Cout == 128 && a.KCH <= 16) return launch_tconv_fwd3<8, 2, 16>(label, a, st);
    }
#endif
    return a.Cout == 64 ? launch_tconv_fwd_nt<2>(label, a, st) : launch_tconv_fwd_nt<4>(label, a, st);
}

// graph-conv launch geometry: a (b, t) slab is split over `parts` workgroups (part p owns node tiles p, p + parts, ..) of
// `waves` waves with up to `maxq` tiles per wave.  The path offers only B*T slabs (192-320 at C2) for 256 CUs and one
// slab's 13 node tiles do not divide over 4 SIMDs, so the forward runs 4 waves x ~1 tile per workgroup (parts = tiles/4:
// every workgroup puts one tile on each SIMD, ~5 workgroups per CU: 27.5 -> 22.6 us at C2 block 0); the backward, whose
// parts would each re-stage X_k / dY and re-form all G_k, stays at one workgroup per slab (2 parts: 29 -> 35 us).
// STGCN_GC_PARTS=<f>,<b> overrides (tuning knob).
struct GcGeom { int parts, waves, maxq; };
inline GcGeom gc_geom(int HT, int want_parts) {
    GcGeom g;
    g.parts = want_parts < 1 ? 1 : (want_parts > HT ? HT : want_parts);
    const int per = (HT + g.parts - 1) / g.parts;          // tiles of the largest part
    g.waves = per > 16 ? 8 : (per < 4 ? 4 : per);
    g.maxq = (per + g.waves - 1) / g.waves;
    return g;
}
inline void gc_parts_override(int& fwd, int& bwd) {
    static int f = 0, b = 0, init = 0;
    if (!init) {
        init = 1;
        const char* e = getenv("STGCN_GC_PARTS");
        if (e) sscanf(e, "%d,%d", &f, &b);
    }
    if (f > 0) fwd = f;
    if (b > 0) bwd = b;
}

// ---- tiled graph conv (stgcn_kernels_gctile.hip.h): one GEMM launch per operator term + one row pass -----------------
template <int NTW>
int launch_gso_gemm_ntw(const char* label, GsoGemmArgs g, hipStream_t st) {
    g.col_tiles = cdiv(g.slabs, 2 * NTW);
    STGCN_LAUNCH(label, st, (gso_gemm_kernel<NTW>), dim3((unsigned)(g.row_tiles * g.col_tiles)), dim3(256), gt_lds_floats(NTW) * sizeof(float), g);
    return STGCN_OK;
}
int launch_gso_gemm(const char* label, const float* M, const float* X, float alpha, const float* Z1, float b1, const float* Z2, float b2,
                    float* out, int N, int NP, long slabs, hipStream_t st) {
    STGCN_F32_ONLY(label);
    GsoGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = M; g.X = X; g.Z1 = Z1; g.Z2 = Z2; g.out = out; g.alpha = alpha; g.b1 = b1; g.b2 = b2;
    g.N = N; g.NP = NP; g.slabs = slabs;
    g.row_tiles = cdiv(N, kGtBM);
    // column extent of the workgroup tile (2 * NTW slabs): the launch costs rounds-of-resident-workgroups x work per
    // workgroup (~NTW); ties go to the wider tile (fewer operator re-reads).  STGCN_GEMM_NTW=<3|4|5> forces one (tuning knob).
    static const int force = getenv("STGCN_GEMM_NTW") ? atoi(getenv("STGCN_GEMM_NTW")) : 0;
    int best = 4;
    long best_cost = -1;
    for (int ntw = 5; ntw >= 3; --ntw) {
        const long wgs = (long)g.row_tiles * cdiv(slabs, 2 * ntw);
        const int cap = ntw == 5 ? wg_capacity(gso_gemm_kernel<5>, 256, gt_lds_floats(5) * sizeof(float))
                      : ntw == 4 ? wg_capacity(gso_gemm_kernel<4>, 256, gt_lds_floats(4) * sizeof(float))
                                 : wg_capacity(gso_gemm_kernel<3>, 256, gt_lds_floats(3) * sizeof(float));
        const long cost = (long)rounds_of(wgs, cap) * ntw;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = ntw; }
    }
    if (force >= 3 && force <= 5) best = force;
    if (best == 5) return launch_gso_gemm_ntw<5>(label, g, st);
    if (best == 3) return launch_gso_gemm_ntw<3>(label, g, st);
    return launch_gso_gemm_ntw<4>(label, g, st);
}
// bf16 / bf16x3 operator product (g_gc_precision 2 / 1): operand planes as laid out by stgcn_gso_prepare (second matrix
// slot: hi plane then lo plane) and by gc_pack_operand_kernel / the previous GEMM's epilogue
struct OperandBuf { float* hi; float* lo; };
inline OperandBuf operand_buf(float* XT, int which, long CP, int NP) {
    const size_t LD = (size_t)gc_plane_ld(NP);
    const size_t rows = (size_t)(CP + 384);        // gc_operand_alloc: slack rows behind CP for the wide column tiles
    float* base = XT + (size_t)which * rows * LD;   // rows * LD bf16 per plane = rows * LD / 2 floats, two planes per buffer
    return OperandBuf{base, base + rows * LD / 2};
}
int launch_pack_operand(const float* X, int N, int NP, long slabs, OperandBuf o, hipStream_t st) {
    STGCN_LAUNCH_ET("gc_pack_operand", st, (gc_pack_operand_kernel<ET>), dim3((unsigned)cdiv(NP, 256), (unsigned)slabs), dim3(256), 256 * 17 * sizeof(float),
                    X, N, NP, gc_plane_ld(NP), o.hi, o.lo);
    return STGCN_OK;
}
int launch_gso_gemm_bf16(const char* label, const float* Mpad, OperandBuf x, float alpha, const float* Z1, float b1, const float* Z2, float b2,
                         float* out, const OperandBuf* next, int N, int NP, long slabs, hipStream_t st) {
    GsoGemmBfArgs g;
    memset(&g, 0, sizeof(g));
    const size_t M = (size_t)NP * NP;
    g.LD = gc_plane_ld(NP);
    g.Mh = Mpad + M; g.Ml = Mpad + M + (size_t)NP * g.LD / 2;
    g.Xh = x.hi; g.Xl = x.lo;
    if (next) { g.Oh = next->hi; g.Ol = next->lo; }
    g.Z1 = Z1; g.Z2 = Z2; g.out = out; g.alpha = alpha; g.b1 = b1; g.b2 = b2;
    g.N = N; g.NP = NP; g.slabs = slabs;
    g.row_tiles = cdiv(N, 128);
    g.col_tiles = (int)(gc_operand_cols(slabs) / 128);
    const dim3 grid((unsigned)(g.row_tiles * g.col_tiles));
    // depth of a pipeline step: 32 (more workgroups per CU) or 64 bf16 (half as many barriers); STGCN_GEMM_BF16_BK forces one
    static const int force_bk = getenv("STGCN_GEMM_BF16_BK") ? atoi(getenv("STGCN_GEMM_BF16_BK")) : 0;
    const int bk = force_bk == 32 || force_bk == 64 ? force_bk : kGbDefaultBK;
    const int split = g_gc_precision == 1 && !g_bf16;   // (bf16 activations: operands are bf16 numbers already, one MFMA per product)
    {   // 256 x (32 * NT) tiles, one workgroup per CU (gso_gemm_bf16_big_kernel); STGCN_GEMM_BIG=0: the 128 x 128 kernel
        static const int off = STGCN_EXP_ENV("STGCN_GEMM_BIG") ? atoi(STGCN_EXP_ENV("STGCN_GEMM_BIG")) == 0 : 0;
        static const int env_nt = STGCN_EXP_ENV("STGCN_GEMM_BIG_NT") ? atoi(STGCN_EXP_ENV("STGCN_GEMM_BIG_NT")) : 0;
        const int force_nt = g_gemm_big_nt ? g_gemm_big_nt : env_nt;   // stgcn_set_gemm_big_nt (tests force every instance) before the environment
        static const int big_bk = getenv("STGCN_GEMM_BIG_BK") && atoi(getenv("STGCN_GEMM_BIG_BK")) == 32 ? 32 : 64;   // r3-17: 64-deep steps 6 % faster
        if (!off && !split && NP % kGbBigBM == 0) {
            const long CP = gc_operand_cols(slabs);
            const int rts = NP / kGbBigBM, cus = device_cus();
            int best = 0;
            double best_cost = 0;
            const int cand[5] = {10, 8, 6, 5, 4};
            for (int i = 0; i < 5; ++i) {   // cost ~ rounds of resident workgroups x columns per tile; ties go to the wider tile
                const int nt = cand[i];
                const long tiles = (long)rts * cdiv(CP, 32 * nt);
                const double cost = (double)rounds_of(tiles, cus) * nt;
                if (force_nt ? nt == force_nt : (!best || cost < best_cost)) { best = nt; best_cost = cost; }
            }
            if (best) {
                g.row_tiles = rts;
                g.col_tiles = cdiv(CP, 32 * best);
                g.Ol = nullptr;   // (no split product reads the low plane of the result)
                const dim3 gridb((unsigned)(g.row_tiles * g.col_tiles));
                const size_t ldsb = gb_big_lds_bytes(best, big_bk);
#define STGCN_BIG_CASE(NTV) \
    case NTV: \
        if (big_bk == 64) STGCN_LAUNCH_ET(label, st, (gso_gemm_bf16_big_kernel<NTV, ET, 64>), gridb, dim3(512), ldsb, g); \
        else STGCN_LAUNCH_ET(label, st, (gso_gemm_bf16_big_kernel<NTV, ET, 32>), gridb, dim3(512), ldsb, g); \
        break;
                switch (best) {
                    STGCN_BIG_CASE(10)
                    STGCN_BIG_CASE(8)
                    STGCN_BIG_CASE(6)
                    STGCN_BIG_CASE(5)
                    default:
                    STGCN_BIG_CASE(4)
                }
#undef STGCN_BIG_CASE
                return STGCN_OK;
            }
        }
    }
    const size_t lds = gb_lds_floats(split, bk) * sizeof(float);
    if (split && bk == 64) STGCN_LAUNCH(label, st, (gso_gemm_bf16_kernel<1, 64, float>), grid, dim3(256), lds, g);
    else if (split) STGCN_LAUNCH(label, st, (gso_gemm_bf16_kernel<1, 32, float>), grid, dim3(256), lds, g);
    else if (bk == 64) STGCN_LAUNCH_ET(label, st, (gso_gemm_bf16_kernel<0, 64, ET>), grid, dim3(256), lds, g);
    else STGCN_LAUNCH_ET(label, st, (gso_gemm_bf16_kernel<0, 32, ET>), grid, dim3(256), lds, g);
    return STGCN_OK;
}

int launch_gconv_fwd_tiled(const GconvFwdArgs& a, hipStream_t st) {
    if (a.Ks > kGcMaxTerms) return fail(STGCN_ERR_UNSUPPORTED, "tiled graph conv with %d terms (supported: up to %d)", a.Ks, kGcMaxTerms);
    if (a.Ks > 1 && !a.Xk) return fail(STGCN_ERR_INVALID, "tiled graph conv needs the X_k buffers");
    const long ks = a.slabs * a.N * 16;                 // elements per term
    const long ksf = g_bf16 ? ks / 2 : ks;              // ... in the 4-byte units of the float* slots
    const bool bf = (g_gc_precision > 0 || g_bf16) && a.Ks > 1;   // bf16 activations: the operator products run on the bf16 matrix cores
    if (bf && !a.XT) return fail(STGCN_ERR_INVALID, "tiled graph conv (bf16 operator products) needs the operand workspace");
    const long CP = gc_operand_cols(a.slabs);
    if (bf) {
        const int rc = launch_pack_operand(a.A, a.N, a.NP, a.slabs, operand_buf(a.XT, 0, CP, a.NP), st);
        if (rc) return rc;
    }
    // X_1 = L X_0 ; X_k = 2 L X_{k-1} - X_{k-2}   (layers.py:153-161; GraphConv: X_1 = A_hat X_0, layers.py:198)
    for (int k = 1; k < a.Ks; ++k) {
        const float* Xm1 = k == 1 ? a.A : a.Xk + (size_t)(k - 2) * ksf;
        const float* Xm2 = k == 1 ? nullptr : (k == 2 ? a.A : a.Xk + (size_t)(k - 3) * ksf);
        float* out = a.Xk + (size_t)(k - 1) * ksf;
        int rc;
        if (bf) {
            const OperandBuf next = operand_buf(a.XT, k & 1, CP, a.NP);   // term k reads buffer (k - 1) & 1
            rc = launch_gso_gemm_bf16("gso_gemm_fwd", a.Lp, operand_buf(a.XT, (k - 1) & 1, CP, a.NP), k == 1 ? 1.f : 2.f, Xm2, -1.f, nullptr, 0.f,
                                      out, k + 1 < a.Ks ? &next : nullptr, a.N, a.NP, a.slabs, st);
        } else {
            (void)Xm1;
            rc = launch_gso_gemm("gso_gemm_fwd", a.Lp, Xm1, k == 1 ? 1.f : 2.f, Xm2, -1.f, nullptr, 0.f, out, a.N, a.NP, a.slabs, st);
        }
        if (rc) return rc;
    }
    GcRowsFwdArgs r;
    memset(&r, 0, sizeof(r));
    r.X0 = a.A; r.Xk = a.Xk; r.W = a.W; r.bias = a.bias; r.G = a.G; r.rows = a.slabs * a.N; r.kstride = ks; r.terms = a.Ks; r.kipf = a.kipf;
    const long tiles = (r.rows + 15) / 16;
    const long wgs = (tiles + 3) / 4;
    STGCN_LAUNCH_ET("gconv_rows_fwd", st, (gconv_rows_fwd_kernel<ET>), dim3((unsigned)(wgs < 4096 ? wgs : 4096)), dim3(256), (size_t)4 * 16 * 20 * sizeof(float), r);
    return STGCN_OK;
}
// backward: row pass (g_k, parameter-gradient partials), then dA = sum_k T_k(L^T) g_k by the Clenshaw recurrence
//     b_K = g_K ; b_k = g_k + 2 L^T b_{k+1} - b_{k+2} (in place over g_k) ; dA = g_0 + L^T b_1 - b_2
int launch_gconv_bwd_tiled(const GconvBwdArgs& a, hipStream_t st) {
    if (a.Ks > kGcMaxTerms) return fail(STGCN_ERR_UNSUPPORTED, "tiled graph conv with %d terms (supported: up to %d)", a.Ks, kGcMaxTerms);
    if (!a.Gk || a.wgs < 1 || a.tiles_per_wg < 1) return fail(STGCN_ERR_INVALID, "tiled graph-conv backward: missing workspace / geometry");
    const long ks = a.slabs * a.N * 16;                 // elements per term
    const long ksf = g_bf16 ? ks / 2 : ks;              // ... in the 4-byte units of the float* slots
    const int K = a.Ks - 1;
    GcRowsBwdArgs r;
    memset(&r, 0, sizeof(r));
    r.dY = a.dY; r.X0 = a.X0; r.Xk = a.Xk; r.W = a.W; r.part = a.part; r.rows = a.slabs * a.N; r.kstride = ks; r.gstride = ks;
    r.Gk = K == 0 ? a.dA : a.Gk;   // a single term: g_0 (+ dY) is dA itself
    r.terms = a.Ks; r.kipf = a.kipf; r.tiles_per_wg = a.tiles_per_wg;
    STGCN_LAUNCH_ET("gconv_rows_bwd", st, (gconv_rows_bwd_kernel<ET>), dim3((unsigned)a.wgs), dim3(256), gc_rows_bwd_lds_bytes(r.terms), r);
    if (K == 0) return STGCN_OK;
    auto gk = [&](int k) { return a.Gk + (size_t)k * ksf; };
    if (g_gc_precision > 0 || g_bf16) {   // bf16 / bf16x3 products: b_{k+1} travels in operand form from epilogue to epilogue
        if (!a.XT) return fail(STGCN_ERR_INVALID, "tiled graph-conv backward (bf16 operator products) needs the operand workspace");
        const long CP = gc_operand_cols(a.slabs);
        int cur = 0;
        int rc = launch_pack_operand(gk(K), a.N, a.NP, a.slabs, operand_buf(a.XT, cur, CP, a.NP), st);
        if (rc) return rc;
        for (int k = K - 1; k >= 1; --k) {
            const OperandBuf next = operand_buf(a.XT, cur ^ 1, CP, a.NP);
            rc = launch_gso_gemm_bf16("gso_gemm_bwd", a.LTp, operand_buf(a.XT, cur, CP, a.NP), 2.f, gk(k), 1.f, k + 2 <= K ? gk(k + 2) : nullptr, -1.f,
                                      gk(k), &next, a.N, a.NP, a.slabs, st);
            if (rc) return rc;
            cur ^= 1;
        }
        return launch_gso_gemm_bf16("gso_gemm_bwd", a.LTp, operand_buf(a.XT, cur, CP, a.NP), 1.f, gk(0), 1.f, K >= 2 ? gk(2) : nullptr, -1.f, a.dA,
                                    nullptr, a.N, a.NP, a.slabs, st);
    }
    for (int k = K - 1; k >= 1; --k) {
        const int rc = launch_gso_gemm("gso_gemm_bwd", a.LTp, gk(k + 1), 2.f, gk(k), 1.f, k + 2 <= K ? gk(k + 2) : nullptr, -1.f, gk(k), a.N, a.NP,
                                       a.slabs, st);
        if (rc) return rc;
    }
    return launch_gso_gemm("gso_gemm_bwd", a.LTp, gk(1), 1.f, gk(0), 1.f, K >= 2 ? gk(2) : nullptr, -1.f, a.dA, a.N, a.NP, a.slabs, st);
}

int launch_gconv_fwd(GconvFwdArgs a, hipStream_t st) {
    if (gc_is_tiled(a.N, a.Ks)) return launch_gconv_fwd_tiled(a, st);
    const int HT = a.NP / 16;
    int pf = (HT + 3) / 4, pb = 1;
    gc_parts_override(pf, pb);
#ifdef STGCN_EXPERIMENTS
    {   // operator-stationary variant: the wave's operator fragments in registers, several slabs per workgroup
        static const int per_cu = getenv("STGCN_GC_REG") ? atoi(getenv("STGCN_GC_REG")) : 0;   // opt-in (measured equal / slower at C2 and at bs 128): workgroups per CU
        const int off = per_cu <= 0;
        const int nterm = a.Ks - 1, KCH = a.NP / 16;
        if (!off && (nterm == 1 || nterm == 2) && KCH <= 21 && a.NP * 4 <= 1024) {
            const int parts = (HT + 3) / 4;
            a.parts = parts;
            static int cus = 0;
            if (!cus) {
                int dev = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            }
            // about one workgroup per CU: groups = CUs / parts slab ranges
            int groups = per_cu * cus / parts;
            if (groups < 1) groups = 1;
            if (groups > a.slabs) groups = (int)a.slabs;
            const int spw = (int)((a.slabs + groups - 1) / groups);
            groups = (int)((a.slabs + spw - 1) / spw);
            const size_t lds = (size_t)2 * 16 * (a.NP + 4) * sizeof(float);
            const dim3 grid((unsigned)(groups * parts)), blk(256);
            if (KCH <= 13 && nterm == 2) STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd_reg_kernel<13, 2>), grid, blk, lds, a, spw);
            else if (KCH <= 13) STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd_reg_kernel<13, 1>), grid, blk, lds, a, spw);
            else if (nterm == 2) STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd_reg_kernel<21, 2>), grid, blk, lds, a, spw);
            else STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd_reg_kernel<21, 1>), grid, blk, lds, a, spw);
            return STGCN_OK;
        }
    }
#endif
#ifdef STGCN_EXPERIMENTS
    {   // persistent operator-stationary form: one workgroup per CU walking >= 2 slabs (gconv_fwd_pers_kernel; measured slower, pass r6-09)
        static const int pers = getenv("STGCN_GC_PERS") ? atoi(getenv("STGCN_GC_PERS")) : 0;
        const int nterm = a.Ks - 1;
        const long sets = device_cus() / 4;
        if (pers && (nterm == 1 || nterm == 2) && HT <= 16 && !(g_slab_gc_precision > 0 && !g_bf16) && a.slabs >= 2 * sets &&
            gconv_fwd_pers_lds_bytes(a.NP, a.Ks) <= 160 * 1024) {
            STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_pers_kernel<ET>), dim3((unsigned)(4 * sets)), dim3(256), gconv_fwd_pers_lds_bytes(a.NP, a.Ks), a, (int)sets);
            return STGCN_OK;
        }
    }
#endif
    const GcGeom g = gc_geom(HT, pf);
    a.parts = g.parts;
    const dim3 grid((unsigned)(a.slabs * g.parts)), blk(g.waves * 64);
    if (!g_bf16 && g_slab_gc_precision > 0 && a.Ks > 1 && g.maxq <= 3 && gs16_np32(a.N) * 2 <= 4 * g.waves * 64) {   // operator products on the bf16 matrix cores (bf16x3; four tiles per wave would spill: those graphs keep the fp32 kernel)
        const size_t lds16 = gconv_fwd16_lds_bytes(a.NP, a.N);
        if (g.maxq <= 1) STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd16_kernel<1, 16>), grid, blk, lds16, a);
        else if (g.maxq <= 2) STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd16_kernel<2, 8>), grid, blk, lds16, a);
        else STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd16_kernel<3, 8>), grid, blk, lds16, a);
        return STGCN_OK;
    }
    const size_t lds = (size_t)16 * (a.NP + 4) * sizeof(float);   // X0 transposed
    // STGCN_GC_SP=2 (opt-in): two slabs per workgroup, every operator fragment a wave loads feeds both.  Measured equal at C2 and 3 % slower at
    // C3 (r3-26 / r3-27): the product loop was bound by the latency of its own fragment loads, not by their volume
#ifdef STGCN_EXPERIMENTS   // (retired from the product build in round 5: a knob that never won)
    static const int force_sp = getenv("STGCN_GC_SP") ? atoi(getenv("STGCN_GC_SP")) : 0;
    const bool sp2 = a.Ks > 1 && g.maxq <= 2 && force_sp == 2;
    if (sp2) {
        const dim3 grid2((unsigned)(cdiv(a.slabs, 2) * g.parts));
        if (g.maxq <= 1) STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_kernel<1, 16, ET, 2>), grid2, blk, 2 * lds, a);
        else STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_kernel<2, 8, ET, 2>), grid2, blk, 2 * lds, a);
        return STGCN_OK;
    }
#endif
    {   // bf16 activations: the operator products from the operator's bf16 fragment plane on 32-deep MFMAs (gconv_fwd_body B16P; STGCN_GC_B16P=0: the 16-deep form)
        const char* e = STGCN_EXP_ENV("STGCN_GC_B16P");
        if (g_bf16 && a.Ks > 1 && g.maxq <= 2 && !(e && e[0] == '0')) {
            const size_t ldsp = gconv_fwd_b16p_lds_bytes(a.NP, a.N);
            if (g.maxq <= 1) STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd_b16p_kernel<1, 16>), grid, blk, ldsp, a);
            else STGCN_LAUNCH("gconv_fwd", st, (gconv_fwd_b16p_kernel<2, 8>), grid, blk, ldsp, a);
            return STGCN_OK;
        }
    }
    if (g.maxq <= 1) STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_kernel<1, 16, ET, 1>), grid, blk, lds, a);
    else if (g.maxq <= 2) STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_kernel<2, 8, ET, 1>), grid, blk, lds, a);
    else if (g.maxq <= 3) STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_kernel<3, 8, ET, 1>), grid, blk, lds, a);
    else if (g.maxq <= 4) STGCN_LAUNCH_ET("gconv_fwd", st, (gconv_fwd_kernel<4, 8, ET, 1>), grid, blk, lds, a);
    else return fail(STGCN_ERR_UNSUPPORTED, "graph convolution with %d nodes (supported: up to 512)", a.N);
    return STGCN_OK;
}

int launch_bwd_data(const char* label, const TconvBwdDataArgs& a, int ntt, hipStream_t st) {
    STGCN_F32_ONLY(label);
    const long tiles = cdiv(a.ts.rows, kTileRows);
    const dim3 grid((unsigned)tiles);
    const size_t lds = kTileLdsFloats * sizeof(float);
    if (ntt == 1) STGCN_LAUNCH(label, st, (tconv_bwd_data_kernel<1, 1, 1>), grid, dim3(256), lds, a);
    else if (ntt == 2) STGCN_LAUNCH(label, st, (tconv_bwd_data_kernel<2, 1, 2>), grid, dim3(256), lds, a);
    else if (ntt == 4) {
        // 8 waves x 2 m-tiles (more waves in flight) unless its register budget leaves part of the grid to a second round
        // and the 4-wave x 4 m-tile variant does not
        static const int force_w = getenv("STGCN_BWD_DATA_WAVES") ? atoi(getenv("STGCN_BWD_DATA_WAVES")) : 0;   // tuning knob
        const int r8 = force_w ? (force_w == 8 ? 0 : 2) : rounds_of(tiles, wg_capacity(tconv_bwd_data_kernel<2, 1, 0, 8>, 512, lds));
        const int r4 = force_w ? 1 : rounds_of(tiles, wg_capacity(tconv_bwd_data_kernel<4, 1, 0, 4>, 256, lds));
        if (r8 <= r4) STGCN_LAUNCH(label, st, (tconv_bwd_data_kernel<2, 1, 0, 8>), grid, dim3(512), lds, a);
        else STGCN_LAUNCH(label, st, (tconv_bwd_data_kernel<4, 1, 0, 4>), grid, dim3(256), lds, a);
    } else if (ntt == 8) STGCN_LAUNCH(label, st, (tconv_bwd_data_kernel<2, 2, 0, 8>), grid, dim3(512), lds, a);
    else return fail(STGCN_ERR_UNSUPPORTED, "backward-data with %d input channel tiles (supported: 1, 2, 4, 8)", ntt);
    return STGCN_OK;
}

// Chained forward launches (stgcn_device.hip.h): 0 = every stage its own launch, 1 = tmp_conv1 + graph conv in one launch, 2 = the whole
// forward of a block (tmp_conv1 + graph conv + tmp_conv2 / LayerNorm / dropout) where the shapes allow.  STGCN_CHAIN overrides (A/B runs);
// the hand-off relies on write-through stores, so a build with STGCN_WT_STORES=0 never chains.
inline int fwd_chain_mode() {
#if !STGCN_WT_STORES
    return 0;
#else
    const char* e = getenv("STGCN_CHAIN");
    return e ? atoi(e) : kChainDefault;
#endif
}
// The head's forward as ONE launch (head_fwd_kernel, stgcn_kernels_fwd.hip.h): default on; STGCN_HEAD_FUSE=0 restores the conv + fc launches
// (A/B runs), 2 / 4 admit only the 32- / 64-row tile (tests).  The exchange of the row statistics relies on write-through stores.
inline int head_fuse_mode() {
#if !STGCN_WT_STORES
    return 0;
#else
    const char* e = getenv("STGCN_HEAD_FUSE");
    return e ? atoi(e) : 1;
#endif
}
// job waves of gconv_bwd2_kernel: one per parameter-gradient job of the part (Ks weight terms + the bias, spread over the parts), but never
// more than the kernel's __launch_bounds__(768) leaves beside the tile waves -- the job loop strides by the job-wave count, so fewer waves
// only walk more jobs each (Ks >= 4 with one part asked for 13+ waves and the launch failed)
inline int gcbwd2_job_waves(int Ks, int parts, int nwa) {
    const int want = (Ks + 1 + parts - 1) / parts, room = 768 / 64 - nwa;
    return want < room ? want : room;
}
int launch_gconv_bwd(GconvBwdArgs a, hipStream_t st) {
    if (gc_is_tiled(a.N, a.Ks)) return launch_gconv_bwd_tiled(a, st);
    const int HT = a.NP / 16;
    int pf = 0, pb = 1;
    gc_parts_override(pf, pb);
    {   // round 3: slab split over parts, parameter-gradient jobs on their own waves (gconv_bwd2_kernel); STGCN_GCBWD2=0: the one-workgroup-per-slab kernel
        static const int off = STGCN_EXP_ENV("STGCN_GCBWD2") ? atoi(STGCN_EXP_ENV("STGCN_GCBWD2")) == 0 : 0;
        const int force_parts = getenv("STGCN_GCBWD2_PARTS") ? atoi(getenv("STGCN_GCBWD2_PARTS")) : 0;   // (read per call: the tests force geometries)
        // One node tile per tile wave (<= 8 tile waves per workgroup), and the whole grid resident in ONE round: every part re-stages the
        // slab's dY and re-forms all G_k, so a grid of SEVERAL parts that needs more rounds than the slab kernel costs more than it returns
        // (measured at the C3 size, 640 slabs x 3 parts on 512 slots: 120 us against 76 us; at C2, 320 x 2 on 768 slots: 27.2 against 30.0 us).
        const char* eb = STGCN_EXP_ENV("STGCN_GC_B16P");
        const bool b16p = g_bf16 && a.Ks > 1 && !(eb && eb[0] == '0');   // bf16 activations: G_k as bf16 planes, products from the operator's bf16 plane (32-deep MFMAs)
        const size_t lds2 = gconv_bwd2_lds_bytes(a.NP, a.N, a.Ks, b16p);   // (G_k tiles + dY rows + the job waves' transposition tiles)
        if (!off && lds2 <= 150 * 1024 && HT <= 64) {
            int best = 0;
            for (int parts = (HT + 7) / 8; parts <= HT && !force_parts; ++parts) {
                const int per = (HT + parts - 1) / parts, njw = gcbwd2_job_waves(a.Ks, parts, per > 8 ? 8 : per);
                if (per < 4 && parts > (HT + 7) / 8) break;     // keep >= 4 tile waves per workgroup
                const int cap = b16p ? wg_capacity(gconv_bwd2_kernel<1, bf16, true>, ((per > 8 ? 8 : per) + njw) * 64, lds2)
                                     : STGCN_ETB_VALUE(wg_capacity(gconv_bwd2_kernel<1, ET>, ((per > 8 ? 8 : per) + njw) * 64, lds2));
                if (a.slabs * parts <= cap) best = parts;
            }
            if (force_parts > 0 && force_parts <= HT && (HT + force_parts - 1) / force_parts <= 24) best = force_parts;   // (tuning: up to 3 tiles per wave)
            // no split fits one round (C3: 640 slabs): ONE part per slab still beats the kernel below since the job waves fetch whole tiles
            // (r3-49: 54.9 / 38.8 -> 47.5 / 33.5 us; 70 KB of LDS = two workgroups per CU instead of one)
            if (!best && !force_parts && HT <= 24) best = 1;
            if (best > 0) {
                const int per = (HT + best - 1) / best, nwa = per > 8 ? 8 : per, maxq = (per + nwa - 1) / nwa, njw = gcbwd2_job_waves(a.Ks, best, nwa);
                a.parts = best;
                const dim3 grid2((unsigned)(a.slabs * best)), blk2((nwa + njw) * 64);
                if (b16p) {
                    if (maxq <= 1) STGCN_LAUNCH("gconv_bwd", st, (gconv_bwd2_kernel<1, bf16, true>), grid2, blk2, lds2, a, nwa);
                    else if (maxq <= 2) STGCN_LAUNCH("gconv_bwd", st, (gconv_bwd2_kernel<2, bf16, true>), grid2, blk2, lds2, a, nwa);
                    else STGCN_LAUNCH("gconv_bwd", st, (gconv_bwd2_kernel<3, bf16, true>), grid2, blk2, lds2, a, nwa);
                    return STGCN_OK;
                }
                if (maxq <= 1) STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd2_kernel<1, ET>), grid2, blk2, lds2, a, nwa);
                else if (maxq <= 2) STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd2_kernel<2, ET>), grid2, blk2, lds2, a, nwa);
                else STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd2_kernel<3, ET>), grid2, blk2, lds2, a, nwa);
                return STGCN_OK;
            }
        }
    }
    const GcGeom g = gc_geom(HT, pb);
    a.parts = g.parts;
    const size_t lds = ((size_t)a.Ks * 16 * (a.NP + 4) + (size_t)a.NP * 20) * sizeof(float);
    if (lds > 160 * 1024) return fail(STGCN_ERR_UNSUPPORTED, "graph-conv backward needs %zu bytes of LDS (N=%d, terms=%d)", lds, a.N, a.Ks);
    const dim3 grid((unsigned)(a.slabs * g.parts)), blk(g.waves * 64);
    if (g.maxq <= 1) STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd_kernel<1, 16, ET>), grid, blk, lds, a);
    else if (g.maxq <= 2) STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd_kernel<2, 8, ET>), grid, blk, lds, a);
    else if (g.maxq <= 3) STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd_kernel<3, 8, ET>), grid, blk, lds, a);
    else if (g.maxq <= 4) STGCN_LAUNCH_ETB("gconv_bwd", st, (gconv_bwd_kernel<4, 8, ET>), grid, blk, lds, a);
    else return fail(STGCN_ERR_UNSUPPORTED, "graph convolution with %d nodes (supported: up to 512)", a.N);
    return STGCN_OK;
}

template <int MTW>
int launch_bwd_weight_n(const char* label, const TconvBwdWeightArgs& a, const WgradGeom& w, hipStream_t st) {
    const dim3 grid(w.chunks, w.mchunks), blk(kThreads * kWgradGroups);   // (16-byte im2col loads: c_in % 4 == 0, checked by the caller)
    // (exact fp32 products in every backward-precision mode: only the paired head launch below takes the bf16x3 form)
    if (a.NC == 128) STGCN_LAUNCH_ET(label, st, (tconv_bwd_weight_kernel<MTW, 2, true, ET>), grid, blk, wgrad_lds_bytes(MTW, 2), a);
    else STGCN_LAUNCH_ET(label, st, (tconv_bwd_weight_kernel<MTW, 4, true, ET>), grid, blk, wgrad_lds_bytes(MTW, 4), a);
    return STGCN_OK;
}
int launch_bwd_weight(const char* label, const TconvBwdWeightArgs& a, const WgradGeom& w, hipStream_t st) {
    if (a.NC != 128 && a.NC != 256) return fail(STGCN_ERR_UNSUPPORTED, "weight gradient with %d output channels", a.NC);
    if ((a.ts.C & 3) != 0) {   // scalar im2col loads: check_desc only admits such inputs with Kt*c_in <= 16, i.e. ONE m-tile
        if (w.MTW != 1) return fail(STGCN_ERR_UNSUPPORTED, "weight gradient: c_in=%d needs Kt*c_in <= 16", a.ts.C);
        if (a.NC == 128) STGCN_LAUNCH_ET(label, st, (tconv_bwd_weight_kernel<1, 2, false, ET>), dim3(w.chunks, w.mchunks), dim3(kThreads * kWgradGroups), wgrad_lds_bytes(1, 2), a);
        else STGCN_LAUNCH_ET(label, st, (tconv_bwd_weight_kernel<1, 4, false, ET>), dim3(w.chunks, w.mchunks), dim3(kThreads * kWgradGroups), wgrad_lds_bytes(1, 4), a);
        return STGCN_OK;
    }
    switch (w.MTW) {
        case 1: return launch_bwd_weight_n<1>(label, a, w, st);
        case 2: return launch_bwd_weight_n<2>(label, a, w, st);
        case 3: return launch_bwd_weight_n<3>(label, a, w, st);
        default: return launch_bwd_weight_n<4>(label, a, w, st);
    }
}

// two independent weight gradients (the head's conv and fc1) in ONE launch when both take the <4 m-tiles> variants
inline bool wgrad_pair_ok(const TconvBwdWeightArgs& a1, const WgradGeom& w1, const TconvBwdWeightArgs& a2, const WgradGeom& w2) {
    static const int off = STGCN_EXP_ENV("STGCN_WGRAD_PAIR") ? atoi(STGCN_EXP_ENV("STGCN_WGRAD_PAIR")) == 0 : 0;   // A/B knob
    return !off && a1.NC == 256 && w1.MTW == 4 && a2.NC == 128 && w2.MTW == 4 && (a1.ts.C & 3) == 0 && (a2.ts.C & 3) == 0;
}
int launch_wgrad_pair(const char* label, const TconvBwdWeightArgs& a1, const WgradGeom& w1, const TconvBwdWeightArgs& a2, const WgradGeom& w2,
                      hipStream_t st) {
    const int n1 = w1.chunks * w1.mchunks, n2 = w2.chunks * w2.mchunks;
    const size_t l1 = wgrad_lds_bytes(4, 4), l2 = wgrad_lds_bytes(4, 2);
    STGCN_LAUNCH_ETB(label, st, (wgrad_pair_kernel<4, 4, 4, 2, ET>), dim3((unsigned)(n1 + n2)), dim3(kThreads * kWgradGroups), l1 > l2 ? l1 : l2, a1, n1,
                    w1.mchunks, a2, n2, w2.mchunks);
    return STGCN_OK;
}

// ---- final deterministic reduction: job lists shared by the backward entry points and stgcn_grad_flush ----------------
struct ReduceList {
    ReduceArgs ra;
    int nj;
    bool overflow;
    ReduceList() : nj(0), overflow(false) { memset(&ra, 0, sizeof(ra)); }
    // job = (dst, src, partial count/stride, enumeration sizes n0 n1 n2 (n2 contiguous in src), src strides, dst strides)
    void add(float* dst, const float* src, int Pn, long pstride, int n0, int n1, int n2, long s0, long s1, long s2, long t0, long t1,
             long t2) {
        if (!dst) return;
        if (nj >= kMaxReduceJobs) { overflow = true; return; }
        ReduceJob& j = ra.job[nj];
        j.src = src; j.dst = dst; j.P = Pn; j.pstride = pstride; j.n0 = n0; j.n1 = n1; j.n2 = n2;
        j.s0 = (int)s0; j.s1 = (int)s1; j.s2 = (int)s2; j.t0 = (int)t0; j.t1 = (int)t1; j.t2 = (int)t2;
        ra.start[nj + 1] = ra.start[nj] + reduce_job_setup(j);
        ++nj;
    }
    void add_flat(float* dst, const float* src, int Pn, long pstride, int n) { add(dst, src, Pn, pstride, 1, 1, n, 0, 0, 1, 0, 0, 1); }
};
int launch_reduce_list(const char* label, ReduceList& L, hipStream_t st) {
    if (L.overflow) return fail(STGCN_ERR_INVALID, "%s: more than %d reduction jobs in one launch", label, kMaxReduceJobs);
    if (L.nj == 0) return STGCN_OK;
    L.ra.njobs = L.nj;
    static const bool log_jobs = getenv("STGCN_REDUCE_LOG") != nullptr;   // diagnostic: the partial-sum tables one launch folds
    if (log_jobs)
        for (int i = 0; i < L.nj; ++i) {
            const ReduceJob& j = L.ra.job[i];
            fprintf(stderr, "[stgcn reduce] %s job %d: %ld elements x %d partials (%.2f MB read), %d workgroups\n", label, i,
                    (long)j.n0 * j.n1 * j.n2, j.P, 4e-6 * (double)j.n0 * j.n1 * j.n2 * j.P, L.ra.start[i + 1] - L.ra.start[i]);
        }
    STGCN_LAUNCH(label, st, reduce_kernel, dim3(L.ra.start[L.nj]), dim3(kThreads), kThreads * 4 * sizeof(float), L.ra);
    return STGCN_OK;
}
void reduce_jobs_block(ReduceList& L, const stgcn_stblock_desc* d, const Derived& v, const BwdGeom& bg, const float* part,
                       const stgcn_stblock_grads* G) {
    // dW_eff partials wp[P][..][NC] (ps floats apart), db_eff partials bp[P][NC] (pb floats apart)
    // The weight-gradient partials are stored TRANSPOSED, dW_eff^T[o][k*Cin + i] with LD floats per column o: the 4 rows a lane of the producing
    // MFMA tile holds for its column are then 16 contiguous bytes (one write-through store instead of four scattered dwords)
    auto add_tconv_at = [&](const float* wp, const float* bp, int Pn, long ps, long pb, int NC, long LD, int Cin, int Cout, float* gw, float* gb, float* gaw,
                            float* gab) {
        // enumerate (o, k, i): src dW_eff^T[o][k*Cin + i] -> dst conv_w[o][i][k]
        L.add(gw, wp, Pn, ps, NC, d->Kt, Cin, LD, Cin, 1, (long)Cin * d->Kt, 1, d->Kt);
        L.add_flat(gb, bp, Pn, pb, NC);
        if (Cin > Cout) {   // the residual branch is a live 1x1 conv: it shares tap Kt-1 of the P half
            L.add(gaw, wp + (long)(d->Kt - 1) * Cin, Pn, ps, 1, Cout, Cin, 0, LD, 1, 0, Cin, 1);   // aw[o][i] <- column o, row (Kt-1)*Cin + i
            L.add_flat(gab, bp, Pn, pb, Cout);
        }
    };
    auto add_tconv = [&](const WgradGeom& w, int Cin, int Cout, float* gw, float* gb, float* gaw, float* gab) {
        const float* wp = part + w.off;
        add_tconv_at(wp, wp + (long)w.chunks * w.Mpad * w.NC, w.chunks, (long)w.Mpad * w.NC, w.NC, w.NC, w.Mpad, Cin, Cout, gw, gb, gaw, gab);
    };
    if (bg.k3) {     // per-workgroup partials of tc1_bwd_kernel: dW_eff1 [Kt*c_in][NC1] | db_eff1 [NC1] | dWa [c0][c1] | dba [c1]
        const float* wp = part + bg.off_k3;
        add_tconv_at(wp, wp + (long)d->Kt * d->c_in * v.NC1, bg.k3_wgs, bg.k3_stride, bg.k3_stride, v.NC1, (long)d->Kt * d->c_in, d->c_in, d->c0, G->tc1_w,
                     G->tc1_b, G->tc1_aw, G->tc1_ab);
    } else if (bg.thin) {   // dW_eff (16 padded rows) and db_eff sit behind dWa | dba in the per-workgroup partials
        const float* wp = part + bg.off_al + (long)d->c0 * 16 + 16;
        L.add(G->tc1_w, wp, bg.al_wgs, bg.al_stride, d->Kt, d->c_in, v.NC1, (long)d->c_in * v.NC1, v.NC1, 1, 1, d->Kt, (long)d->c_in * d->Kt);
        L.add_flat(G->tc1_b, wp + 16 * v.NC1, bg.al_wgs, bg.al_stride, v.NC1);
    } else {
        add_tconv(bg.w1, d->c_in, d->c0, G->tc1_w, G->tc1_b, G->tc1_aw, G->tc1_ab);
    }
    if (bg.k1) {   // per-(window, node tile) partials of tc2_bwd_kernel: [Kt*16][NC2] then [NC2]
        const float* wp = part + bg.off_k1;
        add_tconv_at(wp, wp + (long)d->Kt * 16 * v.NC2, bg.k1_wgs, bg.k1_stride, bg.k1_stride, v.NC2, (long)d->Kt * 16, d->c1, d->c2, G->tc2_w, G->tc2_b,
                     G->tc2_aw, G->tc2_ab);
    } else {
        add_tconv(bg.w2, d->c1, d->c2, G->tc2_w, G->tc2_b, G->tc2_aw, G->tc2_ab);
    }
    if (bg.k3 && d->c0 > d->c1) {
        const float* ap = part + bg.off_k3 + (long)d->Kt * d->c_in * v.NC1 + v.NC1;
        L.add(G->al_w, ap, bg.k3_wgs, bg.k3_stride, 1, d->c0, d->c1, 0, d->c1, 1, 0, 1, d->c0);
        L.add_flat(G->al_b, ap + (long)d->c0 * d->c1, bg.k3_wgs, bg.k3_stride, d->c1);
    } else if (d->c0 > d->c1) {
        // enumerate (i, j): src dWa[i][j] -> dst al_w[j][i]
        L.add(G->al_w, part + bg.off_al, bg.al_wgs, bg.al_stride, 1, d->c0, d->c1, 0, d->c1, 1, 0, 1, d->c0);
        L.add_flat(G->al_b, part + bg.off_al + (long)d->c0 * d->c1, bg.al_wgs, bg.al_stride, d->c1);
    }
    if (d->graph_conv == STGCN_GC_KIPF) L.add_flat(G->gc_w, part + bg.off_gc + 256, bg.gc_count, bg.gc_stride, 256);
    else L.add_flat(G->gc_w, part + bg.off_gc, bg.gc_count, bg.gc_stride, v.terms * 256);
    L.add_flat(G->gc_b, part + bg.off_gc + (long)v.terms * 256, bg.gc_count, bg.gc_stride, 16);
    const int n = d->N * d->c2;
    L.add_flat(G->ln_w, part + bg.off_ln_g, bg.ln_sg, n, n);
    L.add_flat(G->ln_b, part + bg.off_ln_b, bg.ln_sg, n, n);
}
LnRowstatOut rowstat_out(const stgcn_ln_hook* h) {
    LnRowstatOut o;
    memset(&o, 0, sizeof(o));
    if (!h || !h->rowstat) return o;
    o.rowstat = reinterpret_cast<float2*>(h->rowstat); o.y = h->y; o.gamma = h->gamma; o.beta = h->beta;
    o.N = h->N; o.C = h->C; o.training = h->training && h->droprate > 0.f;
    o.keep_scale = 1.0f / (1.0f - h->droprate); o.thresh = drop_thresh(h->droprate); o.seed = h->seed; o.offset = h->offset;
    o.offset_dev = h->offset_dev;
    o.mask_from_y = hook_mask_from_y() ? 1 : 0;
    return o;
}
}  // namespace

extern "C" {

int stgcn_version(void) { return STGCN_ABI_VERSION; }
const char* stgcn_backend(void) { return STGCN_BACKEND_NAME; }
const char* stgcn_last_error(void) { return g_err; }

int stgcn_profile_enable(int on) {
    g_prof_on = on ? 1 : 0;
    if (on) g_prof_n = 0;
    return STGCN_OK;
}

int stgcn_profile_collect(char* buf, size_t cap) {
    if (!buf || cap < 64) return fail(STGCN_ERR_INVALID, "stgcn_profile_collect: buffer too small");
    struct Agg { const char* name; int tag; int calls; double ms; };
    Agg agg[96];
    int na = 0;
    for (int i = 0; i < g_prof_n; ++i) {
        (void)hipEventSynchronize(g_prof[i].e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, g_prof[i].e0, g_prof[i].e1);
        int k = 0;
        for (; k < na; ++k)
            if (agg[k].tag == g_prof[i].tag && strcmp(agg[k].name, g_prof[i].name) == 0) break;
        if (k == na) {
            if (na == 96) continue;
            agg[na].name = g_prof[i].name; agg[na].tag = g_prof[i].tag; agg[na].calls = 0; agg[na].ms = 0.0;
            ++na;
        }
        agg[k].calls++;
        agg[k].ms += ms;
    }
    size_t o = 0;
    o += snprintf(buf + o, cap - o, "{");
    for (int k = 0; k < na && o + 96 < cap; ++k)
        o += snprintf(buf + o, cap - o, "%s\"%s@%d\": {\"calls\": %d, \"total_ms\": %.6f}", k ? ", " : "", agg[k].name, agg[k].tag, agg[k].calls, agg[k].ms);
    snprintf(buf + o, cap - o, "}");
    g_prof_n = 0;
    return STGCN_OK;
}

int stgcn_stblock_plan_query(const stgcn_stblock_desc* d, stgcn_stblock_plan* p) {
    int rc = check_desc(d);
    if (rc) return rc;
    if (!p) return fail(STGCN_ERR_INVALID, "plan is NULL");
    const Derived v = derive(d);
    memset(p, 0, sizeof(*p));
    p->T1 = v.T1; p->T2 = v.T2; p->rows1 = v.rows1; p->rows2 = v.rows2; p->NP = v.NP;
    p->y_floats = v.rows2 * d->c2;
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t at = o; o += rup(n, 64); return at; };   // 256-byte aligned carve
    const bool bf = d->dtype == STGCN_DTYPE_BF16;
    auto act = [&](int64_t n) { return bf ? (n + 1) / 2 : n; };   // 4-byte units of an activation tensor of n elements
    p->recompute_tc1 = (d->Kt * d->c_in <= 16) ? 1 : 0;   // K <= 16: recomputing Z in backward beats storing 2 x rows1 x c0
    p->sv_U1 = take(act(p->recompute_tc1 ? 0 : v.rows1 * d->c0));
    p->sv_S1 = take(act(p->recompute_tc1 ? 0 : v.rows1 * d->c0));
    p->sv_A = take(act(v.rows1 * d->c1));
    p->sv_Xk = take((int64_t)(v.terms - 1) * act(v.rows1 * d->c1));
    p->sv_G = take(act(v.rows1 * d->c1));
    const BwdGeom bgs = bwd_geom(d->B, d->T, d->N, d->c_in, d->c0, d->c1, d->c2, d->Kt, v.terms, d->need_dx);
    p->stored_US2 = (!tc2_ln_fwd_fused_ok(d->c1, d->c2, d->Kt, d->N) || !bgs.k1 || g_debug_stages || !tc2_recompute(bf)) ? 1 : 0;
    p->sv_U2 = take(act(p->stored_US2 ? v.rows2 * d->c2 : 0));
    p->sv_S2 = take(act(p->stored_US2 ? v.rows2 * d->c2 : 0));
    p->sv_mean = take(v.slabs2);
    p->sv_rstd = take(v.slabs2);
    p->sv_rowstat = take(2 * v.rows2);
    p->saved_floats = o;
    o = 0;
    // control words of the chained launches FIRST: their offset must not depend on need_dx / training, because the pack launch that zeroes
    // them may have been planned with other flags than the forward that uses them (stgcn_prepack packs a whole model with need_dx = 1)
    // control words of the forward: header, the arrival counters of A[slab] and G[slab] (chained launches, experiments), then the exchange
    // words of tmp_conv2 + LayerNorm when several workgroups share a slab (64-bit, up to kTc2LnMaxPeers per slab)
    p->chain_words = kChainHdr + 2 * v.slabs1 + 2 * kTc2LnMaxPeers * v.slabs2;
    p->ws_chain = take(p->chain_words);
    p->ws_W1p = take((int64_t)v.NC1 * v.KP1);
    p->ws_W1d = take((int64_t)d->Kt * v.NC1 * v.CP_in);
    p->ws_b1 = take(v.NC1);
    p->ws_Wap = take((int64_t)d->c0 * d->c1);
    p->ws_WaT = take((int64_t)v.CP1 * d->c0);
    p->ws_ba = take(d->c1);
    p->ws_W2p = take((int64_t)v.NC2 * v.KP2);
    p->ws_W2d = take((int64_t)d->Kt * v.NC2 * v.CP1);
    p->ws_b2 = take(v.NC2);
    const bool k3s = (tc1_bwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt) || tc1_fwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt)) && (d->Kt * d->c_in > 4);
    p->ws_W1dense = take((p->recompute_tc1 || k3s) ? (int64_t)v.KP1 * v.NC1 : 0);
    const BwdGeom bgq = bwd_geom(d->B, d->T, d->N, d->c_in, d->c0, d->c1, d->c2, d->Kt, v.terms, d->need_dx);
    p->thin_tc1 = bgq.thin;
    p->fused_tc2_bwd = bgq.k1;
    p->ws_W2dense = take(bgq.k1 ? (int64_t)v.KP2 * v.NC2 : 0);
    p->ws_WaDense = take((p->thin_tc1 || k3s) ? (int64_t)d->c0 * d->c1 : 0);
    p->fused_tc1_bwd = bgq.k3;
    p->ws_rowstat_b = take(2 * v.rows2 + 2 * v.slabs2);   // row partials, then the per-slab constants (big slabs only)
    p->ws_dZ2 = take(act(v.rows2 * v.NC2));
    p->ws_dYg = take(act(v.rows1 * d->c1));
    p->ws_dA = take(act(v.rows1 * d->c1));
    p->ws_dZ1 = take(act(v.rows1 * v.NC1));
    p->tiled_gc = v.tiled;
    p->ws_Gk = take(v.tiled ? (int64_t)v.terms * act(v.rows1 * d->c1) : 0);
    p->ws_XT = take(v.tiled && v.terms > 1 ? 2 * gc_operand_alloc(v.slabs1) * (int64_t)gc_plane_ld(v.NP) : 0);
    p->part_floats = bwd_partial_floats(d->B, d->T, d->N, d->c_in, d->c0, d->c1, d->c2, d->Kt, v.terms, d->need_dx);
    p->ws_part = take(p->part_floats);
    p->ws_floats = o;
    return STGCN_OK;
}

int stgcn_set_debug_stages(int32_t on) {
    const int prev = g_debug_stages;
    if (on == 0 || on == 1) g_debug_stages = on;
    return prev;
}

int64_t stgcn_set_chain_spin_ticks(int64_t ticks) {
    if (ticks == 0) return g_chain_spin_ticks.load();
    return g_chain_spin_ticks.exchange(ticks);
}

int stgcn_set_tc1_bwd_wgs(int32_t n) {
    const int prev = g_tc1_bwd_wgs;
    if (n >= 0) g_tc1_bwd_wgs = n;
    return prev;
}

int stgcn_set_tc2ln_peers(int32_t n) {
    const int prev = g_tc2ln_peers;
    if (n == 0 || n == 1 || n == 2 || n == 4) g_tc2ln_peers = n;
    return prev;
}

int stgcn_set_gc_tiled_min_nodes(int32_t n) {
    const int prev = g_gc_tiled_min_n;
    if (n >= 1) g_gc_tiled_min_n = n;
    return prev;
}

int stgcn_set_gc_ld_pad(int32_t pad) {
    const int prev = g_gc_ld_pad;
    if (pad >= 0 && pad <= 65536 && (pad & 7) == 0) g_gc_ld_pad = pad;
    return prev;
}

int stgcn_set_gemm_big_nt(int32_t nt) {
    const int prev = g_gemm_big_nt;
    if (nt == 0 || nt == 4 || nt == 5 || nt == 6 || nt == 8 || nt == 10) g_gemm_big_nt = nt;
    return prev;
}

int stgcn_set_gc_precision(int32_t mode) {
    const int prev = g_gc_precision;
    if (mode >= 0 && mode <= 2) g_gc_precision = mode;
    return prev;
}

int stgcn_set_bwd_precision(int32_t mode) {
    const int prev = g_bwd_precision;
    if (mode >= 0 && mode <= 1) g_bwd_precision = mode;
    return prev;
}

int stgcn_set_slab_gc_precision(int32_t mode) {
    const int prev = g_slab_gc_precision;
    if (mode >= 0 && mode <= 1) g_slab_gc_precision = mode;
    return prev;
}

int stgcn_gso_layout(int32_t N, int32_t terms, int64_t* NP, int64_t* mats, int64_t* scratch_mats, int32_t* tiled) {
    if (N < 1 || N > 32768 || terms < 1 || terms > 9) return fail(STGCN_ERR_INVALID, "stgcn_gso_layout: bad arguments (1 <= N <= 32768, 1 <= terms <= 9)");
    const bool t = gc_is_tiled(N, terms);
    if (NP) *NP = gc_padded_nodes(N, terms);
    // tiled: fp32 matrix, then its bf16 hi / lo planes with leading dimension gc_plane_ld (NP * LD floats for both)
    const int64_t np = gc_padded_nodes(N, terms);
    // slab-resident: the fp32 fragments of T_1 .. T_{terms-1}, then their bf16 hi / lo fragment planes (stgcn_kernels_gcslab16.hip.h)
    const int64_t m16 = terms > 1 ? ((int64_t)(terms - 1) * (int64_t)gs16_term_floats((int)np, N) + np * np - 1) / (np * np) : 0;
    if (mats) *mats = t ? 1 + (gc_plane_ld((int)np) + np - 1) / np : (terms > 1 ? terms - 1 + m16 : 1);
    if (scratch_mats) *scratch_mats = t ? 0 : 3;
    if (tiled) *tiled = t ? 1 : 0;
    return STGCN_OK;
}

int stgcn_gso_prepare(const float* gso, int32_t N, int32_t terms, float* gso_pad, float* gso_t_pad, float* scratch, void* stream) {
    if (!gso || !gso_pad || !gso_t_pad || (!scratch && !gc_is_tiled(N, terms)) || N < 1 || N > 32768 || terms < 1 || terms > 9)
        return fail(STGCN_ERR_INVALID, "stgcn_gso_prepare: bad arguments (N >= 1, 1 <= terms <= 9)");
    hipStream_t st = (hipStream_t)stream;
    const int NP = gc_padded_nodes(N, terms);
    const size_t M = (size_t)NP * NP;
    const dim3 grid(cdiv((int64_t)M, kThreads)), blk(kThreads);
    if (gc_is_tiled(N, terms)) {   // dense padded operator and its transpose; the recursion runs on the activations
        STGCN_LAUNCH("gso_dense", st, gso_dense_kernel, grid, blk, 0, gso, (int)N, NP, gso_pad);
        STGCN_LAUNCH("gso_dense_t", st, gso_dense_t_kernel, grid, blk, 0, gso, (int)N, NP, gso_t_pad);
        unsigned short* hp = reinterpret_cast<unsigned short*>(gso_pad + M);
        unsigned short* ht = reinterpret_cast<unsigned short*>(gso_t_pad + M);
        const int LD = gc_plane_ld(NP);
        STGCN_LAUNCH("gso_bf16", st, gso_bf16_kernel, grid, blk, 0, gso, (int)N, NP, LD, 0, hp, hp + (size_t)NP * LD);
        STGCN_LAUNCH("gso_bf16_t", st, gso_bf16_kernel, grid, blk, 0, gso, (int)N, NP, LD, 1, ht, ht + (size_t)NP * LD);
        return STGCN_OK;
    }
    float* D[3] = {scratch, scratch + M, scratch + 2 * M};   // D[0] = L (kept), D[1] / D[2]: T_{k-1} / T_{k-2} ring
    STGCN_LAUNCH("gso_dense", st, gso_dense_kernel, grid, blk, 0, gso, (int)N, NP, D[0]);
    const float* tm1 = D[0];      // T_1
    const float* tm2 = nullptr;   // T_0 = I
    for (int k = 1; k < terms; ++k) {
        const float* tk = tm1;
        if (k >= 2) {
            float* out = (tm1 == D[1]) ? D[2] : D[1];
            STGCN_LAUNCH("cheb_next", st, cheb_next_kernel, grid, blk, 0, (const float*)D[0], tm1, tm2, (int)N, NP, out);
            tm2 = tm1;
            tm1 = out;
            tk = out;
        }
        STGCN_LAUNCH("gso_frag", st, gso_frag_kernel, grid, blk, 0, tk, NP, gso_pad + (size_t)(k - 1) * M, gso_t_pad + (size_t)(k - 1) * M);
        const size_t t16 = gs16_term_floats(NP, N), o16 = (size_t)(terms - 1) * M + (size_t)(k - 1) * t16;
        const int KC32 = gs16_np32(N) / 32;
        STGCN_LAUNCH("gso_frag16", st, gso_frag16_kernel, dim3(cdiv((int64_t)(NP / 16) * KC32 * 512, kThreads)), blk, 0, tk, NP, KC32, gso_pad + o16,
                     gso_t_pad + o16);
    }
    return STGCN_OK;
}

int stgcn_dropout_mask(float* out, int64_t n, float droprate, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
    if (!out || n < 0 || (n & 3)) return fail(STGCN_ERR_INVALID, "stgcn_dropout_mask: n must be a non-negative multiple of 4");
    if (n == 0) return STGCN_OK;
    STGCN_LAUNCH("dropout_mask", (hipStream_t)stream, dropout_mask_kernel, dim3(cdiv(n / 4, kThreads)), dim3(kThreads), 0, out,
                 (long)(n / 4), seed, offset, offset_dev, drop_thresh(droprate), 1.0f / (1.0f - droprate));
    return STGCN_OK;
}

int stgcn_stblock_ln_hook(const stgcn_stblock_desc* d, const stgcn_stblock_params* P, const float* y, float* ws, uint64_t seed, uint64_t offset,
                          const uint64_t* offset_dev, stgcn_ln_hook* h) {
    stgcn_stblock_plan pl;
    int rc = stgcn_stblock_plan_query(d, &pl);
    if (rc) return rc;
    if (!P || !P->ln_w || !P->ln_b || !y || !ws || !h) return fail(STGCN_ERR_INVALID, "stgcn_stblock_ln_hook: NULL argument");
    memset(h, 0, sizeof(*h));
    h->rowstat = ws + pl.ws_rowstat_b; h->y = y; h->gamma = P->ln_w; h->beta = P->ln_b;
    h->N = d->N; h->C = d->c2; h->training = d->training; h->droprate = d->droprate; h->dtype = d->dtype;
    h->seed = seed; h->offset = offset; h->offset_dev = offset_dev;
    return STGCN_OK;
}

int stgcn_stblock_chain_status(const stgcn_stblock_desc* d, const float* ws, uint32_t* sticky, void* stream) {
    STGCN_FLUSH_PENDING_PACK();
    stgcn_stblock_plan pl;
    int rc = stgcn_stblock_plan_query(d, &pl);
    if (rc) return rc;
    if (!ws || !sticky) return fail(STGCN_ERR_INVALID, "stgcn_stblock_chain_status: NULL argument");
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(STGCN_ERR_LAUNCH, "stgcn_stblock_chain_status: stream synchronisation failed");
    unsigned w = 0;
    if (hipMemcpy(&w, reinterpret_cast<const unsigned*>(ws + pl.ws_chain) + 2, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(STGCN_ERR_LAUNCH, "stgcn_stblock_chain_status: copy failed");
    *sticky = w;
    return STGCN_OK;
}

int stgcn_stblock_forward(const stgcn_stblock_desc* d, const stgcn_stblock_params* P, const float* x, const float* gso_pad, float* y,
                          float* saved, float* ws, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
    stgcn_stblock_plan pl;
    int rc = stgcn_stblock_plan_query(d, &pl);
    if (rc) return rc;
    if (!P || !x || !gso_pad || !y || !saved || !ws) return fail(STGCN_ERR_INVALID, "stgcn_stblock_forward: NULL buffer");
    if (!P->tc1_w || !P->tc1_b || !P->gc_w || !P->tc2_w || !P->tc2_b || !P->ln_w || !P->ln_b)
        return fail(STGCN_ERR_INVALID, "stgcn_stblock_forward: missing parameter pointer");
    if (d->c_in > d->c0 && !P->tc1_aw) return fail(STGCN_ERR_INVALID, "tc1_aw required when c_in > c0");
    if (d->c0 > d->c1 && !P->al_w) return fail(STGCN_ERR_INVALID, "al_w required when c0 > c1");
    if (d->c1 > d->c2 && !P->tc2_aw) return fail(STGCN_ERR_INVALID, "tc2_aw required when c1 > c2");
    const Derived v = derive(d);
    hipStream_t st = (hipStream_t)stream;
    g_prof_tag = d->reserved;
    g_bf16 = d->dtype == STGCN_DTYPE_BF16;

    // a parked model-level pack (stgcn_prepack): fused with this block's thin first layer when that is what comes next, else launched now
    const bool thin_first = pl.thin_tc1 && thin_wave_tiles() &&
                            !(!pl.recompute_tc1 && !d->x_bstride && !d->x_index_dev && tc1_fwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt));
    const bool fuse_pack = g_pending_pack.valid && d->prepacked && thin_first && g_pending_pack.st == st && pack_fusion_on() &&
                           d->c0 == 64 && d->c1 == 16 && d->c_in < d->c0;
    if (!fuse_pack) STGCN_FLUSH_PENDING_PACK();
    rc = d->prepacked ? STGCN_OK : launch_pack(d, P, pl, ws, st);
    if (rc) return rc;

#ifdef STGCN_EXPERIMENTS   // (round 5: the chained launch of round 4 -- measured slower, DESIGN.md section 3c -- left the product build)
    // ---- chained forward: tmp_conv1 + Align -> graph conv [-> tmp_conv2 + LayerNorm + dropout] as roles of ONE launch -----------------
    {
        const int mode = fwd_chain_mode();
        const int HT = v.NP / 16;
        const bool tc1_ts = !pl.recompute_tc1 && !d->x_bstride && !d->x_index_dev && tc1_fwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt);
        const bool slab_gc = !gc_is_tiled(d->N, v.terms) && !(g_slab_gc_precision > 0 && !g_bf16);
        const bool tc2_one = tc2_ln_fwd_fused_ok(d->c1, d->c2, d->Kt, d->N) && d->N <= 256 && v.slabs2 <= 2L * device_cus();
        if (mode > 0 && tc1_ts && slab_gc && d->c_in == 64 && d->c0 == 64 && d->Kt == 3 && d->act == STGCN_ACT_GLU && HT <= 64) {
            const bool with_tc2 = mode >= 2 && tc2_one && HT <= 16;
            Tc1FwdArgs f1;
            memset(&f1, 0, sizeof(f1));
            f1.x = x; f1.Wp = ws + pl.ws_W1p; f1.bias = ws + pl.ws_b1; f1.WaD = ws + pl.ws_WaDense; f1.ba = ws + pl.ws_ba;
            f1.U = saved + pl.sv_U1; f1.S = saved + pl.sv_S1; f1.A = saved + pl.sv_A;
            f1.B = d->B; f1.T = d->T; f1.T1 = v.T1; f1.N = d->N; f1.node_tiles = (d->N + 15) / 16;
            f1.chain_out = 0;                                   // counters [0, slabs1): node tiles of A[slab]
            GconvFwdArgs f2;
            memset(&f2, 0, sizeof(f2));
            f2.A = saved + pl.sv_A; f2.Lp = gso_pad; f2.W = P->gc_w; f2.bias = P->gc_b; f2.Xk = saved + pl.sv_Xk; f2.G = saved + pl.sv_G;
            f2.N = d->N; f2.NP = v.NP; f2.Ks = v.terms; f2.kipf = d->graph_conv == STGCN_GC_KIPF; f2.slabs = v.slabs1;
            f2.chain_in = 0; f2.chain_expect = (unsigned)f1.node_tiles;
            f2.chain_out = with_tc2 ? (int)v.slabs1 : -1;       // counters [slabs1, 2 slabs1): parts of G[slab]
            // graph-conv geometry: with the third role every workgroup reserves tmp_conv2's 1024 threads and ~60 KB (two per CU), so a slab is ONE
            // workgroup with a wave per node tile; without it the stand-alone split (4-wave parts, ~5 per CU) stays
            const GcGeom gg = gc_geom(HT, with_tc2 ? 1 : (HT + 3) / 4);
            if (gg.maxq == 1) {
                f2.parts = gg.parts;
                const int gc_threads = gg.waves * 64;
                Tc2LnFwdArgs f3;
                memset(&f3, 0, sizeof(f3));
                f3.G = saved + pl.sv_G; f3.Wp = ws + pl.ws_W2p; f3.bias = ws + pl.ws_b2; f3.gamma = P->ln_w; f3.beta = P->ln_b;
                f3.U = pl.stored_US2 ? saved + pl.sv_U2 : nullptr; f3.S = pl.stored_US2 ? saved + pl.sv_S2 : nullptr;
                f3.y = y; f3.mean = saved + pl.sv_mean; f3.rstd = saved + pl.sv_rstd;
                f3.T1 = v.T1; f3.T2 = v.T2; f3.N = d->N; f3.NPR = (int)rup(d->N, 16); f3.act = d->act; f3.training = d->training && d->droprate > 0.f;
                f3.eps = d->ln_eps; f3.keep_scale = 1.0f / (1.0f - d->droprate); f3.thresh = drop_thresh(d->droprate);
                f3.seed = seed; f3.offset = offset; f3.offset_dev = offset_dev;
                f3.chain_in = (int)v.slabs1; f3.chain_expect = (unsigned)gg.parts;
                const long items = (long)d->B * f1.node_tiles;
                static const int force_per_cu = getenv("STGCN_TC1_FWD_PER_CU") ? atoi(getenv("STGCN_TC1_FWD_PER_CU")) : 0;
                const long want = (long)device_cus() * (force_per_cu > 0 ? force_per_cu : (g_bf16 ? 2 : 1));
                const int n1 = (int)(items < want ? items : want), n2 = (int)(v.slabs1 * gg.parts), n3 = with_tc2 ? (int)v.slabs2 : 0;
                size_t lds = tc1_fwd_lds_bytes(d->c_in, d->Kt);
                const size_t l2 = gconv_fwd_lds_bytes(v.NP, 1, gg.waves, with_tc2), l3 = with_tc2 ? tc2_ln_fwd_lds_bytes(d->Kt, d->N) : 0;
                lds = lds > l2 ? lds : l2;
                lds = lds > l3 ? lds : l3;
                const int slot = (int)(lds / 4);                // the ticket's LDS word sits behind every role's own LDS
                lds += 16;
                ChainCtl cc;
                cc.words = reinterpret_cast<unsigned*>(ws + pl.ws_chain);
                cc.ncount = with_tc2 ? 2 * (int)v.slabs1 : (int)v.slabs1;
                cc.total = (unsigned)(n1 + n2 + n3);
                cc.spin = g_chain_spin_ticks;
                const dim3 grid(cc.total);
                if (with_tc2) STGCN_LAUNCH_ET("stblock_fwd", st, (stblock_fwd_chain_kernel<64, 3, 4, true, ET>), grid, dim3(1024), lds, f1, f2, f3, cc, n1, n2, gc_threads, slot);
                else STGCN_LAUNCH_ET("tc1_gconv_fwd", st, (stblock_fwd_chain_kernel<64, 3, 4, false, ET>), grid, dim3(512), lds, f1, f2, f3, cc, n1, n2, gc_threads, slot);
                if (with_tc2) return STGCN_OK;
                goto after_gconv;
            }
        }
    }
#endif

    {
    // ---- tmp_conv1 + GLU + Align(c0 -> c1) -----------------------------------------------------
    if (!pl.recompute_tc1 && !d->x_bstride && !d->x_index_dev && tc1_fwd_shape_ok(d->c_in, d->c0, d->c1, d->Kt)) {
        // time-stepping kernel: weights stationary in registers, every input tile read once (stgcn_kernels_tstep.hip.h)
        Tc1FwdArgs f;
        memset(&f, 0, sizeof(f));
        f.x = x; f.Wp = ws + pl.ws_W1p; f.bias = ws + pl.ws_b1; f.WaD = ws + pl.ws_WaDense; f.ba = ws + pl.ws_ba;
        f.U = saved + pl.sv_U1; f.S = saved + pl.sv_S1; f.A = saved + pl.sv_A;
        f.B = d->B; f.T = d->T; f.T1 = v.T1; f.N = d->N; f.node_tiles = (d->N + 15) / 16;
        const long items = (long)d->B * f.node_tiles;
        // workgroups per CU: one for fp32 (its steps are bound by the fp32 matrix pipe: a second chain on the CU made it 12 % slower), two for
        // bf16 activations (8 x cheaper MFMAs leave a latency chain: C3 34.2 -> 27.1 us, r3-31); STGCN_TC1_FWD_PER_CU forces
#ifdef STGCN_EXPERIMENTS
        static const int force_per_cu = getenv("STGCN_TC1_FWD_PER_CU") ? atoi(getenv("STGCN_TC1_FWD_PER_CU")) : 0;
#else
        constexpr int force_per_cu = 0;   // (round 5: the per-CU count is decided by the activation type; the knob left the product build)
#endif
        const int fwd_per_cu = force_per_cu > 0 ? force_per_cu : (g_bf16 ? 2 : 1);
        const long want = (long)device_cus() * fwd_per_cu;                                    // (stgcn_set_tc1_bwd_wgs overrides the CU count in tests)
        const long by_steps = items * (long)v.T1 / tc1_min_steps(), most = items > by_steps ? items : by_steps;   // (small batches: ranges cut inside items)
        const dim3 grid((unsigned)(most < want ? most : want)), blk(512);                     // equal (item, step) ranges, one workgroup per CU
        const bool x6 = mfma_x6() && !g_bf16;
        const size_t lds = tc1_fwd_lds_bytes(d->c_in, d->Kt, x6);
#define STGCN_TC1_FWD(CIN_)                                                                                   \
        do {                                                                                                  \
            if (x6 && d->act == STGCN_ACT_GLU) STGCN_LAUNCH("tconv_fwd.tc1", st, (tc1_fwd_x6_kernel<64, CIN_, 3, 0>), grid, blk, lds, f);   \
            else if (x6) STGCN_LAUNCH("tconv_fwd.tc1", st, (tc1_fwd_x6_kernel<64, CIN_, 3, 1>), grid, blk, lds, f);      \
            else if (d->act == STGCN_ACT_GLU) STGCN_LAUNCH_ET("tconv_fwd.tc1", st, (tc1_fwd_kernel<64, CIN_, 3, 0, ET>), grid, blk, lds, f);   \
            else STGCN_LAUNCH_ET("tconv_fwd.tc1", st, (tc1_fwd_kernel<64, CIN_, 3, 1, ET>), grid, blk, lds, f);      \
        } while (0)
        if (d->c_in == 64) STGCN_TC1_FWD(64); else if (d->c_in == 32) STGCN_TC1_FWD(32); else STGCN_TC1_FWD(16);
#undef STGCN_TC1_FWD
    } else if (pl.thin_tc1 && thin_wave_tiles()) {
        // thin first layer (K = Kt * c_in <= 4): one wave per 16-row tile, no LDS, no barrier (stgcn_kernels_thin.hip.h)
        ThinFwdArgs f;
        memset(&f, 0, sizeof(f));
        f.ts.src = x; f.ts.C = d->c_in; f.ts.taps = d->Kt; f.ts.N = d->N; f.ts.Tsrc = d->T; f.ts.Tdst = v.T1; f.ts.dir = 1; f.ts.rows = v.rows1;
        f.ts.bstride = d->x_bstride; f.ts.idx_dev = reinterpret_cast<const long*>(d->x_index_dev); f.ts.idx_stride = d->x_index_stride;
        f.Wd = ws + pl.ws_W1dense; f.bias = ws + pl.ws_b1; f.Wap = ws + pl.ws_Wap; f.ba = ws + pl.ws_ba; f.A = saved + pl.sv_A;
        if (fuse_pack) {
            // ONE launch: the layer (operands straight from the parameters) + the parked pack jobs (pack_thin_fwd_kernel)
            f.native = 1; f.cw = P->tc1_w; f.cb = P->tc1_b; f.aw = P->al_w; f.ab = P->al_b;
            PackList& PL = g_pending_pack.L;
            PL.pa.njobs = PL.nj;
            const int n_thin = thin_fwd_wgs(v.rows1);
            PackSync sy{0, nullptr, 0u, 0, 0};
            for (int k = 0; k < PL.pa.ncounters; ++k)
                if (f.ts.idx_dev && PL.pa.cptr[k] == f.ts.idx_dev) {   // the window index of a captured step is one of the pack's counters
                    sy.on = 1; sy.idx_ptr = f.ts.idx_dev; sy.expected = (unsigned)(4 * n_thin); sy.inc = PL.pa.cinc[k]; sy.mod = PL.pa.cmod[k];
                }
            g_pending_pack.valid = false;
            const dim3 gridf((unsigned)(n_thin + PL.pa.start[PL.nj]));
            if (d->act == STGCN_ACT_GLU) STGCN_LAUNCH_ET("tconv_fwd.tc1", st, (pack_thin_fwd_kernel<ET, 0>), gridf, dim3(256), 0, PL.pa, f, n_thin, sy);
            else STGCN_LAUNCH_ET("tconv_fwd.tc1", st, (pack_thin_fwd_kernel<ET, 1>), gridf, dim3(256), 0, PL.pa, f, n_thin, sy);
        } else
        if (d->act == STGCN_ACT_GLU) STGCN_LAUNCH_ET("tconv_fwd.tc1", st, (thin_tc1_fwd_kernel<ET, 0>), dim3(thin_fwd_wgs(v.rows1)), dim3(256), 0, f);
        else STGCN_LAUNCH_ET("tconv_fwd.tc1", st, (thin_tc1_fwd_kernel<ET, 1>), dim3(thin_fwd_wgs(v.rows1)), dim3(256), 0, f);
    } else
    {
        TconvFwdArgs t1;
        memset(&t1, 0, sizeof(t1));
        t1.ts.src = x; t1.ts.C = d->c_in; t1.ts.taps = d->Kt; t1.ts.N = d->N; t1.ts.Tsrc = d->T; t1.ts.Tdst = v.T1; t1.ts.dir = 1;
        t1.ts.rows = v.rows1;
        t1.ts.bstride = d->x_bstride; t1.ts.idx_dev = reinterpret_cast<const long*>(d->x_index_dev); t1.ts.idx_stride = d->x_index_stride;
        t1.Wp = ws + pl.ws_W1p; t1.bias = ws + pl.ws_b1; t1.KCH = v.KP1 / 16; t1.Cout = d->c0; t1.act = d->act;
        t1.U = pl.recompute_tc1 ? nullptr : saved + pl.sv_U1; t1.S = pl.recompute_tc1 ? nullptr : saved + pl.sv_S1; t1.H = nullptr;
        t1.Wap = ws + pl.ws_Wap; t1.ba = ws + pl.ws_ba; t1.A = saved + pl.sv_A; t1.c1 = d->c1;
        rc = launch_tconv_fwd("tconv_fwd.tc1", t1, st);
        if (rc) return rc;
    }

    // ---- graph conv + residual + relu -----------------------------------------------------------
    GconvFwdArgs gc;
    memset(&gc, 0, sizeof(gc));
    gc.A = saved + pl.sv_A; gc.Lp = gso_pad; gc.W = P->gc_w; gc.bias = P->gc_b;
    gc.Xk = saved + pl.sv_Xk; gc.G = saved + pl.sv_G;
    gc.N = d->N; gc.NP = v.NP; gc.Ks = v.terms; gc.kipf = d->graph_conv == STGCN_GC_KIPF; gc.slabs = v.slabs1;
    gc.XT = pl.tiled_gc && v.terms > 1 ? ws + pl.ws_XT : nullptr;
    rc = launch_gconv_fwd(gc, st);
    if (rc) return rc;
    }
#ifdef STGCN_EXPERIMENTS
after_gconv:
#endif
    if (tc2_ln_fwd_fused_ok(d->c1, d->c2, d->Kt, d->N)) {
        // ---- tmp_conv2 + GLU + LayerNorm([N, c2]) + dropout: one workgroup per (b, t2) slab ------------------------------------
        Tc2LnFwdArgs f;
        memset(&f, 0, sizeof(f));
        f.G = saved + pl.sv_G; f.Wp = ws + pl.ws_W2p; f.bias = ws + pl.ws_b2; f.gamma = P->ln_w; f.beta = P->ln_b;
        f.U = pl.stored_US2 ? saved + pl.sv_U2 : nullptr; f.S = pl.stored_US2 ? saved + pl.sv_S2 : nullptr;
        f.y = y; f.mean = saved + pl.sv_mean; f.rstd = saved + pl.sv_rstd;
        f.T1 = v.T1; f.T2 = v.T2; f.N = d->N; f.NPR = (int)rup(d->N, 16); f.act = d->act; f.training = d->training && d->droprate > 0.f;
        f.eps = d->ln_eps; f.keep_scale = 1.0f / (1.0f - d->droprate); f.thresh = drop_thresh(d->droprate);
        f.seed = seed; f.offset = offset; f.offset_dev = offset_dev;
        const bool x6 = mfma_x6() && !g_bf16 && d->N <= 256 && v.slabs2 <= 2L * device_cus() && d->Kt <= 3;   // "bf16x6" products: the 16-wave forms of up to 256 nodes (4 taps: 8 weight plane triples spill)
        const size_t lds = tc2_ln_fwd_lds_bytes(d->Kt, d->N, x6);
        // workgroups per slab (round 6): a slab is one serial chain of ~15 us whatever the batch, so a launch that leaves compute units idle
        // (block 1 of C2: 128 slabs; every small batch) cuts the slab's node tiles over PP workgroups that exchange their statistics
        const int pp = tc2_ln_peers(d->N, v.slabs2);
        f.peer.words = reinterpret_cast<unsigned*>(ws + pl.ws_chain); f.peer.ncount = (int)(pp * v.slabs2); f.peer.total = (unsigned)(pp * v.slabs2);
        f.peer.spin = g_chain_spin_ticks;
        f.peer_slots = reinterpret_cast<unsigned long long*>(ws + pl.ws_chain + kChainHdr + 2 * v.slabs1);
        const dim3 grid((unsigned)(v.slabs2 * pp));
        // 16 waves (4 tile groups) when the grid leaves room for it: at most ~2 workgroups per CU (measured at C2: 25.8 -> ?? us)
        static const int hv_force = STGCN_EXP_ENV("STGCN_TC2LN_HV") ? atoi(STGCN_EXP_ENV("STGCN_TC2LN_HV")) : 0;
        // (up to 384 nodes = 6 tiles per wave of a four-group workgroup: C3's 325-node slabs ran on 8 waves, one workgroup per CU, two rounds)
        // (the 6-tile form only for bf16 activations: with fp32 fragments it needs more than the 128 registers of a 16-wave workgroup)
        const bool wide = d->N <= (g_bf16 ? 384 : 256) && (hv_force ? hv_force == 4 : v.slabs2 <= 2L * device_cus());
        const bool small = d->N <= 224;   // 7 row tiles per wave of a two-group workgroup
#define STGCN_TC2LN(KT_)                                                                                  \
        do {                                                                                              \
            if (x6 && pp == 2) STGCN_LAUNCH("tc2_ln_fwd", st, (tc2_ln_fwd_x6_kernel<64, (KT_ <= 3 ? KT_ : 3), 2, 4, 2>), grid, dim3(1024), lds, f); \
            else if (x6 && pp == 4) STGCN_LAUNCH("tc2_ln_fwd", st, (tc2_ln_fwd_x6_kernel<64, (KT_ <= 3 ? KT_ : 3), 1, 4, 4>), grid, dim3(1024), lds, f); \
            else if (x6 && wide) STGCN_LAUNCH("tc2_ln_fwd", st, (tc2_ln_fwd_x6_kernel<64, (KT_ <= 3 ? KT_ : 3), 4, 4, 1>), grid, dim3(1024), lds, f); \
            else if (pp == 2 && d->N <= 256) STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 2, 4, 2, ET>), grid, dim3(1024), lds, f); \
            else if (pp == 4 && d->N <= 256) STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 1, 4, 4, ET>), grid, dim3(1024), lds, f); \
            else if (pp == 2) STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 3, 4, 2, ET>), grid, dim3(1024), lds, f); \
            else if (pp == 4) STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 2, 4, 4, ET>), grid, dim3(1024), lds, f); \
            else if (wide && d->N <= 256) STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 4, 4, 1, ET>), grid, dim3(1024), lds, f); \
            else if (wide) STGCN_LAUNCH("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 6, 4, 1, bf16>), grid, dim3(1024), lds, f); \
            else if (small) STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 7, 2, 1, ET>), grid, dim3(512), lds, f); \
            else STGCN_LAUNCH_ET("tc2_ln_fwd", st, (tc2_ln_fwd_kernel<64, KT_, 14, 2, 1, ET>), grid, dim3(512), lds, f);           \
        } while (0)
        if (d->Kt == 2) STGCN_TC2LN(2); else if (d->Kt == 3) STGCN_TC2LN(3); else STGCN_TC2LN(4);
#undef STGCN_TC2LN
        return STGCN_OK;
    }

    // ---- tmp_conv2 + GLU ---------------------------------------------------------------------------
    TconvFwdArgs t2;
    memset(&t2, 0, sizeof(t2));
    t2.ts.src = saved + pl.sv_G; t2.ts.C = d->c1; t2.ts.taps = d->Kt; t2.ts.N = d->N; t2.ts.Tsrc = v.T1; t2.ts.Tdst = v.T2; t2.ts.dir = 1;
    t2.ts.rows = v.rows2;
    t2.Wp = ws + pl.ws_W2p; t2.bias = ws + pl.ws_b2; t2.KCH = v.KP2 / 16; t2.Cout = d->c2; t2.act = d->act;
    t2.U = saved + pl.sv_U2; t2.S = saved + pl.sv_S2; t2.rowstat = reinterpret_cast<float2*>(saved + pl.sv_rowstat);
    rc = launch_tconv_fwd("tconv_fwd.tc2", t2, st);
    if (rc) return rc;

    // ---- LayerNorm([N, c2]) + dropout ----------------------------------------------------------------
    LnFwdArgs ln;
    memset(&ln, 0, sizeof(ln));
    ln.U = saved + pl.sv_U2; ln.S = saved + pl.sv_S2; ln.gamma = P->ln_w; ln.beta = P->ln_b; ln.y = y;
    ln.mean = saved + pl.sv_mean; ln.rstd = saved + pl.sv_rstd; ln.rowstat = reinterpret_cast<const float2*>(saved + pl.sv_rowstat);
    ln.n = d->N * d->c2; ln.N = d->N; ln.C = d->c2; ln.act = d->act; ln.training = d->training && d->droprate > 0.f;
    ln.eps = d->ln_eps; ln.keep_scale = 1.0f / (1.0f - d->droprate); ln.thresh = drop_thresh(d->droprate);
    ln.seed = seed; ln.offset = offset; ln.offset_dev = offset_dev;
    rc = launch_ln_fwd("ln_fwd", ln, v.slabs2, st);
    if (rc) return rc;
    return STGCN_OK;
}

#include "stgcn_capi_bwd.inc"
#include "stgcn_capi_head.inc"

}  // extern "C"
