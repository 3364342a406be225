// Fiber scheduler behind tests/emu/hip/hip_runtime.h  --  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <limits>

#if !defined(__x86_64__)
#error "the emulator's context switch is written for x86-64 (System V ABI)"
#endif

namespace stgcn { alignas(64) float stgcn_smem[emu::kLdsBytes / sizeof(float)]; }

// emu_switch(&save_sp, new_sp): push the callee-saved registers and the FP control words, store the stack pointer, continue on new_sp.
// (swapcontext() costs two sigprocmask system calls per switch; a wave-level rendezvous of the emulated MFMA / shuffle instructions
// switches 128 times, so the kernels with ~1000 MFMAs per workgroup spent seconds in the kernel.)
extern "C" void emu_switch(void** save_sp, void* new_sp);
asm(R"(
    .text
    .globl emu_switch
    .type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");

namespace emu {
State g;

static void fiber_entry() {
    g.body();
    g.cur->done = true;
    // a finished thread no longer takes part in workgroup barriers (the hardware does not count ended waves in s_barrier: the roles of a
    // chained launch that are narrower than the launch's block let their surplus waves return at once)
    if (--g.live > 0 && g.bar_arrived == g.live) {
        g.bar_arrived = 0;
        g.bar_gen++;
    }
    emu_switch(&g.cur->sp, g.main_sp);
    abort();   // a finished fiber is never resumed
}

void yield() { emu_switch(&g.cur->sp, g.main_sp); }

// block until *gen moves on from `my`: the scheduler skips the fiber meanwhile (no switch into a fiber that would only yield again)
static void wait_for(const unsigned* gen, unsigned my) {
    Fiber* f = g.cur;
    f->wait_gen = gen;
    f->wait_val = my;
    while (*gen == my) yield();
    f->wait_gen = nullptr;
}

void block_barrier() {
    const unsigned my = g.bar_gen;
    const unsigned me = g.cur->tid;
    if (++g.bar_arrived == g.live) {
        g.bar_arrived = 0;
        g.bar_gen++;
    } else {
        wait_for(&g.bar_gen, my);
    }
    if (me < g.bpass.size()) g.bpass[me]++;      // (race check: this thread is past one more workgroup barrier)
}

void wave_barrier() {
    const unsigned me = g.cur->tid;
    WaveState& w = g.waves[me / kWave];
    const unsigned my = w.gen;
    if (++w.arrived == kWave) {
        w.arrived = 0;
        w.gen++;
    } else {
        wait_for(&w.gen, my);
    }
    if (me < g.wpass.size()) g.wpass[me]++;
}

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define STGCN_EMU_HAS_ASAN 1
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

static void fiber_init(Fiber& f) {
#ifdef STGCN_EMU_HAS_ASAN
    // a fiber abandoned in the middle of a frame (peer_defer, the end of a workgroup) leaves its frames' redzones poisoned in this memory
    __asan_unpoison_memory_region(f.stack.data(), f.stack.size());
#endif
    // initial frame, as emu_switch expects to find it: [mxcsr | x87 cw] r15 r14 r13 r12 rbx rbp <return address = fiber_entry>; after the
    // `ret` the stack pointer is 8 below a 16-byte boundary, as at any function entry
    uintptr_t top = reinterpret_cast<uintptr_t>(f.stack.data() + f.stack.size()) & ~uintptr_t(15);
    uint64_t* sp = reinterpret_cast<uint64_t*>(top - 16);
    *sp = reinterpret_cast<uint64_t>(&fiber_entry);
    for (int i = 0; i < 6; ++i) *--sp = 0;
    --sp;
    reinterpret_cast<uint32_t*>(sp)[0] = 0x1F80;   // MXCSR: default rounding, exceptions masked
    reinterpret_cast<uint32_t*>(sp)[1] = 0x037F;   // x87 control word
    f.sp = sp;
}

// ================================================================================================
// LDS race check (tests/emu/build_emu.py --race; ADVICE r4 asked for "an emulator or TSAN-style test").  The kernel sources are compiled
// with -fsanitize=thread, which makes clang call __tsan_read* / __tsan_write* before every memory access -- and those hooks are OURS (no
// ThreadSanitizer runtime is linked): an access to the emulated LDS array is checked against the last write / the last reads of the same
// 4-byte word.  Two accesses by different threads, at least one a write, are ORDERED if a workgroup barrier lies between them (the later
// thread has passed more __syncthreads() than the earlier one had when it made its access) or, inside one wave, a wave-level meeting point
// (wave_lds_sync, an MFMA, a shuffle).  Anything else between two WAVES is a race on the hardware, whatever order the emulator's fibers
// happened to run in -- e.g. tc2_bwd_kernel re-staging its G tiles for the next item while a slower wave still reads them.
// Conflicts between lanes of ONE wave are in-order LDS instructions of one instruction stream on the hardware; they are only counted.
// ================================================================================================
#ifdef STGCN_EMU_RACE
#define EMU_NOSAN __attribute__((no_sanitize("thread")))
namespace {
constexpr unsigned kNone = 0xffffffffu, kMulti = 0xfffffffeu;
struct Shadow { unsigned wt, wb, ww, rt, rwave, rb, rw; };
Shadow g_shadow[kLdsBytes / 4];
bool g_race_on = false;
long g_launch_races = 0;
EMU_NOSAN void race_word(size_t word, bool is_write) {
    const unsigned t = g.cur->tid, v = t / kWave, pb = g.bpass[t], pw = g.wpass[t];
    Shadow& s = g_shadow[word];
    auto conflict = [&](unsigned ot, unsigned owave, unsigned ob, unsigned ow, const char* what) EMU_NOSAN {
        if (ot == kNone || ot == t) return;
        if (ob < pb) return;                                   // a workgroup barrier lies between
        if (owave == v) {                                      // same wave: ordered by a wave-level meeting point, else lockstep on the hardware
            if (!(ow < pw)) g.races_intra++;
            return;
        }
        if (g_launch_races++ < 8)
            fprintf(stderr, "emu-race: %s, workgroup %u: LDS word %zu: %s by thread %u (wave %u) and an earlier %s by thread %s%u (wave %u) with no workgroup "
                    "barrier between them (both after %u barriers)\n", g.kname, g.blockIdx_.x, word, is_write ? "write" : "read", t, v, what,
                    ot == kMulti ? "(several) " : "", ot == kMulti ? 0u : ot, owave, pb);
        g.races++;
    };
    conflict(s.wt, s.wt == kNone ? 0 : s.wt / kWave, s.wb, s.ww, "write");
    if (is_write) {
        conflict(s.rt, s.rwave, s.rb, s.rw, "read");
        s.wt = t; s.wb = pb; s.ww = pw;
        s.rt = kNone;
    } else if (s.rt == kNone || s.rb < pb) {
        s.rt = t; s.rwave = v; s.rb = pb; s.rw = pw;
    } else {
        if (s.rt != t) s.rt = kMulti;
        if (s.rwave != v) s.rwave = kMulti;                    // readers of several waves: never "same wave" for a later writer
        if (pw > s.rw) s.rw = pw;
    }
}
EMU_NOSAN void race_access(const void* addr, size_t size, bool is_write) {
    if (!g_race_on) return;
    const uintptr_t a = reinterpret_cast<uintptr_t>(addr), lo = reinterpret_cast<uintptr_t>(stgcn::stgcn_smem);
    if (a < lo || a >= lo + kLdsBytes) return;
    for (size_t w = (a - lo) / 4, we = (a - lo + size + 3) / 4; w < we && w < kLdsBytes / 4; ++w) race_word(w, is_write);
}
}  // namespace
EMU_NOSAN bool same_bytes(const void* a, const void* b, size_t n) { return memcmp(a, b, n) == 0; }
EMU_NOSAN void race_block_begin() {
    const size_t words = (g.lds_used + 3) / 4 < kLdsBytes / 4 ? (g.lds_used + 3) / 4 : kLdsBytes / 4;
    for (size_t i = 0; i < words; ++i) { g_shadow[i].wt = kNone; g_shadow[i].rt = kNone; }
}
void race_launch_end() {
    if (g_launch_races > 0) {
        fprintf(stderr, "emu-race: %ld cross-wave LDS race(s) in %s\n", g_launch_races, g.kname);
        g_launch_races = 0;
        if (!getenv("STGCN_EMU_RACE_WARN")) abort();
    }
}
static void race_enable(bool on) { g_race_on = on; }
extern "C" {
EMU_NOSAN void __tsan_init() {}
EMU_NOSAN void __tsan_func_entry(void*) {}
EMU_NOSAN void __tsan_func_exit() {}
EMU_NOSAN void __tsan_vptr_update(void**, void*) {}
EMU_NOSAN void __tsan_vptr_read(void**) {}
#define EMU_RW(N)                                                                           \
    EMU_NOSAN void __tsan_read##N(void* a) { emu::race_access(a, N, false); }               \
    EMU_NOSAN void __tsan_write##N(void* a) { emu::race_access(a, N, true); }               \
    EMU_NOSAN void __tsan_unaligned_read##N(void* a) { emu::race_access(a, N, false); }     \
    EMU_NOSAN void __tsan_unaligned_write##N(void* a) { emu::race_access(a, N, true); }     \
    EMU_NOSAN void __tsan_read##N##_pc(void* a, void*) { emu::race_access(a, N, false); }   \
    EMU_NOSAN void __tsan_write##N##_pc(void* a, void*) { emu::race_access(a, N, true); }
EMU_RW(1) EMU_RW(2) EMU_RW(4) EMU_RW(8) EMU_RW(16)
// (function-local statics: the guard check is an atomic load under -fsanitize=thread.  Plain volatile accesses here: atomics are instrumented
//  even inside no_sanitize functions -- __atomic_load_n in these bodies would call the hook it implements; one thread, x86: a plain load is enough)
EMU_NOSAN char __tsan_atomic8_load(const volatile char* a, int) { return *a; }
EMU_NOSAN void __tsan_atomic8_store(volatile char* a, char v, int) { *a = v; }
EMU_NOSAN int __tsan_atomic32_load(const volatile int* a, int) { return *a; }
EMU_NOSAN void __tsan_atomic32_store(volatile int* a, int v, int) { *a = v; }
EMU_NOSAN long __tsan_atomic64_load(const volatile long* a, int) { return *a; }
EMU_NOSAN void __tsan_atomic64_store(volatile long* a, long v, int) { *a = v; }
EMU_NOSAN void __tsan_read_range(void* a, size_t n) { emu::race_access(a, n, false); }
EMU_NOSAN void __tsan_write_range(void* a, size_t n) { emu::race_access(a, n, true); }
// (a copy that leaves the destination as it is -- the pipelined GEMM's copy slots past the last block repeat that block: "same bytes to the
//  same place" from another wave, benign by construction -- is checked as a read of the destination, not as a write)
EMU_NOSAN void* __tsan_memcpy(void* d, const void* s, size_t n) {
    emu::race_access(s, n, false);
    emu::race_access(d, n, memcmp(d, s, n) != 0);
    return memcpy(d, s, n);
}
EMU_NOSAN void* __tsan_memmove(void* d, const void* s, size_t n) { emu::race_access(s, n, false); emu::race_access(d, n, true); return memmove(d, s, n); }
EMU_NOSAN void* __tsan_memset(void* d, int c, size_t n) { emu::race_access(d, n, true); return memset(d, c, n); }
}
#else
bool same_bytes(const void*, const void*, size_t) { return false; }
void race_block_begin() {}
void race_launch_end() {}
static void race_enable(bool) {}
#endif

void peer_defer() {
    g.defer_req = true;
    for (;;) yield();   // (the scheduler abandons the workgroup: this fiber is never resumed)
}

bool run_block() {
    const int n = g.nthreads;
    constexpr size_t kStack = 256 * 1024;
    while ((int)g.fibers.size() < n) {
        g.fibers.emplace_back();
        g.fibers.back().stack.resize(kStack);
    }
    g.waves.assign(n / kWave, WaveState());
    g.bar_arrived = 0;
    g.live = n;
    g.defer_req = false;
    g.pub_seen = 0;
    g.bpass.assign(n, 0u);
    g.wpass.assign(n, 0u);
    race_block_begin();
    // poison LDS so that reads of never-written shared memory show up as NaN
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    for (size_t i = 0; i < kLdsBytes / sizeof(float); ++i) stgcn::stgcn_smem[i] = qnan;
    for (int t = 0; t < n; ++t) {
        Fiber& f = g.fibers[t];
        f.done = false;
        f.tid = (unsigned)t;
        f.wait_gen = nullptr;
        fiber_init(f);
    }
    int alive = n;
    long rounds = 0;
    while (alive > 0) {
        alive = 0;
        int resumed = 0;
        // alternate the sweep direction so that a missing barrier is exposed whichever side the
        // producer thread sits on
        const bool rev = (rounds++ & 1);
        for (int i = 0; i < n; ++i) {
            const int t = rev ? n - 1 - i : i;
            Fiber& f = g.fibers[t];
            if (f.done) continue;
            if (f.wait_gen && *f.wait_gen == f.wait_val) { ++alive; continue; }   // still blocked
            g.cur = &f;
            g.threadIdx_ = dim3((unsigned)t, 0, 0);
            race_enable(true);
            emu_switch(&g.main_sp, f.sp);
            race_enable(false);
            if (g.defer_req) return false;   // (the fibers are re-initialised by the next run_block)
            ++resumed;
            if (!f.done) ++alive;
        }
        if (alive > 0 && resumed == 0) {
            fprintf(stderr, "emu: deadlock: %d threads wait at barriers that nobody will complete (divergent __syncthreads / wave op?)\n", alive);
            abort();
        }
    }
    return true;
}
}  // namespace emu
