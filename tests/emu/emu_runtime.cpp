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
    if (++g.bar_arrived == g.live) {
        g.bar_arrived = 0;
        g.bar_gen++;
        return;
    }
    wait_for(&g.bar_gen, my);
}

void wave_barrier() {
    WaveState& w = g.waves[g.cur->tid / kWave];
    const unsigned my = w.gen;
    if (++w.arrived == kWave) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    wait_for(&w.gen, my);
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
            emu_switch(&g.main_sp, f.sp);
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
