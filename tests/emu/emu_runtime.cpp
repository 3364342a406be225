// Fiber scheduler behind tests/emu/hip/hip_runtime.h  --  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <limits>

namespace stgcn { alignas(64) float stgcn_smem[emu::kLdsBytes / sizeof(float)]; }

namespace emu {
State g;

static void fiber_entry() {
    g.body();
    g.cur->done = true;
    swapcontext(&g.cur->ctx, &g.main_ctx);
}

void yield() { swapcontext(&g.cur->ctx, &g.main_ctx); }

void block_barrier() {
    const unsigned my = g.bar_gen;
    if (++g.bar_arrived == g.nthreads) {
        g.bar_arrived = 0;
        g.bar_gen++;
        return;
    }
    while (g.bar_gen == my) yield();
}

void wave_barrier() {
    WaveState& w = g.waves[g.cur->tid / kWave];
    const unsigned my = w.gen;
    if (++w.arrived == kWave) {
        w.arrived = 0;
        w.gen++;
        return;
    }
    while (w.gen == my) yield();
}

void run_block() {
    const int n = g.nthreads;
    constexpr size_t kStack = 256 * 1024;
    if ((int)g.fibers.size() != n) {
        g.fibers.assign(n, Fiber());
        for (auto& f : g.fibers) f.stack.resize(kStack);
    }
    g.waves.assign(n / kWave, WaveState());
    g.bar_arrived = 0;
    // poison LDS so that reads of never-written shared memory show up as NaN
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    for (size_t i = 0; i < kLdsBytes / sizeof(float); ++i) stgcn::stgcn_smem[i] = qnan;
    for (int t = 0; t < n; ++t) {
        Fiber& f = g.fibers[t];
        f.done = false;
        f.tid = (unsigned)t;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.data();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link = &g.main_ctx;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    int alive = n;
    long rounds = 0;
    while (alive > 0) {
        alive = 0;
        // alternate the sweep direction so that a missing barrier is exposed whichever side the
        // producer thread sits on
        const bool rev = (rounds++ & 1);
        for (int i = 0; i < n; ++i) {
            const int t = rev ? n - 1 - i : i;
            Fiber& f = g.fibers[t];
            if (f.done) continue;
            g.cur = &f;
            g.threadIdx_ = dim3((unsigned)t, 0, 0);
            swapcontext(&g.main_ctx, &f.ctx);
            if (!f.done) ++alive;
        }
        if (rounds > 50000000L) { fprintf(stderr, "emu: deadlock suspected\n"); abort(); }
    }
}
}  // namespace emu
