// hipemu runtime: fibers (hand-rolled x86-64 context switch), block scheduler, barriers, wave exchange.
// TEST INFRASTRUCTURE ONLY -- see hipemu.h.
#include "hipemu.h"

#include <sys/mman.h>

#include <atomic>
#include <thread>
#include <vector>

// void hipemu_switch(void** save_sp, void* load_sp): save callee-saved regs on the current stack,
// store rsp to *save_sp, switch to load_sp, restore regs, return on the new stack.
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
extern "C" void hipemu_switch(void** save_sp, void* load_sp);

namespace hipemu {

thread_local Lane* cur = nullptr;
static thread_local void* sched_sp = nullptr;

static constexpr size_t kStack = 256 * 1024;

static inline void yield_to_sched() {
    Lane* me = cur;
    hipemu_switch(&me->sp, sched_sp);
    cur = me;
}

static void lane_exit_accounting(Lane* l) {
    Block* b = l->blk;
    WaveState& w = b->waves[l->wave];
    b->alive--;
    w.alive--;
    b->progress++;
    // a finished lane no longer takes part in barriers: release any barrier it was the last hold-out of
    if (b->alive > 0 && b->arrived == b->alive) { b->arrived = 0; b->gen++; }
    if (w.alive > 0 && w.arrived == w.alive) { w.arrived = 0; w.gen++; }
}

static void fiber_entry() {
    Lane* l = cur;
    (*l->blk->body)();
    l->done = true;
    lane_exit_accounting(l);
    hipemu_switch(&l->sp, sched_sp);
    abort();   // never resumed
}

void syncthreads() {
    Block* b = cur->blk;
    unsigned g = b->gen;
    if (++b->arrived == b->alive) {
        b->arrived = 0;
        b->gen++;
        b->progress++;
        return;
    }
    while (b->gen == g) yield_to_sched();
}

void wave_sync() {
    WaveState& w = cur->blk->waves[cur->wave];
    unsigned g = w.gen;
    if (++w.arrived == w.alive) {
        w.arrived = 0;
        w.gen++;
        cur->blk->progress++;
        return;
    }
    while (w.gen == g) yield_to_sched();
}

const unsigned char (*wave_allgather(const void* in, size_t bytes))[64] {
    WaveState& w = cur->blk->waves[cur->wave];
    unsigned par = w.gen & 1;
    memcpy(w.xbuf[par][cur->lane], in, bytes);
    wave_sync();
    // the buffer of parity `par` is rewritten only two wave ops later, and every lane passes the
    // next op's sync (parity par^1) first, so reading it after this sync is race-free.
    return w.xbuf[par];
}

float atomic_add(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        float s = f + v;
        memcpy(&nw, &s, 4);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}
double atomic_add(double* p, double v) {
    unsigned long long* u = reinterpret_cast<unsigned long long*>(p);
    unsigned long long old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    double f;
    do {
        memcpy(&f, &old, 8);
        double s = f + v;
        memcpy(&nw, &s, 8);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}
int atomic_add(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
unsigned atomic_add(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
float atomic_max(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do {
        memcpy(&f, &old, 4);
        float s = f > v ? f : v;
        memcpy(&nw, &s, 4);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}

struct Worker {
    std::vector<Lane> lanes;
    char* stacks = nullptr;
    size_t nstacks = 0;
    char* smem = nullptr;
    size_t smem_cap = 0;
    Block blk;

    ~Worker() {
        if (stacks) munmap(stacks, nstacks * kStack);
        free(smem);
    }

    void run_block(dim3 grid, dim3 bdim, unsigned bx, unsigned by, unsigned bz, size_t smem_bytes,
                   const std::function<void()>& body) {
        int n = int(bdim.x * bdim.y * bdim.z);
        if ((size_t)n > nstacks) {
            if (stacks) munmap(stacks, nstacks * kStack);
            nstacks = n;
            stacks = (char*)mmap(nullptr, nstacks * kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (stacks == MAP_FAILED) { perror("hipemu mmap"); abort(); }
            lanes.resize(n);
        }
        if (smem_bytes + 64 > smem_cap) {
            free(smem);
            smem_cap = smem_bytes + 64;
            smem = (char*)aligned_alloc(64, (smem_cap + 63) / 64 * 64);
        }
        memset(smem, 0xA5, smem_bytes);      // poison: LDS is uninitialised on hardware
        blk.bid = {bx, by, bz};
        blk.bdim = {bdim.x, bdim.y, bdim.z};
        blk.gdim = {grid.x, grid.y, grid.z};
        blk.smem = smem;
        blk.nthreads = blk.alive = n;
        blk.arrived = 0;
        blk.gen = 0;
        blk.progress = 0;
        blk.body = &body;
        int nw = (n + 63) / 64;
        if (nw > 16) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
        for (int w = 0; w < nw; ++w) {
            int c = (w == nw - 1) ? n - w * 64 : 64;
            blk.waves[w].nlanes = blk.waves[w].alive = c;
            blk.waves[w].arrived = 0;
            blk.waves[w].gen = 0;
        }
        for (int i = 0; i < n; ++i) {
            Lane& l = lanes[i];
            l.stack = stacks + (size_t)i * kStack;
            l.lin = i;
            l.tid = {unsigned(i % bdim.x), unsigned((i / bdim.x) % bdim.y), unsigned(i / (bdim.x * bdim.y))};
            l.wave = i / 64;
            l.lane = i % 64;
            l.done = false;
            l.blk = &blk;
            // initial frame: [6 callee-saved regs][ret -> fiber_entry][fake return address]
            uintptr_t top = (uintptr_t)(l.stack + kStack) & ~(uintptr_t)15;
            void** sp = (void**)(top - 8);
            *sp = nullptr;                    // fake return address (keeps rsp % 16 == 8 at entry)
            *--sp = (void*)&fiber_entry;
            for (int r = 0; r < 6; ++r) *--sp = nullptr;
            l.sp = sp;
        }
        while (blk.alive > 0) {
            unsigned long before = blk.progress;
            for (int i = 0; i < n; ++i) {
                Lane& l = lanes[i];
                if (l.done) continue;
                cur = &l;
                hipemu_switch(&sched_sp, l.sp);
            }
            if (blk.alive > 0 && blk.progress == before) {
                fprintf(stderr, "hipemu: DEADLOCK in block (%u,%u,%u): a barrier/wave op is not reached by all live lanes\n", bx, by, bz);
                abort();
            }
        }
        cur = nullptr;
    }
};

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("HIPEMU_THREADS")) hw = (unsigned)atoi(e);
    size_t nt = hw ? hw : 1;
    if (nt > nblocks) nt = nblocks;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Worker wk;
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            unsigned bx = unsigned(b % grid.x), by = unsigned((b / grid.x) % grid.y), bz = unsigned(b / ((size_t)grid.x * grid.y));
            wk.run_block(grid, block, bx, by, bz, smem_bytes, body);
        }
    };
    if (nt == 1) { work(); return; }
    std::vector<std::thread> th;
    for (size_t i = 0; i < nt; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}

}  // namespace hipemu
