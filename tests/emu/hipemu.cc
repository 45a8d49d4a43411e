// hipemu.cc -- fiber scheduler behind tests/emu/hip/hip_runtime.h (TEST INFRASTRUCTURE ONLY).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
namespace {

const size_t kStack = 64 * 1024;
const int kWave = 64;

struct Fiber {
    ucontext_t ctx;
    bool done = false;
    int coll_seq = 0;        // number of wave collectives this lane has entered
    int bar_seq = 0;         // number of block barriers this lane has entered
};

struct WaveSlot {            // one in-flight collective of one wave
    int seq = -1;
    int arrived = 0;
    int consumed = 0;
    uint64_t val[kWave];
};

ucontext_t g_sched;
std::vector<Fiber> g_fibers;
std::vector<WaveSlot> g_slots;       // 2 per wave
char *g_stacks = nullptr;
size_t g_stack_fibers = 0;
int g_cur = -1;
int g_nthreads = 0;
int g_bar_arrived = 0, g_bar_gen = 0;
std::vector<char> g_smem;
const std::function<void()> *g_body = nullptr;

void yield() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

void trampoline()
{
    (*g_body)();
    g_fibers[g_cur].done = true;
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

int wave_size(int wave)
{
    int lo = wave * kWave;
    int n = g_nthreads - lo;
    return n > kWave ? kWave : n;
}

// deposit v for this lane in the wave's current collective, wait for the whole wave,
// return the slot (values of all lanes)
WaveSlot &collective(uint64_t v)
{
    Fiber &f = g_fibers[g_cur];
    const int wave = g_cur / kWave, lane = g_cur % kWave;
    const int seq = f.coll_seq++;
    WaveSlot &s = g_slots[wave * 2 + (seq & 1)];
    if (s.seq != seq) {      // first lane to arrive at this collective: recycle the slot
        s.seq = seq;
        s.arrived = 0;
        s.consumed = 0;
    }
    s.val[lane] = v;
    s.arrived++;
    const int need = wave_size(wave);
    while (s.arrived < need) yield();
    return s;
}

} // namespace

void *dyn_smem() { return g_smem.data(); }

static int g_or_acc[2];      // __syncthreads_or: the OR of generation g collects in slot g & 1

void barrier()
{
    const int gen = g_bar_gen;
    if (++g_bar_arrived == g_nthreads) {
        g_bar_arrived = 0;
        g_or_acc[(gen + 1) & 1] = 0;     // (the next generation's slot; everyone has read its previous use by now)
        g_bar_gen++;
        return;
    }
    while (g_bar_gen == gen) yield();
}

int barrier_or(int pred)
{
    const int gen = g_bar_gen, slot = gen & 1;
    if (pred) g_or_acc[slot] = 1;
    barrier();
    return g_or_acc[slot];
}

unsigned long long ballot(int pred)
{
    WaveSlot &s = collective(pred ? 1 : 0);
    const int n = wave_size(g_cur / kWave);
    unsigned long long m = 0;
    for (int l = 0; l < n; l++) if (s.val[l]) m |= 1ull << l;
    return m;
}

uint64_t shuffle(uint64_t v, int arg, int width, int mode)
{
    WaveSlot &s = collective(v);
    const int lane = g_cur % kWave;
    const int base = lane & ~(width - 1);
    const int rel = lane - base;
    int src;
    switch (mode) {
    case 0: src = arg & (width - 1); break;
    case 1: src = rel + arg; if (src >= width) src = rel; break;
    case 2: src = rel - arg; if (src < 0) src = rel; break;
    default: src = (rel ^ arg); if (src >= width) src = rel; break;
    }
    return s.val[base + src];
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
{
    const int nt = (int)(block.x * block.y * block.z);
    if (nt <= 0 || grid.x == 0) return;
    if ((size_t)nt > g_stack_fibers) {
        if (g_stacks) munmap(g_stacks, g_stack_fibers * kStack);
        g_stacks = (char *)mmap(nullptr, (size_t)nt * kStack, PROT_READ | PROT_WRITE,
                                MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        g_stack_fibers = (size_t)nt;
    }
    g_smem.assign(shmem + 64, 0);
    g_nthreads = nt;
    blockDim = block;
    gridDim = grid;
    g_body = &body;
    const int nwaves = (nt + kWave - 1) / kWave;
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        blockIdx = dim3(bx, by, 0);
        g_fibers.assign((size_t)nt, Fiber());
        g_slots.assign((size_t)nwaves * 2, WaveSlot());
        g_bar_arrived = 0;
        g_bar_gen = 0;
        for (int t = 0; t < nt; t++) {
            Fiber &f = g_fibers[t];
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * kStack;
            f.ctx.uc_stack.ss_size = kStack;
            f.ctx.uc_link = nullptr;
            makecontext(&f.ctx, trampoline, 0);
        }
        int remaining = nt;
        while (remaining > 0) {
            int progressed = 0;
            for (int t = 0; t < nt; t++) {
                Fiber &f = g_fibers[t];
                if (f.done) continue;
                g_cur = t;
                threadIdx = dim3((unsigned)(t % (int)block.x), (unsigned)(t / (int)block.x), 0);
                swapcontext(&g_sched, &f.ctx);
                if (f.done) { remaining--; }
                progressed++;
            }
            if (!progressed) break;
        }
    }
    g_body = nullptr;
}

} // namespace hipemu
