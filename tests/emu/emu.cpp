// TEST INFRASTRUCTURE ONLY — wavefront emulator runtime (see hip_shim.h).
#include <chrono>

#include "hip_shim.h"

double emu_now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

unsigned long long emu_ticks_100mhz() { return (unsigned long long)(emu_now() * 1e8); }

namespace emu {
thread_local Idx t_threadIdx{0, 0, 0}, t_blockIdx{0, 0, 0}, t_blockDim{1, 1, 1}, t_gridDim{1, 1, 1};

// Fiber switch.  glibc's swapcontext saves and restores the signal mask with two system calls per switch, and the
// emulator switches 64 fibers at every wave collective: a third of the CPU test-suite's time was sigprocmask.  On x86-64
// (outside sanitizer builds, whose runtime wants to see the ucontext calls) the switch is six pushes and a stack-pointer swap.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(EMU_UCONTEXT)
#define EMU_FAST_SWITCH 1
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
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch,.-emu_switch
)");
#endif

namespace {
constexpr int kLanes = 64;
constexpr size_t kStack = 512 * 1024;
struct Sched {
#ifdef EMU_FAST_SWITCH
    void* main_sp = nullptr;
    void* sp[kLanes];
#else
    ucontext_t main_ctx;
    ucontext_t ctx[kLanes];
#endif
    char* stacks[kLanes] = {nullptr};
    int state[kLanes];   // 0 runnable, 1 waiting at a rendezvous, 2 done
    int site[kLanes];
    void* ra[kLanes];
    uint64_t buf[2][kLanes];
    int par[kLanes];
    int gen = 0;
    int cur = -1;
    bool wave_mode = false;
    const std::function<void()>* body = nullptr;
};
thread_local Sched* g = nullptr;

#ifdef EMU_FAST_SWITCH
inline void to_main(int lane) { emu_switch(&g->sp[lane], g->main_sp); }
inline void to_lane(int lane) { emu_switch(&g->main_sp, g->sp[lane]); }
#else
inline void to_main(int lane) { swapcontext(&g->ctx[lane], &g->main_ctx); }
inline void to_lane(int lane) { swapcontext(&g->main_ctx, &g->ctx[lane]); }
#endif

void fiber_entry() {
    (*g->body)();
    g->state[g->cur] = 2;
    to_main(g->cur);
    abort();      // a finished fiber is never resumed
}

void rendezvous(uint64_t contribution, int site, void* ra) {
    if (!g || !g->wave_mode) {
        fprintf(stderr, "emu: wave collective outside a 64-lane launch\n");
        abort();
    }
    int lane = g->cur;
    g->par[lane] = g->gen & 1;
    g->buf[g->par[lane]][lane] = contribution;
    g->site[lane] = site;
    g->ra[lane] = ra;
    g->state[lane] = 1;
    to_main(lane);
    // resumed: every lane has arrived
}
}  // namespace

__attribute__((noinline)) void collective_begin(uint64_t contribution, int site) { rendezvous(contribution, site, __builtin_return_address(0)); }
uint64_t collective_read(int lane) { return g->buf[g->par[g->cur]][lane & 63]; }
__attribute__((noinline)) void fence_rendezvous() {
    if (g && g->wave_mode) rendezvous(0, 99, __builtin_return_address(0));
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    Sched* prev = g;
    static thread_local Sched sched;
    g = &sched;
    g->body = &body;
    t_blockDim = Idx{block.x, block.y, block.z};
    t_gridDim = Idx{grid.x, grid.y, grid.z};
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned b = 0; b < grid.x; b++) {
        t_blockIdx = Idx{b, by, 0};
        if (block.x != (unsigned)kLanes) {
            g->wave_mode = false;
            for (unsigned t = 0; t < block.x; t++) {
                t_threadIdx = Idx{t, 0, 0};
                body();
            }
            continue;
        }
        g->wave_mode = true;
        for (int l = 0; l < kLanes; l++) {
            if (!g->stacks[l]) g->stacks[l] = (char*)malloc(kStack);
#ifdef EMU_FAST_SWITCH
            {   // initial frame: six callee-saved registers (zero), then the address emu_switch's `ret` jumps to; the
                // stack pointer is 16-byte aligned + 8 at fiber_entry's first instruction, as after a call
                uintptr_t top = ((uintptr_t)g->stacks[l] + kStack) & ~(uintptr_t)15;
                void** f = (void**)(top - 8 - 7 * sizeof(void*));
                for (int i = 0; i < 6; i++) f[i] = nullptr;
                f[6] = (void*)&fiber_entry;
                g->sp[l] = (void*)f;
            }
#else
            getcontext(&g->ctx[l]);
            g->ctx[l].uc_stack.ss_sp = g->stacks[l];
            g->ctx[l].uc_stack.ss_size = kStack;
            g->ctx[l].uc_link = &g->main_ctx;
            makecontext(&g->ctx[l], fiber_entry, 0);
#endif
            g->state[l] = 0;
        }
        for (;;) {
            int n_done = 0, n_wait = 0, first_site = -1;
            bool mismatch = false;
            for (int l = 0; l < kLanes; l++) {
                if (g->state[l] == 2) { n_done++; continue; }
                g->cur = l;
                t_threadIdx = Idx{(unsigned)l, 0, 0};
                g->state[l] = 0;
                to_lane(l);
                if (g->state[l] == 2) { n_done++; continue; }
                n_wait++;
                if (first_site < 0) first_site = g->site[l];
                else if (first_site != g->site[l]) mismatch = true;
            }
            if (n_done == kLanes) break;
            if (n_done != 0 && n_wait != 0) {
                fprintf(stderr, "emu: %d lanes left the kernel while %d wait at a collective (site %d)\n", n_done, n_wait, first_site);
                abort();
            }
            if (mismatch) {
                fprintf(stderr, "emu: lanes are at different collectives (divergent wave-level call); block %u sites:", b);
                for (int l = 0; l < kLanes; l++) fprintf(stderr, " %d@%p", g->state[l] == 2 ? -1 : g->site[l], g->ra[l]);
                fprintf(stderr, "\n");
                abort();
            }
            g->gen++;
        }
        g->wave_mode = false;
    }
    g = prev;
}
}  // namespace emu
