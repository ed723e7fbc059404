// TEST INFRASTRUCTURE ONLY -- NOT PART OF THE PRODUCT.  See hip/hip_runtime.h in this directory.
// Fiber scheduler + grid runner for the host emulation of HIP kernels.
#include <hip/hip_runtime.h>

#include <mutex>
#include <sys/mman.h>

namespace hipemu {

thread_local Idx t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local Worker* t_worker = nullptr;

// x86-64 SysV context switch: saves callee-saved registers on the current stack, stores the stack
// pointer through save_sp, installs new_sp and returns into the other fiber.
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch, .-hipemu_switch
)");

static void fiber_main() {
    Worker* w = t_worker;
    (*w->body)();
    w = t_worker;
    w->fibers[w->cur].state = DONE;
    yield_to_scheduler();
    fprintf(stderr, "hipemu: resumed a finished fiber\n");
    abort();
}

static void prepare_fiber(Fiber& f, unsigned tid) {
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + Worker::kStack) & ~uintptr_t(15);
    void** sp = reinterpret_cast<void**>(top - 64);
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    sp[6] = reinterpret_cast<void*>(&fiber_main);
    sp[7] = nullptr;
    f.sp = sp;
    f.state = RUNNABLE;
    f.tid = tid;
    f.dma.clear();   // (LDS-DMA pieces a finished kernel never waited for)
}

static void run_block(Worker& w, dim3 block) {
    const int n = w.nthreads;
    const int nwaves = (n + 63) / 64;
    w.blk_arrived = 0;
    for (int i = 0; i < n; ++i) prepare_fiber(w.fibers[i], (unsigned)i);
    for (int v = 0; v < nwaves; ++v) {
        w.waves[v].arrived = 0;
        w.waves[v].lanes = (v == nwaves - 1) ? n - 64 * v : 64;
    }
    int ndone = 0;
    long guard = 0;
    while (ndone < n) {
        bool any = false;
        for (int v = 0; v < nwaves; ++v) {
            bool active = true;
            while (active) {
                active = false;
                for (int l = 0; l < w.waves[v].lanes; ++l) {
                    int i = v * 64 + l;
                    Fiber& f = w.fibers[i];
                    if (f.state != RUNNABLE) continue;
                    any = true;
                    w.cur = i;
                    t_threadIdx.x = i % block.x;
                    t_threadIdx.y = (i / block.x) % block.y;
                    t_threadIdx.z = i / (block.x * block.y);
                    hipemu_switch(&w.sched_sp, f.sp);
                    if (f.state == DONE) ++ndone;
                    else if (f.state == RUNNABLE) active = true;
                }
                if (++guard > (1L << 40)) { fprintf(stderr, "hipemu: livelock (divergent barrier?)\n"); abort(); }
            }
        }
        if (!any && ndone < n) {
            fprintf(stderr, "hipemu: deadlock -- %d of %d work-items wait at a barrier the others never reach\n",
                    n - ndone, n);
            abort();
        }
    }
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nblocks <= 0 || nthreads <= 0) return;
    // HIPEMU_NOEXEC=1: launches return at once -- what is left of a call is the HOST side of the engine (plan logic, tile
    // selection, launch bookkeeping): tools/host_overhead.py times it on a machine without a GPU
    static const bool noexec = [] { const char* e = getenv("HIPEMU_NOEXEC"); return e && atoi(e) != 0; }();
    if (noexec) return;
    int nworkers = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("HIPEMU_THREADS")) nworkers = atoi(e);
    if (nworkers < 1) nworkers = 1;
    if (nworkers > nblocks) nworkers = (int)nblocks;
    std::atomic<long> next{0};

    auto work = [&]() {
        Worker w;
        w.nthreads = nthreads;
        w.body = &body;
        w.fibers.resize(nthreads);
        w.waves.resize((nthreads + 63) / 64);
        size_t bytes = Worker::kStack * (size_t)nthreads;
        char* arena = (char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (arena == (char*)MAP_FAILED) { perror("hipemu mmap"); abort(); }
        for (int i = 0; i < nthreads; ++i) w.fibers[i].stack = arena + Worker::kStack * (size_t)i;
        Worker* saved = t_worker;
        t_worker = &w;
        t_blockDim = Idx{block.x, block.y, block.z};
        t_gridDim = Idx{grid.x, grid.y, grid.z};
        for (;;) {
            long b = next.fetch_add(1);
            if (b >= nblocks) break;
            t_blockIdx.x = (unsigned)(b % grid.x);
            t_blockIdx.y = (unsigned)((b / grid.x) % grid.y);
            t_blockIdx.z = (unsigned)(b / ((long)grid.x * grid.y));
            run_block(w, block);
        }
        t_worker = saved;
        munmap(arena, bytes);
    };

    if (nworkers == 1) { work(); return; }
    std::vector<std::thread> ts;
    for (int i = 0; i < nworkers; ++i) ts.emplace_back(work);
    for (auto& t : ts) t.join();
}

}  // namespace hipemu
