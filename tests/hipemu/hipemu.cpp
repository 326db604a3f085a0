// hipemu.cpp (tests/hipemu) -- TEST INFRASTRUCTURE ONLY: fibre scheduler + host stand-ins for the few HIP
// runtime calls csrc/capi.hip makes.  See hip/hip_runtime.h in this directory.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

namespace hipemu {

extern "C" void hipemu_switch(void **save_sp, void *load_sp);
// x86-64 SysV: save the callee-saved registers on the current stack, swap stack pointers, restore
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
.size hipemu_switch,.-hipemu_switch
)");

enum State { READY = 0, WAIT_WAVE, WAIT_BLOCK, DONE };

struct Wave {
    Xchg in;    // deposits of the operation in progress
    Xchg out;   // snapshot handed to the lanes
    uint64_t arrived = 0;
    uint64_t alive = 0;
    int tag = 0;
};

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    int state = READY;
    int lane = 0;
    Wave *wave = nullptr;
    Ids ids;
};

Fiber *g_cur = nullptr;
static void *g_sched_sp = nullptr;
static void (*g_tramp)(void *) = nullptr;
static void *g_closure = nullptr;
static std::vector<char> g_lds;
static constexpr size_t kStack = 256 * 1024;
static std::vector<char *> g_stacks;

const Ids &ids() { return g_cur->ids; }
void *dyn_lds() { return g_lds.data(); }
int lane_id() { return g_cur->lane; }

static void yield_to_scheduler() {
    Fiber *f = g_cur;
    hipemu_switch(&f->sp, g_sched_sp);
}

static void fiber_main() {
    g_tramp(g_closure);
    g_cur->state = DONE;
    yield_to_scheduler();
    abort();  // a finished fibre is never resumed
}

const Xchg &wave_exchange(int tag, uint32_t val, uint32_t arg) {
    Fiber *f = g_cur;
    Wave *w = f->wave;
    if (w->arrived == 0) {
        w->tag = tag;
    } else if (w->tag != tag) {
        fprintf(stderr, "hipemu: divergent wavefront: lane %d is at cross-lane operation %d while others are at %d "
                        "(block %u)\n", f->lane, tag, w->tag, f->ids.block.x);
        abort();
    }
    w->in.val[f->lane] = val;
    w->in.arg[f->lane] = arg;
    w->arrived |= 1ull << f->lane;
    f->state = WAIT_WAVE;
    yield_to_scheduler();
    return w->out;
}

static int g_block_waiting = 0;

void block_barrier() {
    g_cur->state = WAIT_BLOCK;
    ++g_block_waiting;
    yield_to_scheduler();
}

static void prepare_stack(Fiber &f) {
    // layout expected by hipemu_switch's epilogue: six callee-saved slots, then the return address
    uintptr_t top = (uintptr_t)(f.stack + kStack) & ~(uintptr_t)15;
    void **sp = (void **)(top - 8 * sizeof(void *));
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    sp[6] = (void *)&fiber_main;
    sp[7] = nullptr;
    f.sp = sp;
}

// kernels run to completion inside the launch call, one launch at a time: host threads that launch
// concurrently (each on its own stream in the real runtime) simply take turns here
static std::mutex g_launch_mutex;

void launch_impl(dim3 grid, dim3 block, size_t lds_bytes, void (*tramp)(void *), void *closure) {
    std::lock_guard<std::mutex> one_at_a_time(g_launch_mutex);
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024 || (nthreads % 64) != 0) {
        fprintf(stderr, "hipemu: block of %d work-items (must be a multiple of 64, <= 1024)\n", nthreads);
        abort();
    }
    if (g_cur) {
        fprintf(stderr, "hipemu: nested launch\n");
        abort();
    }
    g_tramp = tramp;
    g_closure = closure;
    int lane_order = 0;
    uint64_t shuffle_state = 0;
    if (const char *e = getenv("FCD_EMU_LANE_ORDER")) {
        lane_order = !strcmp(e, "reverse") ? 1 : (!strcmp(e, "swap-halves") ? 2 : 0);
        if (!strncmp(e, "random:", 7)) {  // a fresh pseudo-random order for every sweep over the fibres
            lane_order = 3;
            shuffle_state = strtoull(e + 7, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
        }
    }
    std::vector<int> perm;
    g_lds.assign(lds_bytes + 64, 0);
    while ((int)g_stacks.size() < nthreads) {
        void *p = nullptr;
        if (posix_memalign(&p, 64, kStack)) abort();
        g_stacks.push_back((char *)p);
    }
    const int nwaves = nthreads / 64;
    std::vector<Fiber> fibers(nthreads);
    std::vector<Wave> waves(nwaves);
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    for (uint64_t b = 0; b < nblocks; ++b) {
        for (int w = 0; w < nwaves; ++w) {
            waves[w].arrived = 0;
            waves[w].alive = ~0ull;
        }
        g_block_waiting = 0;
        for (int t = 0; t < nthreads; ++t) {
            Fiber &f = fibers[t];
            f.stack = g_stacks[t];
            f.state = READY;
            f.lane = t & 63;
            f.wave = &waves[t >> 6];
            f.ids.thread = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            f.ids.block = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((uint64_t)grid.x * grid.y)));
            f.ids.bdim = block;
            f.ids.gdim = grid;
            prepare_stack(f);
        }
        int n_done = 0;
        while (n_done < nthreads) {
            bool progressed = false;
            if (lane_order == 3) {
                perm.resize(nthreads);
                for (int i = 0; i < nthreads; ++i) perm[i] = i;
                for (int i = nthreads - 1; i > 0; --i) {  // Fisher-Yates on an xorshift stream
                    shuffle_state ^= shuffle_state << 13;
                    shuffle_state ^= shuffle_state >> 7;
                    shuffle_state ^= shuffle_state << 17;
                    std::swap(perm[i], perm[(int)(shuffle_state % (uint64_t)(i + 1))]);
                }
            }
            for (int t0 = 0; t0 < nthreads; ++t0) {
                // A fibre runs until its next cross-lane operation, so between two such operations the lanes' plain
                // stores land in SCHEDULING order, not in the per-instruction order of true lockstep: a store of lane 3
                // that the GPU performs after an earlier store of lane 40 lands before it here.  Code that is correct in
                // lockstep but sensitive to that (r04: half 0's padding stores over half 1's ids) shows up as a result
                // that depends on the order the fibres are taken in: FCD_EMU_LANE_ORDER=reverse takes them from the
                // last lane down, =swap-halves exchanges the two halves of every wavefront, =random:SEED shuffles every sweep.
                const int t = lane_order == 1 ? nthreads - 1 - t0 : (lane_order == 2 ? (t0 ^ 32) : (lane_order == 3 ? perm[t0] : t0));
                Fiber &f = fibers[t];
                if (f.state != READY) continue;
                progressed = true;
                g_cur = &f;
                hipemu_switch(&g_sched_sp, f.sp);
                g_cur = nullptr;
                Wave *w = f.wave;
                if (f.state == DONE) {
                    ++n_done;
                    w->alive &= ~(1ull << f.lane);
                }
                // the wavefront's operation completes when every live lane has arrived
                if (w->arrived != 0 && (w->arrived & w->alive) == w->alive) {
                    w->out = w->in;
                    w->out.alive = w->alive;
                    w->arrived = 0;
                    const int base = (t >> 6) << 6;
                    for (int l = 0; l < 64; ++l)
                        if (fibers[base + l].state == WAIT_WAVE) fibers[base + l].state = READY;
                }
                if (g_block_waiting > 0 && g_block_waiting == nthreads - n_done) {
                    g_block_waiting = 0;
                    for (int u = 0; u < nthreads; ++u)
                        if (fibers[u].state == WAIT_BLOCK) fibers[u].state = READY;
                }
            }
            if (!progressed) {
                fprintf(stderr, "hipemu: deadlock in block %llu: ", (unsigned long long)b);
                for (int t = 0; t < nthreads; ++t) fprintf(stderr, "%d", fibers[t].state);
                fprintf(stderr, "\n");
                abort();
            }
        }
    }
}

}  // namespace hipemu

// ---- host runtime stand-ins -------------------------------------------------------------------
// Devices: FCD_EMU_DEVICES (default 1) emulated devices.  The current device is per host thread, as in HIP; streams
// and allocations remember the device they were made on, and an operation on them under ANOTHER current device aborts
// with a message -- on the GPU that is a wrong-context launch or an invalid-device-pointer error, the class of defect a
// one-GPU box can never show (device index > 0 had never executed).
struct hipemu_stream { int device; };
struct hipemu_event { double t_ms; };

static thread_local int t_device = 0;
static int emu_devices() {
    const char *e = getenv("FCD_EMU_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n >= 1 && n <= 16 ? n : 1;
}
extern "C" int hipemu_current_device() { return t_device; }  // (for the tests: "the caller's device is put back")
static std::mutex g_alloc_mu;
static std::map<uintptr_t, std::pair<size_t, int>> g_allocs;  // device allocations: start -> (bytes, device)
static void check_device_ptr(const void *p, const char *what) {
    if (emu_devices() < 2 || !p) return;
    std::lock_guard<std::mutex> g(g_alloc_mu);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return;  // host memory
    --it;
    if ((uintptr_t)p >= it->first + it->second.first) return;
    if (it->second.second != t_device) {
        fprintf(stderr, "hipemu: %s touches memory of device %d while device %d is current\n", what, it->second.second, t_device);
        abort();
    }
}
static void check_stream(hipStream_t s, const char *what) {
    if (s && s->device != t_device) {
        fprintf(stderr, "hipemu: %s on a stream of device %d while device %d is current\n", what, s->device, t_device);
        abort();
    }
}
namespace hipemu {
void check_launch_stream(hipStream_t s) { check_stream(s, "kernel launch"); }
}

static double now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

hipError_t hipGetDeviceCount(int *n) { *n = emu_devices(); return hipSuccess; }
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= emu_devices()) return hipErrorInvalidValue;
    t_device = d;
    return hipSuccess;
}
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) {
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1)) { *p = nullptr; return hipErrorOutOfMemory; }
    memset(q, 0xA5, n);  // device memory arrives uninitialised: make reliance on zeros visible
    *p = q;
    if (emu_devices() > 1) {
        std::lock_guard<std::mutex> g(g_alloc_mu);
        g_allocs[(uintptr_t)q] = std::make_pair(n ? n : (size_t)1, t_device);
    }
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (emu_devices() > 1 && p) {
        std::lock_guard<std::mutex> g(g_alloc_mu);
        g_allocs.erase((uintptr_t)p);
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    *free_b = (size_t)1 << 30;  // small on purpose: exercises the chunked-workspace paths
    *total_b = (size_t)2 << 30;
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t st) {
    check_stream(st, "hipMemcpyAsync");
    check_device_ptr(dst, "hipMemcpyAsync");
    check_device_ptr(src, "hipMemcpyAsync");
    memcpy(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t st) {
    check_stream(st, "hipMemsetAsync");
    check_device_ptr(dst, "hipMemsetAsync");
    memset(dst, v, n);
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int v, size_t n) { memset(dst, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new hipemu_stream{t_device}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 1; *greatest = -1; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t st) { check_stream(st, "hipStreamSynchronize"); return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event{0.0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st) { check_stream(st, "hipEventRecord"); e->t_ms = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t, unsigned) { check_stream(st, "hipStreamWaitEvent"); return hipSuccess; }  // (nothing is ever pending)
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }  // (launches run to completion: whatever was recorded has happened)
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipemu error"; }
