// SIMT interpreter for the specforge_amd HIP kernels.  TEST INFRASTRUCTURE ONLY.
//
// The kernels under specforge_amd/csrc are written against sf_platform.h.  With
// -DSF_EMU that header includes this file instead of <hip/hip_runtime.h> and the
// very same kernel source is compiled for the x86 host (amdclang++ -x c++): every
// GPU thread becomes a ucontext fiber, workgroups are distributed over a few OS
// threads, and __syncthreads() / wave shuffles / MFMA / LDS-DMA are rendezvous
// points between the fibers of a workgroup.  This is a kernel *debugger* (index
// logic, divergent barriers, out-of-bounds via -fsanitize=address); it is not a
// product path: specforge_amd/_lib.py only ever loads libsfhip.so, the tests
// inject the emulated library explicitly.
//
// Lane layouts emulated here are the ones the CDNA4 guide documents:
//   mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+j]  B[k=8*(l>>4)+j][n=l&15]
//                           D[row=4*(l>>4)+r][col=l&15]
//   mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+j]  B[k=8*(l>>5)+j][n=l&31]
//                           D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
//   global_load_lds 16B   : LDS dst = wave-uniform base + 16*lane, src per lane
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <utility>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SFEMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#endif

// Fiber switch.  glibc's swapcontext saves and restores the signal mask: two rt_sigprocmask system calls per switch, and a 256-thread
// workgroup switches fibers ~10^5 times per launch -- a third of the CPU time of the interpreter tests was spent in the kernel.  On x86-64
// (outside AddressSanitizer builds, which need the ucontext path for their fiber annotations) a switch is the System V callee-saved state:
// rbx, rbp, r12 - r15, the stack pointer, MXCSR and the x87 control word.
#if defined(__x86_64__) && !defined(SFEMU_ASAN)
#define SFEMU_FASTSWITCH 1
__attribute__((naked, noinline, unused)) static void sfemu_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\t"
        "pushq %rbx\n\t"
        "pushq %r12\n\t"
        "pushq %r13\n\t"
        "pushq %r14\n\t"
        "pushq %r15\n\t"
        "subq $8, %rsp\n\t"
        "stmxcsr (%rsp)\n\t"
        "fnstcw 4(%rsp)\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "ldmxcsr (%rsp)\n\t"
        "fldcw 4(%rsp)\n\t"
        "addq $8, %rsp\n\t"
        "popq %r15\n\t"
        "popq %r14\n\t"
        "popq %r13\n\t"
        "popq %r12\n\t"
        "popq %rbx\n\t"
        "popq %rbp\n\t"
        "ret\n\t");
}
#endif

namespace sfemu {

struct uint3e {
    unsigned x, y, z;
};
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

enum FiberState { RUNNABLE = 0, AT_BLOCK_BARRIER = 1, AT_WAVE_SYNC = 2, DONE = 3 };

struct Fiber {
#ifdef SFEMU_FASTSWITCH
    void* sp = nullptr;         // saved stack pointer (sfemu_switch)
#else
    ucontext_t ctx;
#endif
    char* stack = nullptr;
    int state = RUNNABLE;
    uint3e tid{0, 0, 0};
    int linear = 0;
    unsigned coll_ops = 0;
    void* asan_fake = nullptr;
};

static constexpr size_t kStackBytes = 64 * 1024;
static constexpr int kSlotBytes = 128;  // per-lane exchange payload

struct Wave {
    alignas(16) unsigned char slot[2][64][kSlotBytes];
};

struct BlockCtx {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
#ifdef SFEMU_FASTSWITCH
    void* sched_sp = nullptr;
#else
    ucontext_t sched;
#endif
    void* sched_asan_fake = nullptr;
    const void* sched_stack_bottom = nullptr;
    size_t sched_stack_size = 0;
    int nthreads = 0;
    Fiber* cur = nullptr;
    uint3e bid{0, 0, 0};
    dim3 bdim, gdim;
    const std::function<void()>* body = nullptr;
    std::vector<char> dyn_smem;
    char* stack_pool = nullptr;     // nthreads x kStackBytes, from the process-wide pool below (not zeroed: a launch's workers used to
                                    // allocate and clear 16 MiB each)
    size_t stack_pool_bytes = 0;
};

// stack arenas are recycled across launches (workers are fresh OS threads per launch, so a thread_local would not survive)
struct StackPool {
    std::atomic_flag lock = ATOMIC_FLAG_INIT;
    std::vector<std::pair<char*, size_t>> free_list;
    char* take(size_t bytes) {
        while (lock.test_and_set(std::memory_order_acquire)) {}
        char* p = nullptr;
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].second == bytes) { p = free_list[i].first; free_list[i] = free_list.back(); free_list.pop_back(); break; }
        lock.clear(std::memory_order_release);
        if (!p && posix_memalign((void**)&p, 64, bytes) != 0) { fprintf(stderr, "[sfemu] out of memory (fiber stacks)\n"); abort(); }
        return p;
    }
    void give(char* p, size_t bytes) {
        while (lock.test_and_set(std::memory_order_acquire)) {}
        if (free_list.size() < 64) { free_list.emplace_back(p, bytes); p = nullptr; }
        lock.clear(std::memory_order_release);
        if (p) free(p);
    }
};
inline StackPool& stack_pool() {
    static StackPool sp;
    return sp;
}

inline BlockCtx*& tls_ctx() {
    static thread_local BlockCtx* c = nullptr;
    return c;
}

inline void switch_to_sched() {
    BlockCtx* c = tls_ctx();
    Fiber* f = c->cur;
#ifdef SFEMU_ASAN
    __sanitizer_start_switch_fiber(f->state == DONE ? nullptr : &f->asan_fake, c->sched_stack_bottom, c->sched_stack_size);
#endif
#ifdef SFEMU_FASTSWITCH
    sfemu_switch(&f->sp, c->sched_sp);
#else
    swapcontext(&f->ctx, &c->sched);
#endif
#ifdef SFEMU_ASAN
    __sanitizer_finish_switch_fiber(f->asan_fake, nullptr, nullptr);
#endif
}

inline void fiber_entry() {
    BlockCtx* c = tls_ctx();
#ifdef SFEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &c->sched_stack_bottom, &c->sched_stack_size);
#endif
    (*c->body)();
    c->cur->state = DONE;
    switch_to_sched();
}

inline void run_block(BlockCtx* c) {
    const int n = c->nthreads;
    for (int i = 0; i < n; ++i) {
        Fiber& f = c->fibers[i];
        f.state = RUNNABLE;
        f.coll_ops = 0;
        f.asan_fake = nullptr;
#ifdef SFEMU_FASTSWITCH
        // the frame sfemu_switch pops on the first switch into this fiber: [mxcsr | x87 cw][r15 r14 r13 r12 rbx rbp][return = fiber_entry][0];
        // after its `ret` the stack pointer is top - 8, i.e. what the ABI promises a function at entry (rsp + 8 a multiple of 16)
        uintptr_t top = ((uintptr_t)f.stack + kStackBytes) & ~(uintptr_t)15;
        void** q = (void**)top;
        *--q = nullptr;                          // fiber_entry's (never used) return address
        *--q = (void*)&fiber_entry;
        for (int r = 0; r < 6; ++r) *--q = nullptr;
        --q;
        unsigned csr = 0x1f80;                   // MXCSR default: all exceptions masked, round to nearest
        unsigned short cw = 0x037f;              // x87 control word default
        std::memcpy((char*)q, &csr, 4);
        std::memcpy((char*)q + 4, &cw, 2);
        f.sp = (void*)q;
#else
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
    }
    const int nwaves = (n + 63) / 64;
    for (;;) {
        bool ran = false;
        for (int i = 0; i < n; ++i) {
            Fiber& f = c->fibers[i];
            if (f.state != RUNNABLE) continue;
            ran = true;
            c->cur = &f;
#ifdef SFEMU_ASAN
            __sanitizer_start_switch_fiber(&c->sched_asan_fake, f.stack, kStackBytes);
#endif
#ifdef SFEMU_FASTSWITCH
            sfemu_switch(&c->sched_sp, f.sp);
#else
            swapcontext(&c->sched, &f.ctx);
#endif
#ifdef SFEMU_ASAN
            __sanitizer_finish_switch_fiber(c->sched_asan_fake, nullptr, nullptr);
#endif
        }
        bool released = false;
        int live = 0, at_bar = 0;
        for (int w = 0; w < nwaves; ++w) {
            int lo = w * 64, hi = std::min(n, lo + 64), wl = 0, ws = 0;
            for (int i = lo; i < hi; ++i) {
                int s = c->fibers[i].state;
                if (s != DONE) ++wl;
                if (s == AT_WAVE_SYNC) ++ws;
                if (s == AT_BLOCK_BARRIER) ++at_bar;
            }
            live += wl;
            if (wl > 0 && ws == wl) {
                for (int i = lo; i < hi; ++i)
                    if (c->fibers[i].state == AT_WAVE_SYNC) c->fibers[i].state = RUNNABLE;
                released = true;
            }
        }
        if (live == 0) break;
        if (at_bar == live) {
            for (int i = 0; i < n; ++i)
                if (c->fibers[i].state == AT_BLOCK_BARRIER) c->fibers[i].state = RUNNABLE;
            released = true;
        }
        if (!released && !ran) {
            fprintf(stderr, "[sfemu] DEADLOCK in block (%u,%u,%u): divergent barrier / wave collective\n", c->bid.x, c->bid.y, c->bid.z);
            for (int i = 0; i < n; ++i) fprintf(stderr, "%d", c->fibers[i].state);
            fprintf(stderr, "\n");
            abort();
        }
    }
}

inline int emu_threads() {
    static int n = [] {
        const char* e = getenv("SFEMU_THREADS");
        int v = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(v, 64));
    }();
    return n;
}

inline void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nblocks <= 0 || nthreads <= 0) return;
    std::atomic<long> next{0};
    auto worker = [&]() {
        BlockCtx ctx;
        ctx.nthreads = nthreads;
        ctx.fibers.resize(nthreads);
        ctx.waves.resize((nthreads + 63) / 64);
        ctx.stack_pool_bytes = (size_t)nthreads * kStackBytes;
        ctx.stack_pool = stack_pool().take(ctx.stack_pool_bytes);
        ctx.dyn_smem.resize(smem + 64);
        ctx.bdim = block;
        ctx.gdim = grid;
        ctx.body = &body;
        for (int i = 0; i < nthreads; ++i) {
            Fiber& f = ctx.fibers[i];
            f.stack = ctx.stack_pool + (size_t)i * kStackBytes;
            f.linear = i;
            f.tid.x = i % block.x;
            f.tid.y = (i / block.x) % block.y;
            f.tid.z = i / (block.x * block.y);
        }
        tls_ctx() = &ctx;
        for (;;) {
            long b = next.fetch_add(1);
            if (b >= nblocks) break;
            ctx.bid.x = (unsigned)(b % grid.x);
            ctx.bid.y = (unsigned)((b / grid.x) % grid.y);
            ctx.bid.z = (unsigned)(b / ((long)grid.x * grid.y));
            run_block(&ctx);
        }
        tls_ctx() = nullptr;
        stack_pool().give(ctx.stack_pool, ctx.stack_pool_bytes);
    };
    int nt = (int)std::min<long>(emu_threads(), nblocks);
    if (nt <= 1) {
        // still run on a fresh OS thread: thread_local __shared__ arrays stay per-launch-thread
        std::thread t(worker);
        t.join();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < nt; ++i) ts.emplace_back(worker);
        for (auto& t : ts) t.join();
    }
}

// ---- what kernels see -------------------------------------------------------
inline uint3e cur_tid() { return tls_ctx()->cur->tid; }
inline uint3e cur_bid() { return tls_ctx()->bid; }
inline dim3 cur_bdim() { return tls_ctx()->bdim; }
inline dim3 cur_gdim() { return tls_ctx()->gdim; }
inline char* dyn_smem_base() {
    char* p = tls_ctx()->dyn_smem.data();
    return (char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
}

inline void block_barrier() {
    tls_ctx()->cur->state = AT_BLOCK_BARRIER;
    switch_to_sched();
}
// cooperative spin-wait step: stay RUNNABLE, let every other fiber of the block run one slice
inline void yield() { switch_to_sched(); }
inline void wave_sync() {
    tls_ctx()->cur->state = AT_WAVE_SYNC;
    switch_to_sched();
}
inline int lane_id() { return tls_ctx()->cur->linear & 63; }
inline int wave_index() { return tls_ctx()->cur->linear >> 6; }
inline int wave_lanes() {
    BlockCtx* c = tls_ctx();
    int lo = wave_index() * 64;
    return std::min(64, c->nthreads - lo);
}

// publish `bytes` of this lane, rendezvous, return the wave's slot table for reading.
// Double-buffered by the per-fiber collective counter: a lane may start collective
// n+1 (other buffer) while slower lanes still read buffer n; buffer n is rewritten
// only by collective n+2, which every lane reaches after passing rendezvous n+1.
typedef unsigned char SlotRow[kSlotBytes];
inline SlotRow* wave_exchange(const void* mine, int bytes) {
    BlockCtx* c = tls_ctx();
    Fiber* f = c->cur;
    Wave& w = c->waves[f->linear >> 6];
    int par = (int)(f->coll_ops++ & 1u);
    memcpy(w.slot[par][f->linear & 63], mine, (size_t)bytes);
    wave_sync();
    return w.slot[par];
}

template <typename T>
inline T shfl(T v, int src_lane) {
    static_assert(sizeof(T) <= kSlotBytes, "payload");
    SlotRow* s = wave_exchange(&v, sizeof(T));
    T r;
    int n = wave_lanes();
    int sl = src_lane & 63;
    if (sl >= n) sl = lane_id();
    memcpy(&r, s[sl], sizeof(T));
    return r;
}
template <typename T>
inline T shfl_xor(T v, int mask) {
    return shfl(v, lane_id() ^ mask);
}
template <typename T>
inline T shfl_down(T v, int d) {
    int l = lane_id() + d;
    return shfl(v, l > 63 ? lane_id() : l);
}

inline float bf16_bits_to_f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

inline v4f mfma_16x16x32_bf16(v8s a, v8s b, v4f c) {
    struct P {
        v8s a, b;
    } p{a, b};
    SlotRow* s = wave_exchange(&p, sizeof(P));
    int l = lane_id();
    int col = l & 15;
    v4f d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r;
        float acc = 0.f;
        for (int k = 0; k < 32; ++k) {
            P pa, pb;
            memcpy(&pa, s[row + 16 * (k >> 3)], sizeof(P));
            memcpy(&pb, s[col + 16 * (k >> 3)], sizeof(P));
            acc += bf16_bits_to_f((unsigned short)pa.a[k & 7]) * bf16_bits_to_f((unsigned short)pb.b[k & 7]);
        }
        d[r] = c[r] + acc;
    }
    return d;
}

inline v16f mfma_32x32x16_bf16(v8s a, v8s b, v16f c) {
    struct P {
        v8s a, b;
    } p{a, b};
    SlotRow* s = wave_exchange(&p, sizeof(P));
    int l = lane_id();
    int col = l & 31;
    v16f d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = 0.f;
        for (int k = 0; k < 16; ++k) {
            P pa, pb;
            memcpy(&pa, s[row + 32 * (k >> 3)], sizeof(P));
            memcpy(&pb, s[col + 32 * (k >> 3)], sizeof(P));
            acc += bf16_bits_to_f((unsigned short)pa.a[k & 7]) * bf16_bits_to_f((unsigned short)pb.b[k & 7]);
        }
        d[r] = c[r] + acc;
    }
    return d;
}

// LDS-DMA: every lane passes its own source; the destination is a WAVE-UNIFORM base (the
// hardware takes it from M0 = readfirstlane(ptr)) + 16*lane.  Lanes disagreeing on the base
// is a bug on real hardware (wrong as soon as the load is exec-masked), so it is fatal here.
inline void global_load_lds16(const void* gsrc, void* lds_base) {
    struct P {
        const void* g;
        void* l;
    } p{gsrc, lds_base};
    SlotRow* s = wave_exchange(&p, sizeof(P));
    const int n = wave_lanes();
    for (int i = 0; i < n; ++i) {
        P o;
        memcpy(&o, s[i], sizeof(P));
        if (o.l != lds_base) {
            fprintf(stderr, "[sfemu] global_load_lds: LDS base is not wave-uniform (lane %d vs %d)\n", lane_id(), i);
            abort();
        }
    }
    memcpy((char*)lds_base + 16 * lane_id(), gsrc, 16);
}

inline void global_load_lds4(const void* gsrc, void* lds_base) {
    struct P {
        void* l;
    } p{lds_base};
    SlotRow* s = wave_exchange(&p, sizeof(P));
    const int n = wave_lanes();
    for (int i = 0; i < n; ++i) {
        P o;
        memcpy(&o, s[i], sizeof(P));
        if (o.l != lds_base) {
            fprintf(stderr, "[sfemu] global_load_lds4: LDS base is not wave-uniform (lane %d vs %d)\n", lane_id(), i);
            abort();
        }
    }
    memcpy((char*)lds_base + 4 * lane_id(), gsrc, 4);
}

// ds_read_b64_tr_b16 (transpose read): see sf_platform.h::sf_ds_read_tr16
typedef short v4s __attribute__((ext_vector_type(4)));
inline v4s ds_read_tr16_b64(const void* lds_ptr) {
    v4s mine;
    memcpy(&mine, lds_ptr, 8);
    SlotRow* s = wave_exchange(&mine, 8);
    const int l = lane_id(), g = l >> 4, i = l & 15;
    v4s r;
    for (int j = 0; j < 4; ++j) {
        v4s o;
        memcpy(&o, s[16 * g + 4 * j + (i >> 2)], 8);
        r[j] = o[i & 3];
    }
    return r;
}

template <typename T>
inline T atomic_add(T* p, T v) {
    if constexpr (std::is_floating_point<T>::value) {
        T old = *p, des;
        do {
            des = old + v;
        } while (!__atomic_compare_exchange(p, &old, &des, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
        return old;
    } else {
        return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
    }
}

}  // namespace sfemu
