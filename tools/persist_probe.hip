// persist_probe.hip -- what does ONE persistent launch per NMF iteration cost on this chip, against the three dependent
// launches the split path makes?  (Round-3 VERDICT next 4b: measure the barrier argument on this kernel's footprint.)
//
// One iteration of a lone small problem is three all-to-all dependent phases (H half-step; W half-step writing slabs; the
// slab sum / apply): every workgroup of a phase reads what OTHER workgroups (on other CUs and XCDs) wrote in the phase before.
// The probe keeps that shape and replaces the arithmetic by its duration: G workgroups (one per CU), three phases of b1 / b2 /
// b3 microseconds of busy time, each ending in a 4-KiB record per workgroup (plain 16-B stores) that the next phase's
// workgroup (id + 37) % G reads in full and CHECKS (a stale read is counted, not ignored).
//   launches : the three phases as three dependent kernel launches (what nmf_split16 + split_apply do today)
//   flat     : one launch per `iters` iterations; phases separated by a grid barrier on one monotonic counter (lane 0: drain,
//              agent-scope release, arrive; relaxed sc1 poll with s_sleep; agent-scope acquire)
//   xcd      : the same with the XCD-hierarchical barrier MI355X_MICROARCH.md prices cheapest (barrier-xcd): per-XCD arrival
//              counter; the last arriver of an XCD releases, arrives at the top counter, waits for all XCDs, acquires and bumps
//              its XCD's generation word; everybody else polls that word and acquires.  XCD membership is counted at run time
//              (placement is not promised).  Every spin is bounded.
// All variants are captured into ONE hipGraph of `iters` iterations and replayed between two hipEvents.
//   hipcc -O3 --offload-arch=gfx950 tools/persist_probe.hip -o /tmp/persist_probe && /tmp/persist_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int REC = 1024;              // floats per record: 4 KiB, one f32x4 per thread
constexpr long kSpinLimit = 4000000;   // ~ a second: give up, flag, carry on (nothing may hang the box)

struct Sync {                          // one block of its own, zeroed before every graph replay
    unsigned flat;                     // monotonic arrival counter of the flat barrier
    unsigned top;                      // arrivals of XCD leaders
    unsigned gave_up;                  // a bounded spin ran out
    unsigned stale;                    // records that did not hold the expected value
    unsigned members[8];               // workgroups per XCD (census)
    unsigned arrive[8 * 16];           // per-XCD arrival counters, 64 B apart
    unsigned gen[8 * 16];              // per-XCD generation words, 64 B apart
};

__device__ __forceinline__ void busy_wait_us(float us) {   // wall_clock64: the constant 100 MHz counter
    const long ticks = (long)(us * 100.f), t0 = (long)wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 7u; }   // HW_REG_XCC_ID, bits 3:0

// phase body: busy time, then check the record of workgroup (id + 37) % G from the previous phase, then write this one's
__device__ __forceinline__ void phase(float *__restrict__ rec_out, const float *__restrict__ rec_in, int G, float us, float expect, float value, Sync *s, bool check) {
    const int tid = threadIdx.x, id = blockIdx.x;
    busy_wait_us(us);
    if (check) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(rec_in + (size_t)((id + 37) % G) * REC + 4 * tid);
        if (v[0] != expect || v[1] != expect || v[2] != expect || v[3] != expect) atomicAdd(&s->stale, 1u);
    }
    *reinterpret_cast<f32x4 *>(rec_out + (size_t)id * REC + 4 * tid) = f32x4{value, value, value, value};
}

__global__ __launch_bounds__(256) void phase_kernel(float *rec_out, const float *rec_in, int G, float us, float expect, float value, Sync *s, int check) {
    phase(rec_out, rec_in, G, us, expect, value, s, check != 0);
}

__device__ __forceinline__ void spin_until(const unsigned *w, unsigned target, Sync *s) {
    long n = 0;
    while ((int)(ld_sc1(w) - target) < 0) { __builtin_amdgcn_s_sleep(1); if (++n > kSpinLimit) { atomicAdd(&s->gave_up, 1u); break; } }
}
// epoch e = 1, 2, ...: all G workgroups arrive, nobody leaves before the last has
__device__ __forceinline__ void barrier_flat(Sync *s, unsigned e, int G) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        spin_until(&s->flat, e * (unsigned)G, s);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
__device__ __forceinline__ void barrier_xcd(Sync *s, unsigned e, unsigned x, unsigned my_members, unsigned n_xcd) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&s->arrive[16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == e * my_members) {      // the last arriver of this XCD speaks for it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // writes back this XCD's L2: every member's stores are drained
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spin_until(&s->top, e * n_xcd, s);
            __hip_atomic_store(&s->gen[16 * x], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            spin_until(&s->gen[16 * x], e, s);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int MODE>   // 1: flat barrier, 2: XCD-hierarchical
__global__ __launch_bounds__(256) void persistent_kernel(float *recA, float *recB, float *recC, int G, float b1, float b2, float b3, int iters, Sync *s) {
    unsigned e = 0, x = 0, mine = 0, nx = 0;
    if (MODE == 2) {   // census of the placement (one flat barrier per launch)
        x = xcc_id();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&s->members[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        barrier_flat(s, 1, G);
        for (int i = 0; i < 8; ++i) { const unsigned m = ld_sc1(&s->members[i]); if (m) ++nx; if ((unsigned)i == x) mine = m; }
    }
    auto bar = [&]() { ++e; if (MODE == 1) barrier_flat(s, e, G); else barrier_xcd(s, e, x, mine, nx); };
    for (int it = 0; it < iters; ++it) {
        const float v = (float)(3 * it);
        phase(recA, recC, G, b1, v - 1.f, v + 1.f, s, it > 0);   // "H half-step": reads the apply's output of the iteration before
        bar();
        phase(recB, recA, G, b2, v + 1.f, v + 2.f, s, true);     // "W half-step": reads H, leaves slabs
        bar();
        phase(recC, recB, G, b3, v + 2.f, v + 2.f, s, true);     // "apply": reads the slabs   (value v + 2 = next v - 1)
        bar();
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200, reps = 5;
    struct Shape { const char *name; int G; float b1, b2, b3; };
    // busy times: the measured kernel durations of profiles/r03_small_*_timeline.txt less ~1.7 us of per-launch prologue / first touch
    const Shape shapes[] = {{"cfg2 1024x4096x64 ", 256, 12.9f, 12.6f, 3.2f}, {"gold 4096x350x128 ", 242, 10.8f, 3.1f, 13.0f}, {"paper 512x3445x30 ", 256, 5.6f, 5.6f, 3.7f}};
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *rec[3];
    for (auto &r : rec) CHK(hipMalloc(&r, sizeof(float) * REC * 256));
    Sync *s;
    CHK(hipMalloc(&s, sizeof(Sync)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (const Shape &sh : shapes) {
        double us[3] = {0, 0, 0};
        unsigned bad[3] = {0, 0, 0}, quit[3] = {0, 0, 0};
        for (int mode = 0; mode < 3; ++mode) {
            hipGraph_t g; hipGraphExec_t ge;
            CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            CHK(hipMemsetAsync(s, 0, sizeof(Sync), st));
            if (mode == 0) {
                for (int it = 0; it < iters; ++it) {
                    const float v = (float)(3 * it);
                    hipLaunchKernelGGL(phase_kernel, dim3(sh.G), dim3(256), 0, st, rec[0], rec[2], sh.G, sh.b1, v - 1.f, v + 1.f, s, it > 0 ? 1 : 0);
                    hipLaunchKernelGGL(phase_kernel, dim3(sh.G), dim3(256), 0, st, rec[1], rec[0], sh.G, sh.b2, v + 1.f, v + 2.f, s, 1);
                    hipLaunchKernelGGL(phase_kernel, dim3(sh.G), dim3(256), 0, st, rec[2], rec[1], sh.G, sh.b3, v + 2.f, v + 2.f, s, 1);
                }
            } else if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(sh.G), dim3(256), 0, st, rec[0], rec[1], rec[2], sh.G, sh.b1, sh.b2, sh.b3, iters, s);
            else hipLaunchKernelGGL(persistent_kernel<2>, dim3(sh.G), dim3(256), 0, st, rec[0], rec[1], rec[2], sh.G, sh.b1, sh.b2, sh.b3, iters, s);
            CHK(hipStreamEndCapture(st, &g));
            CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));   // warm
            float best = 1e30f;
            for (int r = 0; r < reps; ++r) {
                CHK(hipEventRecord(e0, st)); CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            Sync h; CHK(hipMemcpy(&h, s, sizeof h, hipMemcpyDeviceToHost));
            us[mode] = best * 1e3 / iters; bad[mode] = h.stale; quit[mode] = h.gave_up;
            CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        }
        const double work = sh.b1 + sh.b2 + sh.b3;
        printf("%s G=%3d busy %4.1f us/iteration: three launches %6.2f us | one launch, flat barrier %6.2f us | one launch, XCD-hierarchical barrier %6.2f us   "
               "(per seam beyond the busy time: %.2f / %.2f / %.2f us; stale records %u / %u / %u; spins given up %u / %u / %u)\n",
               sh.name, sh.G, work, us[0], us[1], us[2], (us[0] - work) / 3, (us[1] - work) / 3, (us[2] - work) / 3, bad[0], bad[1], bad[2], quit[0], quit[1], quit[2]);
    }
    return 0;
}
