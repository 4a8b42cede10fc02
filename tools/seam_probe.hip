// seam_probe.hip -- what does it cost on this chip to combine split-K slabs INSIDE the launch that wrote them, against the
// separate apply launch the split path uses?  ("fold the apply into the slab-writing launch")
//
// The shape of the cfg2 W-step: G row groups x S splits workgroups (64 x 4 = 256, one per CU), each leaves a 16 x K slab
// (K = 64: 4 KiB) after `busy` microseconds of work; the S slabs of a group are summed in fixed order and applied to U.
//   two   : writer kernel, then an apply kernel (a dependent launch: what nmf_split16.hip + split_apply_kernel do)
//   fence : one kernel; plain slab stores, agent-scope release, ticket (atomic add), the last arriver of a group acquires,
//           re-reads the S slabs, applies (deterministic: fixed summation order whoever arrives last); no workgroup ever waits
//   sc1   : the same with write-through (sc1) slab stores and sc1 loads instead of the release / acquire fences
// Each variant's chain of `iters` dependent repetitions (writer [+ apply] + a consumer launch that reads all of U, standing in
// for the next half-step) is captured into ONE hipGraph and replayed between two hipEvents, so the host's launch rate
// (~3.4 us per eager launch) is out of the picture; printed: microseconds per repetition; the result is checked on the host.  Build + run:  hipcc -O3 --offload-arch=gfx950 tools/seam_probe.hip -o /tmp/seam_probe && /tmp/seam_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64, ROWS = 16, SLAB = ROWS * K;      // floats per slab: 4 KiB

__device__ __forceinline__ void busy_wait(long ticks) {   // wall_clock64: the constant 100 MHz counter
    const long t0 = (long)wall_clock64();
    while ((long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void store_sc1(float *p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 load_sc1(const float *p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

// slab of (group, split): value = seed + group + split / 8 at every entry (cheap to verify)
template <int MODE>   // 0: writer only, 1: fence combine, 2: sc1 combine
__global__ __launch_bounds__(256) void writer_kernel(float *__restrict__ slabs, float *__restrict__ U, unsigned *__restrict__ ticket, int S, float seed, long busy_cycles) {
    __shared__ unsigned last;
    const int group = blockIdx.x / S, split = blockIdx.x % S, tid = threadIdx.x;
    busy_wait(busy_cycles);
    float *mine = slabs + ((size_t)group * S + split) * SLAB;
    const f32x4 v = {seed + group + split * 0.125f, seed + group + split * 0.125f, seed + group + split * 0.125f, seed + group + split * 0.125f};
    if (MODE == 2) store_sc1(mine + 4 * tid, v); else *reinterpret_cast<f32x4 *>(mine + 4 * tid) = v;
    if (MODE == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const unsigned t = __hip_atomic_fetch_add(ticket + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (unsigned)S - 1) ? 1u : 0u;
        if (last) {
            __hip_atomic_store(ticket + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
    }
    __syncthreads();
    if (!last) return;
    const float *base = slabs + (size_t)group * S * SLAB + 4 * tid;
    f32x4 s = MODE == 2 ? load_sc1(base) : *reinterpret_cast<const f32x4 *>(base);
    for (int sp = 1; sp < S; ++sp) s += MODE == 2 ? load_sc1(base + (size_t)sp * SLAB) : *reinterpret_cast<const f32x4 *>(base + (size_t)sp * SLAB);
    float *u = U + (size_t)group * SLAB + 4 * tid;
    *reinterpret_cast<f32x4 *>(u) = *reinterpret_cast<const f32x4 *>(u) * 0.5f + s;
}
__global__ __launch_bounds__(256) void apply_kernel(const float *__restrict__ slabs, float *__restrict__ U, int S) {
    const int group = blockIdx.x, tid = threadIdx.x;
    const float *base = slabs + (size_t)group * S * SLAB + 4 * tid;
    f32x4 s = *reinterpret_cast<const f32x4 *>(base);
    for (int sp = 1; sp < S; ++sp) s += *reinterpret_cast<const f32x4 *>(base + (size_t)sp * SLAB);
    float *u = U + (size_t)group * SLAB + 4 * tid;
    *reinterpret_cast<f32x4 *>(u) = *reinterpret_cast<const f32x4 *>(u) * 0.5f + s;
}
// stands in for the next half-step: every workgroup reads all of U (the factor the previous step produced)
__global__ __launch_bounds__(256) void consumer_kernel(const float *__restrict__ U, float *__restrict__ sink, int G, long busy_cycles) {
    float a = 0.f;
    for (int i = threadIdx.x; i < G * SLAB; i += 256 * 16) a += U[i];
    busy_wait(busy_cycles);
    if (a == 12345.678f) sink[blockIdx.x] = a;
}

int main(int argc, char **argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 64, S = argc > 2 ? atoi(argv[2]) : 4, iters = argc > 3 ? atoi(argv[3]) : 400;
    const double busy_us = argc > 4 ? atof(argv[4]) : 10.0;
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const long busy_cycles = (long)(busy_us * 100.0);      // s_memtime counts at 100 MHz on gfx950
    float *slabs, *U, *sink;
    unsigned *ticket;
    CHK(hipMalloc(&slabs, sizeof(float) * (size_t)G * S * SLAB));
    CHK(hipMalloc(&U, sizeof(float) * (size_t)G * SLAB));
    CHK(hipMalloc(&sink, sizeof(float) * 1024));
    CHK(hipMalloc(&ticket, sizeof(unsigned) * G));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("%s: %d row groups x %d splits = %d workgroups, slab %d B, %.0f us of work per workgroup, %d dependent repetitions\n", prop.name, G, S, G * S, SLAB * 4, busy_us, iters);
    const char *names[3] = {"two launches (writer; apply)", "one launch, fences + ticket   ", "one launch, sc1 + ticket      "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CHK(hipMemsetAsync(U, 0, sizeof(float) * (size_t)G * SLAB, st));
            CHK(hipMemsetAsync(ticket, 0, sizeof(unsigned) * G, st));
            hipGraph_t graph;
            hipGraphExec_t exec;
            CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int it = 0; it < iters; ++it) {
                const float seed = (float)(it % 7);
                if (mode == 0) {
                    hipLaunchKernelGGL(writer_kernel<0>, dim3(G * S), dim3(256), 0, st, slabs, U, ticket, S, seed, busy_cycles);
                    hipLaunchKernelGGL(apply_kernel, dim3(G), dim3(256), 0, st, slabs, U, S);
                } else if (mode == 1) hipLaunchKernelGGL(writer_kernel<1>, dim3(G * S), dim3(256), 0, st, slabs, U, ticket, S, seed, busy_cycles);
                else hipLaunchKernelGGL(writer_kernel<2>, dim3(G * S), dim3(256), 0, st, slabs, U, ticket, S, seed, busy_cycles);
                hipLaunchKernelGGL(consumer_kernel, dim3(256), dim3(256), 0, st, U, sink, G, busy_cycles);
            }
            CHK(hipStreamEndCapture(st, &graph));
            CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int pass = 0; pass < 2; ++pass) {      // pass 0: warm-up
                CHK(hipEventRecord(e0, st));
                CHK(hipGraphLaunch(exec, st));
                CHK(hipEventRecord(e1, st));
                CHK(hipEventSynchronize(e1));
            }
            CHK(hipGraphExecDestroy(exec));
            CHK(hipGraphDestroy(graph));
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            // verify: U after 2 * iters repetitions of u = u / 2 + sum_s (seed + g + s / 8)
            std::vector<float> h((size_t)G * SLAB);
            CHK(hipMemcpy(h.data(), U, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
            int bad = 0;
            for (int g = 0; g < G; ++g) {
                float u = 0.f;
                for (int pass = 0; pass < 2; ++pass)
                    for (int it = 0; it < iters; ++it) { float s = 0.f; for (int sp = 0; sp < S; ++sp) { const float v = (float)(it % 7) + g + sp * 0.125f; s = sp == 0 ? v : s + v; } u = u * 0.5f + s; }
                for (int i = 0; i < SLAB; ++i) if (h[(size_t)g * SLAB + i] != u) ++bad;
            }
            printf("  %s: %7.2f us per repetition (incl. %.0f + %.0f us of work and the consumer launch)%s\n", names[mode], ms * 1e3 / iters, busy_us, busy_us, bad ? "   RESULT MISMATCH" : "");
        }
    return 0;
}
