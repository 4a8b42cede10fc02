// seg_bw_probe.hip -- what does HBM deliver for the H-step's X access pattern?  A wave of the 64-column kernel reads, per 32-row chunk,
// 16 columns x 128 contiguous bytes of a column-major X (column stride 4 M bytes: 16 KB at M = 4096) and walks down the rows chunk by
// chunk.  This probe streams a 4096 x 65536 fp32 X that way with SEG = 128, 256 or 512 contiguous bytes per column and visit
// (= 32, 64, 128-row chunks), nothing else in the kernel, and reports TB/s -- the ceiling a half-step at small K could reach by
// changing its chunk height.    hipcc -O3 --offload-arch=gfx950 tools/seg_bw_probe.hip -o /tmp/seg_bw && /tmp/seg_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave owns 16 columns; lanes: LPC = SEG / 16 lanes per column, 64 / LPC columns per pass, 16 / (64 / LPC) passes per visit
template <int SEG>
__global__ __launch_bounds__(256) void stream_kernel(const float *__restrict__ X, int M, int N, float *__restrict__ out) {
    constexpr int LPC = SEG / 16, CPP = 64 / LPC, PASSES = 16 / CPP, ROWS = SEG / 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q0 = (blockIdx.x * 4 + wave) * 16;
    if (q0 >= N) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t ld = (size_t)M;
    for (int p0 = 0; p0 < M; p0 += ROWS) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int col = q0 + ps * CPP + lane / LPC;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(X + (size_t)col * ld + p0 + 4 * (lane % LPC));
            acc += v;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}

// the W-step's pattern: a wave owns 16 ROWS (64 contiguous bytes of every column), the workgroup's four waves 64 adjacent rows (256 B per
// column); per chunk the wave reads 32 columns (4 lanes per column, 16 columns per pass), walking along the columns
__global__ __launch_bounds__(256) void stream_w_kernel(const float *__restrict__ X, int M, int N, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + wave) * 16;
    const int nsplit = gridDim.y, c_per = N / nsplit, c0 = blockIdx.y * c_per;
    if (r0 >= M) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = c0; p0 < c0 + c_per; p0 += 32) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int col = p0 + ps * 16 + (lane >> 2);
            acc += *reinterpret_cast<const f32x4 *>(X + (size_t)col * (size_t)M + r0 + 4 * (lane & 3));
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1.f;
}
static void run_w(const float *X, int M, int N, float *out, int nsplit) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const dim3 grid(M / 64, nsplit), block(256);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_w_kernel, grid, block, 0, 0, X, M, N, out);
    (void)hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_w_kernel, grid, block, 0, 0, X, M, N, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("  W-step pattern (64 B per wave and column, 256 B per workgroup; %d cuts of the columns = %d workgroups): %.3f ms = %.2f TB/s\n", nsplit, (M / 64) * nsplit, ms, 4.0 * M * N / (ms * 1e-3) / 1e12);
}

template <int SEG>
static void run(const float *X, int M, int N, float *out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid((N / 16 + 3) / 4), block(256);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_kernel<SEG>, grid, block, 0, 0, X, M, N, out);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(stream_kernel<SEG>, grid, block, 0, 0, X, M, N, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("  %3d contiguous bytes per column and visit (%3d-row chunks): %.3f ms = %.2f TB/s\n", SEG, SEG / 4, ms, 4.0 * M * N / (ms * 1e-3) / 1e12);
}

int main() {
    for (int M : {4096, 8192}) {
        const int N = M == 4096 ? 65536 : 32768;
        float *X, *out;
        hipMalloc(&X, (size_t)M * N * 4); hipMalloc(&out, 256);
        hipMemset(X, 0, (size_t)M * N * 4);
        printf("X %d x %d fp32 column-major (%.2f GB), one wave per 16 columns, 8 waves / SIMD possible:\n", M, N, 4.0 * M * N / 1e9);
        run<128>(X, M, N, out); run<256>(X, M, N, out); run<512>(X, M, N, out);
        run<128>(X, M, N, out);
        run_w(X, M, N, out, 16); run_w(X, M, N, out, 32); run_w(X, M, N, out, 64);
        hipFree(X); hipFree(out);
    }
    return 0;
}
