// query_race_probe.cpp -- which answer does hipStreamQuery give on a healthy stream while OTHER threads of the process capture graphs?
// Round 4 saw one such answer (code not recorded) in the four-thread rehearsal; round 5 removed the poll from the library.  This probe
// reproduces the setting outside the library's wait: thread A keeps a stream busy with tiny kernels and polls it with hipStreamQuery
// as fast as it can; threads B.. do what the rehearsal's ranks do, chosen by argv[1]:
//   capture   hipStreamBeginCapture(ThreadLocal) / a few launches / EndCapture / GraphInstantiate / GraphLaunch / destroy, in a loop
//   streams   hipStreamCreateWithFlags / a launch / synchronize / hipStreamDestroy, in a loop
//   library   update_div_ex through the in-library driver with a one-rank RCCL communicator (n_devices = 1 + a device list), in a loop
// Every answer other than hipSuccess / hipErrorNotReady is counted by name.   argv[2] = seconds (default 10), argv[3] = worker threads (3)
//   hipcc -O2 -Iinclude tools/query_race_probe.cpp -Lnmf-gpu_amd -lnmf_mi355x -Wl,-rpath,$PWD/nmf-gpu_amd -o /tmp/qrp && /tmp/qrp capture
#include <hip/hip_runtime.h>
#include "nmf_mi355x.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

__global__ void tick(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "capture";
    const double seconds = argc > 2 ? atof(argv[2]) : 10.0;
    const int workers = argc > 3 ? atoi(argv[3]) : 3;
    std::atomic<bool> stop{false};
    std::mutex mu;
    std::map<std::string, long> odd;
    std::atomic<long> polls{0}, work{0}, work_err{0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    std::thread poller([&] {
        hipStream_t st; float *buf;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&buf, 64 * sizeof(float)) != hipSuccess) { fprintf(stderr, "poller set-up failed\n"); return; }
        (void)hipMemset(buf, 0, 64 * sizeof(float));
        while (!stop) {
            for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(tick, dim3(64), dim3(64), 0, st, buf);
            for (;;) {
                const hipError_t q = hipStreamQuery(st);
                ++polls;
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { std::lock_guard<std::mutex> lk(mu); ++odd[hipGetErrorName(q)]; (void)hipGetLastError(); }
                if (stop) break;
            }
        }
        (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); (void)hipFree(buf);
    });

    std::vector<std::thread> th;
    for (int w = 0; w < workers; ++w) th.emplace_back([&, w] {
        if (mode == "library") {
            const int M = 512 + 128 * w, N = 2048, K = 64;
            std::vector<float> W((size_t)M * K), H((size_t)K * N), X((size_t)M * N);
            unsigned s = 12345u + w;
            auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f + 1e-3f; };
            for (auto &v : W) v = rnd(); for (auto &v : H) v = rnd(); for (auto &v : X) v = rnd();
            while (!stop) {
                std::vector<float> w2 = W, h2 = H;
                matrix mw = {w2.data(), nullptr, {M, K}}, mh = {h2.data(), nullptr, {K, N}}, mx = {X.data(), nullptr, {M, N}};
                nmf_opts o; nmf_default_opts(&o);
                int dev[1] = {0};
                o.max_iter = 80; o.n_devices = 1; o.devices = dev; o.use_graph = 1; o.converge_thresh = 1e-30f; o.iter_check = 40;
                nmf_result r;
                if (update_div_ex(mw, mh, mx, &o, &r) != NMF_OK) { ++work_err; fprintf(stderr, "worker %d: %s\n", w, nmf_last_error()); }
                ++work;
            }
            return;
        }
        float *buf;
        if (hipMalloc(&buf, 64 * sizeof(float)) != hipSuccess) return;
        hipStream_t st = nullptr;
        if (mode == "capture" && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
        while (!stop) {
            if (mode == "capture") {
                hipGraph_t g = nullptr; hipGraphExec_t e = nullptr;
                bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
                for (int i = 0; ok && i < 16; ++i) hipLaunchKernelGGL(tick, dim3(64), dim3(64), 0, st, buf);
                ok = hipStreamEndCapture(st, &g) == hipSuccess && ok;
                ok = ok && hipGraphInstantiate(&e, g, nullptr, nullptr, 0) == hipSuccess;
                ok = ok && hipGraphLaunch(e, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
                if (e) (void)hipGraphExecDestroy(e);
                if (g) (void)hipGraphDestroy(g);
                if (!ok) { ++work_err; (void)hipGetLastError(); }
            } else {
                hipStream_t s2;
                bool ok = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess;
                if (ok) { hipLaunchKernelGGL(tick, dim3(64), dim3(64), 0, s2, buf); ok = hipStreamSynchronize(s2) == hipSuccess; (void)hipStreamDestroy(s2); }
                if (!ok) { ++work_err; (void)hipGetLastError(); }
            }
            ++work;
        }
        if (st) (void)hipStreamDestroy(st);
        (void)hipFree(buf);
    });

    const double t0 = now();
    while (now() - t0 < seconds) std::this_thread::sleep_for(std::chrono::milliseconds(50));
    stop = true;
    for (auto &t : th) t.join();
    poller.join();
    printf("mode %s, %d workers, %.0f s: %ld polls, %ld worker rounds (%ld failed); answers other than Success / NotReady:", mode.c_str(), workers, seconds, polls.load(), work.load(), work_err.load());
    if (odd.empty()) printf(" none\n");
    else { printf("\n"); for (auto &kv : odd) printf("    %s x %ld\n", kv.first.c_str(), kv.second); }
    return 0;
}
