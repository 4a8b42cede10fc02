// Race test of nmf_comm_abort against ranks on their way into / inside ncclAllReduce (round-3 VERDICT, weak 5: the abort used to
// free a communicator handle that a peer thread had just read without a lock).  Links a ThreadSanitizer build of the product's
// nmf_comm.cpp (host side only) against tests/cpu_sanitize/fake_rccl.c posing as librccl.so.1; no GPU, no HIP call is made.
//   argv[1] = rounds.  Every round: an ncclCommInitAll group of 4, one thread per rank calling the f32 all-reduce in a loop,
//   an abort from a fifth thread (or from one of the ranks, via the NMF_FAULT_ALLREDUCE-style failure path) at a random time.
#include "../../nmf-gpu_amd/csrc/nmf_comm.h"
#include "../../include/nmf_mi355x.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

void nmf_internal_set_error(const char *) {}
// the emulated group's device-side sum (nmf_kernels.hip) is not part of this host-only build and is never reached here
hipError_t nmf_emu_sum_launch(const void *const *, int, void *, size_t, bool, hipStream_t) { return hipErrorNotSupported; }
hipError_t nmf_flag_store_launch(unsigned *, unsigned, hipStream_t) { return hipErrorNotSupported; }

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    const int G = 4;
    int devs[G] = {0, 0, 0, 0};
    long total_ok = 0, total_err = 0;
    for (int r = 0; r < rounds; ++r) {
        nmf_comm *c[G] = {};
        if (nmf_comm_init_all(c, G, devs) != NMF_OK) { fprintf(stderr, "init_all failed\n"); return 2; }
        std::atomic<long> ok{0}, err{0};
        std::atomic<bool> go{false};
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g)
            th.emplace_back([&, g]() {
                float buf[16] = {0};
                while (!go.load()) std::this_thread::yield();
                for (int i = 0; i < 100000; ++i) {
                    const int st = nmf_comm_allreduce_f32(c[g], buf, 16, nullptr);
                    if (st != NMF_OK) { err++; break; }
                    ok++;
                    // a rank that fails by itself aborts the group from its own thread (nmf_multi.cpp does this)
                    if (g == (r % G) && (r & 1) && i == 50 + (r * 37) % 400) { nmf_comm_abort(c[g]); }
                }
            });
        go = true;
        if (!(r & 1)) {   // even rounds: the abort comes from outside (the watchdog / a rank's deadline)
            std::this_thread::sleep_for(std::chrono::microseconds(200 + (r * 131) % 1500));
            nmf_comm_abort(c[(r / 2) % G]);
        }
        for (auto &t : th) t.join();
        for (int g = 0; g < G; ++g) if (!nmf_comm_aborted(c[g])) { fprintf(stderr, "round %d: rank %d not aborted\n", r, g); return 3; }
        for (int g = 0; g < G; ++g) nmf_comm_destroy(c[g]);
        total_ok += ok; total_err += err;
    }
    printf("comm race driver: %d rounds, %ld all-reduces completed, %ld ended by the abort\n", rounds, total_ok, total_err);
    return 0;
}
