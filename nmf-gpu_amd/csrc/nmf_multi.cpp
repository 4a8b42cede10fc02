// nmf_multi.cpp -- update_div over the GPUs of one node, inside the library (SURVEY 8b "Threading": multi-GPU is
// callee-internal, invisible at the boundary): ONE process, one host thread per device, column shards of X and H,
// a full copy of W everywhere, and per iteration one RCCL all-reduce of [Z_g*H_g' ; rowsum(H_g)] captured inside each
// device's hipGraph.  No Python, no torch: the caller is the reference's plain `main` (cuda/nmf.cu:30-51).
//
// Each rank runs the ordinary resident solver (nmf_solver_*, nmf_host.cpp) with nmf_opts.comm set; this file only shards,
// spawns, and gathers.  With nmf_opts.emulate_shards = G the G ranks live on ONE device and the all-reduce is a device-side
// sum behind host rendezvous (nmf_comm.cpp: EmuGroup) -- the same driver, testable on a one-GPU box.
#include "../../include/nmf_mi355x.h"
#include "nmf_comm.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rank {
    int status = NMF_OK;
    char err[512] = "";
    nmf_result res;
    std::vector<float> w_copy;   // every rank's W after the run (replica check)
    std::vector<float> h_copy;   // this rank's column block of H: the caller's H.mat is written only once EVERY rank has succeeded
};

// All ranks meet here after their set-up; if any failed, every rank learns of it and none enters the loop (a rank that
// returned early would leave the others waiting inside a collective).
struct SetupGate {
    std::mutex mu; std::condition_variable cv; int n, arrived = 0; bool failed = false;
    explicit SetupGate(int n_) : n(n_) {}
    bool pass(bool ok) {   // returns true if every rank is ok
        std::unique_lock<std::mutex> lk(mu);
        if (!ok) failed = true;
        if (++arrived == n) cv.notify_all(); else cv.wait(lk, [&] { return arrived == n; });
        return !failed;
    }
};

// contiguous column blocks; the first N % G ranks get one more column (sharded.py: column_shards)
void column_shards(int N, int G, std::vector<int> &start, std::vector<int> &count) {
    start.resize((size_t)G); count.resize((size_t)G);
    const int base = N / G, extra = N % G;
    int s = 0;
    for (int g = 0; g < G; ++g) { count[(size_t)g] = base + (g < extra ? 1 : 0); start[(size_t)g] = s; s += count[(size_t)g]; }
}

}  // namespace

// "only where N is large enough to amortise the all-reduce" (north_star): per-GPU compute of an iteration at ~1e14 flop/s
// against 2 x 4 (M K + K) bytes at ~1e11 B/s per xGMI link plus ~20 us of latency; shard when compute >= 4 x that
extern "C" int nmf_worth_sharding(int M, int N, int K, int n_devices) {
    if (n_devices <= 1 || N < n_devices) return 0;
    const double compute_s = 8.0 * M * ((double)N / n_devices) * K / 1.0e14;
    const double comm_s = 2.0 * 4.0 * ((double)M * K + K) / 1.0e11 + 20e-6;
    return compute_s >= 4.0 * comm_s ? 1 : 0;
}

int nmf_update_div_multi(matrix W, matrix H, matrix X, const nmf_opts &o, const int *devices, int G, bool emulate, nmf_result *res) {
    const double t_begin = now_s();
    const int M = W.dim[0], K = W.dim[1], N = H.dim[1];
    std::vector<int> start, count;
    column_shards(N, G, start, count);
    std::vector<nmf_comm *> comm((size_t)G, nullptr);
    if (emulate && hipSetDevice(devices[0]) != hipSuccess) return NMF_ERR_HIP;
    const int cst = emulate ? nmf_comm_create_emulated(comm.data(), G) : nmf_comm_init_all(comm.data(), G, devices);
    if (cst != NMF_OK) return cst;
    std::vector<Rank> rank((size_t)G);
    std::vector<std::thread> th;
    // 0 = setting up (uploads: no collective can block yet), 1 = in the loop (watched), 2 = finished
    std::vector<std::atomic<int>> phase((size_t)G);
    for (auto &p : phase) p = 0;
    int stall_rank = -1;
    if (const char *e = getenv("NMF_FAULT_STALL_RANK")) stall_rank = atoi(e);
    SetupGate gate(G);
    SetupGate captured(stall_rank >= 0 && stall_rank < G ? G - 1 : G);   // the rank of the stall injection never gets that far
    const double t_setup = now_s() - t_begin;
    for (int g = 0; g < G; ++g) {
        th.emplace_back([&, g]() {
            Rank &r = rank[(size_t)g];
            memset(&r.res, 0, sizeof r.res);
            auto fail = [&](int st) { r.status = st; snprintf(r.err, sizeof r.err, "rank %d: %s", g, nmf_last_error()); };
            if (hipSetDevice(devices[g]) != hipSuccess) { r.status = NMF_ERR_HIP; snprintf(r.err, sizeof r.err, "rank %d: hipSetDevice(%d) failed", g, devices[g]); }
            nmf_opts og = o;
            og.device = devices[g];
            og.comm = comm[(size_t)g];
            og.stream = nullptr;
            og.n_devices = 1; og.emulate_shards = 0;
            // every rank takes the same decision about convergence checks (they all-reduce three doubles: a rank that skipped
            // them would pair its next collective with the others' check); only rank 0 prints (verbose = 2: check, silently)
            og.verbose = o.verbose ? (g == 0 ? 1 : 2) : 0;
            if (!nmf_comm_capturable(comm[(size_t)g])) og.use_graph = 0;
            nmf_solver *s = nullptr;
            int st = r.status;
            const double t0 = now_s();
            if (st == NMF_OK) st = nmf_solver_create(&s, M, count[(size_t)g], K, &og);
            // W is broadcast once (every rank uploads the same host copy); H and X by column block: column-major, so a block is contiguous
            if (st == NMF_OK) st = nmf_solver_upload(s, W.mat, H.mat + (size_t)start[(size_t)g] * K, X.mat + (size_t)start[(size_t)g] * M);
            if (st == NMF_OK) st = nmf_solver_sync(s);
            r.res.t[NMF_T_H2D] = now_s() - t0;
            if (st != NMF_OK) fail(st);
            struct Done { std::atomic<int> &p; ~Done() { p = 2; } } done_guard{phase[(size_t)g]};
            if (!gate.pass(st == NMF_OK)) { if (s) nmf_solver_destroy(s); if (r.status == NMF_OK) { r.status = NMF_ERR_COMM; snprintf(r.err, sizeof r.err, "rank %d: another rank failed during set-up", g); } return; }
            phase[(size_t)g] = 1;
            // Every rank brings up its communicator (the eager first all-reduce, waited for with the deadline) and captures ALL the
            // graphs its run will replay -- one capture at a time, process-wide (nmf_host.cpp: capture_graph) -- and only when every
            // rank has done so does any rank replay or wait on a stream: no thread of this call captures while another one launches,
            // polls or synchronises (round-4 VERDICT weak 2; the reference's capture, cuda/nmf.cu:100-115, has one thread and no such question).
            if (stall_rank != g) {
                st = nmf_solver_prepare(s, o.max_iter < 41 ? o.max_iter : 41);   // 32 + 8 + 1: every graph length
                if (st != NMF_OK) { fail(st); nmf_comm_abort(comm[(size_t)g]); }
                if (!captured.pass(st == NMF_OK)) { nmf_solver_destroy(s); if (r.status == NMF_OK) { r.status = NMF_ERR_COMM; snprintf(r.err, sizeof r.err, "rank %d: another rank failed while capturing its graphs", g); } return; }
            }
            if (stall_rank == g) {   // fault injection (NMF_FAULT_STALL_RANK): this rank never reaches its first collective
                const double t_s = now_s();
                while (!nmf_comm_aborted(comm[(size_t)g]) && now_s() - t_s < 20.0 * nmf_comm_timeout_s()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
                r.status = NMF_ERR_COMM; snprintf(r.err, sizeof r.err, "rank %d: stalled before its first collective (NMF_FAULT_STALL_RANK)", g);
                nmf_comm_abort(comm[(size_t)g]); nmf_solver_destroy(s);
                return;
            }
            nmf_result rr;
            st = nmf_solver_run(s, o.converge_thresh, o.max_iter, o.iter_check, og.verbose, &rr);
            if (st != NMF_OK) { fail(st); nmf_comm_abort(comm[(size_t)g]); nmf_solver_destroy(s); return; }
            const double t1 = now_s();
            r.w_copy.resize((size_t)M * K);
            r.h_copy.resize((size_t)K * count[(size_t)g]);
            phase[(size_t)g] = 2;   // no collective from here on
            st = nmf_solver_download(s, r.w_copy.data(), r.h_copy.data());
            const double h2d = r.res.t[NMF_T_H2D];
            r.res = rr;
            r.res.t[NMF_T_H2D] = h2d;
            r.res.t[NMF_T_D2H] = now_s() - t1;
            if (st != NMF_OK) fail(st);
            nmf_solver_destroy(s);
        });
    }
    // The calling thread is the watchdog.  A rank's own waits carry deadlines (nmf_comm_wait, the emulated rendezvous), but a
    // rank blocked INSIDE a host call -- RCCL connects transports inside the first collective's enqueue and waits there for
    // its peers -- can be freed only from another thread: if a rank in the loop shows no sign of life (nmf_comm_heartbeat)
    // for three time-outs, abort the group; every blocked call then returns an error and the threads end.
    {
        std::vector<long> last((size_t)G, -1);
        std::vector<double> since((size_t)G, now_s());
        const double limit = 3.0 * nmf_comm_timeout_s();
        bool fired = false;
        for (;;) {
            bool all_done = true;
            for (int g = 0; g < G; ++g) {
                const int ph = phase[(size_t)g].load();
                if (ph != 2) all_done = false;
                if (ph != 1 || fired) { since[(size_t)g] = now_s(); continue; }
                const long b = nmf_comm_heartbeat(comm[(size_t)g]);
                if (b != last[(size_t)g]) { last[(size_t)g] = b; since[(size_t)g] = now_s(); }
                else if (now_s() - since[(size_t)g] > limit) {
                    fprintf(stderr, "nmf: rank %d has not moved for %.0f s (blocked inside a collective?); aborting the communicator group\n", g, limit);
                    nmf_comm_abort(comm[(size_t)g]);
                    fired = true;
                }
            }
            if (all_done) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(fired ? 5 : 20));
        }
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < G; ++g) nmf_comm_destroy(comm[(size_t)g]);
    int st = NMF_OK;
    for (int g = 0; g < G && st == NMF_OK; ++g)
        if (rank[(size_t)g].status != NMF_OK) { st = rank[(size_t)g].status; fprintf(stderr, "nmf: %s\n", rank[(size_t)g].err); nmf_internal_set_error(rank[(size_t)g].err); }
    if (st != NMF_OK) return st;   // the caller's W.mat / H.mat are untouched: a fallback may start again from them
    // every rank applied the same update to the same all-reduced operand: the replicas of W must agree bit for bit
    int identical = 1;
    for (int g = 1; g < G; ++g)
        if (memcmp(rank[(size_t)g].w_copy.data(), rank[0].w_copy.data(), sizeof(float) * (size_t)M * K) != 0) identical = 0;
    if (identical) {   // H is gathered, W copied, only now
        memcpy(W.mat, rank[0].w_copy.data(), sizeof(float) * (size_t)M * K);
        for (int g = 0; g < G; ++g) memcpy(H.mat + (size_t)start[(size_t)g] * K, rank[(size_t)g].h_copy.data(), sizeof(float) * (size_t)K * count[(size_t)g]);
    }
    if (res) {
        *res = rank[0].res;
        res->n_shards = G;
        res->w_replicas_identical = identical;
        res->t[NMF_T_SETUP] += t_setup;
        res->t[NMF_T_TOTAL] = now_s() - t_begin;
    }
    if (!identical) { fprintf(stderr, "nmf: the replicas of W differ between shards\n"); return NMF_ERR_COMM; }
    return NMF_OK;
}
