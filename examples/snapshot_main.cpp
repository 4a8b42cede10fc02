// The snapshot's main() (cuda/nmf.cu:30-51) on its own C++ surface -- Matrix, read_matrix, run_async, write_matrix -- through
// include/nmf_mi355x.hpp: what a maintainer of recoord/nmf-gpu keeps when cuda/matrix.cu and the bodies in cuda/nmf.cu are replaced by
// libnmf_mi355x.so.  Gone: the stream and the cuBLAS handle (cuda/nmf.cu:31-35,47-48) and the arguments that carried them.
//   snapshot_main                     the reference's run: ../X.bin ../H.bin ../W.bin -> ../Wout.bin ../Hout.bin, MAX_ITER iterations
//   snapshot_main operators <iters>   the same files through the reference's own sixteen-operator iteration (cuda/nmf.cu:118-176,
//                                     written against the mirrored operators in nmf_mi355x.hpp) instead of the fused loop
// Build:  g++ -O2 -std=c++17 -Iinclude examples/snapshot_main.cpp -Lnmf-gpu_amd -lnmf_mi355x -Wl,-rpath,$PWD/nmf-gpu_amd -o snapshot_main
#include "nmf_mi355x.hpp"

#include <cstring>

using namespace nmf_ref;

#define MAX_ITER 200      // cuda/nmf.cu:10
#define CONVERGE_THRESH 0 // cuda/nmf.cu:11

int main(int argc, char *argv[]) {
    Matrix X = read_matrix("../X.bin");
    Matrix H = read_matrix("../H.bin");
    Matrix W = read_matrix("../W.bin");

    if (argc > 2 && !std::strcmp(argv[1], "operators")) {
        // the temporaries of run_async (cuda/nmf.cu:91-98) and its loop body (:104-107), eagerly
        Matrix Z(X.rows, X.cols), sumW(1, W.cols), WtZ(H.rows, H.cols), sumH2(H.rows, 1), ZHt(W.rows, W.cols);
        for (int i = 0; i < std::atoi(argv[2]); ++i) {
            update_h(&W, &H, &X, &Z, &sumW, &WtZ);
            update_w(&W, &H, &X, &Z, &sumH2, &ZHt);
        }
    } else {
        // Run iterative nmf minimization
        run_async(&W, &H, &X, CONVERGE_THRESH, MAX_ITER);
    }

    write_matrix(&W, "../Wout.bin");
    write_matrix(&H, "../Hout.bin");
    return 0;
}
