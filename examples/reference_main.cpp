// The reference's main() (cuda/nmf.cu:30-51) rebound to libnmf_mi355x.so, exactly as INTEGRATION.md section 1 shows it:
// same hard-coded file names, same MAX_ITER / CONVERGE_THRESH defines, update_div in place of run_async.
// Build:  g++ -O2 -Iinclude examples/reference_main.cpp -Lnmf-gpu_amd -lnmf_mi355x -Wl,-rpath,$PWD/nmf-gpu_amd -o nmf_ref_main
#include "nmf_mi355x.h"            // was: error-check.hpp, matrix.cuh

#define MAX_ITER 200               // cuda/nmf.cu:10
#define CONVERGE_THRESH 0          // cuda/nmf.cu:11

int main() {
    matrix X, H, W;
    if (nmf_read_matrix(&X, "../X.bin") || nmf_read_matrix(&H, "../H.bin") || nmf_read_matrix(&W, "../W.bin"))
        return 1;                                  // was read_matrix(), cuda/nmf.cu:37-39 (fopen unchecked there)
    update_div(W, H, X, CONVERGE_THRESH, MAX_ITER, /*t=*/nullptr, /*verbose=*/0);   // was run_async(), cuda/nmf.cu:42
    nmf_write_matrix(W, "../Wout.bin");            // cuda/nmf.cu:44
    nmf_write_matrix(H, "../Hout.bin");            // cuda/nmf.cu:45
    nmf_destroy_matrix(&X); nmf_destroy_matrix(&H); nmf_destroy_matrix(&W);
    return 0;
}
