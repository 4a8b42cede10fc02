/* nmf_mi355x.hpp -- the snapshot's C++ surface (cuda/matrix.cuh:18-52, cuda/nmf.cu:13-28) over the C ABI of nmf_mi355x.h.
 *
 * Header-only, plain C++17, no HIP type in sight: a maintainer of recoord/nmf-gpu keeps `Matrix`, `matrix_multiply`,
 * `element_divide`, `row_divide`, `read_matrix`, `run_async` ... with the reference's argument order and meaning, and drops only
 * what belonged to CUDA -- the cublasHandle_t / cudaStream_t parameters, the launch-geometry arrays (`block_size`, `params`, the
 * scratch `Memory`: cuda/nmf.cu:53-74,91), and the ×32 padding (rows_padded == rows here: padding is internal to the library).
 * Error behaviour is the reference's: a message on stderr and exit(code) (error-check.hpp:12-17; cuda/matrix.cu:130-134).
 *
 *   #include "nmf_mi355x.hpp"
 *   using namespace nmf_ref;
 *   Matrix X = read_matrix("../X.bin"), H = read_matrix("../H.bin"), W = read_matrix("../W.bin");   // cuda/nmf.cu:37-39
 *   run_async(&W, &H, &X, CONVERGE_THRESH, MAX_ITER);                                              // cuda/nmf.cu:42
 *   write_matrix(&W, "../Wout.bin"); write_matrix(&H, "../Hout.bin");                              // cuda/nmf.cu:44-45
 *
 * update_h / update_w below are the reference's sixteen-operator iteration (cuda/nmf.cu:118-176) written against this surface, in
 * its intended ("spec") form -- what run_async computes with two fused kernels instead; tests/test_gpu_cxx_surface.py holds the
 * two to each other. */
#ifndef NMF_MI355X_HPP
#define NMF_MI355X_HPP
#include "nmf_mi355x.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace nmf_ref {

inline void check(int status, const char *what) {            /* cudaAssert, error-check.hpp:9-17 */
    if (status == NMF_OK) return;
    std::fprintf(stderr, "%s: %s (%s)\n", what, nmf_status_string(status), nmf_last_error());
    std::exit(status);
}

class Matrix {                                               /* cuda/matrix.cuh:18-39 */
  public:
    float *data = nullptr;                                   /* device buffer, column-major, ld = rows */
    uint32_t rows = 0, cols = 0, rows_padded = 0, cols_padded = 0;

    Matrix(uint32_t rows_, uint32_t cols_) {                 /* cuda/matrix.cu:42-51: uninitialised */
        matrix m;
        check(nmf_matrix_alloc_device(&m, (int)rows_, (int)cols_), "Matrix(rows, cols)");
        adopt(m);
    }
    Matrix(const float *host_data, uint32_t rows_, uint32_t cols_) {   /* cuda/matrix.cu:53-67: copy of a column-major host array */
        matrix m = {const_cast<float *>(host_data), nullptr, {(int)rows_, (int)cols_}};
        check(nmf_matrix_to_device(&m), "Matrix(host_data, rows, cols)");
        adopt(m);
    }
    Matrix(float value, uint32_t rows_, uint32_t cols_) {    /* cuda/matrix.cu:69-80: filled */
        std::vector<float> v((size_t)rows_ * cols_, value);
        matrix m = {v.data(), nullptr, {(int)rows_, (int)cols_}};
        check(nmf_matrix_to_device(&m), "Matrix(value, rows, cols)");
        adopt(m);
    }
    /* owning, hence move-only (the reference's class is copyable with an owning destructor: SURVEY 8 a1) */
    Matrix(Matrix &&o) noexcept { *this = std::move(o); }
    Matrix &operator=(Matrix &&o) noexcept {
        if (this != &o) { release(); data = o.data; rows = o.rows; cols = o.cols; rows_padded = o.rows_padded; cols_padded = o.cols_padded; o.data = nullptr; }
        return *this;
    }
    Matrix(const Matrix &) = delete;
    Matrix &operator=(const Matrix &) = delete;
    ~Matrix() { release(); }

    matrix c() const { return matrix{nullptr, data, {(int)rows, (int)cols}}; }   /* the C struct view (device side only) */
    std::vector<float> to_host() const {                     /* cuda/nmf.cu:228-232 */
        std::vector<float> v((size_t)rows * cols);
        matrix m = {v.data(), data, {(int)rows, (int)cols}};
        check(nmf_matrix_from_device(&m), "Matrix::to_host");
        return v;
    }
    void set_epsilon() { check(nmf_set_epsilon(c(), nullptr), "set_epsilon"); }                       /* cuda/matrix.cu:182-201 */
    void sum_cols(Matrix *output) { check(nmf_sum_cols(c(), output->c(), nullptr), "sum_cols"); }    /* cuda/matrix.cu:261-377: output 1 x cols */
    void sum_rows(Matrix *output) { check(nmf_sum_rows(c(), output->c(), nullptr), "sum_rows"); }    /* cuda/matrix.cu:379-503: output rows x 1 */

  private:
    void adopt(const matrix &m) { data = m.mat_d; rows = rows_padded = (uint32_t)m.dim[0]; cols = cols_padded = (uint32_t)m.dim[1]; }
    void release() { if (data) { matrix m = c(); (void)nmf_matrix_free_device(&m); data = nullptr; } }
};

/* sgemms (cuda/matrix.cuh:41-44) */
inline void matrix_multiply(Matrix *a, Matrix *b, Matrix *c) { check(nmf_matrix_multiply(a->c(), b->c(), c->c(), nullptr), "matrix_multiply"); }
inline void matrix_multiply_AtB(Matrix *a, Matrix *b, Matrix *c) { check(nmf_matrix_multiply_AtB(a->c(), b->c(), c->c(), nullptr), "matrix_multiply_AtB"); }
inline void matrix_multiply_ABt(Matrix *a, Matrix *b, Matrix *c) { check(nmf_matrix_multiply_ABt(a->c(), b->c(), c->c(), nullptr), "matrix_multiply_ABt"); }
/* element operations (cuda/matrix.cuh:46-48) */
inline void element_multiply(Matrix *a, Matrix *b, Matrix *c) { check(nmf_element_multiply(a->c(), b->c(), c->c(), nullptr), "element_multiply"); }
inline void element_divide(Matrix *a, Matrix *b, Matrix *c) { check(nmf_element_divide(a->c(), b->c(), c->c(), nullptr), "element_divide"); }
/* row / col-wise (cuda/matrix.cuh:50-52): c[i,j] = a[i,j] / b[j] resp. a[i,j] / b[i] */
inline void row_divide(Matrix *a, Matrix *b, Matrix *c) { check(nmf_row_divide(a->c(), b->c(), c->c(), nullptr), "row_divide"); }
inline void col_divide(Matrix *a, Matrix *b, Matrix *c) { check(nmf_col_divide(a->c(), b->c(), c->c(), nullptr), "col_divide"); }

/* cuda/nmf.cu:188-218: header of two uint32, column-major float32 data; every entry clamped to EPS on the device copy (:211) */
inline Matrix read_matrix(const std::string &file) {
    matrix m;
    check(nmf_read_matrix(&m, file.c_str()), "read_matrix");
    Matrix out(m.mat, (uint32_t)m.dim[0], (uint32_t)m.dim[1]);
    nmf_destroy_matrix(&m);
    out.set_epsilon();
    return out;
}
/* cuda/nmf.cu:220-259 */
inline void write_matrix(Matrix *a, const std::string &file) {
    std::vector<float> v = a->to_host();
    matrix m = {v.data(), nullptr, {(int)a->rows, (int)a->cols}};
    check(nmf_write_matrix(m, file.c_str()), "write_matrix");
}

/* cuda/nmf.cu:118-146: H = H .* (W' * (X ./ max(W*H, EPS))) ./ max(colsum(W), EPS), with the reference's temporaries */
inline void update_h(Matrix *W, Matrix *H, Matrix *X, Matrix *Z, Matrix *sumW, Matrix *WtZ) {
    matrix_multiply(W, H, Z);          /* Z = W*H            :125 */
    Z->set_epsilon();                  /*                    :128 */
    element_divide(X, Z, Z);           /* Z = X ./ Z         :131 */
    W->sum_cols(sumW);                 /* sumW = colsum(W)   :134 */
    sumW->set_epsilon();               /*                    :135 */
    matrix_multiply_AtB(W, Z, WtZ);    /* WtZ = W'*Z         :138 */
    col_divide(WtZ, sumW, WtZ);        /* WtZ[k,:] /= sumW[k] :141 */
    element_multiply(H, WtZ, H);       /* H = H .* WtZ       :144 */
}
/* cuda/nmf.cu:148-176: W = W .* ((X ./ max(W*H, EPS)) * H') ./ max(rowsum(H), EPS) */
inline void update_w(Matrix *W, Matrix *H, Matrix *X, Matrix *Z, Matrix *sumH2, Matrix *ZHt) {
    matrix_multiply(W, H, Z);          /* :155 */
    Z->set_epsilon();                  /* :158 */
    element_divide(X, Z, Z);           /* :161 */
    H->sum_rows(sumH2);                /* :164 */
    sumH2->set_epsilon();              /* :165 */
    matrix_multiply_ABt(Z, H, ZHt);    /* :168 */
    row_divide(ZHt, sumH2, ZHt);       /* ZHt[:,k] /= sumH2[k]  :171 */
    element_multiply(W, ZHt, W);       /* :174 */
}

/* cuda/nmf.cu:76-116: `max_iter` iterations of update_h, update_w on device-resident W, H (updated in place), X read-only;
 * thresh > 0 adds the README's convergence test every 25 iterations (README.md:51; the snapshot's body ignores `thresh`).
 * The fused hipGraph loop of the library, not the sixteen operators above. */
inline void run_async(Matrix *W, Matrix *H, Matrix *X, const float thresh, const uint32_t max_iter) {
    nmf_opts o;
    nmf_default_opts(&o);
    o.converge_thresh = thresh;
    o.max_iter = (int)max_iter;
    nmf_result r;
    check(update_div_ex(W->c(), H->c(), X->c(), &o, &r), "run_async");
}

}  // namespace nmf_ref
#endif /* NMF_MI355X_HPP */
