// nmf_fused16.hip -- dispatch of the 16-column fused half-step (nmf_fused16_impl.h) over its K = 16 KT instantiations
// (nmf_fused16_inst.hip): half-steps, the KL check and C = W * H.
#include "nmf_fused16_impl.h"

namespace nmf {

#define NMF_K16_EXTERN(KT)                                                                                                   \
    extern template hipError_t launch_fused_k16<KT>(const FusedArgs &, bool, hipStream_t);                                   \
    extern template hipError_t launch_check_k16<KT>(const float *, const float *, const float *, int, int, int, double *, hipStream_t, int, size_t, size_t, int); \
    extern template hipError_t launch_gemm_k16<KT>(const float *, const float *, float *, int, int, int, hipStream_t);
NMF_K16_ALL(NMF_K16_EXTERN)
#undef NMF_K16_EXTERN

// the K the 16-column kernel computes on for a logical K: the next multiple of 16 (every one up to kMaxK16 = 576 is instantiated), 0 = none
int fused16_compute_k(int K) {
    if (K > kMaxK16 || K < 1) return 0;
    return (K + 15) & ~15;
}
static bool k16_shape_ok(int Kp, int Kc) { return Kc >= 16 && Kc <= kMaxK16 && Kc == fused16_compute_k(Kc) && Kp == ((Kc + 31) & ~31); }

// C(Mp x Np) = A(Mp x Kp) * B(Kp x Np), all contiguous column-major; Mp % 32 == 0, Np % 16 == 0, Kp a multiple of 32 in [64, 512]
// (the kernel stages whole 32-column pieces of A: no padding inside a caller's matrix to read instead)
hipError_t launch_gemm_nn16(const float *A, const float *B, float *C, int Mp, int Np, int Kp, hipStream_t stream) {
    if ((Mp & 31) || (Np & 15) || Np < 16 || (Kp % 32) || !k16_shape_ok(Kp, Kp) || (size_t)Mp * (size_t)Kp >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    switch (Kp / 16) {
#define NMF_K16_CASE(KT) case KT: return launch_gemm_k16<KT>(A, B, C, Mp, Np, Kp, stream);
        NMF_K16_ALL(NMF_K16_CASE)
#undef NMF_K16_CASE
        default: return hipErrorInvalidValue;
    }
}
bool gemm_nn16_eligible(int m, int n, int k, long lda, long ldb, long ldc) {
    return (m % 32 == 0) && (n % 16 == 0) && n >= 16 && (k % 32 == 0) && k16_shape_ok(k, k) && lda == m && ldb == k && ldc == m &&
           (size_t)m * (size_t)k < ((size_t)1 << 31);
}

hipError_t launch_fused16(const FusedArgs &a, bool wstep, hipStream_t stream) {
    const int kc = a.Kc > 0 ? a.Kc : a.Kp;
    if (!k16_shape_ok(a.Kp, kc)) return hipErrorInvalidValue;
    switch (kc / 16) {
#define NMF_K16_CASE(KT) case KT: return launch_fused_k16<KT>(a, wstep, stream);
        NMF_K16_ALL(NMF_K16_CASE)
#undef NMF_K16_CASE
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_check16(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, int Kc, double *part, hipStream_t stream,
                          int batch, size_t strideW, size_t strideH, int nsplit) {
    const int kc = Kc > 0 ? Kc : Kp;
    if (!k16_shape_ok(Kp, kc)) return hipErrorInvalidValue;
    switch (kc / 16) {
#define NMF_K16_CASE(KT) case KT: return launch_check_k16<KT>(W, H, X, Mp, Np, Kp, part, stream, batch, strideW, strideH, nsplit);
        NMF_K16_ALL(NMF_K16_CASE)
#undef NMF_K16_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace nmf
