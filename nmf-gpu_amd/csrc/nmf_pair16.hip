// nmf_pair16.hip -- dispatch of the wave-pair fused half-step (nmf_pair16_impl.h) over its K = 32 KTH instantiations
// (nmf_pair16_inst.hip): half-steps and the KL check for 576 < K <= 1024.
#include "nmf_pair16_impl.h"

namespace nmf {

#define NMF_P16_EXTERN(KTH)                                                                      \
    extern template hipError_t launch_pair_kth<KTH>(const FusedArgs &, bool, hipStream_t);       \
    extern template hipError_t launch_check_kth<KTH>(const float *, const float *, const float *, int, int, int, double *, hipStream_t);
NMF_P16_ALL(NMF_P16_EXTERN)
#undef NMF_P16_EXTERN

// the K the wave-pair kernel computes on for a logical K in (576, 1024]: the next multiple of 32 (the reference's PAD_MULT,
// cuda/matrix.cuh:7); the factors are 64-padded in HBM (pair_pad_k): the staging moves whole 64-column pieces.  0 = none
int pair_compute_k(int K) { return (K > kMaxK16 && K <= 1024) ? ((K + 31) & ~31) : 0; }
int pair_pad_k(int K) { return (K > kMaxK16 && K <= 1024) ? ((K + 63) & ~63) : 0; }
static bool pair_shape_ok(int Kp, int Kc) { return Kc > kMaxK16 && Kc <= 1024 && Kc % 32 == 0 && Kp == ((Kc + 63) & ~63); }

hipError_t launch_fused_pair(const FusedArgs &a, bool wstep, hipStream_t stream) {
    const int kc = a.Kc > 0 ? a.Kc : a.Kp;
    if (!pair_shape_ok(a.Kp, kc) || (size_t)a.Mp * (size_t)a.Kp >= ((size_t)1 << 30) || (size_t)a.Np * (size_t)a.Kp >= ((size_t)1 << 62)) return hipErrorInvalidValue;
    switch (kc / 32) {
#define NMF_P16_CASE(KTH) case KTH: return launch_pair_kth<KTH>(a, wstep, stream);
        NMF_P16_ALL(NMF_P16_CASE)
#undef NMF_P16_CASE
        default: return hipErrorInvalidValue;
    }
}
hipError_t launch_check_pair(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, int Kc, double *part, hipStream_t stream) {
    const int kc = Kc > 0 ? Kc : Kp;
    if (!pair_shape_ok(Kp, kc)) return hipErrorInvalidValue;
    switch (kc / 32) {
#define NMF_P16_CASE(KTH) case KTH: return launch_check_kth<KTH>(W, H, X, Mp, Np, Kp, part, stream);
        NMF_P16_ALL(NMF_P16_CASE)
#undef NMF_P16_CASE
        default: return hipErrorInvalidValue;
    }
}

}  // namespace nmf
