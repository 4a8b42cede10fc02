// nmf_fused16_inst.hip -- the instantiations of the 16-column fused half-step (nmf_fused16_impl.h), compiled once per group
// of KT values (-DNMF_K16_GROUP=0..3, csrc/Makefile) so that the groups build in parallel.
#include "nmf_fused16_impl.h"

#ifndef NMF_K16_GROUP
#error "compile with -DNMF_K16_GROUP=0..3"
#endif

namespace nmf {

#define NMF_K16_INSTANTIATE(KT)                                                                                              \
    template hipError_t launch_fused_k16<KT>(const FusedArgs &, bool, hipStream_t);                                          \
    template hipError_t launch_check_k16<KT>(const float *, const float *, const float *, int, int, int, double *, hipStream_t, int, size_t, size_t, int); \
    template hipError_t launch_gemm_k16<KT>(const float *, const float *, float *, int, int, int, hipStream_t);

#if NMF_K16_GROUP == 0
NMF_K16_GROUP0(NMF_K16_INSTANTIATE)
#elif NMF_K16_GROUP == 1
NMF_K16_GROUP1(NMF_K16_INSTANTIATE)
#elif NMF_K16_GROUP == 2
NMF_K16_GROUP2(NMF_K16_INSTANTIATE)
#elif NMF_K16_GROUP == 3
NMF_K16_GROUP3(NMF_K16_INSTANTIATE)
#else
#error "NMF_K16_GROUP out of range"
#endif

}  // namespace nmf
