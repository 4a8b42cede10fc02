// nmf_pair16_inst.hip -- the instantiations of the wave-pair fused half-step (nmf_pair16_impl.h), compiled once per group of
// KTH values (-DNMF_P16_GROUP=0..3, csrc/Makefile) so that the groups build in parallel.
#include "nmf_pair16_impl.h"

#ifndef NMF_P16_GROUP
#error "compile with -DNMF_P16_GROUP=0..3"
#endif

namespace nmf {

#define NMF_P16_INSTANTIATE(KTH)                                                          \
    template hipError_t launch_pair_kth<KTH>(const FusedArgs &, bool, hipStream_t);       \
    template hipError_t launch_check_kth<KTH>(const float *, const float *, const float *, int, int, int, double *, hipStream_t);

#if NMF_P16_GROUP == 0
NMF_P16_GROUP0(NMF_P16_INSTANTIATE)
#elif NMF_P16_GROUP == 1
NMF_P16_GROUP1(NMF_P16_INSTANTIATE)
#elif NMF_P16_GROUP == 2
NMF_P16_GROUP2(NMF_P16_INSTANTIATE)
#elif NMF_P16_GROUP == 3
NMF_P16_GROUP3(NMF_P16_INSTANTIATE)
#else
#error "NMF_P16_GROUP out of range"
#endif

}  // namespace nmf
