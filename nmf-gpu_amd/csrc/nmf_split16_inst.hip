// nmf_split16_inst.hip -- the instantiations of the split half-step (nmf_split16_impl.h), compiled once per group of
// configurations (-DNMF_S16_GROUP=0..3, csrc/Makefile) so that the groups build in parallel.
#include "nmf_split16_impl.h"

#ifndef NMF_S16_GROUP
#error "compile with -DNMF_S16_GROUP=0..3"
#endif

namespace nmf {

#define NMF_S16_INSTANTIATE(KT, NW, OCC, DB) template hipError_t launch_split_k16<KT, NW, OCC, DB>(const SplitArgs &, bool, hipStream_t);

#if NMF_S16_GROUP == 0
NMF_S16_GROUP0(NMF_S16_INSTANTIATE)
#elif NMF_S16_GROUP == 1
NMF_S16_GROUP1(NMF_S16_INSTANTIATE)
#elif NMF_S16_GROUP == 2
NMF_S16_GROUP2(NMF_S16_INSTANTIATE)
#elif NMF_S16_GROUP == 3
NMF_S16_GROUP3(NMF_S16_INSTANTIATE)
#else
#error "NMF_S16_GROUP out of range"
#endif

}  // namespace nmf
