#!/bin/bash
# builds ab/libnmf_pair_ablate_<n>.so: the shipped objects with the wave-pair kernel's instantiations recompiled under -DNMF_PAIR_ABLATE=<n>
# (0 = as shipped; 1 = no quotient, 2 = no LDS exchange of the halves of S, 3 = no end-of-chunk barrier, 4 = no nops behind product 1,
#  5 = no ds_writes of the next image, 6 = no global loads in the loop).  Timing only: the ablated kernels compute nonsense.
set -e
cd /root/repo/nmf-gpu_amd/csrc; mkdir -p ../../scratch_ab
mkdir -p /tmp/pab && cp nmf_pair16_impl.h /tmp/pab/ && patch -s /tmp/pab/nmf_pair16_impl.h ../../tools/ablation/pair_ablate.patch && cp nmf_pair16_inst.hip /tmp/pab/
FL="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-result -mllvm -enable-misched=false -mllvm -pragma-unroll-threshold=1000000"
for n in "$@"; do
  for g in 0 1 2 3; do
    /opt/rocm/bin/hipcc $FL -I/tmp/pab -I. -DNMF_PAIR_ABLATE=$n -DNMF_P16_GROUP=$g -c /tmp/pab/nmf_pair16_inst.hip -o /tmp/pab/p16_${n}_$g.o &
  done
  wait
  others=$(ls *.o | grep -v nmf_pair16_inst)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch_ab/libnmf_pair_ablate_$n.so $others /tmp/pab/p16_${n}_0.o /tmp/pab/p16_${n}_1.o /tmp/pab/p16_${n}_2.o /tmp/pab/p16_${n}_3.o -ldl -lpthread
  echo built $n
done
