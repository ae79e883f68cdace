#!/bin/bash
# GPU box: VL-CABS head variants (tools/build_variant.py libraries) against the default build, in isolation (tools/kvlcabs.py): cfg 2, cfg 4, cfg 5 shape, 518^2 x 64, one image.
O=gpurun_out/r06; mkdir -p $O
: > $O/vlcabs_ab.txt
for lib in "" radzero_amd/libradzero_hip_vcacc4.so radzero_amd/libradzero_hip_vcpipe.so radzero_amd/libradzero_hip_vcpipe4.so; do
  for shape in "32 1024 14" "16 1024 64" "1 1536 193" "64 518 14" "1 1024 14" "1 518 14"; do
    set -- $shape
    RZ_LIB_PATH=$lib timeout -k 10 120 python3 tools/kvlcabs.py --batch $1 --side $2 --prompts $3 2>&1 | grep "vlcabs B=" >> $O/vlcabs_ab.txt
  done
done
cat $O/vlcabs_ab.txt
