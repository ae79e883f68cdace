#!/bin/bash
# GPU box: the 128 x 128 GEMM kernel's LDS-DMA ring, 4 stages (product) against 2 (radzero_amd/libradzero_hip_s2.so: tools/build_variant.py s2 gemm.hip -DRZ_SMALL_STAGES=2),
# inside the step at the shapes that kernel serves, prompt encode, one request.   bash tools/small_ring_ab.sh
O=gpurun_out/r06; mkdir -p $O
: > $O/small_ring_ab.txt
for lib in radzero_amd/libradzero_hip_s2.so ""; do
  echo "#### library: ${lib:-product (4 stages)}" >> $O/small_ring_ab.txt
  for shape in "518 1 bf16" "518 2 bf16" "518 4 bf16" "518 8 bf16" "518 16 bf16" "224 1 bf16" "224 8 bf16" "224 32 bf16" "1024 1 bf16" "1024 2 bf16" "1024 4 bf16" "1536 1 f16" "518 1 f32" "518 4 f32" "1024 1 f32" "224 8 f32"; do
    set -- $shape
    echo "== side $1 batch $2 $3" >> $O/small_ring_ab.txt
    RZ_LIB_PATH=$lib timeout -k 10 170 python3 tools/step_ab.py "gemm_variant=0" --dtype $3 --side $1 --batch $2 --rounds 3 --steps 6 2>&1 | grep "gemm_variant=" >> $O/small_ring_ab.txt
  done
  RZ_LIB_PATH=$lib timeout -k 10 170 python3 tools/text_encode_time.py 2>&1 | grep "T=" | grep -v "split_guard\|gemm_f32_split" >> $O/small_ring_ab.txt
done
cat $O/small_ring_ab.txt
