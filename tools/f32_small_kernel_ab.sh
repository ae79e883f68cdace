#!/bin/bash
# GPU box: fp32 (1e-3) mode, MX form — automatic kernel choice (round 6: 128 x 128 MX kernel at small shapes) against the persistent 256 x 256 kernel forced
# (gemm_variant = 8, the only MX kernel of rounds 4-5) and the small one forced (1), inside the step.   bash tools/f32_small_kernel_ab.sh
O=gpurun_out/r06; mkdir -p $O
: > $O/f32_small_mx_kernel_step_ab.txt
for shape in "1024 1" "1024 2" "1024 4" "1024 8" "1024 32" "518 1" "518 2" "518 4" "518 8" "518 16" "518 64" "224 2" "224 8" "224 32" "1536 1"; do
  set -- $shape
  echo "== side $1 batch $2" >> $O/f32_small_mx_kernel_step_ab.txt
  timeout -k 10 170 python3 tools/step_ab.py "gemm_variant=0" "gemm_variant=8" "gemm_variant=1" --dtype f32 --side $1 --batch $2 --rounds 4 --steps 4 --check 2>&1 | grep "gemm_variant=" >> $O/f32_small_mx_kernel_step_ab.txt
done
cat $O/f32_small_mx_kernel_step_ab.txt
