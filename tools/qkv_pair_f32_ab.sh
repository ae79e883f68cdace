#!/bin/bash
# GPU box: fp32 mode, q|k and v projections as one launch of the 128x128 family (gemm_qkv_pair 1) against two (0), inside the step.
O=gpurun_out/r06; mkdir -p $O
F=$O/qkv_pair_f32_step_ab.txt
: > $F
for shape in "224 1" "224 2" "224 3" "224 4" "518 1" "518 2" "518 3" "518 4" "1024 1"; do
  set -- $shape
  echo "== side $1 batch $2 f32" >> $F
  timeout -k 10 200 python3 tools/step_ab.py "gemm_qkv_pair=0" "gemm_qkv_pair=1" --dtype f32 --side $1 --batch $2 --rounds 5 --steps 6 --check 2>&1 | grep "gemm_qkv_pair=" | cut -c1-200 >> $F
done
cat $F
