#!/bin/bash
# GPU box: q|k and v projections as one launch of the 128x128 family: 0 = two launches (rounds 1-5) | 1 = the shipped rule (also in front of the persistent kernel's merged
# projection up to 448 tiles) | 2 = in front of it wherever the pair applies; inside the step.
O=gpurun_out/r06; mkdir -p $O
F=$O/qkv_pair_step_ab.txt
: > $F
for shape in "224 1 bf16" "224 2 bf16" "224 3 bf16" "224 4 bf16" "224 8 bf16" "224 16 bf16" "518 1 bf16" "518 2 bf16" "518 3 bf16" "518 4 bf16" "1024 1 bf16" "224 1 f16" "518 1 f16"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $F
  timeout -k 10 200 python3 tools/step_ab.py "gemm_qkv_pair=0" "gemm_qkv_pair=1" "gemm_qkv_pair=2" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 --check 2>&1 | grep "gemm_qkv_pair=" | cut -c1-200 >> $F
done
cat $F
