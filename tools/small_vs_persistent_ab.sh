#!/bin/bash
# GPU box: after the small-tile geometries, is the kernel-choice cost model (gemm.hip big_tiles_pay / gemm_qkv_fused_ok) still right?  Step time with everything
# forced onto the 128x128 family (gemm_variant=1) against the automatic choice.
O=gpurun_out/r06; mkdir -p $O
F=$O/small_vs_persistent_step_ab.txt
: > $F
for shape in "224 2 bf16" "224 4 bf16" "224 8 bf16" "224 16 bf16" "224 32 bf16" "518 2 bf16" "518 4 bf16" "518 6 bf16" "518 8 bf16" "518 12 bf16" "1024 1 bf16" "1024 2 bf16" "1024 3 bf16"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $F
  timeout -k 10 200 python3 tools/step_ab.py "gemm_variant=0" "gemm_variant=1" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 2>&1 | grep "gemm_variant=" | cut -c1-250 >> $F
done
cat $F
