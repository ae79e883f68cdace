#!/bin/bash
# GPU box: tile geometry of the 128x128-kernel family (gemm_small_tile 0 = by grid size | 1 = 128x128 | 2 = 64x64, one wave | 3 = 128x64, two waves; ring depth automatic) inside the step.
O=gpurun_out/r06; mkdir -p $O
F=$O/small_tile_step_ab3.txt
: > $F
for shape in "518 1 bf16" "518 2 bf16" "518 4 bf16" "518 5 bf16" "224 1 bf16" "224 2 bf16" "224 4 bf16" "224 8 bf16" "224 16 bf16" "1024 1 bf16" "518 1 f32" "518 2 f32" "518 4 f32" "224 1 f32" "224 4 f32" "1024 1 f32"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $F
  timeout -k 10 200 python3 tools/step_ab.py "gemm_small_tile=0" "gemm_small_tile=1" "gemm_small_tile=2" "gemm_small_tile=3" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 --check 2>&1 | grep "gemm_small_tile=" | cut -c1-110 >> $F
done
cat $F
