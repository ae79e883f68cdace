#!/bin/bash
# GPU box: where the 128x64 tile stops paying (N = 768 GEMMs of 330 / 396 / 432 / 504 tiles of 128x128)
O=gpurun_out/r06; mkdir -p $O
F=$O/small_tile_step_ab2.txt
: > $F
for shape in "518 5 bf16" "518 6 bf16" "224 24 bf16" "1024 2 bf16" "518 6 f32" "1024 2 f32"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $F
  timeout -k 10 200 python3 tools/step_ab.py "gemm_small_tile=1" "gemm_small_tile=3" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 --check 2>&1 | grep "gemm_small_tile=" | cut -c1-250 >> $F
done
cat $F
