#!/bin/bash
# GPU box: the 128x128 kernel family's launch rule (gemm_small_tile 0) against its round-5 form (21 = 128x128 tiles, two stages) and the forced corners, inside the step.
O=gpurun_out/r06; mkdir -p $O
F=$O/small_ring_step_ab.txt
: > $F
for shape in "518 1 bf16" "518 2 bf16" "518 3 bf16" "518 4 bf16" "518 5 bf16" "224 1 bf16" "224 2 bf16" "224 4 bf16" "224 8 bf16" "224 16 bf16" "1024 1 bf16" "1536 1 bf16" "518 1 f16" "518 1 f32" "518 2 f32" "518 4 f32" "224 1 f32" "224 4 f32" "1024 1 f32"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $F
  timeout -k 10 200 python3 tools/step_ab.py "gemm_small_tile=21" "gemm_small_tile=0" "gemm_small_tile=41" "gemm_small_tile=23" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 --check 2>&1 | grep "gemm_small_tile=" | cut -c1-200 >> $F
done
python3 tools/text_encode_time.py >> $F 2>&1
cat $F
