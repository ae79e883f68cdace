#!/bin/bash
# GPU box: token rows per image padded to 128 (pad_rows=128) or by the shipped rule (0: to 256 for odd batches at 518^2), one process per configuration (flipping pad_rows
# inside a process re-allocates tables and workspaces), inside the step.
O=gpurun_out/r06; mkdir -p $O
F=$O/pad_rows_step_ab.txt
: > $F
for shape in "518 1 bf16" "518 3 bf16" "518 5 bf16" "518 7 bf16" "518 1 f32" "518 3 f32"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $F
  for p in 0 128 0 128; do
    timeout -k 10 200 python3 tools/step_ab.py "pad_rows=$p" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 2>&1 | grep "pad_rows=" | cut -c1-160 >> $F
  done
done
cat $F
