#!/bin/bash
# GPU box: flash attention with 64-row workgroups (automatic below 384 blocks of 128 rows; attn_variant 401 / 402 force the 64- / 128-row form) inside the step, 16-bit modes.
O=gpurun_out/r06; mkdir -p $O
: > $O/attn_qblock_step_ab.txt
for shape in "518 1 bf16" "518 2 bf16" "518 4 bf16" "224 1 bf16" "224 8 bf16" "224 32 bf16" "1024 1 bf16" "1024 2 bf16" "1536 1 f16" "518 1 f16"; do
  set -- $shape
  echo "== side $1 batch $2 $3" >> $O/attn_qblock_step_ab.txt
  timeout -k 10 170 python3 tools/step_ab.py "attn_variant=0" "attn_variant=402" "attn_variant=401" --dtype $3 --side $1 --batch $2 --rounds 5 --steps 6 --check 2>&1 | grep "attn_variant=" >> $O/attn_qblock_step_ab.txt
done
cat $O/attn_qblock_step_ab.txt
