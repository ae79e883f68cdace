#!/bin/bash
# GPU box: the fp32 (1e-3) mode's operand form at small batches — option gemm_f32_mx 1 (MX form from 64 row tiles of 256 on) against 2 (wherever the
# rows are a multiple of 256), inside the step (tools/step_ab.py, interleaved rounds).   bash tools/f32_form_sweep.sh
O=gpurun_out/r06; mkdir -p $O
: > $O/f32_form_sweep.txt
for shape in "1024 1" "1024 2" "1024 3" "1024 4" "1024 8" "518 1" "518 2" "518 4" "518 8" "518 16" "518 32" "224 2" "224 8" "224 32" "224 128"; do
  set -- $shape
  echo "== side $1 batch $2" >> $O/f32_form_sweep.txt
  timeout -k 10 170 python3 tools/step_ab.py "gemm_f32_mx=1" "gemm_f32_mx=2" --dtype f32 --side $1 --batch $2 --rounds 4 --steps 4 2>&1 | grep "gemm_f32_mx=" >> $O/f32_form_sweep.txt
done
cat $O/f32_form_sweep.txt
