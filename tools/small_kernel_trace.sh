#!/bin/bash
# GPU box: kernel traces of the step with the 128x128 family's geometry / ring depth forced (gemm_small_tile = 10 S + geometry), parsed into per-GEMM medians.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06/small_trace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in "518 1" "518 2" "518 4" "224 2" "224 8" "1024 1"; do
  set -- $shape
  for cfg in 21 22 23 31 32 33 41 42 43; do
    timeout -k 10 120 rocprofv3 --kernel-trace -d $O/s$1b$2__$cfg -o t --output-format csv -- python3 $R/tools/step_ab.py "gemm_small_tile=$cfg" --dtype bf16 --side $1 --batch $2 --rounds 1 --steps 12 > $O/log_s$1b$2_$cfg.txt 2>&1 || echo "run failed: $shape $cfg"
  done
  echo "done $shape"
done
cd $R && python3 tools/small_kernel_trace.py parse $O > gpurun_out/r06/small_kernel_per_gemm.txt 2>&1 || { cat gpurun_out/r06/small_kernel_per_gemm.txt; exit 1; }
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
cat gpurun_out/r06/small_kernel_per_gemm.txt
