#!/bin/bash
# GPU box: the profiling passes whose summaries are committed under profiles/<round>/ (run from the repo root via gpurun):
#   bash tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command           -> kernel_stats.csv (+ the bench line it printed)
# 2. two --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass)      -> HBM bytes per launch (tools/pmc_summary.py)
# 3. one --pmc pass of SQ / GRBM counters                                     -> MFMA busy, VALU active, wait fractions, clock
# Counter passes carry --kernel-trace only (no other trace domain), and python3 is the program after `--`.  The FETCH / WRITE passes keep the MANGLED kernel
# names (-M): rocprofv3 7.x garbles the demangled template arguments of gemm_kernel_v8 (r04: "<bool _Accum, int, ELb0ELb0E, -1>"), and the per-epilogue
# traffic table of tools/pmc_summary.py is keyed on the EPI template argument.
set -e -o pipefail
R=${1:-r02}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs"
PMCB="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-events"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
echo "[profile] kernel trace done"
rocprofv3 --kernel-trace --mangled-kernels --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMCB > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --mangled-kernels --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMCB > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "[profile] WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $PMCB > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
echo "[profile] SQ pass done"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "[profile] default bench done"
python3 tools/pmc_summary.py $R $OUT
# 4. the other single-GPU BASELINE workloads: kernel trace + stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg4 -- python3 bench.py --batch 16 --prompts 64 --min-len 8 --max-len 32 --maps upsample --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/bench_cfg4_under_rocprof.json 2> $OUT/trace_cfg4.err
echo "[profile] cfg4 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -- python3 bench.py --dtype f16 --batch 1 --side 1536 --prompts 193 --min-len 6 --max-len 16 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/bench_cfg5_under_rocprof.json 2> $OUT/trace_cfg5.err
echo "[profile] cfg5 trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_f32 -- python3 bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench_f32_under_rocprof.json 2> $OUT/trace_f32.err
echo "[profile] fp32-mode trace done"
# 5. (round 6) the fp32 (1e-3) mode's HBM traffic: FETCH / WRITE passes of `bench.py --dtype f32` -> hbm_traffic_pmc_f32.json (bench.py's fp32 `roofline.traffic`)
PMCF="python3 bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-kernel-events"
rocprofv3 --kernel-trace --mangled-kernels --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_f32 -- $PMCF > $OUT/pmc_fetch_f32.json 2> $OUT/pmc_fetch_f32.err
rocprofv3 --kernel-trace --mangled-kernels --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_f32 -- $PMCF > $OUT/pmc_write_f32.json 2> $OUT/pmc_write_f32.err
rocprofv3 --kernel-trace --mangled-kernels --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq_f32 -- $PMCF > $OUT/pmc_sq_f32.json 2> $OUT/pmc_sq_f32.err
python3 tools/pmc_summary_f32.py $OUT > /dev/null
echo "[profile] fp32-mode FETCH / WRITE passes done"
for c in cfg4 cfg5 f32; do f=$(find $OUT/trace_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/summary/kernel_stats_$c.csv; cp $OUT/bench_${c}_under_rocprof.json $OUT/summary/ || true; done
# drop the bulky per-dispatch traces, keep the summaries (gpurun merges <= 64 MiB back)
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
ls -la $OUT | head -30
