#!/bin/bash
# rocprofv3 evidence for BASELINE config 5's shape on ONE GPU (1B clustered fp16 points / 50M targets / k = 32): tools/profile_c5.sh r04
# kernel stats of tools/run_step.py C5 2 (first build + two steps), then separate PMC passes (FETCH_SIZE / WRITE_SIZE / one SQ set) of one
# step, the per-target instrumentation (records looked at, phase times, sort-merges: make visits) and the bench line; under gpurun_out/<tag>_c5/
set -e
tag=${1:-r04}
out=gpurun_out/${tag}_c5; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 tools/run_step.py C5 2 > $out/run.log 2>&1
echo stats done >> $out/progress.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 tools/run_step.py C5 1 > $out/fetch.log 2>&1
echo fetch done >> $out/progress.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 tools/run_step.py C5 1 > $out/write.log 2>&1
echo write done >> $out/progress.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq1 -o sq1 -- python3 tools/run_step.py C5 1 > $out/sq1.log 2>&1
echo sq1 done >> $out/progress.log
if [ -f tools/_ab/libpt_visits.so ]; then
  PT_HIP_LIB=tools/_ab/libpt_visits.so python3 tools/probe_wave_visits.py 1e9 5e7 32 8192 > $out/visits.log 2>&1
  PT_HIP_LIB=tools/_ab/libpt_visits.so python3 tools/probe_wave_visits.py 1e9 5e7 32 8192 dup_runs=0 > $out/visits_no_dup_runs.log 2>&1
  echo visits done >> $out/progress.log
fi
python3 bench.py --workload C5 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
find $out -name "*.csv" | head -20
