#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box from the repo root): tools/profile_round.sh r02 [C4]
# kernel stats of bench.py, then separate PMC passes (FETCH_SIZE / WRITE_SIZE / two SQ sets) of tools/run_step.py; CSVs under gpurun_out/<tag>/
set -e
tag=${1:-r02}; wl=${2:-C4}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_under_profiler.json 2> $out/stats.err
echo stats done >> $out/progress.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 tools/run_step.py $wl 1 > $out/fetch.log 2>&1
echo fetch done >> $out/progress.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 tools/run_step.py $wl 1 > $out/write.log 2>&1
echo write done >> $out/progress.log
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq1 -o sq1 -- python3 tools/run_step.py $wl 1 > $out/sq1.log 2>&1
echo sq1 done >> $out/progress.log
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq2 -o sq2 -- python3 tools/run_step.py $wl 1 > $out/sq2.log 2>&1
echo sq2 done >> $out/progress.log
python3 bench.py --workload $wl > $out/bench.json 2> $out/bench.err
find $out -name "*.csv" | head -40
