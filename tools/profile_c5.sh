#!/bin/bash
# rocprofv3 kernel stats of one build + query of BASELINE config 5's shape on one GPU (1B clustered fp16 / 50M / k=32): tools/profile_c5.sh r02
tag=${1:-r02}
out=gpurun_out/${tag}_c5; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 tools/run_clustered.py 1e9 5e7 32 8192 > $out/run.log 2>&1
find $out -name "*kernel_stats.csv" | head -3
