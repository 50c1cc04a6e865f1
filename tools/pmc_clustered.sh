#!/bin/bash
# PMC passes of the query kernels on the clustered cloud (dev helper): tools/pmc_clustered.sh n m k thr
export TMPDIR=/tmp
out=gpurun_out/pmc_cl; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -o f -- python3 tools/run_clustered.py $1 $2 $3 $4 > $out/f.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq1 -o sq1 -- python3 tools/run_clustered.py $1 $2 $3 $4 > $out/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS --output-format csv -d $out/sq2 -o sq2 -- python3 tools/run_clustered.py $1 $2 $3 $4 > $out/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -o tcc -- python3 tools/run_clustered.py $1 $2 $3 $4 > $out/tcc.log 2>&1
for f in $(find $out -name "*counter_collection.csv"); do echo "== $f"; python3 tools/pmc_all.py $f knn_wave; done
