#!/bin/bash
# dev probe: two builds of libpt_hip.so on the SAME box, alternating (boxes differ by a few per cent; runs on one box do not):
#   bash tools/ab_bench.sh tools/_ab/libpt_sw_old.so [workload] [repetitions]     -- against the in-tree library
# (an "old" build: git stash; bash tools/sweep_pend.sh old:""; git stash pop; make -C 3d-reconstruction-from-point-cloud_amd/csrc)
other=$1; wl=${2:-C4}; reps=${3:-3}
for rep in $(seq $reps); do for lib in 3d-reconstruction-from-point-cloud_amd/libpt_hip.so $other; do
  PT_HIP_LIB=$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; a=json.loads(sys.stdin.read()); print('$lib', round(a['ms_per_step'],2), a['kernels_ms'])"
done; done
