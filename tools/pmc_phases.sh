#!/bin/bash
# Per-phase PMC table of the tile kernel at C4 (run on the GPU box from the repo root, after `make -C .../csrc ablate`):
#   tools/pmc_phases.sh r03
# The timing-only builds (-DPT_ABLATE=1/2/3: staging / + pass 1 / + pass 2) and the full kernel (query only, and with the fused blend),
# each under two rocprofv3 --pmc passes (instruction counts; LDS activity) -> gpurun_out/<tag>/phases/*.csv and a summary table.
set -e
tag=${1:-r03}
out=gpurun_out/$tag/phases; mkdir -p $out
export TMPDIR=/tmp
run() {   # label, library ('' = the product library), extra arg
  local label=$1 lib=$2 extra=$3
  for set in "inst:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "lds:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"; do
    local name=${set%%:*} ctr=${set#*:}
    if [ -n "$lib" ]; then
      PT_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/${label}_$name -o p -- python3 tools/probe_ablate.py $label $extra > $out/${label}_$name.log 2>&1
    else
      rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/${label}_$name -o p -- python3 tools/probe_ablate.py $label $extra > $out/${label}_$name.log 2>&1
    fi
    echo "$label $name done" >> $out/progress.log
  done
}
run stage tools/_ab/libpt_ab1.so ""
run pass1 tools/_ab/libpt_ab2.so ""
run pass2 tools/_ab/libpt_ab3.so ""
run full "" ""
run fullblend "" blend
python3 tools/pmc_phase_table.py $out | tee $out/table.txt
