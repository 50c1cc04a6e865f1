#!/bin/bash
# (rounds 2 - 4: the sorted list with sort-merges.  The final tree of round 4 keeps an unsorted pool; the one parameter left is -DPT_PEND_FLUSH,
#  swept with `make ab NAME=pf24 DEF=-DPT_PEND_FLUSH=24` and tools/ab_wave.sh -- DESIGN.md section 10, item 10.)
# dev probe: the wave kernel's selection parameters (-DPT_MERGE_MIN / -DPT_PEND_FLUSH / -DPT_PEND_MIN_K) on the clustered generator.
#   bash tools/sweep_pend.sh m20f44:"-DPT_MERGE_MIN=20 -DPT_PEND_FLUSH=44" none:"-DPT_PEND_MIN_K=99" ...   (builds here, then: gpurun -- 'bash tools/sweep_pend.sh run')
set -e
csrc=3d-reconstruction-from-point-cloud_amd/csrc
if [ "$1" != run ]; then
  make -C $csrc -j8 >/dev/null; mkdir -p $csrc/_build/ab tools/_ab; rm -f tools/_ab/libpt_sw_*.so
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    ( cd $csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c pt_query.hip -o _build/ab/pt_query_sw_$name.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/libpt_sw_$name.so _build/ab/pt_query_sw_$name.o _build/pt_grid.o _build/pt_attr.o _build/pt_bake.o _build/pt_exchange.o _build/pt_refine.o _build/pt_api.o ) &
  done
  wait; exit 0
fi
out=gpurun_out/sweep_pend.log; : > $out
for lib in $csrc/../libpt_hip.so tools/_ab/libpt_sw_*.so; do
  for cfg in "1e9 5e7 32 8192" "1e8 5e6 20 8192" "1e8 5e6 8 8192"; do
    echo -n "$(basename $lib) $cfg: " >> $out
    PT_HIP_LIB=$lib timeout -k 10 120 python tools/run_clustered.py $cfg 2>/dev/null | tail -1 >> $out || exit 1
  done
done
cat $out
