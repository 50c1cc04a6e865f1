#!/bin/bash
# A/B of two builds of the query kernels on ONE box, alternating: shell (50M / 2.5M / K = 20), clustered 100M / 5M / k = 8, C5's shape.
# usage: tools/ab_wave.sh libA.so libB.so [c5]   (paths relative to the repo root; "" = the shipped library)
A=$1; B=$2
for rep in 1 2; do
  for L in "$A" "$B"; do
    export PT_HIP_LIB=$L; [ -z "$L" ] && unset PT_HIP_LIB
    echo "== lib ${L:-shipped}"
    timeout -k 10 200 python tools/probe_r4.py shell 50e6 default 2>&1 | grep -v amdgpu.ids
    timeout -k 10 200 python tools/probe_r4.py clus 100e6 default 2>&1 | grep -v amdgpu.ids
    if [ "$3" = "c5" ]; then timeout -k 10 300 python tools/probe_r4.py c5 1e9 default 2>&1 | grep -v amdgpu.ids; fi
  done
done
