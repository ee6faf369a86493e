#!/bin/bash
# builds experimental variants of the library: tools/exp_build.sh name "-DFLAG ..." [name2 "flags2" ...]  -> build/exp/<name>/liblidarshooter_hip.so
set -e
cd "$(dirname "$0")/../lidarshooter_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-result"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  out=../../build/exp/$name; mkdir -p $out
  for f in ls_project.hip ls_kernels.hip ls_tracer.cpp; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c -o $out/${f%.*}.o $f &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblidarshooter_hip.so $out/ls_kernels.o $out/ls_project.o $out/ls_tracer.o
  echo built $out
done
