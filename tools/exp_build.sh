#!/bin/bash
# builds experimental variants of the library: tools/exp_build.sh name "-DFLAG ..." [name2 "flags2" ...]  -> build/exp/<name>/liblidarshooter_hip.so
# Variants are built with -DLS_EXPERIMENTAL: the tuning knobs of lidarshooter_amd/csrc/ls_tuning.h read the environment
# (the shipped library ignores it).  tools/exp_run.sh runs them on the GPU box through LS_LIB_PATH.
set -e
cd "$(dirname "$0")/../lidarshooter_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-result -DLS_EXPERIMENTAL"
SRCS="ls_project.hip ls_kernels.hip ls_sort.hip ls_handle.cpp ls_registry.cpp ls_commit.cpp ls_trace.cpp ls_host_pool.cpp ls_debug.cpp"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  out=../../build/exp/$name; mkdir -p $out
  objs=""
  for f in $SRCS; do
    [ -f $f ] || continue
    /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c -o $out/${f%.*}.o $f &
    objs="$objs $out/${f%.*}.o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblidarshooter_hip.so $objs -lpthread
  echo built $out
done
