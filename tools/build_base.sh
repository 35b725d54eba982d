#!/bin/bash
# Builds the library of a git revision (default HEAD) into gpurun_ab/libevodiff_hip_base.so for tools/ab_step.py
set -e
rev=${1:-HEAD}
root=$(git rev-parse --show-toplevel)
tmp=/tmp/evodiff_base_build
rm -rf $tmp && mkdir -p $tmp/evodiff_amd/csrc $tmp/include
for f in $(git -C $root ls-tree -r --name-only $rev evodiff_amd/csrc include); do git -C $root show $rev:$f > $tmp/$f; done
make -C $tmp/evodiff_amd/csrc > $tmp/build.log 2>&1 || { tail -20 $tmp/build.log; exit 1; }
mkdir -p $root/gpurun_ab && cp $tmp/evodiff_amd/csrc/libevodiff_hip.so $root/gpurun_ab/libevodiff_hip_base.so
echo "built $rev -> gpurun_ab/libevodiff_hip_base.so"
