#!/bin/bash
# Builds tools/bin/exp_fast_base from the detection kernel of a git revision (default HEAD), beside
# tools/bin/exp_fast from the working tree, so that one GPU run times both on the same device:
#   bash tools/ab_fast.sh [rev] && gpurun -- 'tools/bin/exp_fast_base; tools/bin/exp_fast'
set -e
cd "$(dirname "$0")"
REV=${1:-HEAD}
mkdir -p bin /tmp/ab_fast
git show "$REV:orbslam2_amd/csrc/fast.hip" > /tmp/ab_fast/fast.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -std=c++17 -ffp-contract=off -I/tmp/ab_fast -I../include -I../orbslam2_amd/csrc \
    exp_fast.hip -L../orbslam2_amd -lorbx -Wl,-rpath,'$ORIGIN/../../orbslam2_amd' -o bin/exp_fast_base
make bin/exp_fast
