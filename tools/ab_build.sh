#!/bin/bash
# Build a variant of libmi355cd.so with extra compiler flags into gpu-computing-course_amd/ab/libmi355cd_NAME.so (A/B runs: tools/ab_lib.py).
# usage: tools/ab_build.sh NAME "-DFOO=1 -DBAR"
set -e
cd "$(dirname "$0")/../gpu-computing-course_amd"; mkdir -p ab
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -w -I../include -pthread -shared $2 -o ab/libmi355cd_$1.so csrc/mi355cd.hip host/load_obj_fast.cpp
echo built ab/libmi355cd_$1.so "$2"
