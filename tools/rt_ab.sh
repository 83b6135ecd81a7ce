#!/bin/bash
# Builds variants of libmi355rt.so (HERE, in the build container: hipcc cross-compiles) into gpu-computing-course_amd/ab/ :   rt_ab.sh build NAME "-DFLAGS" [NAME "-DFLAGS" ...]
# and times them ON THE GPU BOX with tools/rt_time.py (MI355RT_LIB):                                                          rt_ab.sh run NAME [NAME ...]
R=$(cd $(dirname $0)/.. && pwd); D=$R/gpu-computing-course_amd
if [ "$1" = build ]; then
  shift; mkdir -p $D/ab
  while [ $# -gt 1 ]; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -I$R/include $2 -shared -o $D/ab/librt_$1.so $D/csrc/mi355rt.hip && echo "built $1 ($2)"
    shift 2
  done
else
  shift
  for rep in 1 2; do for N in "$@"; do echo "== $N"; MI355RT_LIB=$D/ab/librt_$N.so python3 $R/tools/rt_time.py 2>/dev/null; done; done
fi
