#!/bin/bash
# Runs ON THE GPU BOX: per-kernel averages (tools/ktrace_opts.sh) of builds made by tools/ab_build.sh, on the meshes given.
# usage: ab_ktrace.sh "NAME1 NAME2 ..." "MESH1 MESH2 ..." [kernel name filter for the printout]
R=$GRAFT_REPO_ROOT
for M in $2; do for N in $1; do
  echo "#### build $N mesh $M"
  MI355CD_LIB=$R/gpu-computing-course_amd/ab/libmi355cd_$N.so STEPS=${STEPS:-60} bash $R/tools/ktrace_opts.sh ab_${N}_$M $M "" | grep -E "${3:-.}"
done; done
