#!/bin/bash
# Runs ON THE GPU BOX: per-kernel average durations of the fused step at 1 M, 4 M and 8 M triangles (kernel trace of tools/trace_steps.py).
# usage: size_ktrace.sh TAG [MESH ...]
R=$GRAFT_REPO_ROOT; TAG=$1; shift
MESHES=${@:-"cloth1M cloth4M cfg4_8M soup8M"}
for M in $MESHES; do
  STEPS=60 bash $R/tools/ktrace_opts.sh ${TAG}_$M $M ""
done
