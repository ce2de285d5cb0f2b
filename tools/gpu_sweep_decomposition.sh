#!/bin/bash
# usage (on the GPU box): bash tools/gpu_sweep_decomposition.sh <tag>
tag=${1:-x}
out=gpurun_out/${tag}_sweep_decomposition.txt
: > $out
for lib in "" $(ls nonlinearsolve.jl_amd/lib/libmi355x_nk_exp*.so); do
  if [ -n "$lib" ]; then export NK_LIB_PATH=$lib; fi
  timeout 300 python tools/sweep_decomposition.py >> $out 2>&1
done
cat $out
