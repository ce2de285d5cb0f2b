#!/bin/bash
# Round 6, GPU call D: why the contract's clock (no event records in the timed loop) reads 25 % more time per step than the
# per-step events — the host's liveness query of the stream in nk_spin_wait and the event records, crossed.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/gpu_ab.sh r06_d_a 2 "NK_SPIN_QUERY_POLLS=16384" "" -- --step-events
mv gpurun_out/r06_d_a_ab.txt gpurun_out/r06_d_events_ab.txt
tools/gpu_ab.sh r06_d_b 2 "NK_SPIN_QUERY_POLLS=16384" "" "NK_SOLVE_HEAD=0" "NK_SPIN_QUERY_POLLS=16384 NK_SOLVE_HEAD=0"
mv gpurun_out/r06_d_b_ab.txt gpurun_out/r06_d_noevents_ab.txt
