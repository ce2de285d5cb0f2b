#!/bin/bash
# Round 6, GPU call E: what lets the runtime retire its launch records during the steps — contract clock per variant
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
tools/gpu_ab.sh r06_e_retire 2 "NK_STEP_RETIRE=0" "NK_STEP_RETIRE=1" "NK_STEP_RETIRE=2" "NK_STEP_RETIRE=3"
