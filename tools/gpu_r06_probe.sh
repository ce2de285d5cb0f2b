#!/bin/bash
# Round 6: the discriminating experiments for the shared-device corruption (tools/shared_device_probe.py). One JSON line per
# experiment into gpurun_out/r06_probe/probe.jsonl. Usage: tools/gpu_r06_probe.sh [set]   (set: a | b | c …)
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06_probe
mkdir -p $OUT
SET=${1:-a}
run() {  # label, args…
  local P="python tools/shared_device_probe.py --grid ${GRID:-512} --trials ${TRIALS:-30} --steps ${STEPS:-10} --timeout ${TMO:-150}"
  local label=$1; shift
  echo "== $label: $*" >&2
  timeout 400 $P --label "$label" "$@" >> $OUT/probe_$SET.jsonl 2>> $OUT/probe_$SET.err || echo "{\"label\": \"$label\", \"failed_rc\": $?}" >> $OUT/probe_$SET.jsonl
}
case $SET in
a)
  run solo_alone            --mode solo --competitor none   --env NK_DEVICE_SHARED=0
  run ranks2_torch          --mode ranks --transport torch  --env NK_DEVICE_SHARED=0
  run ranks2_peer           --mode ranks --transport peer   --env NK_DEVICE_SHARED=0
  run threads2_callbacks    --mode threads                  --env NK_DEVICE_SHARED=0
  run solo_vs_stream        --mode solo --competitor stream --env NK_DEVICE_SHARED=0
  run solo_vs_gemm          --mode solo --competitor gemm   --env NK_DEVICE_SHARED=0
  run solo_vs_solver        --mode solo --competitor solver --env NK_DEVICE_SHARED=0
  run ranks2_torch_mm0      --mode ranks --transport torch  --env NK_DEVICE_SHARED=0 --env NK_SS_MM=0
  run ranks2_torch_dcgs2    --mode ranks --transport torch  --env NK_DEVICE_SHARED=0 --ortho dcgs2
  ;;
b)
  for i in 1 2 3; do run threads2_default_$i --mode threads --detail --env NK_DEVICE_SHARED=0; done
  run threads2_mm0          --mode threads --detail --env NK_DEVICE_SHARED=0 --env NK_SS_MM=0
  run threads2_kconst0      --mode threads --detail --env NK_DEVICE_SHARED=0 --env NK_SS_KCONST=0
  run threads2_implicit0    --mode threads --detail --env NK_DEVICE_SHARED=0 --env NK_SS_IMPLICIT=0
  run threads2_sstep8       --mode threads --detail --env NK_DEVICE_SHARED=0 --sstep 8
  run threads2_dcgs2        --mode threads --detail --env NK_DEVICE_SHARED=0 --ortho dcgs2
  run threads1              --mode threads --world 1 --detail --env NK_DEVICE_SHARED=0
  TMO=70 run ranks2_torch_default  --mode ranks --transport torch --detail --stall-dump 30 --env NK_DEVICE_SHARED=0
  TMO=70 run ranks2_torch_kconst0  --mode ranks --transport torch --detail --stall-dump 30 --env NK_DEVICE_SHARED=0 --env NK_SS_KCONST=0
  run solo_vs_solver_nopowers --mode solo --competitor solver --env NK_DEVICE_SHARED=0 --env NK_SPMV_POWERS=0
  ;;
esac
echo done >&2
