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
c)  # after the stream-ordered memsets (nk_memset / nk_memcpy)
  export TMO=60
  run threads1              --mode threads --world 1 --detail --stall-dump 25
  for i in 1 2 3; do run threads2_default_$i --mode threads --detail --stall-dump 25 --env NK_DEVICE_SHARED=0; done
  run threads2_mm0          --mode threads --detail --stall-dump 25 --env NK_DEVICE_SHARED=0 --env NK_SS_MM=0
  run threads2_fused0       --mode threads --detail --stall-dump 25 --env NK_DEVICE_SHARED=0 --env NK_SS_FUSED=0
  for i in 1 2 3; do run ranks2_torch_default_$i --mode ranks --transport torch --detail --stall-dump 25 --env NK_DEVICE_SHARED=0; done
  run ranks2_torch_mm0      --mode ranks --transport torch --detail --stall-dump 25 --env NK_DEVICE_SHARED=0 --env NK_SS_MM=0
  run solo_fused0           --mode solo --detail --env NK_SS_FUSED=0
  GRID=362 run solo_fused0_362 --mode solo --detail --env NK_SS_FUSED=0
  ;;
d)  # micro-reproducer: one sweep launch beside a competitor (tools/sweep_race_probe.py)
  sw() { local label=$1; shift; echo "== $label: $*" >&2; timeout 300 python tools/sweep_race_probe.py --label "$label" "$@" >> $OUT/sweep_$SET.jsonl 2>> $OUT/sweep_$SET.err || echo "{\"label\": \"$label\", \"failed_rc\": $?}" >> $OUT/sweep_$SET.jsonl; }
  sw alone_k16
  sw thread_sweep_k16        --competitor thread-sweep
  sw proc_sweep_k16          --competitor proc-sweep
  sw proc_stream_k16         --competitor proc-stream
  sw proc_gemm_k16           --competitor proc-gemm
  sw proc_sweep_k16_mm0      --competitor proc-sweep --env NK_SS_MM=0
  sw proc_sweep_k1           --competitor proc-sweep --k 1
  sw proc_sweep_A_k16        --competitor proc-sweep --mode 0 --cmode 1
  sw proc_sweep_k16_2tiles   --competitor proc-sweep --n 262144
  sw proc_sweep_k16_8tiles   --competitor proc-sweep --n 1048576 --reps 60
  ;;
e)  # audit: which kernel's output differs first
  export TMO=60
  for i in 1 2 3; do run ranks2_torch_audit_$i --mode ranks --transport torch --audit --stall-dump 25 --env NK_DEVICE_SHARED=0; done
  run ranks2_torch_mm0_audit --mode ranks --transport torch --audit --stall-dump 25 --env NK_DEVICE_SHARED=0 --env NK_SS_MM=0
  run solo_fused0_audit     --mode solo --audit --env NK_SS_FUSED=0
  ;;
f)  # after the fix of the factorisation's write-after-read race: every form that failed before
  export TMO=60
  for i in 1 2 3; do run ranks2_torch_fixed_$i --mode ranks --transport torch --audit --stall-dump 25 --env NK_DEVICE_SHARED=0; done
  TRIALS=60 run ranks2_peer_fixed --mode ranks --transport peer --stall-dump 25 --env NK_DEVICE_SHARED=0
  TRIALS=60 run ranks2_peer_workaround_default --mode ranks --transport peer --stall-dump 25
  GRID=1024 TRIALS=12 run ranks2_peer_1024 --mode ranks --transport peer --stall-dump 40 --env NK_DEVICE_SHARED=0
  GRID=1024 TRIALS=12 run ranks2_torch_1024 --mode ranks --transport torch --stall-dump 40 --env NK_DEVICE_SHARED=0
  run solo_vs_solver_fixed  --mode solo --competitor solver --env NK_DEVICE_SHARED=0
  ;;
esac
echo done >&2
