#!/bin/bash
# Development: SQ counters of the sweep kernels of one bench run (one rocprofv3 --pmc pass per counter set, kernel trace only)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/${1:-r06_p}_sweep_pmc.txt
: > $OUT
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z0-9_]*(MFMA|LDS|WAIT|BUSY|WAVE|VMEM|VALU)[A-Z0-9_]*" | sort -u | tr '\n' ' ' >> $OUT; echo >> $OUT
CMD="python $REPO/bench.py --steps 8 --warmup 3 --cpu-seconds 0 --no-profile-pass --no-ttt --no-spmv-hbm --pmc off"
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_MFMA"; do
  rm -rf /tmp/pm
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o p -- $CMD > /dev/null 2>&1
  F=$(find /tmp/pm -name "p_counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_ss_block" in k or "k_multiaxpy" in k:
        acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:48s} {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
cat $OUT
