set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_b; mkdir -p $O
for s in 1 2 4 8 15; do timeout 120 python tools/powers_bench.py 1024 $s 60 < /dev/null 2>&1 | tail -1 | tee -a $O/powers_scan.jsonl; done
