cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_gpu_sstep.py -x -q < /dev/null 2>&1 | tail -12 | cut -c1-400
timeout 60 python bench.py --cpu-seconds 0 --no-ttt < /dev/null 2>/dev/null | tail -1 | cut -c1-130
NK_SS_FUSED=0 timeout 60 python bench.py --cpu-seconds 0 --no-ttt < /dev/null 2>/dev/null | tail -1 | cut -c1-130
