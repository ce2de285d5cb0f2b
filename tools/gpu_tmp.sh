cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_gpu_sstep.py -x -q < /dev/null 2>&1 | tail -3 | cut -c1-300
timeout 60 python bench.py --cpu-seconds 0 --no-ttt < /dev/null 2>/dev/null | tail -1 | cut -c1-130
