cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_gpu_sstep.py -x -q < /dev/null 2>&1 | tail -8 | cut -c1-400
timeout 60 python tools/ss_test.py < /dev/null 2>&1 | grep "n=2\^20" | cut -c1-200
timeout 60 python bench.py --cpu-seconds 0 --no-ttt < /dev/null 2>/dev/null | tail -1 | cut -c1-130
