# a 20-second sanity pass on a GPU box: smoke(), the s-step and polyalgorithm suites, the default bench line
cd $GRAFT_REPO_ROOT
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -2 | cut -c1-300
timeout 100 python -m pytest tests/test_gpu_sstep.py tests/test_gpu_polyalg.py -x -q < /dev/null 2>&1 | tail -3 | cut -c1-300
timeout 60 python bench.py --cpu-seconds 0 --no-ttt < /dev/null 2>/dev/null | tail -1 | cut -c1-130
