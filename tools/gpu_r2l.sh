set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_direct.py -x -q -s > $O/pytest_direct.log 2>&1; tail -6 $O/pytest_direct.log
python tools/c2_direct.py 256 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
cat > /tmp/fac.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import numpy as np, scipy.sparse as sp, torch
import nonlinearsolve_jl_amd as nls
from oracle import reference_restatement as R
pb = R.Bratu2D(256, 6.0)
J = sp.csr_matrix(pb.jac(0.3*np.random.default_rng(0).standard_normal(pb.n)))
A = nls.CSRMatrix.from_scipy(J); F = nls.BandedLU(A)
for _ in range(5): F.factor()
b = torch.randn(J.shape[0], dtype=torch.float64, device="cuda")
for _ in range(5): F.solve(b)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o kt -- python /tmp/fac.py > /dev/null 2>&1
find /tmp/kt2 -name "kt_kernel_stats.csv" -exec head -9 {} \; | cut -c1-150 > $GRAFT_REPO_ROOT/$O/bcr_kernel_stats.csv; cat $GRAFT_REPO_ROOT/$O/bcr_kernel_stats.csv
