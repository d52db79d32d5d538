#!/bin/bash
# round-3 experiment: MFMA spreader with all operand reads in flight before the MFMAs; stagger of the second block per CU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_spread_mfma.py -x -q -m gpu > $O/e1_tests.log 2>&1 || { tail -20 $O/e1_tests.log; exit 1; }
tail -3 $O/e1_tests.log
python tools/spread_diag.py 10000000 > $O/e1_diag.log 2>&1
cat $O/e1_diag.log
for st in 1 2 4; do
  echo "stagger $st"
  EFGP_MFMA_STAGGER=$st python tools/spread_only.py 10000000 5 >/dev/null 2>&1
  EFGP_MFMA_STAGGER=$st python - <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gp-quadrature_amd"))
import torch
from efgp_hip import NufftPlan, PointSet, kernel_timing, kernel_timing_read
N = 10_000_000
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).cuda()
y = torch.randn(N, generator=g, dtype=torch.float64).cuda()
pts = PointSet(x, values=y)
plan = NufftPlan(x, 0.346, 6e-8, points=pts)
for _ in range(3):
    plan.type1_pair(y, (23, 23), (45, 45))
kernel_timing(True, only="spread")
for _ in range(10):
    plan.type1_pair(y, (23, 23), (45, 45))
ms, n = kernel_timing_read("spread")
print(f"stagger={os.environ.get('EFGP_MFMA_STAGGER')} spread launch {1e3*ms/n:.1f} us", flush=True)
PY
done 2>&1 | tee $O/e1_stagger.log
python tools/gather_order.py 10000000 2>&1 | tee $O/e1_gather_order.log
