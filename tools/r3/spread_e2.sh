#!/bin/bash
# round-3 experiment: one-cell bands (W + 1 tile columns, 3 waves per SIMD) against 8-cell bands, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_spread_mfma.py -x -q -m gpu > $O/e2_tests.log 2>&1 || { tail -30 $O/e2_tests.log; exit 1; }
tail -3 $O/e2_tests.log
for N in 10000000 1000000 4000000; do
for bc in 8 1; do
for per in 2 3; do
  [ $bc = 8 ] && [ $per = 3 ] && continue
  EFGP_MFMA_BAND_CELLS=$bc EFGP_MFMA_BLOCKS_PER_CU=$per python - $N <<'PY'
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gp-quadrature_amd"))
import torch
from efgp_hip import NufftPlan, PointSet, kernel_timing, kernel_timing_read
N = int(sys.argv[1])
g = torch.Generator().manual_seed(0)
x = (torch.rand(N, 2, generator=g, dtype=torch.float64) * 2 - 1).cuda()
y = torch.randn(N, generator=g, dtype=torch.float64).cuda()
pts = PointSet(x, values=y)
plan = NufftPlan(x, 0.346, 1e-7, points=pts)
for _ in range(3):
    plan.type1_pair(y, (23, 23), (45, 45))
kernel_timing(True, only="spread")
for _ in range(10):
    plan.type1_pair(y, (23, 23), (45, 45))
ms, n = kernel_timing_read("spread")
print(f"N={N} band_cells={os.environ.get('EFGP_MFMA_BAND_CELLS')} blocks/CU={os.environ.get('EFGP_MFMA_BLOCKS_PER_CU')} spread launch {1e3*ms/n:.1f} us", flush=True)
PY
done; done; done 2>&1 | grep -v amdgpu.ids | tee $O/e2_times.log
