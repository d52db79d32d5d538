#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_spread_mfma.py tests/test_gpu_nufft.py -x -q -m gpu > $O/e4_tests.log 2>&1 || { tail -30 $O/e4_tests.log; exit 1; }
tail -2 $O/e4_tests.log
python -m pytest tests/test_gpu_fullsize_properties.py -x -q -m gpu > $O/e4_full.log 2>&1 || { tail -30 $O/e4_full.log; exit 1; }
tail -2 $O/e4_full.log
for N in 10000000 4000000 1000000; do
  python tools/r3/northstar_times.py $N 2>&1 | grep -v amdgpu.ids
  EFGP_B64_GATHER=0 python tools/r3/northstar_times.py $N 2>&1 | grep -v amdgpu.ids | sed 's/^/   [two-copy gather] /'
  EFGP_G2M_SPLIT=1 python tools/r3/northstar_times.py $N 2>&1 | grep -v amdgpu.ids | sed 's/^/   [g2m split 1] /'
  EFGP_NO_DENSE_SIGMA=1 python tools/r3/northstar_times.py $N 2>&1 | grep -v amdgpu.ids
done | tee $O/e4_times.log
