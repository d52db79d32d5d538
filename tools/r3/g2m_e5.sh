#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof_g2m2
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_spread_mfma.py tests/test_gpu_nufft.py tests/test_gpu_sharded.py -x -q -m gpu -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
grep -v amdgpu.ids $O/tests.log | grep "passed\|iterations" | tail -8
cd /tmp && export TMPDIR=/tmp
for N in 1000000 10000000; do
for v in new old; do
  if [ $v = old ]; then export EFGP_G2M_V1=1; else unset EFGP_G2M_V1; fi
  rocprofv3 --kernel-trace --stats -d $O/s_${N}_$v -o run -- python3 $R/bench.py --main-only --steps 20 --warmup 5 --global-n $N > $O/s_${N}_$v.json 2> $O/s_${N}_$v.err || echo "failed"
  db=$(find $O/s_${N}_$v -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/ks_${N}_$v.csv
  echo "N=$N g2m=$v: $(grep 'grid_to_modes\|grid_rows\|rows_modes' $O/ks_${N}_$v.csv | cut -d, -f1-4 | cut -c1-110)  step: $(cat $O/s_${N}_$v.json | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  rm -rf $O/s_${N}_$v
done; done
