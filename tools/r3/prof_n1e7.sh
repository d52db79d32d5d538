#!/bin/bash
# round 3: rocprofv3 kernel table of the fit + mean step at N = 1e7 with the dense-sigma rule on and off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export EFGP_NO_DENSE_SIGMA=1; else unset EFGP_NO_DENSE_SIGMA; fi
  rocprofv3 --kernel-trace --stats -d $O/stats_$mode -o run -- python3 $R/bench.py --main-only --steps 20 --warmup 5 --global-n 10000000 > $O/stats_$mode.json 2> $O/stats_$mode.err || echo "stats $mode failed"
  db=$(find $O/stats_$mode -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats_n1e7_dense_$mode.csv || true
  find $O/stats_$mode -name "*.db" -size +20M -delete
  head -14 $O/kernel_stats_n1e7_dense_$mode.csv
done
