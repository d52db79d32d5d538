#!/bin/bash
# Round-2 profiles on the GPU box: rocprofv3 kernel tables and HBM-traffic counters of the fit + mean step at N = 1e6 and
# N = 1e7 (bench.py --main-only: every launch belongs to the step).  Counters in their own passes (no trace domains with --pmc).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 1000000 10000000; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$N -o run -- python3 $R/bench.py --main-only --steps 20 --warmup 5 --global-n $N > $O/stats_$N.json 2> $O/stats_$N.err || echo "stats $N failed"
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$N -o run -- python3 $R/bench.py --main-only --steps 10 --warmup 2 --global-n $N > /dev/null 2> $O/fetch_$N.err || echo "fetch $N failed"
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$N -o run -- python3 $R/bench.py --main-only --steps 10 --warmup 2 --global-n $N > /dev/null 2> $O/write_$N.err || echo "write $N failed"
  python3 $R/tools/pmc_hbm.py $O/fetch_$N $O/write_$N $O/pmc_$N.json > /dev/null || echo "pmc post $N failed"
  f=$(find $O/stats_$N -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_$N.csv
done
ls $O
