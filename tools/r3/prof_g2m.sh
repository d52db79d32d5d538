#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof_g2m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export EFGP_B64_GATHER=0
for N in 1000000 10000000; do
for sp in 1 2 4 8; do
  export EFGP_G2M_SPLIT=$sp
  rocprofv3 --kernel-trace --stats -d $O/s_${N}_$sp -o run -- python3 $R/bench.py --main-only --steps 20 --warmup 5 --global-n $N > $O/s_${N}_$sp.json 2> $O/s_${N}_$sp.err || echo "failed"
  db=$(find $O/s_${N}_$sp -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/ks_${N}_$sp.csv
  echo "N=$N split=$sp: $(grep grid_to_modes $O/ks_${N}_$sp.csv | cut -d, -f1-4 | cut -c1-120)  step: $(cat $O/s_${N}_$sp.json | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  rm -rf $O/s_${N}_$sp
done; done
