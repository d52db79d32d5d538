#!/bin/bash
# usage: prof_cg3.sh mtot   -> kernel table of tools/r3/cg3_iter.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof_cg3_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/st -o run -- python3 $R/tools/r3/cg3_iter.py $1 > $O/prof.log 2>&1 || echo failed
db=$(find $O/st -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats.csv
rm -rf $O/st
grep -v amdgpu.ids $O/prof.log
cut -c1-150 $O/kernel_stats.csv | head -20
