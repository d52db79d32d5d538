#!/bin/bash
# usage: prof_nd.sh d N [ls]   -> gpurun_out/r3/prof_nd_d<d>_n<N>/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof_nd_d$1_n$2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/r3/step_nd.py "$@" 2>&1 | grep -v amdgpu.ids | tee $O/wall.txt
rocprofv3 --kernel-trace --stats -d $O/st -o run -- python3 $R/tools/r3/step_nd.py "$@" > $O/prof.log 2>&1 || echo failed
db=$(find $O/st -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats.csv
rm -rf $O/st
cut -c1-160 $O/kernel_stats.csv | head -30
