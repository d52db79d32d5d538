#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/prof_c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/r3/c5_step.py 5000000 3 1e-3 2>&1 | grep -v amdgpu.ids | tee $O/c5_wall.txt
rocprofv3 --kernel-trace --stats -d $O/st -o run -- python3 $R/tools/r3/c5_step.py 5000000 3 1e-3 > $O/c5_prof.log 2>&1 || echo failed
db=$(find $O/st -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/c5_kernel_stats.csv
rm -rf $O/st
cut -c1-150 $O/c5_kernel_stats.csv | head -24
