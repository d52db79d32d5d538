#!/bin/bash
# Round-3 profiles on the GPU box (everything lands under gpurun_out/r3/final; the summaries are copied to profiles/r3_* by hand):
#   * bench.py (the driver's command) -> bench.json
#   * rocprofv3 kernel tables + HBM-traffic counters of the fit + mean step at N = 1e6 and N = 1e7 (bench.py --main-only: every
#     launch belongs to the step); counters in their own passes (no trace domains with --pmc)
#   * SQ counters of the new spread / gather kernels at N = 1e7
#   * kernel table of BASELINE configs[4] (3-D Matern-3/2, N = 5e6, 128^3 circulant grid) and of configs[3]'s hard case (2-D SE
#     l = 0.05, 256^2 circulant grid, N = 1e6)
#   * tools/fuzz_nufft.py, seeds 0-2
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3/final
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err || echo "bench failed"
cd /tmp && export TMPDIR=/tmp
for N in 1000000 10000000; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$N -o run -- python3 $R/bench.py --main-only --steps 20 --warmup 5 --global-n $N > $O/stats_$N.json 2> $O/stats_$N.err || echo "stats $N failed"
  db=$(find $O/stats_$N -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats_$N.csv || true
  rm -rf $O/stats_$N
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$N -o run -- python3 $R/bench.py --main-only --steps 10 --warmup 2 --global-n $N > /dev/null 2> $O/fetch_$N.err || echo "fetch $N failed"
  rocprofv3 --pmc WRITE_SIZE -d $O/write_$N -o run -- python3 $R/bench.py --main-only --steps 10 --warmup 2 --global-n $N > /dev/null 2> $O/write_$N.err || echo "write $N failed"
  python3 $R/tools/pmc_hbm.py $O/fetch_$N $O/write_$N $O/pmc_$N.json > /dev/null || echo "pmc post $N failed"
  rm -rf $O/fetch_$N $O/write_$N
done
echo "profiles n1e6 / n1e7 done"
SKIP_GATHER= bash $R/tools/r3/pmc_sq.sh > $O/sq_counters.txt 2>&1 || echo "sq failed"
cp $R/gpurun_out/r3/pmc/gather_summary.txt $O/sq_gather_n1e7.txt 2>/dev/null
cp $R/gpurun_out/r3/pmc/spread_summary.txt $O/sq_spread_n1e7.txt 2>/dev/null
echo "sq done"
rocprofv3 --kernel-trace --stats -d $O/c5 -o run -- python3 $R/tools/r3/c5_step.py 5000000 3 1e-3 > $O/c5_wall.txt 2>&1 || echo "c5 failed"
db=$(find $O/c5 -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats_c5_3d_n5e6.csv || true
rm -rf $O/c5
cd $R
rocprofv3 --kernel-trace --stats -d $O/c4 -o run -- python3 $R/tools/config4_phase.py 1000000 > $O/c4_wall.txt 2>&1 || echo "c4 failed"
db=$(find $O/c4 -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats_c4_256sq_n1e6.csv || true
rm -rf $O/c4
echo "c5 / c4 done"
for s in 0 1 2; do python tools/fuzz_nufft.py 100 $s 2>&1 | grep -v amdgpu.ids | tail -3; done > $O/fuzz.txt
tail -9 $O/fuzz.txt
ls $O
