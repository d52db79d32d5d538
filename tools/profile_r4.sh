#!/bin/bash
# Round-4 profiles on the GPU box (everything lands under gpurun_out/r4/final; the summaries are copied to profiles/r4_* by hand):
#   * bench.py (the driver's command) -> bench.json
#   * rocprofv3 kernel tables + HBM-traffic counters of the fit + mean step at N = 1e6 and N = 1e7 (bench.py --main-only)
#   * kernel table AND per-kernel HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) of BASELINE configs[4] (3-D
#     Matern-3/2, N = 5e6: pair pass, own_fft_pass_kernel, cg3h_*) and of configs[3]'s hard case (2-D SE l = 0.05, 256^2
#     circulant grid, N = 1e6: cg_coop2d_herm_kernel)
# usage: bash tools/profile_r4.sh [part ...]   parts: bench main c5 c4 (default: all)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4/final
mkdir -p $O
PARTS=${@:-bench main c5 c4}
has() { [[ " $PARTS " == *" $1 "* ]]; }
cd $R
if has bench; then python bench.py > $O/bench.json 2> $O/bench.err || echo "bench failed"; echo "bench done"; fi
cd /tmp && export TMPDIR=/tmp
trace() {   # trace <tag> <program args...>: kernel table + per-kernel traffic of one command
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/st_$tag -o run -- python3 "$@" > $O/${tag}_wall.txt 2> $O/${tag}_trace.err || echo "trace $tag failed"
  local db=$(find $O/st_$tag -name "*_results.db" | head -1); python3 $R/tools/kernel_stats_from_db.py $db $O/kernel_stats_$tag.csv || true
  rm -rf $O/st_$tag
  rocprofv3 --pmc FETCH_SIZE -d $O/fe_$tag -o run -- python3 "$@" > /dev/null 2> $O/${tag}_fetch.err || echo "fetch $tag failed"
  rocprofv3 --pmc WRITE_SIZE -d $O/wr_$tag -o run -- python3 "$@" > /dev/null 2> $O/${tag}_write.err || echo "write $tag failed"
  python3 $R/tools/r4/pmc_all.py $O/fe_$tag $O/wr_$tag $O/kernel_stats_$tag.csv $O/pmc_$tag.json > $O/pmc_$tag.txt || echo "pmc post $tag failed"
  rm -rf $O/fe_$tag $O/wr_$tag
  echo "$tag done"
}
if has main; then
  for N in 1000000 10000000; do
    trace n$N $R/bench.py --main-only --steps 20 --warmup 5 --global-n $N
    # the file bench.py reads `roofline.traffic` from keeps its round-3 layout
    python3 - $O/pmc_n$N.json $O/pmc_hbm_n$N.json <<'PY'
import json, sys
src = json.load(open(sys.argv[1]))
out = {"note": src["note"]}
for k, e in src["kernels"].items():
    if k.startswith("spread_mfma_kernel"):
        out["spread_kernel"] = e
        out["hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
    if k.startswith("interp_real2_pair_kernel"):
        out["interp_kernel"] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
  done
fi
if has c5; then trace c5_3d_n5e6 $R/tools/r3/c5_step.py 5000000 3 1e-3; fi
if has c4; then cd $R; trace c4_256sq_n1e6 $R/tools/config4_phase.py 1000000; fi
ls $O
