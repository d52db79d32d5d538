#!/bin/bash
# Round-4 profiles on the GPU box (everything lands under gpurun_out/r4/final; the summaries are copied to profiles/r4_* by hand):
#   * bench.py (the driver's command) -> bench.json
#   * rocprofv3 kernel tables + HBM-traffic counters of the fit + mean step at N = 1e6 and N = 1e7 (bench.py --main-only)
#   * kernel table AND per-kernel HBM traffic (separate --pmc FETCH_SIZE / WRITE_SIZE passes) of BASELINE configs[4] (3-D
#     Matern-3/2, N = 5e6: pair pass, own_fft_pass_kernel, cg3h_*) and of configs[3]'s hard case (2-D SE l = 0.05, 256^2
#     circulant grid, N = 1e6: cg_coop2d_herm_kernel)
#   * kernel table and the device timeline of one synchronised hyper-gradient step (T = 5, N = 1e6; efgp_gradient_step) and of one
#     fit of the headline model (part `grad`)
# usage: bash tools/profile_r4.sh [part ...]   parts: bench main c5 c4 grad (default: all)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4/final
mkdir -p $O
PARTS=${@:-bench main c5 c4 grad}
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
if has grad; then
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_grad -o kt -- python3 $R/tools/r4/grad_steps_plain.py 300 > /dev/null 2> $O/grad_trace.err || echo "grad trace failed"
  f=$(find $O/kt_grad -name "*kernel_trace.csv" | head -1); python3 $R/tools/r4/step_timeline.py $f > $O/gradient_step_timeline.txt || true
  f=$(find $O/kt_grad -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_gradient_step.csv
  rm -rf $O/kt_grad
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_fit -o kt -- python3 $R/bench.py --main-only --steps 300 > /dev/null 2> $O/fit_trace.err || echo "fit trace failed"
  f=$(find $O/kt_fit -name "*kernel_trace.csv" | head -1); python3 $R/tools/r4/step_timeline.py $f > $O/fit_step_timeline.txt || true
  rm -rf $O/kt_fit
  rocprofv3 --kernel-trace --output-format csv -d $O/kt_hard -o kt -- python3 $R/tools/r4/hard_fit_steps.py 60 > /dev/null 2> $O/hard_trace.err || echo "hard trace failed"
  f=$(find $O/kt_hard -name "*kernel_trace.csv" | head -1); python3 $R/tools/r4/step_timeline.py $f > $O/hard_fit_timeline.txt || true
  rm -rf $O/kt_hard
  echo "grad done"
fi
ls $O
