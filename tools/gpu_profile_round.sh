#!/bin/bash
# Collect the judged artefacts of a round on the GPU box (from the repo root):
#   bench lines (cfg2, cfg3), rocprofv3 kernel stats of the default bench command, PMC traffic passes.
# usage: tools/gpu_profile_round.sh r01
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
# the default command (cfg2 line + host path + both 512^3 legs + CPU baseline)
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -3 $out/bench_default.err; exit 1; }
for wl in cfg2 cfg3; do
  steps=8; [ $wl = cfg3 ] && steps=4   # same steps / lock-step batch as the bench legs above; fp32 solver storage (the
  # mode the 76 B / update figure is defined on) for both sizes
  # warm-up 0: every k_sor_step dispatch in the stats belongs to the timed region, so rocprof's average
  # duration is directly comparable with the HIP-event figure in the JSON line of the same run
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/_kt_$wl -- python3 bench.py --workload $wl --steps $steps --warmup 0 --solver-fp64 1 --no-cpu-baseline --no-extras > $out/rocprof_$wl.log 2>&1 || exit 1
  grep -o '"avg_launch_us": [0-9.]*\|"launches": [0-9]*' $out/rocprof_$wl.log | tr '\n' ' ' > $out/rocprof_${wl}_hipevents.txt
  cp $(find $out/_kt_$wl -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$wl.csv
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/_pf_$wl -- python3 bench.py --workload $wl --steps 1 --warmup 0 --batch 1 --solver-fp64 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/_pw_$wl -- python3 bench.py --workload $wl --steps 1 --warmup 0 --batch 1 --solver-fp64 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/_pf_$wl > $out/pmc_fetch_$wl.txt
  python3 tools/pmc_summary.py $out/_pw_$wl > $out/pmc_write_$wl.txt
  rm -rf $out/_kt_$wl $out/_pf_$wl $out/_pw_$wl
done
grep -h "sor_step\|axpy" $out/pmc_*.txt | cut -c1-150
head -4 $out/kernel_stats_cfg2.csv | cut -c1-200
tail -c 1500 $out/bench_default.json
