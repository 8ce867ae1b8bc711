#!/bin/bash
# Collect the judged artefacts of a round on the GPU box (from the repo root):
#   default bench line, rocprofv3 kernel stats of the cfg2 / cfg3 bench commands, PMC traffic passes per solver mode.
# usage: tools/gpu_profile_round.sh r04 [nobench|bench] ["cfg2:3 cfg2:1 ..."] [noasmooth]
#   (a gpurun call is limited to 20 minutes: split the pairs over two calls)
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
if [ "$2" != "nobench" ]; then
  # the default command (cfg2 line + host path + the 512^3 legs + a_smooth leg + CPU baseline)
  python3 bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -3 $out/bench_default.err; exit 1; }
fi
# workload : solver mode pairs; the first of each workload is the library's automatic choice
pairs=${3:-"cfg2:3 cfg2:1 cfg3:3 cfg3:1 cfg3:2"}
for pair in $pairs; do
  wl=${pair%%:*}; md=${pair##*:}
  steps=8; [ $wl = cfg3 ] && steps=4   # same steps / lock-step batch as the bench legs
  # warm-up 0: every k_sor_step dispatch in the stats belongs to the timed region, so rocprof's average
  # duration is directly comparable with the HIP-event figure in the JSON line of the same run
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/_kt -- python3 bench.py --workload $wl --steps $steps --warmup 0 --solver-fp64 $md --lanes 1 --no-cpu-baseline --no-extras > $out/rocprof_${wl}_m$md.log 2>&1 || exit 1
  grep -o '"avg_launch_us": [0-9.]*\|"launches": [0-9]*' $out/rocprof_${wl}_m$md.log | tr '\n' ' ' > $out/rocprof_${wl}_m${md}_hipevents.txt
  cp $(find $out/_kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats_${wl}_m$md.csv
  rm -rf $out/_kt
  timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/_pf -- python3 bench.py --workload $wl --steps 1 --warmup 0 --batch 1 --solver-fp64 $md --lanes 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/_pf > $out/pmc_fetch_${wl}_m$md.txt; rm -rf $out/_pf
  timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/_pw -- python3 bench.py --workload $wl --steps 1 --warmup 0 --batch 1 --solver-fp64 $md --lanes 1 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/_pw > $out/pmc_write_${wl}_m$md.txt; rm -rf $out/_pw
  echo "$pair done"
done
# the psi_smooth solver (a_smooth = 0.5) on the cfg2 geometry: kernel stats only
if [ "$4" != "noasmooth" ]; then
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/_kt -- python3 bench.py --workload cfg2 --steps 8 --warmup 0 --a-smooth 0.5 --lanes 1 --no-cpu-baseline --no-extras > $out/rocprof_cfg2_asmooth05.log 2>&1 || exit 1
cp $(find $out/_kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats_cfg2_asmooth05.csv; rm -rf $out/_kt
fi
grep -h "sor_step\|axpy" $out/pmc_*.txt | cut -c1-150
for f in $out/kernel_stats_*.csv; do echo $f; head -3 $f | cut -c1-200; done
if [ -f $out/bench_default.json ]; then tail -c 1500 $out/bench_default.json; fi
