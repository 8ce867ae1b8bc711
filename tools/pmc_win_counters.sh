#!/bin/bash
# Hardware counters of the window sweep kernel (k_sor_win) in separate rocprofv3 --pmc passes (kernel-trace only).
# usage: tools/pmc_win_counters.sh TAG [MODE]
tag=${1:-r04}; mode=${2:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
run() {  # name, counters...
  name=$1; shift
  FR3D_SWEEP=window timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/_pmc_$name -- python3 bench.py --steps 4 --warmup 0 --batch 4 --condition 0 --no-cpu-baseline --no-extras --solver-fp64 $mode --lanes 1 > $out/pmc_$name.log 2>&1 || { tail -5 $out/pmc_$name.log; return 1; }
  python3 tools/pmc_summary.py $out/_pmc_$name sor_win > $out/pmc_win_${name}_m$mode.txt
  rm -rf $out/_pmc_$name
  cat $out/pmc_win_${name}_m$mode.txt | cut -c55-200
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM || exit 1
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
run sq4 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU || exit 1
