#!/bin/bash
# Hardware counters of the SOR sweep kernel (k_sor_step) in separate rocprofv3 --pmc passes
# (kernel-trace only; never combined with other trace domains).  usage: tools/pmc_sor_counters.sh TAG [WORKLOAD]
tag=${1:-r02}; wl=${2:-cfg2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
steps=8; batch=8; [ $wl = cfg3 ] && { steps=1; batch=1; }
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/_pmc_$name -- python3 bench.py --workload $wl --steps $steps --warmup 0 --batch $batch --condition 0 --no-cpu-baseline > $out/pmc_$name.log 2>&1 || { tail -5 $out/pmc_$name.log; return 1; }
  python3 tools/pmc_summary.py $out/_pmc_$name sor_step > $out/pmc_${name}_$wl.txt
  rm -rf $out/_pmc_$name
  cat $out/pmc_${name}_$wl.txt | cut -c55-200
}
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE || exit 1
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
run tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum || exit 1
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
