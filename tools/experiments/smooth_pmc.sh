#!/bin/bash
# HBM traffic of the psi_smooth solver forms (FETCH_SIZE / WRITE_SIZE in separate passes, batch 1, one lane)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-smoothpmc}; mkdir -p $out
for form in split paired; do for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/_p
  FR3D_LIB=flowreg3d_amd/lib/libflowreg3d_hip_exp.so FR3D_SMOOTH=$form timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/_p -- python3 bench.py --steps 1 --warmup 0 --batch 1 --lanes 1 --a-smooth 0.5 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py $out/_p k_smooth | cut -c1-170 | sed "s/^/$form /" | tee -a $out/summary.txt
  python3 tools/pmc_summary.py $out/_p k_axpy | cut -c1-170 | sed "s/^/$form /" >> $out/summary.txt
done; done
rm -rf $out/_p
