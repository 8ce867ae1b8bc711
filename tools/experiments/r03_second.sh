#!/bin/bash
# round 3, fused sweep kernel with phase ordering: tile-shape sweeps + PMC traffic of two shapes
set -e -o pipefail
O=gpurun_out/r03b; mkdir -p $O
export FR3D_LIB=$PWD/flowreg3d_amd/lib/libflowreg3d_hip_exp.so
python tools/experiments/sor_env_probe.py 256 8 FR3D_SOR_SHAPE 2x1,2x2,2x4,4x2,1x4,4x1,1x8 3 > $O/shape_256_m1.jsonl
echo 256 done
FR3D_PROBE_MODE=3 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x4,4x2,2x2 2 > $O/shape_512_m3.jsonl
FR3D_PROBE_MODE=2 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x4,4x2 2 > $O/shape_512_m2.jsonl
FR3D_PROBE_MODE=1 python tools/experiments/sor_env_probe.py 512 4 FR3D_SOR_SHAPE 2x1,2x4,4x2 2 > $O/shape_512_m1.jsonl
echo 512 done
for shp in 2x1 2x4; do
  export FR3D_SOR_SHAPE=$shp
  bash tools/pmc_quick.sh FETCH_SIZE --workload cfg2 --solver-fp64 1 --no-extras > $O/pmc_fetch_cfg2_$shp.txt
  bash tools/pmc_quick.sh WRITE_SIZE --workload cfg2 --solver-fp64 1 --no-extras > $O/pmc_write_cfg2_$shp.txt
done
cat $O/pmc_*.txt
