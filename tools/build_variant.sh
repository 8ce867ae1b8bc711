#!/bin/bash
# build_variant.sh NAME "-DFOO=1 ..." : rebuilds k_sor_band.hip with extra flags and links
# flowreg3d_amd/lib/variants/libfr3d_NAME.so (select with FR3D_LIB=...).  Measurement helper.
set -e
cd "$(dirname "$0")/../flowreg3d_amd/csrc"
make -s -j8
mkdir -p ../lib/variants ../../build/var_$1
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $2 -c k_sor_band.hip -o ../../build/var_$1/k_sor_band.o
OBJS=$(ls ../../build/obj/*.o | grep -v k_sor_band.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libfr3d_$1.so $OBJS ../../build/var_$1/k_sor_band.o
echo built flowreg3d_amd/lib/variants/libfr3d_$1.so
