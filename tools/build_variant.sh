#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." : neumann_amd/lib/variants/libneumann_gpu_NAME.so with nmn_scan_mfma.hip rebuilt under the flags
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/neumann_amd/lib/variants $R/neumann_amd/build/variants
O=$R/neumann_amd/build/variants/nmn_scan_mfma_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include $2 -c $R/neumann_amd/csrc/nmn_scan_mfma.hip -o $O
OBJS=$(ls $R/neumann_amd/build/*.o | grep -v nmn_scan_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/neumann_amd/lib/variants/libneumann_gpu_$1.so $OBJS $O -lpthread -ldl
echo $R/neumann_amd/lib/variants/libneumann_gpu_$1.so
