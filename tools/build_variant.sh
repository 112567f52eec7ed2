#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [SOURCE]: neumann_amd/lib/variants/libneumann_gpu_NAME.so with SOURCE (default
# nmn_scan_mfma) rebuilt under the flags; NEUMANN_GPU_LIB=<that path> makes the Python layer load it
set -e
R=$(cd $(dirname $0)/.. && pwd)
SRC=${3:-nmn_scan_mfma}
mkdir -p $R/neumann_amd/lib/variants $R/neumann_amd/build/variants
O=$R/neumann_amd/build/variants/${SRC}_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include $2 -c $R/neumann_amd/csrc/$SRC.hip -o $O
OBJS=$(ls $R/neumann_amd/build/*.o | grep -v /$SRC.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/neumann_amd/lib/variants/libneumann_gpu_$1.so $OBJS $O -lpthread -ldl
echo $R/neumann_amd/lib/variants/libneumann_gpu_$1.so
