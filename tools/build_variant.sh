#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." [SOURCE]: a copy of libneumann_gpu.so with SOURCE (default nmn_scan_mfma) rebuilt under
# the flags; NEUMANN_GPU_LIB=<that path> makes the Python layer load it.  Objects are built under /tmp (nothing of a variant ends
# up in the snapshot gpurun pushes unless asked for): the library lands in $NMN_VARIANT_DIR, default /tmp/nmn_variants; give
# NMN_VARIANT_DIR=neumann_amd/lib/variants (git-ignored) when the variant has to travel to the GPU box, and delete it afterwards.
set -e
R=$(cd $(dirname $0)/.. && pwd)
SRC=${3:-nmn_scan_mfma}
OUT=${NMN_VARIANT_DIR:-/tmp/nmn_variants}
case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT /tmp/nmn_variants/obj
O=/tmp/nmn_variants/obj/${SRC}_$1.o
EXTRA=""
case "$SRC" in
  nmn_exact|nmn_ingest|nmn_kmeans) EXTRA="-ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt" ;;
  nmn_synth|nmn_ivf) EXTRA="-ffp-contract=off" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include $EXTRA $2 -c $R/neumann_amd/csrc/$SRC.hip -o $O
OBJS=$(ls $R/neumann_amd/build/*.o | grep -v /$SRC.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libneumann_gpu_$1.so $OBJS $O -lpthread -ldl
echo $OUT/libneumann_gpu_$1.so
