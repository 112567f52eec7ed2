#!/bin/bash
# bench's IVF leg (clustered rows): kernel times per probe, current library vs the bitonic final_kernel
OUT=$PWD/gpurun_out/r04z6; mkdir -p $OUT; R=$PWD
V=$R/neumann_amd/lib/variants
cd /tmp; export TMPDIR=/tmp
for tag in current bitonic; do
  if [ $tag = bitonic ]; then export NEUMANN_GPU_LIB=$V/libneumann_gpu_sort_bitonic.so; else unset NEUMANN_GPU_LIB; fi
  rm -rf $OUT/trace_$tag
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$tag -o t -- python $R/tools/ivf_clustered_child.py > $OUT/child_$tag.txt 2>&1
  DB=$(find $OUT/trace_$tag -name "*.db" | head -1)
  python $R/tools/prof_summary.py $DB "ivf clustered, $tag" > $OUT/kernels_$tag.txt 2>&1
  rm -rf $OUT/trace_$tag
  echo "== $tag"; grep -v amdgpu $OUT/child_$tag.txt | tail -1; head -24 $OUT/kernels_$tag.txt | cut -c1-150
done
