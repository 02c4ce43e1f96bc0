#!/bin/bash
# per-kernel times of the coder stage for the default build and every variant under build/variants (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/default -- python3 $R/tools/coder_kernels.py ${2:-512} 2>&1 | grep "so="
for v in $R/build/variants/lib*.so; do
  n=$(basename $v .so)
  LH264_SO=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -- python3 $R/tools/coder_kernels.py ${2:-512} 2>&1 | grep "so="
done
python3 $R/tools/summarize_stats.py "$OUT/*/*/*kernel_stats.csv" | grep "==\|coder_"
