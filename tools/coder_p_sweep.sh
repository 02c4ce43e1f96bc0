#!/bin/bash
# coder kernel times against the number of partitions per stream (run on the GPU box): tools/coder_p_sweep.sh OUT "L1 L2 .." STREAM N [FRAMES]
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT
LS=$2
shift; shift
for L in $LS; do
  LH264_CODER_LOG2P=$L timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l$L -- python3 $R/tools/coder_cfg_once.py "$@" > $OUT/l$L.log 2>&1 || { tail -5 $OUT/l$L.log; exit 1; }
  echo "== log2p $L: $(grep 'decision words' $OUT/l$L.log)"
  python3 $R/tools/summarize_stats.py $OUT/l$L/*/*kernel_stats.csv | grep "coder_resolve\|coder_emit\|coder_count\|coder_scan"
  rm -rf $OUT/l$L
done
