#!/bin/bash
# kernel trace of one launch of the ctx + coder kernels with a forced coder form: tools/coder_path_trace.sh OUT sw|wave STREAM N [FRAMES]
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT
export LH264_CODER_PATH=$2
shift; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/coder_cfg_once.py "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -1 $OUT/run.log
python3 $R/tools/summarize_stats.py $OUT/kernel_stats.csv
