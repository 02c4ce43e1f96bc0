#!/bin/bash
# kernel trace of bench.py on one of the other configs (run on the GPU box): tools/profile_config.sh CONFIG NAME
R=/root/repo
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/$2; mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --config $1 --steps 3 --warmup 1 --no-cpu > $OUT/bench.log 2>&1
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/$2_kernel_stats.csv
python3 $R/tools/summarize_stats.py $OUT/$2_kernel_stats.csv | head -14
