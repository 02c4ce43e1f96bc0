#!/bin/bash
# the profile artifacts of a round (run on the GPU box): kernel trace of the bench command, then HBM counters in their own passes
# usage: tools/profile_round.sh NAME     -> gpurun_out/NAME/{stats,fetch,write}, summaries NAME_kernel_stats.csv / NAME_pmc_hbm.csv
set -e
R=/root/repo
N=$1
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/$N; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/tools/one_launch.py ba 512 3 > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/tools/one_launch.py ba 512 3 > $OUT/write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/fetch $OUT/write > $OUT/${N}_pmc_hbm.csv
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${N}_kernel_stats.csv
tail -1 $OUT/bench.log | cut -c1-400
cat $OUT/${N}_pmc_hbm.csv
head -14 $OUT/${N}_kernel_stats.csv
