#!/bin/bash
# the profile artifacts of a round for one bench configuration (run on the GPU box): kernel trace of the bench command, then HBM
# counters in their own passes (rocprofv3 --pmc over a plain run of the same launches, FETCH_SIZE and WRITE_SIZE separately)
# usage: tools/profile_round.sh NAME [CONFIG]   -> gpurun_out/NAME/NAME_cfgC_{kernel_stats.csv,pmc_hbm.csv,bench.json,build_id.txt}
set -e
R=/root/repo
N=$1
C=${2:-1}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/$N; mkdir -p $OUT
T=${N}_cfg$C
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$C -- python3 $R/bench.py --config $C --steps 5 --warmup 2 --no-cpu --no-host > $OUT/$T.bench.log 2>&1
grep '^{' $OUT/$T.bench.log | tail -1 > $OUT/${T}_bench.json
cp $(ls $OUT/stats$C/*/*kernel_stats.csv | head -1) $OUT/${T}_kernel_stats.csv
rm -rf $OUT/stats$C
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$C -- python3 $R/tools/one_launch.py --config $C --reps 3 > $OUT/$T.fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write$C -- python3 $R/tools/one_launch.py --config $C --reps 3 > $OUT/$T.write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/fetch$C $OUT/write$C > $OUT/${T}_pmc_hbm.csv
grep -o "build [0-9a-f]*" $OUT/$T.fetch.log | tail -1 | cut -d' ' -f2 > $OUT/${T}_build_id.txt
grep -o "streams [0-9]*" $OUT/$T.fetch.log | tail -1 | cut -d' ' -f2 > $OUT/${T}_streams.txt
rm -rf $OUT/fetch$C $OUT/write$C
python3 -c "import json;d=json.load(open('$OUT/${T}_bench.json'));print('config $C:', round(d['value'],1),'MB/s', round(d['ms_per_step'],2),'ms', d['config']['stage_ms'], 'frac', round(d['roofline']['frac'],4))"
python3 $R/tools/summarize_stats.py $OUT/${T}_kernel_stats.csv | head -24
grep "recon_chain" $OUT/${T}_pmc_hbm.csv
