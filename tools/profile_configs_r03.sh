#!/bin/bash
# kernel traces of bench.py on configs 2, 3, 4 (run on the GPU box): tools/profile_configs_r03.sh PREFIX
# -> gpurun_out/PREFIX_cfgN/{PREFIX_cfgN_kernel_stats.csv, bench.json}
R=/root/repo
P=$1
cd /tmp; export TMPDIR=/tmp
for c in 2 3 4; do
  OUT=$R/gpurun_out/${P}_cfg$c; mkdir -p $OUT
  timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu > $OUT/bench.log 2>&1 || { echo "config $c failed"; tail -5 $OUT/bench.log; exit 1; }
  cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${P}_cfg${c}_kernel_stats.csv
  grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
  rm -rf $OUT/stats
  echo "== config $c"; python3 $R/tools/summarize_stats.py $OUT/${P}_cfg${c}_kernel_stats.csv | head -20
  python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['config']['stage_ms'],d['roofline']['frac'])"
done
