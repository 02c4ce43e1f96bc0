#!/bin/bash
# coder parity on the GPU box, then the bench lines of the four configs (kernel traces): tools/gpu_check_coder.sh PREFIX
R=/root/repo
P=$1
cd $R
timeout -k 10 900 python3 -m pytest tests/test_coder_gpu.py tests/test_sweep.py tests/test_ctx_gpu.py -x -q -m gpu > gpurun_out/${P}_pytest.log 2>&1 || { tail -30 gpurun_out/${P}_pytest.log; exit 1; }
tail -3 gpurun_out/${P}_pytest.log
cd /tmp; export TMPDIR=/tmp
for c in 1 2 3 4; do
  OUT=$R/gpurun_out/${P}_cfg$c; mkdir -p $OUT
  timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --config $c --steps 3 --warmup 1 --no-cpu --no-host > $OUT/bench.log 2>&1 || { echo "config $c failed"; tail -5 $OUT/bench.log; exit 1; }
  cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/${P}_cfg${c}_kernel_stats.csv
  grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
  rm -rf $OUT/stats
  echo "== config $c"; python3 $R/tools/summarize_stats.py $OUT/${P}_cfg${c}_kernel_stats.csv | head -20
  python3 -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d['config']['stage_ms'],d['roofline']['frac'])"
done
