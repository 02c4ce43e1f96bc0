#!/bin/bash
# HBM bytes written / fetched by the coder kernels on one launch: tools/coder_write_pmc.sh OUT STREAM N [FRAMES]
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT
shift
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/tools/coder_cfg_once.py "$@" > $OUT/$c.log 2>&1 || { tail -5 $OUT/$c.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT/WRITE_SIZE $OUT/FETCH_SIZE | grep "coder_" > $OUT/hbm.csv
rm -rf $OUT/WRITE_SIZE $OUT/FETCH_SIZE
cat $OUT/hbm.csv
