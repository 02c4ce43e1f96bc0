#!/bin/bash
# lane utilisation of the reconstruct kernel (run on the GPU box): SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), per config
# tools/recon_lanes.sh OUT
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT
for c in 1 2 3; do
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p$c -- python3 $R/tools/one_launch.py --config $c --reps 1 > $OUT/p$c.log 2>&1 || { tail -3 $OUT/p$c.log; exit 1; }
  echo "== config $c: $(tail -1 $OUT/p$c.log)"
  python3 $R/tools/pmc_summary.py $OUT/p$c | grep "recon_chain"
  rm -rf $OUT/p$c
done
