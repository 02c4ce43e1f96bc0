#!/bin/bash
# VALU / SALU / LDS instruction counts of the chain kernel with the in-loop filter switched off by the job flag (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/reconpmc2; mkdir -p $OUT
for fl in 0 2 3; do
  LH264_FLAGS=$fl timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/f$fl -- python3 $R/tools/recon_time.py 1 > $OUT/f$fl.log 2>&1
  echo "flags=$fl"; python3 $R/tools/pmc_summary.py $OUT/f$fl | grep recon_chain
done
