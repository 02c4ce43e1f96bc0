#!/bin/bash
# dynamic instruction counts and time of the chain kernel for every ablation variant under build/variants (run on the GPU box)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/ablate; mkdir -p $OUT
for v in $R/build/variants/lib*.so; do
  n=$(basename $v .so)
  echo "== $n"
  LH264_SO=$v timeout -k 10 120 python3 $R/tools/recon_time.py
  LH264_SO=$v LH264_FLAGS=2 timeout -k 10 120 python3 $R/tools/recon_time.py
  LH264_SO=$v timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/$n -- python3 $R/tools/recon_time.py 1 > $OUT/$n.log 2>&1
  python3 $R/tools/pmc_summary.py $OUT/$n | grep recon_chain
done
