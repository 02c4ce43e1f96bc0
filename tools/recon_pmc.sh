#!/bin/bash
# issue-side counters of the chain kernel on BA_MW_D x 512 (run on the GPU box); a few passes (the SQ has few counter slots)
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/reconpmc; mkdir -p $OUT
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/tools/recon_time.py 1 > $OUT/p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT | grep "recon_chain\|counter"
