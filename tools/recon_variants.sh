#!/bin/bash
# chain-kernel time of every variant under build/variants on BA_MW_D x 512 (run on the GPU box); two rounds to see the noise
cd /root/repo
for round in 1 2; do
for v in build/variants/lib*.so; do
  echo "== $v"
  LH264_SO=$PWD/$v timeout -k 10 120 python3 tools/recon_time.py
done
done
