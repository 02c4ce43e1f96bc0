#!/bin/bash
# one HIP stream against two (reconstruct beside context index + coder) for every variant under build/variants (run on the GPU box)
cd /root/repo
for v in build/variants/lib*.so; do
  echo "== $v"
  LH264_SO=$PWD/$v timeout -k 10 200 python3 tools/overlap_probe.py 512 2>&1 | grep "stream\|recon\|split"
done
