#!/bin/bash
# one stream against two with fewer waves per reconstruct workgroup (so that the resolve kernel fits beside it); run on the GPU box
cd /root/repo
for spec in "libw4.so 8" "libw4.so 6" "libw5.so 8" "libw5.so 6" "libw5.so 7" "libw5.so 5"; do
  set -- $spec
  echo "== $1 waves=$2"
  LH264_SO=$PWD/build/variants/$1 LH264_WAVES=$2 timeout -k 10 200 python3 tools/overlap_probe.py 512 2>&1 | grep "one stream\|recon last\|split"
done
