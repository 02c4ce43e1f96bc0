#!/bin/bash
# time kernel variants built under build/variants (tuning experiments)
for v in build/variants/lib*.so; do
  echo "== $v"
  LH264_SO=$PWD/$v python3 tools/scale_probe.py 2>&1 | grep -E "streams= *(1|256|512|1024) "
done
