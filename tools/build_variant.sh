#!/bin/bash
# build a tuning/diagnostic variant of liblh264.so under build/variants/ (they travel to the GPU box, unlike gpurun_out/)
# usage: tools/build_variant.sh NAME [extra hipcc flags...]      e.g.  tools/build_variant.sh stamp -DLH264_STAMP
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
C=losslessh264_amd/csrc
W="-DLH264_MIN_WAVES=5"           # the product's register cap, unless the variant sets its own
case "$*" in *LH264_MIN_WAVES*) W="";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value $W -Iinclude "$@" \
  $C/lh264_kernels.hip $C/lh264_ctx.hip $C/lh264_coder.hip $C/lh264_coder_sw.hip $C/lh264_capi.hip $C/lh264_compress.hip $C/host/h264_parser.cpp $C/host/isvc_shim.cpp $C/host/pip_symbols.cpp $C/host/pip_restore.cpp -pthread -o build/variants/lib$name.so
echo built build/variants/lib$name.so
