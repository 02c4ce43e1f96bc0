#!/bin/bash
# the range kernels' times with a variant library: tools/variant_trace.sh OUT VARIANT STREAM N [FRAMES]
R=/root/repo
O=$1; V=$2; shift; shift
echo "== $V $*"
LH264_SO=$R/build/variants/lib$V.so bash $R/tools/coder_path_trace.sh $O "" "$@" | grep "coder_range\|coder_accum\|coder_bytes\|coder_resolve" || exit 1
