#!/bin/bash
# resolve kernel time against the window the waves of a stream are kept in: tools/coder_window_sweep.sh OUT "W1 W2 .." STREAM N [FRAMES]
R=/root/repo
O=$1; WS=$2; shift; shift
for w in $WS; do
  echo "== window $w"
  LH264_CODER_WINDOW=$w bash $R/tools/coder_path_trace.sh ${O}_w$w "" "$@" | grep "coder_resolve" || exit 1
done
