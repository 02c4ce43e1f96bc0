#!/bin/bash
# kernel trace of one launch of the ctx + coder kernels with a forced coder form: tools/coder_path_trace.sh OUT sw|wave STREAM N [FRAMES]
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT
export LH264_CODER_PATH=$2
shift; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/coder_cfg_once.py "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cp $(ls $OUT/stats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/stats/*/*kernel_trace.csv')
if f:
    for r in csv.DictReader(open(f[0])):
        n=r['Kernel_Name']
        if 'range_walk1' in n: print('   walk1 launch: grid %s  %.3f ms' % (r.get('Grid_Size_X', r.get('Grid_Size','?')), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
rm -rf $OUT/stats
tail -1 $OUT/run.log
python3 $R/tools/summarize_stats.py $OUT/kernel_stats.csv
