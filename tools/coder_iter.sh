#!/bin/bash
# one tuning iteration on the GPU box: coder parity (both forms), then one traced launch of the ctx + coder kernels on the configs[2] / configs[3] batches
R=/root/repo
cd $R
timeout -k 10 900 python3 -m pytest tests/test_coder_gpu.py tests/test_sweep.py -x -q -m gpu > gpurun_out/$1_pytest.log 2>&1 || { tail -30 gpurun_out/$1_pytest.log; exit 1; }
tail -2 gpurun_out/$1_pytest.log
bash tools/coder_path_trace.sh $1_c3 "" syn1080p_IP_8f.264 256 8 | grep -v "^W2026" | head -14 &&
bash tools/coder_path_trace.sh $1_c2 "" syn720p_allI_4slices_8f.264 1024 4 | grep -v "^W2026" | head -14
python3 $R/tools/coder_parts.py syn1080p_IP_8f.264 256 8 | tail -2
