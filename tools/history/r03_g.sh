#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_gpu_host.py tests/test_gpu_f4.py tests/test_filetuner.py -x -q -m gpu 2>&1 | tail -8
for src in f32 u8; do for late in 0 1 2; do
  WEBRADIO_AUDIO_LATE=$late WR_HOST_BENCH_PROFILE=1 timeout 200 tests/cxx/host_bench 256 30 4000000 $src 2>&1 | grep -E "^\{|^process\(\)"
done; done
