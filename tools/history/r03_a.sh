#!/bin/bash
# r03 first GPU call: the new bench-launcher / RCCL tests, and this box's steady-state baseline
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -15
mkdir -p tools/variants; cp webradio_amd/lib/libwebradio_amd.so tools/variants/base.so
C=1 bash tools/try_steady.sh
C=4 bash tools/try_steady.sh
