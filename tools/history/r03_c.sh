#!/bin/bash
# r03: parity of the role-split DDC kernel, then steady-state timing of its variants
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_tuner.py tests/test_gpu_ring.py tests/test_gpu_fuzz.py tests/test_gpu_f4.py -x -q -m gpu 2>&1 | tail -8
C=1 bash tools/try_steady.sh
C=4 bash tools/try_steady.sh
