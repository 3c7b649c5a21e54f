#!/bin/bash
# one fresh box: tests/test_gpu_blocks.py as the FIRST thing the box does (cold caches, idle GPU), runtime errors to stderr
mkdir -p gpurun_out
stamp=$(date -u +%H%M%S)
LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1 timeout 600 python -m pytest tests/test_gpu_blocks.py -x -q -p no:cacheprovider > /tmp/f.log 2>&1
rc=$?
echo "fresh $stamp rc=$rc $(grep -E 'passed|failed' /tmp/f.log | tail -1)" > gpurun_out/abort_fresh_$stamp.txt
[ $rc -ne 0 ] && grep -v "^  File\|^Extension modules" /tmp/f.log | head -80 >> gpurun_out/abort_fresh_$stamp.txt
cat gpurun_out/abort_fresh_$stamp.txt | head -40
