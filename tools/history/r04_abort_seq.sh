#!/bin/bash
# r04: the module sequence that aborted once in pass 1 of tools/gate_loop.sh -- test_gpu_reference_pin, test_gpu_stage,
# test_gpu_blocks in ONE process -- N times in fresh processes
N=${1:-40}; tag=${2:-seq}
mkdir -p gpurun_out
out=gpurun_out/r04_abort_${tag}.txt
: > $out
for i in $(seq 1 $N); do
  LIBC_FATAL_STDERR_=1 AMD_LOG_LEVEL=1 WR_TEST_KEEP_ORDER=${KEEP-} timeout 300 python -m pytest tests/test_gpu_reference_pin.py tests/test_gpu_stage.py tests/test_gpu_blocks.py -x -q -p no:cacheprovider -p no:randomly > /tmp/s.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E 'passed|failed' /tmp/s.log | tail -1)" >> $out
  if [ $rc -ne 0 ]; then grep -v "^  File\|^Extension modules\|^/root/reference" /tmp/s.log | head -40 >> $out; fi
done
echo "ok $(grep -c 'rc=0' $out) of $N"; grep -v "rc=0" $out | head -50
