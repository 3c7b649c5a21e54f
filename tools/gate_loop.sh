#!/bin/bash
# the GPU suite N times in a row on one lease (flaky tests show up here, not at the driver's gate); one line per pass
N=${1:-4}
mkdir -p gpurun_out
: > gpurun_out/${2:-r05}_gate_loop.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 1200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > /tmp/loop_$i.log 2>&1
  rc=$?
  echo "pass $i rc=$rc $(( $(date +%s) - t0 ))s [$(grep -E 'passed|failed|error' /tmp/loop_$i.log | tail -1)]" | tee -a gpurun_out/${2:-r05}_gate_loop.txt
  if [ $rc -ne 0 ]; then grep -v "^/root/reference" /tmp/loop_$i.log | tail -60 | tee -a gpurun_out/${2:-r05}_gate_loop.txt; fi
done
