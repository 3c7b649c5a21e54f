#!/bin/bash
# r04: hunting the one-in-a-dozen abort of tests/test_gpu_blocks.py::test_u8_ingest_from_page_locked_host_memory seen in
# pass 1 of tools/gate_loop.sh: the module N times in fresh processes, everything the process wrote kept for a failing run
N=${1:-40}
mkdir -p gpurun_out
: > gpurun_out/r04_abort_hunt.txt
for i in $(seq 1 $N); do
  timeout 300 python -m pytest tests/test_gpu_blocks.py -x -q -p no:cacheprovider > /tmp/hunt.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E 'passed|failed' /tmp/hunt.log | tail -1)" >> gpurun_out/r04_abort_hunt.txt
  if [ $rc -ne 0 ]; then
    echo "---- run $i, the process's output up to the traceback:" >> gpurun_out/r04_abort_hunt.txt
    grep -v "^  File\|^Extension modules" /tmp/hunt.log | head -60 >> gpurun_out/r04_abort_hunt.txt
    dmesg 2>/dev/null | tail -5 >> gpurun_out/r04_abort_hunt.txt
  fi
done
grep -c "rc=0" gpurun_out/r04_abort_hunt.txt; grep -v "rc=0" gpurun_out/r04_abort_hunt.txt | head -80
