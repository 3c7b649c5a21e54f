#!/bin/bash
# The driver's round-end GPU gate, as the driver runs it, on a fresh lease: `python -m pytest tests/ -x -q -m gpu` under its
# 1200 s limit, then __graft_entry__.smoke().  Leaves ONE summary line in gpurun_out/gate_<label>.summary (gpurun merges
# files back, it does not append); `cat gpurun_out/gate_*.summary > profiles/r04_gate_leases.txt` collects them.
# Called as  gpurun --timeout 1500 -- 'bash tools/gate_lease.sh <label>'.
label=${1:-lease}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/gate_${label}.log 2>&1
rc=$?
t1=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/gate_${label}_smoke.log 2>&1
src=$?
t2=$(date +%s)
summary=$(grep -E "passed|failed|error" gpurun_out/gate_${label}.log | tail -1)
echo "$(date -u +%FT%TZ) $label head=$(cat .gate_head 2>/dev/null) pytest rc=$rc $((t1-t0))s [$summary] smoke rc=$src $((t2-t1))s" | tee gpurun_out/gate_${label}.summary
tail -22 gpurun_out/gate_${label}.log
