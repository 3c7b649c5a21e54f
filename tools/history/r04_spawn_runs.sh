#!/bin/bash
# r04: the self-spawning N > 1 launch path (spawn_ranks: own children, file store, watchdog), 20 runs of each workload with two
# gloo ranks on the one GPU, wall time and exit code of every run; then the launcher's watchdog and a failing rank.
mkdir -p gpurun_out
{
echo "$(date -u +%FT%TZ) HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY-unset}"
python -c "import torch" ; worst=0
for i in $(seq 1 20); do
  for extra in "" "--workload c5"; do
    t0=$(date +%s%N)
    timeout -k 5 120 python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --spawn-timeout 100 --rdzv-timeout 60 $extra > /tmp/sp.out 2> /tmp/sp.err
    rc=$?
    ms=$(( ($(date +%s%N) - t0) / 1000000 ))
    [ $ms -gt $worst ] && worst=$ms
    echo "$i ${extra:-c2} rc=$rc ${ms} ms n_gpus=$(grep -o '"n_gpus": [0-9]*' /tmp/sp.out | head -1)"
    [ $rc -ne 0 ] && tail -20 /tmp/sp.err
  done
done
echo "worst of 40 runs: $worst ms"
t0=$(date +%s%N); python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --spawn-timeout 0.3 > /dev/null 2> /tmp/sp.err; echo "watchdog: rc=$? $(( ($(date +%s%N) - t0) / 1000000 )) ms: $(tail -1 /tmp/sp.err)"
t0=$(date +%s%N); python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --fail-rank 1 --rdzv-timeout 100 > /dev/null 2> /tmp/sp.err; echo "failing rank: rc=$? $(( ($(date +%s%N) - t0) / 1000000 )) ms: $(tail -1 /tmp/sp.err)"
} 2>&1 | tee gpurun_out/r04_spawn_runs.txt
