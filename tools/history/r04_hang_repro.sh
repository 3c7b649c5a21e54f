#!/bin/bash
# r04: look for the driver-side hang of round 3 (GPUTEST_r03: pytest -m gpu killed at 1200 s inside
# tests/test_gpu_bench.py).  Runs the self-spawning two-rank bench N times under `timeout`, with and
# without HSA_ENABLE_IPC_MODE_LEGACY, and logs wall time + exit code of every run.
N=${1:-10}
OUT=gpurun_out/r04_hang_repro.txt
mkdir -p gpurun_out
{
echo "box: $(hostname)  $(date -u +%FT%TZ)  HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY-unset}"
python -c "import torch; print('torch', torch.__version__, 'gpus', torch.cuda.device_count())"
for i in $(seq 1 $N); do
  for variant in plain c5 launcher; do
    case $variant in
      plain) cmd="python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo" ;;
      c5) cmd="python bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --workload c5" ;;
      launcher) cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 4 --warmup 1 --backend gloo" ;;
    esac
    t0=$(date +%s.%N)
    TORCH_DISTRIBUTED_DEBUG=DETAIL timeout -k 5 120 $cmd > /tmp/repro.out 2> /tmp/repro.err
    rc=$?
    t1=$(date +%s.%N)
    printf "%2d %-8s rc=%d %.1fs %s\n" $i $variant $rc $(echo "$t1 - $t0" | bc) "$(grep -c '^{' /tmp/repro.out) line(s)"
    if [ $rc -ne 0 ]; then echo "---- stderr tail"; tail -30 /tmp/repro.err; fi
  done
done
} 2>&1 | tee $OUT
