timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_ring.py tests/test_gpu_reference_pin.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
for r in 1 2; do for run in 2 0; do
  if [ $run = 0 ]; then unset WR_STREAM_POST_RUN; else export WR_STREAM_POST_RUN=$run; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_p_${run}_$r.json 2>/dev/null
  python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('run=$run K=200 %.2f us/step frac %.4f' % (d['ms_per_step']*1e3, d['roofline']['frac']))"
done; done
