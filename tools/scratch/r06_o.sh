one() { python bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$2 K=$1 %.2f us/step' % (d['ms_per_step']*1e3))"; }
for r in 1 2 3; do for run in 2 4 5 6 8 12; do
  WR_STREAM_POST_RUN=$run one 200 run=$run
  WR_STREAM_POST_RUN=$run one 20 run=$run
done; done
for r in 1 2; do for np in 192 128; do
  WR_STREAM_POST_RUN=4 WR_STREAM_NPOST=$np one 200 run=4,npost=$np
  WR_STREAM_POST_RUN=4 WR_STREAM_NPOST=$np one 20 run=4,npost=$np
done; done
