one() { python bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$2 K=$1 %.2f us/step' % (d['ms_per_step']*1e3))"; }
for r in 1 2 3 4; do
  one 20 drain=2
  WR_STREAM_DRAIN_RUN=1 one 20 drain=1
done
bash tools/scratch/ab.sh 20 4 la3 la4 la5 la6
