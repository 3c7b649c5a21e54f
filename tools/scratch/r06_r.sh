# does the launch's time depend on how much input lies resident (the Infinity Cache's share for the ring)?
for r in 1 2 3; do for nb in 2 6 12 24 48; do
python bench.py --steps 200 --warmup 5 --resident-blocks $nb --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('resident=$nb K=200 %.2f us/step' % (d['ms_per_step']*1e3))"
done; done
