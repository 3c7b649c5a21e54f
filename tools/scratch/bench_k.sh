cd /root/repo
for i in 1 2 3; do for k in 20 200; do python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('K=$k stream %.1f us/step %.0f Msps' % (d['ms_per_step']*1e3, d['value']))"; done; done
