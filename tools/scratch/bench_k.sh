cd /root/repo
for i in $(seq 1 12); do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('K=20 stream %.1f us/step %.0f Msps kernel %.1f us' % (d['ms_per_step']*1e3, d['value'], d['roofline']['kernel_ms']*1e3))"; done
