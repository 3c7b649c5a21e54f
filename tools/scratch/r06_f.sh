R=$GRAFT_REPO_ROOT; round=r06
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${round}_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${round}_s -o s -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${round}_stream_bench.json 2>/tmp/prof_${round}_s.log
cd $R
python3 - "$round" <<'PY'
import csv, sys, glob, os, json
r = sys.argv[1]; R = os.environ['GRAFT_REPO_ROOT']
rows = []
for f in glob.glob('/tmp/prof_%s_s/**/*kernel_trace.csv' % r, recursive=True):
    for x in csv.DictReader(open(f)):
        if 'k_tuner_stream' in x['Kernel_Name']:
            rows.append((int(x['Start_Timestamp']), int(x['End_Timestamp']) - int(x['Start_Timestamp'])))
rows.sort()
line = [l for l in open(R + '/gpurun_out/%s_stream_bench.json' % r) if l.startswith('{')][-1]
j = json.loads(line)
with open(R + '/gpurun_out/%s_stream_launches.txt' % r, 'w') as f:
    f.write('rocprofv3 --kernel-trace of `python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary`: every k_tuner_stream launch, in order, ns\n')
    f.write(' '.join(str(d) for _, d in rows) + '\n')
    last = rows[-1][1]
    f.write('the LAST launch is the timed region\'s (60 blocks): %d ns = %.2f us per block; bench.py\'s own events for it: kernel_ms %.5f = %.2f us per block; '
            'wall clock ms_per_step %.5f\n' % (last, last / 60e3, j['roofline']['kernel_ms'], j['roofline']['kernel_ms'] / 60 * 1e3, j['ms_per_step']))
    settle = [d for _, d in rows[:-1] if d > 2000000]
    if settle:
        f.write('the %d launches of the clock settling (100 blocks each): mean %.1f us per block\n' % (len(settle), sum(settle) / len(settle) / 100e3))
PY
cat gpurun_out/r06_stream_launches.txt
