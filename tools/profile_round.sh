#!/bin/bash
# Produces the rocprof evidence for bench.py's roofline numbers (run on the GPU box via gpurun):
#   gpurun_out/<round>_bench.json            the bench line (C2 headline + secondary.c3 + cpu_baseline)
#   gpurun_out/<round>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (all launches: warm-up,
#                                            clock settling and the timed steps)
#   gpurun_out/<round>_kernel_stats_timed.csv   the same trace, the timed region's launches only
#   gpurun_out/<round>_pmc_fetch_size.txt / _pmc_write_size.txt   FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
round=${1:-r06}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && python bench.py > gpurun_out/${round}_bench.json 2> gpurun_out/${round}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$round
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$round -o $round -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > /tmp/prof_$round.log 2>&1
python3 - "$round" <<'PY'
import csv, sys, os
r = sys.argv[1]; R = os.environ['GRAFT_REPO_ROOT']
src = '/tmp/prof_%s/%s_kernel_stats.csv' % (r, r)
rows = list(csv.DictReader(open(src)))
with open(R + '/gpurun_out/%s_kernel_stats.csv' % r, 'w') as f:
    w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for x in rows:
        name = x['Name']
        short = name.split('(')[0][:90]          # torch's generator kernels have page-long names
        w.writerow([short, x['Calls'], x['TotalDurationNs'], x['AverageNs'], x['Percentage'], x['MinNs'], x['MaxNs']])
PY
# by launch shape: with bench.py's default --blocks-per-launch 4 the dominant kernel runs with the grid of a
# 16 M-frame launch (headline) and, in secondary.c2_one_block_per_launch, with the grid of a 4 M-frame launch.
# Each shape's timed region is its last launches (before them: warm-up and the clock-settling steps, see
# bench.py --settle-ms): the last 15 are reported beside the mean over all.
python3 - "$round" <<'PY'
import csv, sys, glob, os, collections
r = sys.argv[1]; R = os.environ['GRAFT_REPO_ROOT']
rows = []
for f in glob.glob('/tmp/prof_%s/**/*kernel_trace.csv' % r, recursive=True):
    for x in csv.DictReader(open(f)):
        rows.append((int(x['Start_Timestamp']), x['Kernel_Name'].split('(')[0].replace('void ', ''),
                     int(x.get('Grid_Size_X', x.get('Grid_Size', 0)) or 0) // max(1, int(x.get('Workgroup_Size_X', x.get('Workgroup_Size', 1)) or 1)),
                     int(x['End_Timestamp']) - int(x['Start_Timestamp'])))
rows.sort()
g = collections.OrderedDict()
for _, name, wgs, d in rows:
    if name.startswith('k_'):
        g.setdefault((name, wgs), []).append(d)
with open(R + '/gpurun_out/%s_kernel_stats_timed.csv' % r, 'w') as f:
    w = csv.writer(f); w.writerow(['Name', 'Workgroups', 'Calls', 'AverageNs', 'MinNs', 'MaxNs', 'Last15AverageNs'])
    for (name, wgs), d in sorted(g.items()):
        if len(d) < 3 and not name.startswith(('k_fft', 'k_tuner_stream')):
            continue
        last = d[-16:-1] if len(d) > 40 else d
        w.writerow([name, wgs, len(d), '%.1f' % (sum(d) / len(d)), min(d), max(d), '%.1f' % (sum(last) / len(last))])
PY
# the headline's kernel by itself: bench.py --no-secondary, every k_tuner_stream launch in order -- the warm-up's, the 100-block
# streams of the clock settling, and LAST the timed region's one launch of 60 blocks (what bench.py's own events time)
rm -rf /tmp/prof_${round}_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${round}_s -o s -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/${round}_stream_bench.json 2>/tmp/prof_${round}_s.log
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
# PMC passes (separate runs, --kernel-trace only beside --pmc):
#   "stream": the headline's kernel -- ONE streaming launch of $SB blocks (no warm-up, no settling steps: the byte counters do
#             not care about the clock), so that bytes per launch / $SB = bytes per block
#   ""      : bench.py --no-stream --blocks-per-launch 4 -- k_tuner_ddc in its four-block and (secondary) one-block shapes, the FFT passes
SB=24
echo $SB > $R/gpurun_out/${round}_pmc_stream_blocks.txt
for variant in stream ""; do
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  if [ "$variant" = "stream" ]; then
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps $SB --warmup 0 --settle-ms 0 --no-secondary --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  else
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --no-stream --blocks-per-launch 4 --steps 12 --warmup 2 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
  fi
  python3 - "$c" "$round" "$variant" <<'PY'
import csv, sys, glob, collections, os
c, r, variant = sys.argv[1], sys.argv[2], sys.argv[3]; R = os.environ['GRAFT_REPO_ROOT']
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0][:60]
        wgs = int(row.get('Grid_Size', 0) or 0) // max(1, int(row.get('Workgroup_Size', 1) or 1))
        if k.startswith(('k_', 'void k_')):
            acc['%s [%d workgroups]' % (k, wgs)].append(float(row['Counter_Value']))
name = '%s_pmc_%s%s.txt' % (r, 'stream_' if variant else '', c.lower())
with open(R + '/gpurun_out/' + name, 'w') as out:
    for k, v in sorted(acc.items()):
        if len(v) < 3 and 'k_fft' not in k and 'k_tuner_stream' not in k:
            continue
        out.write('%s %s mean=%.1f KB per launch (n=%d)\n' % (k, c, sum(v) / len(v), len(v)))
PY
done
done
# profiles/traffic.json (what bench.py quotes as roofline.traffic) from the two PMC dumps: no hand step
python3 $R/tools/make_traffic.py $round
cat $R/gpurun_out/${round}_bench.json | cut -c1-600; grep -E "k_tuner|k_fft" $R/gpurun_out/${round}_kernel_stats.csv; cat $R/gpurun_out/${round}_kernel_stats_timed.csv; cat $R/gpurun_out/${round}_pmc_*.txt
