#!/bin/bash
# kernel timeline of the last quick_time steps: start/end (us, relative) and queue id per kernel
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $GRAFT_REPO_ROOT/tools/quick_time.py 256 rotate > /tmp/tl.log 2>&1
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/tl/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if n.startswith(('k_tuner', 'void k_tuner')):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0][:28], r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
rows.sort()
rows = rows[-24:]
t0 = rows[0][0]
for s, e, n, q, st in rows:
    print('%9.1f %9.1f  dur %6.1f  q=%s st=%s  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, n))
PY
