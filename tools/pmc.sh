#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <cmd...>  -> gpurun_out/<tag>_pmc.txt (per-kernel counter means)
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -o $tag -- "$@" > /tmp/pmc_$tag.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
python3 - "$tag" <<'PY'
import csv, sys, glob, collections, os
tag = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        wgs = int(r.get('Grid_Size', 0) or 0) // max(1, int(r.get('Workgroup_Size', 1) or 1))
        k = '%s [%d workgroups]' % (r['Kernel_Name'].split('(')[0][:60], wgs)      # by launch shape
        rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/%s_pmc.txt' % tag, 'w')
for k, d in sorted(rows.items()):
    if not k.startswith(('k_', 'void k_')): continue
    line = k + ' : ' + ', '.join('%s=%.4g (n=%d)' % (c, sum(v)/len(v), len(v)) for c, v in sorted(d.items()))
    print(line); out.write(line + '\n')
PY
tail -2 /tmp/pmc_$tag.log
