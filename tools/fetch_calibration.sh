#!/bin/bash
# FETCH_SIZE (and the request counters behind it) against known byte counts, per load shape.
#   gpurun -- 'bash tools/fetch_calibration.sh'   ->   gpurun_out/r05_fetch_calibration.txt
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r05_fetch_calibration.txt
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "FETCH_SIZE calibration on known byte counts (tools/ubench_fetch.hip; 1 GiB buffer, each kernel reads its bytes once; KB = 1024 B)" > $out
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $ctrs | tr ' ' '_')
  rm -rf /tmp/fc_$tag
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/fc_$tag -o fc -- $R/tools/ubench_fetch > /tmp/fc_$tag.log 2>&1
  grep "bytes asked" /tmp/fc_$tag.log | head -1 >> $out
  python3 - "$tag" >> $out <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/fc_%s/**/*counter_collection.csv' % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
asked = {'k_win8': (1 << 30) // 3200 * 512, 'k_win8u': (1 << 30) // 800 * 128}
for k, d in sorted(acc.items()):
    want = asked.get(k.replace('void ', '').split('<')[0], 1 << 30)
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        extra = ''
        if c in ('FETCH_SIZE', 'WRITE_SIZE'):
            extra = '  = %.3f x the bytes asked for (counter in KB)' % (m * 1024 / want)
        elif 'RDREQ' in c or 'WRREQ' in c:
            extra = '  -> %.1f bytes asked for per count' % (want / m if m else 0)
        print('  %-40s %-24s mean %.6g (n=%d)%s' % (k, c, m, len(v), extra))
PY
  tail -1 /tmp/fc_$tag.log | grep -i "error\|fail" >> $out
done
cat $out
