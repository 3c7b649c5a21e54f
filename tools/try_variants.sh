#!/bin/bash
# usage: tools/try_variants.sh  -> per-variant kernel stats (swaps the product .so on the GPU box only)
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  tag=$(basename $v .so)
  $R/tools/prof.sh $tag python $R/tools/quick_time.py 256 split > /dev/null 2>&1
  python3 - "$tag" <<'PY'
import csv, sys, os
tag = sys.argv[1]
out = []
for r in csv.DictReader(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/%s_kernel_stats.csv' % tag)):
    n = r['Name']
    if 'k_' in n[:12]:
        out.append('%s=%.1f' % (n.split('(')[0].replace('void ', '')[:22], float(r['AverageNs']) / 1e3))
print(tag, ' '.join(out))
PY
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
