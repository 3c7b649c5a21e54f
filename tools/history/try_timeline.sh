#!/bin/bash
# usage: tools/try_timeline.sh -> tools/timeline.py under every tools/variants/tl*.so (built with -DDDC_TIMELINE)
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/tl*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  echo "== $(basename $v .so)"
  timeout 120 python $R/tools/timeline.py 2>&1 | tail -40
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
