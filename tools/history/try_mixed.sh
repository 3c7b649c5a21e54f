#!/bin/bash
# usage: tools/try_mixed.sh -> every lane group mixed seven ways (per-lane-taps kernel), and one odd receiver, under every tools/variants/*.so
R=$GRAFT_REPO_ROOT
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  echo "== $(basename $v .so)"
  cd $R && timeout 300 python -m pytest tests/test_gpu_tuner.py -x -q -m gpu -k "mixed or odd or few_distinct or sixteen" 2>&1 | tail -1
  QT_MIXED=1 QT_REPS=800 QT_BLOCKS=12 QT_PROFILE=0 timeout 200 bash $R/tools/kstats.sh python $R/tools/quick_time.py 256 rotate
  QT_ONE_ODD=1 QT_REPS=800 QT_BLOCKS=12 QT_PROFILE=0 timeout 200 bash $R/tools/kstats.sh python $R/tools/quick_time.py 256 rotate | head -1
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
