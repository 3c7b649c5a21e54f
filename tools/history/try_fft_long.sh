#!/bin/bash
# usage: tools/try_fft_long.sh -> C3 per 121-frame block by frames per pass-1 workgroup (WR_FFT_P1_FPW), bench.py's own events + rocprofv3,
# under the shipped library and every tools/variants/*.so
R=$GRAFT_REPO_ROOT
one() { timeout 200 python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --settle-ms 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['secondary']['c3']; print('%.2f us  frac %.4f' % (c['ms_per_block']*1e3, c['roofline']['frac']))"; }
run() {
  for f in 1 2 4; do echo -n "  $f frame(s) per workgroup: "; WR_FFT_P1_FPW=$f one; done
  for f in 1 2 4; do echo -n "  rocprofv3, $f: "; WR_FFT_P1_FPW=$f bash $R/tools/kstats.sh python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --settle-ms 30 | grep pass1; done
}
echo "== shipped"; run
cp $R/webradio_amd/lib/libwebradio_amd.so /tmp/orig.so
for v in $R/tools/variants/*.so; do
  [ -f "$v" ] || continue
  cp $v $R/webradio_amd/lib/libwebradio_amd.so
  echo "== $(basename $v .so)"; run
done
cp /tmp/orig.so $R/webradio_amd/lib/libwebradio_amd.so
