#!/bin/bash
# tools/scratch/ab.sh K REPS name...: bench.py's headline (K steps) with the library variants tools/variants/<name>/, interleaved
K=$1; REPS=$2; shift 2
cp webradio_amd/lib/libwebradio_amd.so /tmp/lib_keep.so
for r in $(seq 1 $REPS); do for v in "$@"; do
  cp tools/variants/$v/libwebradio_amd.so webradio_amd/lib/libwebradio_amd.so
  python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v K=$K %.2f us/step kernel %.2f us/step' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3/$K))"
done; done
cp /tmp/lib_keep.so webradio_amd/lib/libwebradio_amd.so
