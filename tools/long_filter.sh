#!/bin/bash
# r05: what a filter of 128 / 256 taps costs per BASELINE config 2 block in each of the three places the tuner takes one
# (tools/quick_time.py 256 rotate, 12 resident blocks, one event pair per 20 launches; all launches of a block inside).
# gpurun -- 'bash tools/long_filter.sh > gpurun_out/long_filter.txt'
cd "$(dirname "$0")/.."
q() { echo "== $1"; env $1 QT_BLOCKS=12 QT_PROFILE=20 python tools/quick_time.py 256 rotate 2>&1 | tail -1; }
q "QT_L2=64"
for l in 128 256; do
	q "QT_L1=$l"
	q "QT_L2=$l"
	q "QT_L2=$l QT_KEEP=1"
	q "QT_L2=$l WR_POST_FLUSH_RUN=1"
	q "QT_L2=$l WR_POST_FLUSH_RUN=2"
done
q "QT_KEEP=1"
q "QT_L1B=64"
q "QT_L1B=128"
q "QT_L1B=256"
q "QT_L1=256 QT_L1B=256 QT_L2=256"
