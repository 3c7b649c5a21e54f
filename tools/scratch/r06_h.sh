# the block-entry words asked for beside the taps (pf) against asked for with the wave standing still (nopf)
timeout 600 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -4
bash tools/scratch/ab.sh 200 3 nopf pf
bash tools/scratch/ab.sh 20 3 nopf pf
