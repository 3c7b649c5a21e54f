timeout 600 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -5
python tools/scratch/sb2.py 200 4 2>&1 | grep -v amdgpu.ids
WR_STREAM_RUN=4 python tools/scratch/sb2.py 200 3 2>&1 | tail -2
WR_STREAM_RUN=8 python tools/scratch/sb2.py 200 3 2>&1 | tail -2
WR_STREAM_NPOST=128 python tools/scratch/sb2.py 200 3 2>&1 | tail -2
WR_STREAM_NPOST=500 python tools/scratch/sb2.py 200 3 2>&1 | tail -2
WR_STREAM_DBG=1 python tools/scratch/sb2.py 200 3 2>&1 | tail -2
