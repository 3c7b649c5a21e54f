timeout 600 python -m pytest tests/test_gpu_stream.py tests/test_gpu_ring.py -x -q -m gpu 2>&1 | tail -2
bash tools/scratch/ab.sh 20 8 base fix
bash tools/scratch/ab.sh 200 3 base fix
cp webradio_amd/lib/libwebradio_amd.so /tmp/keep.so; cp tools/variants/tl/libwebradio_amd.so webradio_amd/lib/
WR_STREAM_DBG=16 python tools/scratch/tl20.py 20 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so webradio_amd/lib/libwebradio_amd.so
