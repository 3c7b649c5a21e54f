bash tools/scratch/ab.sh 200 3 base st_p1 st_p2 st_p3
for v in st_p1 st_p3; do cp webradio_amd/lib/libwebradio_amd.so /tmp/keep.so; cp tools/variants/$v/libwebradio_amd.so webradio_amd/lib/; echo $v; timeout 300 python -m pytest tests/test_gpu_stream.py -x -q -m gpu 2>&1 | tail -2; cp /tmp/keep.so webradio_amd/lib/libwebradio_amd.so; done
