cp webradio_amd/lib/libwebradio_amd.so /tmp/keep.so; cp tools/variants/tl/libwebradio_amd.so webradio_amd/lib/
WR_STREAM_DBG=16 python tools/scratch/tl20.py 20
cp /tmp/keep.so webradio_amd/lib/libwebradio_amd.so
