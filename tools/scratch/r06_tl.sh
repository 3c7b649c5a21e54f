# profiles/r06_stream_timeline.txt: the streaming launch taken apart with its own per-wave cycle counters (-DSTREAM_TL build, tools/variants/tl)
cp webradio_amd/lib/libwebradio_amd.so /tmp/keep.so; cp tools/variants/tl/libwebradio_amd.so webradio_amd/lib/
echo "== k_tuner_stream<5,2> at C2, -DSTREAM_TL build, WR_STREAM_DBG=16 (tools/scratch/sbench.py 200): per-wave cycle counters of one 200-block streaming launch, r06 final code (ring of 6 blocks, post-stage runs of 5 tiles)"
WR_STREAM_DBG=16 python tools/scratch/sbench.py 200 2>&1 | grep -v amdgpu.ids
echo "== the same with the post stage body skipped (WR_STREAM_DBG=17: results wrong, timing only)"
WR_STREAM_DBG=17 python tools/scratch/sbench.py 200 2>&1 | grep -v amdgpu.ids | head -12
echo "== the same with post-stage runs of 2 tiles as in r05 (WR_STREAM_POST_RUN=2)"
WR_STREAM_POST_RUN=2 WR_STREAM_DBG=16 python tools/scratch/sbench.py 200 2>&1 | grep -v amdgpu.ids | head -12
echo "== a 20-block stream, block by block (tools/scratch/tl20.py; us from the launch's first instruction)"
WR_STREAM_DBG=16 python tools/scratch/tl20.py 20 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so webradio_amd/lib/libwebradio_amd.so
