WR_STREAM_DBG=16 python tools/scratch/sb2.py 200 3 2>&1 | grep -v "amdgpu.ids\|\[wr\]"
WR_STREAM_DBG=17 python tools/scratch/sb2.py 200 3 2>&1 | grep -v "amdgpu.ids\|\[wr\]"
