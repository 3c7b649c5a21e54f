for d in 0 1 4 5; do WR_STREAM_DBG=$d python tools/scratch/sbench.py 200 2>&1 | tail -4; done
WR_STREAM_DBG=0 WR_STREAM_NPOST=64 python tools/scratch/sbench.py 200 2>&1 | tail -2
WR_STREAM_DBG=0 WR_STREAM_RUN=1 python tools/scratch/sbench.py 200 2>&1 | tail -2
WR_STREAM_DBG=0 WR_STREAM_RUN=2000 python tools/scratch/sbench.py 200 2>&1 | tail -2
