cd /root/repo
for l2 in 64 128 256; do for run in 0 1 2 4; do echo "== QT_L2=$l2 WR_POST_FLUSH_RUN=$run"; WR_POST_FLUSH_RUN=$run QT_L2=$l2 QT_BLOCKS=12 QT_PROFILE=20 python tools/quick_time.py 256 rotate 2>&1 | tail -1; [ $l2 = 64 ] && break; done; done
echo "== QT_KEEP=1 (64 taps, two kernels)"; QT_KEEP=1 QT_BLOCKS=12 QT_PROFILE=20 python tools/quick_time.py 256 rotate 2>&1 | tail -1
for l2 in 128 256; do echo "== QT_KEEP=1 QT_L2=$l2 (two kernels)"; QT_KEEP=1 QT_L2=$l2 QT_BLOCKS=12 QT_PROFILE=20 python tools/quick_time.py 256 rotate 2>&1 | tail -1; done
