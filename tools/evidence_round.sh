#!/bin/bash
# The secondary evidence of a round (run on the GPU box via gpurun, after tools/profile_round.sh):
#   <round>_bench_driver_flags.json  bench.py with the driver's flags
#   <round>_c5_1gpu.json             bench.py --workload c5 on one GPU
#   <round>_host_bench.txt           the drop-in C++ path, block in host memory (audio on time / one block late)
#   <round>_mixed_passbands.txt      one odd receiver among 256 / every lane group mixed seven ways
#   <round>_clock_ramp.txt           kernel duration by time since the first launch from an idle GPU
round=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${round}_bench_driver_flags.json 2> /dev/null
timeout 300 python bench.py --workload c5 --steps 30 --warmup 4 > $O/${round}_c5_1gpu.json 2> /dev/null
timeout 300 python bench.py --workload c5 --halo ring --steps 30 --warmup 4 > $O/${round}_c5_1gpu_ring.json 2> /dev/null
( for src in f32 u8; do for late in 0 1 2; do
    WEBRADIO_AUDIO_LATE=$late WR_HOST_BENCH_PROFILE=1 timeout 200 tests/cxx/host_bench 256 300 4000000 $src
  done; done ) 2>&1 | grep -E "^\{|^process\(\)" > $O/${round}_host_bench.txt
( echo "== QT_ONE_ODD=1"; QT_ONE_ODD=1 QT_REPS=1600 QT_BLOCKS=12 QT_PROFILE=0 timeout 200 bash tools/kstats.sh python $R/tools/quick_time.py 256 rotate
  echo "== QT_MIXED=1";   QT_MIXED=1 QT_REPS=1600 QT_BLOCKS=12 QT_PROFILE=0 timeout 200 bash tools/kstats.sh python $R/tools/quick_time.py 256 rotate ) > $O/${round}_mixed_passbands.txt 2>&1
timeout 300 bash tools/clock_ramp.sh > $O/${round}_clock_ramp.txt 2>&1
cut -c1-300 $O/${round}_bench_driver_flags.json; cut -c1-300 $O/${round}_c5_1gpu.json; cat $O/${round}_host_bench.txt $O/${round}_mixed_passbands.txt; cat $O/${round}_clock_ramp.txt
