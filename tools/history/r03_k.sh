#!/bin/bash
# SQ counters of the SpectrumSink passes (inside bench.py's C3 loop) and of the long-filter kernel (quick_time, QT_L1=128):
# gpurun_out/r03_sq_fft_{a,b}_pmc.txt, r03_sq_long_pmc.txt
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --settle-ms 30"
bash $R/tools/pmc.sh r03_sq_fft_a "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $B | grep fft
bash $R/tools/pmc.sh r03_sq_fft_b "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU" $B | grep fft
export QT_REPS=100 QT_BLOCKS=12 QT_PROFILE=0 QT_L1=128
bash $R/tools/pmc.sh r03_sq_long "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD" python $R/tools/quick_time.py 256 rotate | grep long
