#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(SQ\|SQC\)_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/r03_counters.txt
cd $R; $R/tools/ubench_tap > $R/gpurun_out/r03_ubench_tap.txt 2>&1; cat $R/gpurun_out/r03_ubench_tap.txt
export QT_REPS=300 QT_BLOCKS=12 QT_PROFILE=0
bash tools/pmc.sh r03_sq_a "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_VMEM" python $R/tools/quick_time.py 256 rotate | grep ddc
bash tools/pmc.sh r03_sq_b "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" python $R/tools/quick_time.py 256 rotate | grep ddc
bash tools/pmc.sh r03_sq_c "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES" python $R/tools/quick_time.py 256 rotate | grep ddc
cat /tmp/pmc_r03_sq_b.log | tail -5
