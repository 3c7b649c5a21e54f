#!/bin/bash
# SQ counters of ONE streaming launch of 24 C2 blocks (k_tuner_stream<5, 2>), separate passes as tools/pmc.sh makes them:
#   gpurun -- 'bash tools/sq_stream.sh'  ->  gpurun_out/${ROUND:-r06}_sq_stream_{a,b,c}_pmc.txt
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 24 --warmup 0 --settle-ms 0 --no-secondary --no-cpu-baseline"
bash $R/tools/pmc.sh ${ROUND:-r06}_sq_stream_a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH" $CMD
bash $R/tools/pmc.sh ${ROUND:-r06}_sq_stream_b "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" $CMD
bash $R/tools/pmc.sh ${ROUND:-r06}_sq_stream_c "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" $CMD
