for run in 4 5 6; do export WR_STREAM_POST_RUN=$run; echo "== run $run"; bash tools/scratch/ab.sh 200 3 ring6 ring7 ring8; bash tools/scratch/ab.sh 20 3 ring6 ring7 ring8; done
