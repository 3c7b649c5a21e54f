bash tools/scratch/ab.sh 20 12 base rev
