bash tools/scratch/ab.sh 200 6 base ring7 ring6
bash tools/scratch/ab.sh 20 6 base ring7 ring6
