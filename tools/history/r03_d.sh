#!/bin/bash
# r03: ablations of the lean DDC loop (timing only; the NOLDS / NOSEL variants compute wrong results)
R=$GRAFT_REPO_ROOT
cd $R
C=1 bash tools/try_steady.sh
C=4 bash tools/try_steady.sh
echo "== post stage NOT riding (WR_DEFER_POST=0): DDC kernel alone, every variant"
WR_DEFER_POST=0 C=1 bash tools/try_steady.sh
