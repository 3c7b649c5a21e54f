#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
C=1 bash tools/try_steady.sh
C=4 bash tools/try_steady.sh
