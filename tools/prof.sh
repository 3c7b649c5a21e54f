#!/bin/bash
# usage: tools/prof.sh <tag> <cmd...>   -> gpurun_out/<tag>_stats.csv (kernel stats)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- "$@" > /tmp/prof_$tag.log 2>&1
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv \;
tail -5 /tmp/prof_$tag.log
