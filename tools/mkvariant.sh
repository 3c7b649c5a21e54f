#!/bin/bash
# usage: tools/mkvariant.sh <name> [-DFLAG ...] -> tools/variants/<name>.so (the product library with
# wr_kernels.hip compiled under extra flags; timed on the GPU box by tools/try_variants*.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../webradio_amd/csrc"
mkdir -p ../../tools/variants /tmp/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -I../../include -I."
/opt/rocm/bin/hipcc $F "$@" -c wr_kernels.hip -o /tmp/var_$name/wr_kernels.o
# the other objects: rebuilt whenever a source or header is newer (a stale wr_capi.o with an old struct layout
# once cost a 25-minute hang on the GPU box)
O=/tmp/var_common; mkdir -p $O
for f in wr_fft wr_capi wr_ring; do
  if [ ! -f $O/$f.o ] || [ -n "$(find . ../../include -newer $O/$f.o \( -name '*.h' -o -name "$f.hip" \))" ]; then /opt/rocm/bin/hipcc $F -c $f.hip -o $O/$f.o; fi
done
if [ ! -f $O/wr_design.o ] || [ -n "$(find . ../../include -newer $O/wr_design.o \( -name '*.h' -o -name 'wr_design.cpp' \))" ]; then /opt/rocm/bin/hipcc $F -x hip -c wr_design.cpp -o $O/wr_design.o; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/wr_kernels.o $O/wr_fft.o $O/wr_capi.o $O/wr_design.o $O/wr_ring.o -ldl -o ../../tools/variants/$name.so
echo built tools/variants/$name.so
