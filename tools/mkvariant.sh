#!/bin/bash
# usage: tools/mkvariant.sh <name> [-DFLAG ...] -> tools/variants/<name>.so (the product library with
# wr_kernels.hip compiled under extra flags; timed on the GPU box by tools/try_variants*.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../webradio_amd/csrc"
mkdir -p ../../tools/variants /tmp/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -I../../include -I."
/opt/rocm/bin/hipcc $F "$@" -c wr_kernels.hip -o /tmp/var_$name/wr_kernels.o
for f in wr_fft wr_capi; do [ -f $f.o ] || /opt/rocm/bin/hipcc $F -c $f.hip -o $f.o; done
[ -f wr_design.o ] || /opt/rocm/bin/hipcc $F -x hip -c wr_design.cpp -o wr_design.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/wr_kernels.o wr_fft.o wr_capi.o wr_design.o -o ../../tools/variants/$name.so
echo built tools/variants/$name.so
