#!/bin/bash
# usage: tools/mkvariant_fft.sh <name> [-DFLAG ...] -> tools/variants/<name>.so (wr_fft.hip compiled under extra flags)
set -e
name=$1; shift
cd "$(dirname "$0")/../webradio_amd/csrc"
mkdir -p ../../tools/variants /tmp/var_$name
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -I../../include -I."
/opt/rocm/bin/hipcc $F "$@" -c wr_fft.hip -o /tmp/var_$name/wr_fft.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC wr_kernels.o /tmp/var_$name/wr_fft.o wr_capi.o wr_design.o wr_ring.o -ldl -o ../../tools/variants/$name.so
echo built tools/variants/$name.so
