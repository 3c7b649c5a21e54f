#!/bin/bash
# tools/scratch/mkvariant.sh NAME "<extra hipcc flags>": the product library with wr_kernels.hip compiled under extra flags,
# into tools/variants/NAME/libwebradio_amd.so (git-ignored; travels with gpurun) -- for A/B timing of compile-time choices
set -e
cd "$(dirname "$0")/../../webradio_amd/csrc"
mkdir -p ../../tools/variants/$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -I../../include -I. $2 -c wr_kernels.hip -o /tmp/wr_kernels_$1.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/wr_kernels_$1.o wr_fft.o wr_capi.o wr_design.o wr_ring.o -ldl -o ../../tools/variants/$1/libwebradio_amd.so
echo built $1
