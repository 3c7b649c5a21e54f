#!/bin/bash
# as mkvariant.sh, for choices wr_capi.hip has to know about too (WR_STREAM_RING): both objects under the extra flags
set -e
cd "$(dirname "$0")/../../webradio_amd/csrc"
mkdir -p ../../tools/variants/$1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -Wno-pass-failed -I../../include -I. $2"
/opt/rocm/bin/hipcc $F -c wr_kernels.hip -o /tmp/wr_kernels_$1.o 2>/dev/null
/opt/rocm/bin/hipcc $F -c wr_capi.hip -o /tmp/wr_capi_$1.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/wr_kernels_$1.o wr_fft.o /tmp/wr_capi_$1.o wr_design.o wr_ring.o -ldl -o ../../tools/variants/$1/libwebradio_amd.so
echo built $1
