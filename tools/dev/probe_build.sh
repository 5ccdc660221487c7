#!/bin/bash
# developer: the library with ONE source rebuilt with a probe define -> build_variants/libganq_probe.so
# usage: tools/dev/probe_build.sh gemm_h16 HG_PROBE | tools/dev/probe_build.sh hessian_w4 HW_PROBE
set -e
cd "$(dirname "$0")/../.."
src=$1; def=$2
make -C ganq_amd/csrc -j8 >/dev/null
mkdir -p build_variants
/opt/rocm/bin/hipcc -D$def $EXTRA_DEFS --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-result -Wno-pass-failed -c ganq_amd/csrc/$src.hip -o build_variants/${src}_probe.o
objs=$(ls ganq_amd/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libganq_probe.so $objs build_variants/${src}_probe.o
echo built build_variants/libganq_probe.so
