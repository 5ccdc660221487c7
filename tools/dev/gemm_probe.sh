#!/bin/bash
# developer: the library with the four-wave GEMM's cycle probes compiled in (-DHG_PROBE) -> build_variants/libganq_probe.so
set -e
cd "$(dirname "$0")/../.."
make -C ganq_amd/csrc -j8 >/dev/null
mkdir -p build_variants
/opt/rocm/bin/hipcc -DHG_PROBE --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-result -Wno-pass-failed -c ganq_amd/csrc/gemm_h16.hip -o build_variants/gemm_h16_probe.o
objs=$(ls ganq_amd/csrc/*.o | grep -v gemm_h16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libganq_probe.so $objs build_variants/gemm_h16_probe.o
echo built build_variants/libganq_probe.so
