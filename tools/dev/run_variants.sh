#!/bin/bash
# developer: time one script against every library under build_variants/
for d in build_variants/*/; do n=$(basename $d); echo -n "$n: "; GANQ_HIP_LIB=$PWD/$d/libganq_hip.so python "$@" 2>&1 | grep -v amdgpu.ids | tail -1; done
