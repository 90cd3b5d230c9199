#!/bin/bash
# Builds tools/ff_fused_probe_<name> for a list of "name:extra flags" variants of the fused FeedForward probe
# (gcd_amd/csrc/ff_fused_kernel.h knobs FF_ABL / FF_D / FF_DMA / FF_GELU_DEG).  Usage: tools/ff_probe_build.sh name:flags ...
set -e
cd "$(dirname "$0")/.."
F="-O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=fast -fno-slp-vectorize tools/ff_fused_probe.cpp -Iinclude -Igcd_amd/csrc -Lgcd_amd -lgcd_amd -Wl,-rpath,\$ORIGIN/../gcd_amd"
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  ( hipcc $F $flags -o tools/ff_fused_probe_$name 2>&1 | grep -E "error" || true ) &
  while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 0.5; done
done
wait
ls tools/ff_fused_probe_* | tr '\n' ' '
