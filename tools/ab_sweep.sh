#!/bin/bash
# tools/ab_sweep.sh OUT [ROUNDS] — same-box, interleaved A/B of the scheduling variants of the sampler step (run on the
# GPU box through gpurun).  Every variant is one `bench.py` process (10 steps x 2 regions, no CPU leg); the variants are
# run round-robin ROUNDS times so that clock / thermal drift of the box lands on all of them alike.
#   GCD_ZIGZAG=0..4    walk directions of the big streaming launches (gcd_amd/engine.py)
#   GCD_AMD_LIB=...    tools/libgcd_amd_wt7.so: write-through epilogue stores (gemm_common.h, GCD_EPI_WT)
# Prints one line per run and a median table; OUT receives the same text.
OUT=${1:-gpurun_out/ab_sweep.txt}
ROUNDS=${2:-2}
cd "$(dirname "$0")/.."
mkdir -p "$(dirname "$OUT")"
# AB_VARIANTS="name:ENV=.. ENV=..;name2:..." overrides the default list
if [ -n "$AB_VARIANTS" ]; then
  IFS=';' read -r -a VARIANTS <<< "$AB_VARIANTS"
else
  VARIANTS=("base:GCD_ZIGZAG=0" "zz1:GCD_ZIGZAG=1" "zz2:GCD_ZIGZAG=2" "gn2:GCD_ZIGZAG=3" "ln2:GCD_ZIGZAG=4")
  if [ -f tools/libgcd_amd_wt7.so ]; then
    VARIANTS+=("wt7:GCD_ZIGZAG=0 GCD_AMD_LIB=tools/libgcd_amd_wt7.so" "wt7zz1:GCD_ZIGZAG=1 GCD_AMD_LIB=tools/libgcd_amd_wt7.so")
  fi
fi
: > "$OUT"
for r in $(seq 1 "$ROUNDS"); do
  for v in "${VARIANTS[@]}"; do
    name=${v%%:*}
    envs=${v#*:}
    line=$(env $envs python bench.py --steps 10 --warmup 3 --repeats 2 --no-cpu-baseline 2>/dev/null | tail -1)
    ms=$(python - "$line" <<'E'
import json, sys
try:
    d = json.loads(sys.argv[1])
    print("%.3f %.3f %.3f %s" % (d["timing_stats"]["ms_per_step_median"], d["roofline"]["kernel_ms_per_step"],
                                 d["attention"]["kernel_ms_per_step"], d["output_finite"]))
except Exception as e:
    print("FAILED", e)
E
)
    echo "round $r $name ($envs): step_ms gemm_ms attn_ms finite = $ms" | tee -a "$OUT"
  done
done
python - "$OUT" <<'E' | tee -a "$OUT"
import re, sys, statistics
rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"round \d+ (\S+) .* = ([\d.]+) ([\d.]+) ([\d.]+)", ln)
    if m:
        rows.setdefault(m.group(1), []).append(tuple(float(x) for x in m.groups()[1:]))
print("variant   runs  step_ms(median)  gemm_ms  attn_ms")
for k, v in rows.items():
    print("%-9s %4d  %10.3f  %10.3f %8.3f" % (k, len(v), statistics.median(x[0] for x in v),
                                               statistics.median(x[1] for x in v), statistics.median(x[2] for x in v)))
E
