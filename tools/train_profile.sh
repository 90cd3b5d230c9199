#!/bin/bash
# tools/train_profile.sh TAG [ENGINE] — the fine-tune step's bench line and a rocprofv3 kernel trace of the same command
# (10 steps, so that model construction amortises), summarised per step.  Run from the repo root on the GPU box.
TAG=${1:-train}
ENGINE=${2:-planned}
ROOT=$(pwd)
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp GCD_TRAIN_ENGINE=$ENGINE
python tools/train_step_bench.py --steps 5 > $O/train_${ENGINE}.json 2> $O/train_${ENGINE}.err
cat $O/train_${ENGINE}.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $ROOT/tools/train_step_bench.py --steps 9 > /dev/null 2>&1)
python tools/rocpd_stats.py $(find $O/trace -name "*_results.db" | head -1) --steps 10 > $O/train_kernel_stats_${ENGINE}.txt
rm -rf $O/trace
python - <<PY
import re
rows = []
for l in open("$O/train_kernel_stats_${ENGINE}.txt"):
    m = re.match(r'(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', l)
    if m and not l.startswith('#'):
        rows.append((m.group(1)[:64], int(m.group(2)), float(m.group(3)), float(m.group(5))))
print("launches/step (10 steps + construction)", sum(r[1] for r in rows) / 10, "kernel ms/step", round(sum(r[2] for r in rows) / 10, 1))
for n, c, t, a in rows[:40]:
    print(f"{n:64s} {c / 10:7.0f}/step {t / 10:7.2f} ms/step avg {a:7.1f} us")
PY
