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
# a second trace with 5 steps: (dispatches of 10 steps - dispatches of 5 steps) / 5 = launches per step, model construction excluded
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace5 -o t -- python $ROOT/tools/train_step_bench.py --steps 4 > /dev/null 2>&1)
python tools/rocpd_stats.py $(find $O/trace5 -name "*_results.db" | head -1) --steps 5 > $O/train_kernel_stats_${ENGINE}_5steps.txt
python - <<PY | tee $O/launches_per_step_${ENGINE}.txt
import re
def tot(f):
    m = re.search(r"total GPU kernel time ([\d.]+) ms over (\d+) dispatches", open(f).read())
    return float(m.group(1)), int(m.group(2))
t10, n10 = tot("$O/train_kernel_stats_${ENGINE}.txt")
t5, n5 = tot("$O/train_kernel_stats_${ENGINE}_5steps.txt")
print(f"${ENGINE} engine: {(n10 - n5) / 5:.0f} launches per step, {(t10 - t5) / 5:.1f} ms of kernel time per step "
      f"(10-step trace {n10} dispatches / {t10:.0f} ms, 5-step trace {n5} / {t5:.0f} ms; the rest is model construction)")
PY
rm -rf $O/trace $O/trace5
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
