#!/bin/bash
# tools/collect_profiles.sh TAG — everything profiles/TAG_* is made from, in one pass on the GPU box (from the repo
# root): the bench line, the rocprofv3 kernel trace of the same command, the PMC traffic / SQ passes (one counter
# group per pass, --kernel-trace only), the per-shape GEMM tables, the fine-tune step.  Outputs under gpurun_out/TAG/.
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python $ROOT/bench.py --steps 5 --warmup 2 \
    --repeats 1 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err)
python tools/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) --steps 8 > $OUT/kernel_stats.txt
bash tools/collect_pmc.sh $TAG
tools/gemm_bench full 5 > $OUT/gemm_bench_rows.txt 2>&1
# the fine-tune step through the same bench.py the driver runs (bf16 + fp16, C-ABI calls, cfg4 golden parity)
python bench.py --train > $OUT/bench_train.json 2> $OUT/bench_train.err
# the fine-tune step (round 5: planned engine by default; the autograd engine and forced checkpointing beside it)
for dt in fp16 bf16; do python tools/train_step_bench.py --steps 4 --dtype $dt > $OUT/train_step_$dt.json 2>/dev/null; done
python tools/train_step_bench.py --steps 4 --checkpoint on > $OUT/train_step_fp16_checkpointed.json 2>/dev/null
GCD_TRAIN_ENGINE=autograd python tools/train_step_bench.py --steps 4 > $OUT/train_step_fp16_autograd_engine.json 2>/dev/null
GCD_TRAIN_GRAPH=1 python tools/train_step_bench.py --steps 8 > $OUT/train_step_fp16_hipgraph.json 2>/dev/null
for c in 4 8; do python tools/train_step_bench.py --steps 3 --clips $c > $OUT/train_step_fp16_${c}clips.json 2>/dev/null; done
python tools/train_step_bench.py --steps 3 --clips 8 --checkpoint on > $OUT/train_step_fp16_8clips_checkpointed.json 2>/dev/null
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace_train -o t -- python $ROOT/tools/train_step_bench.py --steps 9 \
    > /dev/null 2>&1)
python tools/rocpd_stats.py $(find $OUT/trace_train -name "*_results.db" | head -1) --steps 10 > $OUT/train_kernel_stats.txt
# launches per step without model construction: difference of a 10-step and a 5-step trace (tools/train_profile.sh), both engines
for e in planned autograd; do bash tools/train_profile.sh $TAG/tp_$e $e > /dev/null 2>&1; cp $OUT/tp_$e/launches_per_step_$e.txt $OUT/; done
cat $OUT/launches_per_step_*.txt
# keep the summaries only: gpurun merges at most 64 MiB back
rm -rf $OUT/trace $OUT/trace_train
du -sh $ROOT/gpurun_out
ls -la $OUT
# round 4: the CPU path timed at the metric's own shape beside the GPU number (one oracle step at 14x72x128, minutes)
if [ "${GCD_CPU_FULL:-0}" = "1" ]; then
  python bench.py --steps 10 --warmup 2 --repeats 1 --cpu-baseline-full > $OUT/bench_cpu_full.json 2> $OUT/bench_cpu_full.err
fi
