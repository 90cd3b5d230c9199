#!/bin/bash
# tools/first_multi_gpu.sh [N=8] — everything that has never run on more than one GPU, in one command, for the first
# lease of a multi-GPU node (from the repo root; outputs under gpurun_out/multi_gpu/).  Nothing here has been executed
# on hardware yet: the pool gives one GPU per lease.  What it runs, in order:
#   1. bench.py --gpus 2 / 4 / N   one rank per GPU over RCCL; bench.py exits non-zero unless an executed all-reduce of
#                                  ones over the nccl backend returns N (`rccl_ranks`) — the scaling curve of cfg2
#   2. tests/test_parallel_gpu.py  with GCD_DIST_BACKEND=nccl and one GPU per rank (GCD_TEST_GPUS_PER_RANK=1): clip
#                                  sharding through the fused hipGraph loop, the data-parallel fine-tune step
#   3. bench.py --train            the fine-tune step of cfg4 on ONE GPU of the node (bf16 + fp16, parity beside it): the
#                                  single-GPU figure the DDP step below is compared with
#   4. tools/train_step_bench.py   under torchrun, GradBucketer on RCCL (few-row parameters in their own last buckets):
#                                  step time, exposed all-reduce ms (cfg4's DDP)
# Reference pattern: scripts/test.py:1051-1090 (one replica per GPU, strided clips), main.py:826-843 (DDPStrategy).
set -u
N=${1:-8}
OUT=gpurun_out/multi_gpu
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
have=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $have (asked for $N)" | tee $OUT/summary.txt
if [ "$have" -lt 2 ]; then echo "needs >= 2 GPUs" | tee -a $OUT/summary.txt; exit 2; fi
[ "$N" -gt "$have" ] && N=$have
rc=0
for n in 1 2 4 $N; do
  [ "$n" -gt "$N" ] && continue
  python bench.py --gpus $n --steps 20 --warmup 3 > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err || { echo "bench --gpus $n FAILED" | tee -a $OUT/summary.txt; rc=1; }
  python - $OUT/bench_n$n.json $n <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    n = int(sys.argv[2])
    ok = n == 1 or d.get("rccl_ranks") == n
    print(f"N={n}: {d['value']:.2f} steps/s, {d['ms_per_step']:.2f} ms/step max over ranks, rccl_ranks={d.get('rccl_ranks')}"
          f"{'' if ok else '  <-- RCCL DID NOT RUN ON ALL RANKS'}")
except Exception as e:
    print(f"N={sys.argv[2]}: no bench line ({e})")
PY
done
GCD_DIST_BACKEND=nccl GCD_TEST_GPUS_PER_RANK=1 python -m pytest tests/test_parallel_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/test_parallel_gpu_nccl.log
python bench.py --train > $OUT/bench_train.json 2> $OUT/bench_train.err || { echo "bench --train FAILED" | tee -a $OUT/summary.txt; rc=1; }
python - $OUT/bench_train.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    t = d["train"]
    print("fine-tune step, 1 GPU: " + ", ".join(f"{k} {v['step_s']:.4f} s = {v['tflops']:.0f} TFLOP/s" for k, v in t.items()))
except Exception as e:
    print(f"no bench --train line ({e})")
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29641 \
  tools/train_step_bench.py --steps 5 --ddp > $OUT/train_step_ddp.json 2> $OUT/train_step_ddp.err || { echo "DDP train step FAILED" | tee -a $OUT/summary.txt; rc=1; }
tail -2 $OUT/train_step_ddp.json | tee -a $OUT/summary.txt
exit $rc
