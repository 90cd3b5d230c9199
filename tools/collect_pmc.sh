#!/bin/bash
# tools/collect_pmc.sh TAG — the counter passes of tools/collect_profiles.sh on their own (from the repo root, on the GPU
# box): FETCH_SIZE / WRITE_SIZE / TCC hit-miss (one counter group per rocprofv3 pass, --kernel-trace only) -> hbm_traffic.{txt,
# json} + gemm_shape_traffic.txt, then the two SQ passes -> sq_counters.txt, sq_valu_mix.txt.  Outputs under gpurun_out/TAG/.
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$OUT/pmc_${C%% *}
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -o p -- python $ROOT/bench.py --steps 1 \
      --warmup 1 --repeats 1 --no-cpu-baseline --dump-profile $OUT/launches.json > $D.log 2>&1)
done
F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
H=$(find $OUT/pmc_TCC_HIT_sum -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $F $W --json $OUT/hbm_traffic.json --source profiles/${TAG}_hbm_traffic.txt > $OUT/hbm_traffic.txt
python tools/pmc_gemm_shapes.py $OUT/launches.json $F $W $H > $OUT/gemm_shape_traffic.txt
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d $OUT/pmc_sq -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1)
(cd tools && python pmc_sq.py $(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1) > $OUT/sq_counters.txt)
(cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA \
    SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv \
    -d $OUT/pmc_sq2 -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1)
(cd tools && python pmc_sq2.py $(find $OUT/pmc_sq2 -name "*counter_collection.csv" | head -1) > $OUT/sq_valu_mix.txt)
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum $OUT/pmc_sq $OUT/pmc_sq2
