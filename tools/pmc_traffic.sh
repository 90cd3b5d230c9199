#!/bin/bash
# tools/pmc_traffic.sh TAG — the two PMC passes of bench.py that `roofline.traffic` quotes (run on the GPU
# box through gpurun; FETCH_SIZE and WRITE_SIZE cannot share a pass, and --pmc is never combined with
# the hip/hsa trace domains).  Writes gpurun_out/TAG_hbm_traffic.{txt,json}; copy them to
# profiles/TAG_hbm_traffic.* and profiles/hbm_traffic_latest.json.
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o p -- \
    python bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
f=$(find gpurun_out/pmc_FETCH_SIZE -name 'p_counter_collection.csv' | head -1)
w=$(find gpurun_out/pmc_WRITE_SIZE -name 'p_counter_collection.csv' | head -1)
python tools/pmc_traffic.py "$f" "$w" --json gpurun_out/${TAG}_hbm_traffic.json \
  --source "profiles/${TAG}_hbm_traffic.{txt,json}" | tee gpurun_out/${TAG}_hbm_traffic.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
