set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "walk or reverse" 2>&1 | tail -5 > gpurun_out/r04m/tests_new.txt
timeout 300 python -m pytest tests/test_unet_gpu.py -q -x -k "walk or forward_vs_reference_golden or image_only" 2>&1 | tail -5 >> gpurun_out/r04m/tests_new.txt
cat gpurun_out/r04m/tests_new.txt
timeout 420 bash tools/ab_sweep.sh gpurun_out/r04m/ab_sweep.txt 2
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r04m/tests_kernels.txt
cat gpurun_out/r04m/tests_kernels.txt
