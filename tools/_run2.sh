set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -5 > $O/tests_kernels.txt
cat $O/tests_kernels.txt
timeout 400 python -m pytest tests/test_backward_gpu.py -q -x -k "groupnorm or fused_node or resblock or full_width or layernorm" 2>&1 | tail -5 > $O/tests_backward.txt
cat $O/tests_backward.txt
AB_VARIANTS="base:GCD_ZIGZAG=0 GCD_ATTN_IMPL=16;tmfma:GCD_ZIGZAG=0;tmfma_zz2:GCD_ZIGZAG=2" timeout 400 bash tools/ab_sweep.sh $O/ab_sweep.txt 3
for mt in 192 128 84 48; do
  echo "== GCD_PP_MIN_TILES=$mt" >> $O/train_ab.txt
  GCD_PP_MIN_TILES=$mt timeout 200 python tools/train_step_bench.py --steps 3 2>/dev/null | tail -1 >> $O/train_ab.txt
done
cat $O/train_ab.txt
