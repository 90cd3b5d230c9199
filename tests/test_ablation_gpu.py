"""GPU: kernel variants that exist in the ABLATION build of the library only (tools/libgcd_amd_ablate.so,
`python -m gcd_amd.csrc.build --ablation`; run with GCD_AMD_LIB pointing at it).  The product library refuses their
descriptors, so this module skips as a whole on a normal run — the records of experiments that lost their A/B
 stay testable without thirteen permanently skipped cases in the product suite."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

if "ablate" not in os.environ.get("GCD_AMD_LIB", ""):
    pytest.skip("ablation-only kernel variants: set GCD_AMD_LIB=tools/libgcd_amd_ablate.so", allow_module_level=True)

pytestmark = pytest.mark.gpu
TOL_F32 = 1e-4
TOL_F16 = 6e-4


def _h(t):
    return t.to(torch.float16).float()


def _gen(seed):
    return torch.Generator().manual_seed(seed)


@pytest.fixture(params=[0, 2, 3], ids=["auto", "pingpong", "ring"])
def gemm_impl(request):
    from gcd_amd import ops
    ops.tune_set(ops.TUNE_GEMM_IMPL, request.param)
    yield request.param
    ops.tune_set(ops.TUNE_GEMM_IMPL, 0)


@pytest.mark.parametrize("M,K,with_pos", [(700, 128, False), (1000, 1280, True), (256 * 200, 320, True)])
def test_gemm_fused_layernorm(gpu, gemm_impl, M, K, with_pos):
    """LayerNorm of the freshly written residual rows in the GEMM epilogue (N == 320, ping-pong kernel),
    incl. the x + time_pos_embed form that also returns the fp32 sum (video_attention.py:283-284)."""
    from gcd_amd import ops
    N = 320
    if not ops.gemm_ln_fusable(M, N, K):
        pytest.skip("shape / kernel choice does not take the fused LayerNorm")
    g = _gen(23)
    a = _h(torch.randn(M, K, generator=g))
    w = _h(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    r1 = torch.randn(M, N, generator=g) * 2 + 0.7          # residual with a non-zero mean
    gamma, beta = torch.randn(N, generator=g), torch.randn(N, generator=g)
    rows = 97
    pos = torch.randn((M + rows - 1) // rows, N, generator=g)
    x = a @ w.t() + bias + r1
    z = x + pos.repeat_interleave(rows, 0)[:M] if with_pos else x
    ref16 = F.layer_norm(z, (N,), gamma, beta, 1e-5)
    out = torch.empty(M, N, device=gpu)
    y16 = torch.empty(M, N, device=gpu, dtype=torch.float16)
    zsum = torch.empty(M, N, device=gpu)
    ln = dict(gamma=gamma.to(gpu), beta=beta.to(gpu), out16=y16)
    if with_pos:
        ln.update(addvec=pos.to(gpu), rows_per_vec=rows, sum_out=zsum)
    ops.gemm(a.half().to(gpu), w.half().to(gpu), out, M=M, bias=bias.to(gpu), r1=r1.to(gpu), ln=ln)
    torch.cuda.synchronize()
    assert rel_l2(out, x) < TOL_F32
    if with_pos:
        assert rel_l2(zsum, z) < TOL_F32
    e = rel_l2(y16.float(), ref16)
    assert e < TOL_F16, f"fused LayerNorm rel-L2 {e:.3e}"
