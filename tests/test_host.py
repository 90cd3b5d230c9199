"""CPU: host logic of gcd_amd — plugin surface, weight packing, C-ABI surface (symbols only, no
compute without a GPU), sampler arithmetic of the generic path, workspace placement."""
import ctypes
import math
import re
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden"


# ------------------------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    from gcd_amd import _lib
    header = (ROOT / "include" / "gcd_amd.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(gcd_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()                       # raises if the .so has not been built
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gcd_abi_version() == _lib.ABI_VERSION == 9


def test_train_library_exports_every_declared_symbol():
    """libgcd_amd_train.so (the fine-tune step's own kernels) against include/gcd_amd_train.h; argument validation is
    observable without a GPU."""
    from gcd_amd import _lib
    header = (ROOT / "include" / "gcd_amd_train.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|int64_t|const char\*)\s+(gcd_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.TRAIN_SIGNATURES), (declared ^ set(_lib.TRAIN_SIGNATURES))
    lib = _lib.load_train()
    assert lib.gcd_train_abi_version() == _lib.TRAIN_ABI_VERSION
    assert lib.gcd_wgrad_tr_scratch_floats(43008, 320, 1280) == 64 * 320 * 1280      # 2 x 8 tiles of 160 -> 64 token slices
    assert lib.gcd_wgrad_tr_scratch_floats(10752, 640, 640) == 40 * 640 * 640        # 5 x 5 tiles of 128 -> 40 token slices
    assert lib.gcd_wgrad_tr_f16(16, 320, 16, 1280, 100, 321, 1280, 0, 16, 1280, 16, 1 << 30, None) != 0
    assert b"multiples of 8" in lib.gcd_train_last_error()


def test_gemm_desc_layout_matches_header(tmp_path):
    """ctypes mirror of gcd_gemm_desc has the C struct's offsets and size (checked with gcc)."""
    import shutil
    import subprocess
    from gcd_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    fields = [f[0] for f in _lib.GemmDesc._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "gcd_amd.h"}"',
           'int main(void) {']
    src += [f'  printf("{n} %zu\\n", offsetof(gcd_gemm_desc, {n}));' for n in fields]
    src += ['  printf("sizeof %zu\\n", sizeof(gcd_gemm_desc));', '  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(c)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for n in fields:
        assert int(out[n]) == getattr(_lib.GemmDesc, n).offset, n
    assert int(out["sizeof"]) == ctypes.sizeof(_lib.GemmDesc)


@pytest.mark.parametrize("cname,pyname", [("gcd_pack_entry", "PackEntry"), ("gcd_smallm_problem", "SmallmProblem")])
def test_train_table_structs_match_header(tmp_path, cname, pyname):
    """ctypes mirrors of the device-table structs of include/gcd_amd_train.h (the multi-tensor weight pack, the grouped
    few-row Linears) have the C structs' offsets and sizes (checked with gcc)."""
    import shutil
    import subprocess
    from gcd_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    cls = getattr(_lib, pyname)
    fields = [f[0] for f in cls._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "gcd_amd_train.h"}"',
           'int main(void) {']
    src += [f'  printf("{n} %zu\\n", offsetof({cname}, {n}));' for n in fields]
    src += [f'  printf("sizeof %zu\\n", sizeof({cname}));', '  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(c)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for n in fields:
        assert int(out[n]) == getattr(cls, n).offset, n
    assert int(out["sizeof"]) == ctypes.sizeof(cls)


def test_ops_refuse_cpu_tensors_and_bad_args():
    from gcd_amd import _lib, ops
    a = torch.zeros(64, 64, dtype=torch.float16)
    with pytest.raises(_lib.GcdError, match="no CPU fallback"):
        ops.gemm(a, a, torch.zeros(64, 64), M=64)
    # argument validation happens before any launch, so it is observable without a GPU
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.gcd_gemm_f16(ctypes.byref(d), None) != 0
    assert b"null operand" in lib.gcd_last_error()
    d.A = d.W = d.out = 16
    d.M, d.N, d.K, d.lda, d.ldo = 10, 32, 100, 104, 32
    assert lib.gcd_gemm_f16(ctypes.byref(d), None) != 0
    assert b"multiple of 32" in lib.gcd_last_error()
    assert lib.gcd_attn_temporal_f16(16, 192, 16, 64, 1, 17, 4, 1, None) != 0
    assert b"T=17" in lib.gcd_last_error()
    assert lib.gcd_groupnorm_stats(16, 48, 48, None, 0, 0, 10, 10, 1e-5, 16, 1, 16, None) != 0


# ------------------------------------------------------------------------------------ plugin surface
def test_state_dict_keys_match_reference_fixture():
    """Key set + shapes of the drop-in VideoUNet == those of the reference class (captured in the
    golden fixture from the reference itself)."""
    from gcd_amd.video_model import VideoUNet
    from oracle import svd_unet_ref as O
    g = torch.load(GOLD / "unet_tiny.pt")
    with torch.device("meta"):
        net = VideoUNet(**O.TINY.as_reference_kwargs())
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == g["state_dict_shapes"]
    assert list(mine) == list(g["state_dict_shapes"])
    with torch.device("meta"):
        full = VideoUNet(**O.KUBRIC.as_reference_kwargs())
    sd = full.state_dict()
    assert len(sd) == 1432 and sum(v.numel() for v in full.parameters()) == 1_526_427_882
    # names quoted in SURVEY.md §5 (checkpoint contract)
    assert tuple(sd["input_blocks.1.0.time_stack.in_layers.2.weight"].shape) == (320, 320, 3, 1, 1)
    assert tuple(sd["input_blocks.1.1.time_stack.0.ff_in.net.0.proj.weight"].shape) == (2560, 320)
    assert tuple(sd["input_blocks.1.1.time_mixer.mix_factor"].shape) == (1,)


def test_default_init_is_zero_module_like_reference():
    from gcd_amd.video_model import VideoUNet
    from oracle import svd_unet_ref as O
    net = VideoUNet(**O.TINY.as_reference_kwargs())
    zeros = [k for k, v in net.state_dict().items() if k.endswith(".weight") and v.dim() >= 2
             and float(v.abs().max()) == 0.0]
    assert len(zeros) == 61          # 44 ResBlock out convs + 16 proj_out + final conv


def test_unsupported_configs_and_cpu_forward_raise():
    from gcd_amd import _lib
    from gcd_amd.video_model import VideoUNet
    from oracle import svd_unet_ref as O
    kw = O.TINY.as_reference_kwargs()
    with pytest.raises(NotImplementedError, match="use_scale_shift_norm"):
        VideoUNet(**dict(kw, use_scale_shift_norm=True))
    with pytest.raises(AssertionError):
        VideoUNet(**dict(kw, context_dim=None))
    net = VideoUNet(**kw)
    x = torch.zeros(4, 8, 8, 8)
    with pytest.raises(_lib.GcdError, match="no CPU"):
        net(x, torch.zeros(4), context=torch.zeros(4, 1, 64), y=torch.zeros(4, 128),
            num_video_frames=2, image_only_indicator=torch.zeros(2, 2))


def test_instantiate_from_config_targets():
    from gcd_amd.util import append_dims, instantiate_from_config
    s = instantiate_from_config({
        "target": "gcd_amd.sampling.EulerEDMSampler",
        "params": {"num_steps": 25, "device": "cpu",
                   "discretization_config": {"target": "gcd_amd.discretizer.EDMDiscretization",
                                             "params": {"sigma_max": 700.0}},
                   "guider_config": {"target": "gcd_amd.guiders.LinearPredictionGuider",
                                     "params": {"num_frames": 14, "max_scale": 1.5, "min_scale": 1.0}}}})
    assert s.num_steps == 25 and s.guider.num_frames == 14 and s.guider.max_scale == 1.5
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})
    assert instantiate_from_config("__is_unconditional__") is None
    assert append_dims(torch.zeros(3), 4).shape == (3, 1, 1, 1)
    with pytest.raises(ValueError):
        append_dims(torch.zeros(3, 1), 1)


# ------------------------------------------------------------------------------------ sampler pieces
def test_discretizer_guider_scaling_known_answers():
    from gcd_amd.denoiser_scaling import VScalingWithEDMcNoise
    from gcd_amd.discretizer import EDMDiscretization
    from gcd_amd.guiders import LinearPredictionGuider
    k = torch.load(GOLD / "kat.pt")
    sig = EDMDiscretization(sigma_max=700.0)(25)
    assert torch.equal(sig, k["sigmas_25_700"])
    assert torch.equal(EDMDiscretization(sigma_max=700.0)(25, flip=True), torch.flip(sig, (0,)))
    assert len(EDMDiscretization()(10, do_append_zero=False)) == 10
    g = LinearPredictionGuider(max_scale=1.5, num_frames=14)
    assert torch.equal(g.scale, k["guider_scale_14"])
    sc = torch.stack(VScalingWithEDMcNoise()(k["scaling_sigma"]))
    assert torch.allclose(sc, k["scaling"], rtol=1e-6, atol=0)


def test_generic_sampler_path_matches_reference_golden_with_oracle_network():
    """The drop-in sampler / guider / denoiser / wrapper classes (generic torch path, CPU) driving
    the ORACLE network reproduce the reference's 5-step trajectory: pins the host arithmetic."""
    from gcd_amd.denoiser import Denoiser
    from gcd_amd.sampling import EulerEDMSampler
    from gcd_amd.wrappers import OpenAIWrapper
    from oracle import svd_unet_ref as O, weights
    g = torch.load(GOLD / "sampler_tiny.pt")
    gu = torch.load(GOLD / "unet_tiny.pt")
    sd = weights.synth_state_dict(gu["state_dict_shapes"])
    T = g["T"]

    class OracleNet(torch.nn.Module):
        def forward(self, x, timesteps=None, context=None, y=None, num_video_frames=None,
                    image_only_indicator=None):
            return O.unet_forward(sd, O.TINY, x, timesteps, context, y, num_video_frames,
                                  image_only_indicator)

    noise, c, uc = weights.synth_inputs(1, T, g["h"], g["w"], O.TINY.context_dim,
                                        O.TINY.adm_in_channels + O.TINY.aux_emb_dim, g["input_seed"])
    sampler = EulerEDMSampler(
        discretization_config={"target": "gcd_amd.discretizer.EDMDiscretization",
                               "params": {"sigma_max": 700.0}},
        num_steps=g["steps"],
        guider_config={"target": "gcd_amd.guiders.LinearPredictionGuider",
                       "params": {"num_frames": T, "max_scale": 1.5, "min_scale": 1.0}},
        device="cpu")
    den = Denoiser({"target": "gcd_amd.denoiser_scaling.VScalingWithEDMcNoise"})
    model = OpenAIWrapper(OracleNet())
    extra = {"num_video_frames": T, "image_only_indicator": torch.zeros(2, T)}
    x_in = noise.clone()
    with torch.no_grad():
        out = sampler(lambda i, s, cc: den(model, i, s, cc, **extra), x_in, cond=c, uc=uc)
    assert sampler.last_path == "generic"
    assert rel_l2(out, g["final"]) < 2e-5
    # the caller's tensor is scaled in place, as the reference does (sampling.py:54)
    assert torch.allclose(x_in, noise * math.sqrt(1 + 700.0001220703125 ** 2), rtol=1e-6)


# ------------------------------------------------------------------------------------ packing
def test_pack_conv3x3_k_order_matches_im2col():
    from gcd_amd import packing
    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 5, 3, 3, generator=g)
    x = torch.randn(2, 5, 4, 4, generator=g)
    wp = packing.pack_conv3x3(w).float()
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1)                      # n h w c
    cols = torch.stack([xp[:, dy:dy + 4, dx:dx + 4, :] for dy in range(3) for dx in range(3)], 3)
    out = cols.reshape(2, 4, 4, 45) @ wp.t()
    ref = F.conv2d(x, w.half().float(), padding=1).permute(0, 2, 3, 1)
    assert torch.allclose(out, ref, atol=1e-4)
    pad = packing.pack_conv3x3(w, cin_pad=8, cout_pad=16)
    assert pad.shape == (16, 72) and float(pad[6:].abs().max()) == 0
    assert torch.equal(pad.reshape(16, 9, 8)[:6, :, :5].reshape(6, 45), packing.pack_conv3x3(w))


def test_pack_temporal_qkv_geglu():
    from gcd_amd import packing
    g = torch.Generator().manual_seed(1)
    w = torch.randn(4, 3, 3, 1, 1, generator=g)
    x = torch.randn(1, 3, 5, 2, 1, generator=g)                           # b c t h w
    wp = packing.pack_conv_t3(w).float()
    xp = F.pad(x, (0, 0, 0, 0, 1, 1))[0, :, :, :, 0].permute(1, 2, 0)     # t h c
    cols = torch.cat([xp[dt:dt + 5] for dt in range(3)], -1)              # t h (kt c)
    ref = F.conv3d(x, w.half().float(), padding=(1, 0, 0))[0, :, :, :, 0].permute(1, 2, 0)
    assert torch.allclose(cols @ wp.t(), ref, atol=1e-4)
    q, k, v = (torch.randn(8, 8, generator=g) for _ in range(3))
    assert torch.equal(packing.pack_qkv(q, k, v), torch.cat([q, k, v]).half())
    inner = 32
    wg, bg = torch.randn(2 * inner, 8, generator=g), torch.randn(2 * inner, generator=g)
    wpk, bpk = packing.pack_geglu(wg, bg)
    order = packing.geglu_row_order(inner)
    assert sorted(order.tolist()) == list(range(2 * inner))
    h = torch.randn(3, 8, generator=g) @ wpk.float().t() + bpk
    val = h.reshape(3, inner // 16, 2, 16)[:, :, 0].reshape(3, inner)
    gate = h.reshape(3, inner // 16, 2, 16)[:, :, 1].reshape(3, inner)
    x3 = (h - bpk) @ torch.linalg.pinv(wpk.float().t())                   # recover inputs
    full = x3 @ wg.half().float().t() + bg
    assert torch.allclose(val, full[:, :inner], atol=1e-3)
    assert torch.allclose(gate, full[:, inner:], atol=1e-3)


# ------------------------------------------------------------------------------------ workspace
def test_workspace_replays_identical_placement():
    from gcd_amd.engine import Workspace
    ws = Workspace(torch.device("cpu"))

    def run():
        ws.reset(("sig", 1))
        a = ws.alloc((100, 64), torch.float32)
        b = ws.alloc((50, 64), torch.float16)
        ws.release(a)
        c = ws.alloc((10, 10), torch.float64)      # fits in a's slab
        d = ws.alloc((100, 64), torch.float32)     # needs a new slab on the first run
        ptrs = [t.data_ptr() for t in (a, b, c, d)]
        ws.release(b, c, d)
        ws.finish()
        return ptrs

    first, second, third = run(), run(), run()
    assert first == second == third
    assert first[2] == first[0] and first[3] != first[0]
    n = ws.nbytes()
    ws.reset(("other", 2))                           # new signature: re-plan, may grow
    ws.alloc((1000, 64), torch.float32)
    assert ws.nbytes() > n


def test_pack_qkv_folds_the_softmax_scale_in_fp32():
    """pack_qkv(q_scale): W_q * scale is formed in fp32 and rounded to fp16 ONCE (not fp16(W_q) * scale
    re-rounded), k / v rows are untouched."""
    from gcd_amd import ops, packing
    g = torch.Generator().manual_seed(3)
    wq, wk, wv = (torch.randn(64, 64, generator=g) for _ in range(3))
    s = ops.ATTN_Q_SCALE_LOG2
    assert abs(s - math.log2(math.e) / 8) < 1e-12
    p = packing.pack_qkv(wq, wk, wv, q_scale=s)
    assert p.dtype == torch.float16 and p.shape == (192, 64)
    assert torch.equal(p[:64], (wq * s).half())
    assert torch.equal(p[64:], torch.cat([wk, wv]).half())
    assert torch.equal(packing.pack_qkv(wq, wk, wv), torch.cat([wq, wk, wv]).half())


def test_fused_feedforward_generated_code_keeps_its_hazard_distances():
    """The one-kernel FeedForward issues its MFMAs as inline asm (gcd_amd/csrc/ff_fused_kernel.h), so hipcc pads none of their
    hazards and counts none of the asm loads; tools/ff_isa_audit.py checks the gfx950 code hipcc actually generates — no spill
    or accumulator shuttling in the steady-state iteration, no VALU write of an MFMA operand right in front of the MFMA, no
    VALU read of an MFMA result right behind it, no compiler-inserted counted wait inside the first iteration.  (hipcc
    cross-compiles without a GPU: ~40 s.)"""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "ff_isa_audit.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("0 finding(s)") == 6, r.stdout          # three epilogue forms x (plain, LayerNorm) the library carries


def test_round6_kernels_compile_without_spills():
    """lnqkv.hip, conv_narrow.hip and the wide fp32 Linear hold their operands in registers across long unrolled loops; hipcc's
    scheduling of them is fragile (left alone it hoisted the 40 LayerNorm-affine reads of lnqkv_kernel and spilled 250
    registers into the hot prologue — the source pins the pass order).  A spill there is a silent 2x: check the code
    object metadata of what the library is built from.  (hipcc cross-compiles without a GPU: ~40 s.)"""
    import re
    import subprocess
    import tempfile
    from pathlib import Path
    from gcd_amd.csrc import build as B
    wanted = {"lnqkv.hip": ("lnqkv_kernel",), "conv_narrow.hip": ("conv3x3_narrow_kernel",),
              "elementwise.hip": ("linear_smallm_mfma_kernel",)}
    with tempfile.TemporaryDirectory() as td:
        for src, kernels in wanted.items():
            out = Path(td) / (src + ".s")
            subprocess.check_call([B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S",
                                   str(B.CSRC / src), "-o", str(out)], stderr=subprocess.DEVNULL)
            text = out.read_text()
            found = 0
            for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
                if any(k in m.group(1) for k in kernels):
                    found += 1
                    assert int(m.group(2)) == 0, f"{m.group(1)} spills {m.group(2)} registers"
            assert found >= 1, f"no kernel of {kernels} found in {src}"
