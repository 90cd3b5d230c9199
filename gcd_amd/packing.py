"""Weight re-layout from the reference's torch parameter shapes to the fp16 operands the HIP GEMM
consumes.  Pure tensor reshapes — they run once per `load_state_dict` / device move, on whatever
device the parameters live on (CPU is fine, so these are unit-testable without a GPU).

GEMM weight convention: W16[N, K] row-major with K ordered the way the kernel's A-gather walks it.
"""
from __future__ import annotations

import torch


def pack_linear(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight [N, K] -> fp16 [N, K]."""
    return w.detach().to(torch.float16).contiguous()


def pack_conv3x3(w: torch.Tensor, cin_pad: int = 0, cout_pad: int = 0,
                 dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """nn.Conv2d weight [Cout, Cin, 3, 3] -> fp16 [Cout(+pad), 9*Cin(+pad)], K order (kh, kw, cin).

    cin_pad / cout_pad (absolute sizes) zero-pad the input channels to the 64-channel K granule and
    the output channels to the 16-row N granule (first and last conv of the UNet).
    """
    cout, cin, kh, kw = w.shape
    assert (kh, kw) == (3, 3)
    cin_p = max(cin, cin_pad)
    cout_p = max(cout, cout_pad)
    out = torch.zeros(cout_p, 3, 3, cin_p, dtype=dtype, device=w.device)
    out[:cout, :, :, :cin] = w.detach().permute(0, 2, 3, 1).to(dtype)
    return out.reshape(cout_p, 9 * cin_p).contiguous()


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d 1x1 weight [Cout, Cin, 1, 1] -> fp16 [Cout, Cin]."""
    return w.detach().reshape(w.shape[0], w.shape[1]).to(torch.float16).contiguous()


def pack_conv_t3(w: torch.Tensor, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """nn.Conv3d (3,1,1) weight [Cout, Cin, 3, 1, 1] -> fp16 [Cout, 3*Cin], K order (kt, cin)."""
    cout, cin, kt, kh, kw = w.shape
    assert (kt, kh, kw) == (3, 1, 1)
    return w.detach().reshape(cout, cin, 3).permute(0, 2, 1).reshape(cout, 3 * cin).to(dtype).contiguous()


def pack_qkv(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, q_scale: float = 1.0) -> torch.Tensor:
    """to_q / to_k / to_v weights [C, C] -> one fp16 [3C, C] so q|k|v come out of a single GEMM.

    q_scale (spatial attention: log2(e)/sqrt(d)) is multiplied into W_q in fp32 BEFORE the single
    fp16 rounding, so the attention kernel receives scores that are already exp2 arguments at no
    loss of precision (gcd_attn_spatial_f16, q_prescaled = 1)."""
    wq32 = wq.detach().to(torch.float32)
    if q_scale != 1.0:
        wq32 = wq32 * q_scale
    return torch.cat([wq32, wk.detach().to(torch.float32), wv.detach().to(torch.float32)], 0).to(
        torch.float16).contiguous()


def geglu_row_order(inner: int, device=None) -> torch.Tensor:
    """Row permutation for GEGLU.proj ([2*inner, C]): 16 value rows then their 16 gate rows.

    Packed row 32*j + i (i < 16) is value row 16*j + i; packed row 32*j + 16 + i is gate row
    inner + 16*j + i.  The GEMM epilogue then finds a*gelu(g) operands in the same lane.
    """
    assert inner % 16 == 0
    j = torch.arange(inner // 16, device=device).repeat_interleave(32)
    i = torch.arange(32, device=device).repeat(inner // 16)
    return torch.where(i < 16, 16 * j + i, inner + 16 * j + (i - 16))


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    """GEGLU.proj weight [2*inner, C], bias [2*inner] -> (fp16 W permuted, fp32 bias permuted)."""
    inner = w.shape[0] // 2
    order = geglu_row_order(inner, w.device)
    return (w.detach()[order].to(torch.float16).contiguous(),
            b.detach()[order].to(torch.float32).contiguous())
