"""tools/pmc_traffic.py — HBM-side bytes per kernel family from two rocprofv3 --pmc passes of bench.py
(FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, TCC counter budget):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o p -- \
        python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    (same with WRITE_SIZE)
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv \
        gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv --json profiles/r01v_hbm_traffic.json

Counter units are KB.  gfx950 correction (same guide, HBM section): FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is taken as reported (it matches
the algorithmic output bytes of the GEMM shapes, profiles/r01k_gemm_hbm_traffic.txt).  Infinity-Cache
hits are included in both (they are fabric-side counters).
"""
from __future__ import annotations

import argparse
import csv
import json
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)

FAMILIES = [
    # what bench.py times as the GEMM family (ops._Timed("gemm")): the tile kernels and, from round 6, the one-launch
    # LayerNorm + FeedForward, LayerNorm + q|k|v and the output head
    ("gemm", ("gemm_p8_kernel", "gemm_p8x_kernel", "gemm_pp_kernel", "gemm_f16_kernel", "splitk_reduce_kernel",
              "ff_fused_kernel", "lnqkv_kernel", "conv3x3_narrow_kernel")),
    ("attn_spatial", ("attn_spatial",)),
    ("attn_temporal", ("attn_temporal",)),
    ("groupnorm", ("gn_stats", "gn_apply")),
    ("layernorm", ("layernorm",)),
]


def family(name: str) -> str:
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def load(path: str):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            fam = family(row["Kernel_Name"])
            tot[fam] += float(row["Counter_Value"])
            cnt[fam] += 1
    return tot, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--json", default="")
    ap.add_argument("--source", default="", help="name of the profile this JSON was made from (recorded)")
    a = ap.parse_args()
    f_tot, f_cnt = load(a.fetch_csv)
    w_tot, w_cnt = load(a.write_csv)
    out = {}
    print(f"{'family':14s} {'launches':>8s} {'FETCH MB':>10s} {'x2 MB':>10s} {'WRITE MB':>10s} {'MB/launch':>10s}")
    for fam in [f for f, _ in FAMILIES] + ["other"]:
        n = f_cnt.get(fam, 0)
        if not n:
            continue
        assert w_cnt.get(fam, 0) == n, (fam, n, w_cnt.get(fam))
        fetch_mb = f_tot[fam] * 1024 / 1e6
        write_mb = w_tot[fam] * 1024 / 1e6
        per = (2 * fetch_mb + write_mb) / n
        print(f"{fam:14s} {n:8d} {fetch_mb:10.1f} {2 * fetch_mb:10.1f} {write_mb:10.1f} {per:10.2f}")
        out[fam] = dict(launches=n, fetch_mb_raw=round(fetch_mb, 1), fetch_mb_x2=round(2 * fetch_mb, 1),
                        write_mb=round(write_mb, 1), bytes_per_launch=round(per * 1e6))
    if a.json:
        # stamp with the digest of the sources the counters were collected on (bench.py refuses to
        # quote a profile whose digest differs from the build it runs)
        import pathlib
        sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
        from gcd_amd.csrc import build as _b
        out["sources_digest"] = _b.sources_digest()
        out["source"] = a.source or a.json
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
