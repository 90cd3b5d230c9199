"""tools/pmc_gemm_shapes.py — fabric-side traffic PER GEMM SHAPE of one sampler step, from the same rocprofv3 --pmc
passes of bench.py that tools/pmc_traffic.py sums per kernel family (FETCH_SIZE, WRITE_SIZE, and optionally
TCC_HIT_sum TCC_MISS_sum) plus the per-launch table of the instrumented step (`bench.py --dump-profile`):

    for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${C%% *} -o p -- \
          python bench.py --steps 1 --warmup 1 --no-cpu-baseline --dump-profile gpurun_out/launches.json
    done
    python tools/pmc_gemm_shapes.py gpurun_out/launches.json gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv \
        gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv [gpurun_out/pmc_TCC_HIT_sum/p_counter_collection.csv]

The GEMM-family dispatches of the LAST step of the run (the instrumented eager step) are matched, in launch order,
to the `gemm` records of the launch table (a split-K call is two dispatches: the K-slice kernel + the reduce).
Units as in pmc_traffic.py: counters in KB, FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide
coalesced reads at 64 B).  'alg read' = the A operand once + the weights once + the fp32 residual(s); 'alg write' =
the output once."""
from __future__ import annotations

import csv
import json
import sys
from collections import OrderedDict, defaultdict

csv.field_size_limit(1 << 30)
GEMM_KEYS = ("gemm_p8_kernel", "gemm_p8x_kernel", "gemm_pp_kernel", "gemm_f16_kernel", "splitk_reduce_kernel",
             "ff_fused_kernel", "lnqkv_kernel", "conv3x3_narrow_kernel")      # = pmc_traffic.FAMILIES["gemm"]


def dispatches(path):
    """[(dispatch id, kernel name, {counter: value})] of the GEMM family, in dispatch order."""
    rows = OrderedDict()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if not any(k in r["Kernel_Name"] for k in GEMM_KEYS):
                continue
            d = rows.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], defaultdict(float)])
            d[1][r["Counter_Name"]] += float(r["Counter_Value"])
    return [(k, v[0], v[1]) for k, v in sorted(rows.items())]


def per_call(disp, ncalls):
    """Fold the last step's dispatches into `ncalls` GEMM calls (split-K = K-slice kernel + reduce kernel)."""
    calls = []
    for _, name, c in disp:
        if "splitk_reduce_kernel" in name and calls:
            for k, v in c.items():
                calls[-1][1][k] += v
            calls[-1][0] += " + reduce"
        else:
            calls.append([name, defaultdict(float, c)])
    assert len(calls) % ncalls == 0, f"{len(calls)} GEMM calls in the trace, {ncalls} per step"
    return calls[-ncalls:]


def main():
    launches = [r for r in json.load(open(sys.argv[1])) if r["kind"] == "gemm"]
    fetch = per_call(dispatches(sys.argv[2]), len(launches))
    write = per_call(dispatches(sys.argv[3]), len(launches))
    tcc = per_call(dispatches(sys.argv[4]), len(launches)) if len(sys.argv) > 4 else None
    groups = OrderedDict()
    for i, r in enumerate(launches):
        key = (r["M"], r["N"], r["K"], r["mode"], r.get("out_kind", 0), r.get("nres", 0), r.get("stride", 1), r.get("up", 0))
        g = groups.setdefault(key, dict(n=0, fetch=0.0, write=0.0, hit=0.0, miss=0.0, ms=0.0, kernel=fetch[i][0], rec=r))
        g["n"] += 1
        g["fetch"] += fetch[i][1]["FETCH_SIZE"] * 1024 / 1e6
        g["write"] += write[i][1]["WRITE_SIZE"] * 1024 / 1e6
        g["ms"] += r["ms"]
        if tcc:
            g["hit"] += tcc[i][1]["TCC_HIT_sum"]
            g["miss"] += tcc[i][1]["TCC_MISS_sum"]
    print("# per launch, one EulerEDM step at 14x72x128 (tools/pmc_gemm_shapes.py); mode 0 plain / 1 conv3x3 / 2 (3,1,1); "
          "out 0 fp32 / 1 fp16 / 2 GEGLU")
    print(f"{'M':>7s} {'N':>6s} {'K':>6s} mode out res {'calls':>5s} {'us':>8s} {'TF/s':>7s} {'FETCHx2 MB':>11s} {'alg read':>9s} "
          f"{'ratio':>6s} {'WRITE MB':>9s} {'alg write':>9s} {'L2 hit':>7s}  kernel")
    tot = dict(f=0.0, w=0.0, ar=0.0, aw=0.0)
    for (M, N, K, mode, ok, nres, stride, up), g in groups.items():
        n = g["n"]
        cin = g["rec"].get("cin", K)
        rows_in = M * (stride * stride) // (4 if up else 1) if mode == 1 else M
        a_bytes = rows_in * cin * 2 if mode != 0 else M * K * 2
        alg_r = (a_bytes + N * K * 2 + nres * M * N * 4) / 1e6
        if g["rec"].get("fused_ff"):      # LayerNorm + FeedForward in one launch: the fp32 rows once, the weights, a second stream
            alg_r = (M * 320 * 4 + 320 * 3840 * 2 + (nres - 1) * M * 320 * 4) / 1e6
        elif g["rec"].get("fused_ln"):    # LayerNorm + q|k|v: the fp32 rows once, the weights
            alg_r = (M * 320 * 4 + N * 320 * 2) / 1e6
        alg_w = M * (N // 2 if ok == 2 else N) * (4 if ok == 0 else 2) / 1e6
        f2, w = 2 * g["fetch"] / n, g["write"] / n
        us = g["ms"] / n * 1e3
        hit = g["hit"] / max(g["hit"] + g["miss"], 1.0) if tcc else float("nan")
        kern = " + ".join(k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                          for k in g["kernel"].split(" + "))[:56]
        print(f"{M:7d} {N:6d} {K:6d} {mode:4d} {ok:3d} {nres:3d} {n:5d} {us:8.1f} {2.0 * M * N * K / us / 1e6:7.0f} {f2:11.1f} "
              f"{alg_r:9.1f} {f2 / alg_r:6.2f} {w:9.1f} {alg_w:9.1f} {hit:7.3f}  {kern}")
        tot["f"] += f2 * n
        tot["w"] += w * n
        tot["ar"] += alg_r * n
        tot["aw"] += alg_w * n
    print(f"# step total: FETCHx2 {tot['f'] / 1e3:.1f} GB vs algorithmic reads {tot['ar'] / 1e3:.1f} GB ({tot['f'] / tot['ar']:.2f}x); "
          f"WRITE {tot['w'] / 1e3:.1f} GB vs {tot['aw'] / 1e3:.1f} GB; {len(launches)} GEMM calls")


if __name__ == "__main__":
    main()
