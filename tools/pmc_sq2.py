"""tools/pmc_sq2.py — what the SIMDs' VALU port is busy with, per kernel family, from one more rocprofv3 --pmc pass of
bench.py (companion of tools/pmc_sq.py; 8 SQ counters per pass):

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA \
        SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv \
        -d gpurun_out/pmc_sq2 -o p -- python bench.py --steps 1 --warmup 1 --repeats 1 --no-cpu-baseline
    python tools/pmc_sq2.py gpurun_out/pmc_sq2/p_counter_collection.csv

Instruction counters are per wave-instruction; a transcendental (v_exp_f32 ...) holds the VALU port for 16 cycles =
4 quad-cycles, a full-rate VALU instruction for 1 (MI355X_MICROARCH.md, per-instruction constants), so
  trans share of VALU time ~ 4 * TRANS / (4 * TRANS + (VALU - TRANS - MFMA))     (MFMA instructions are counted in
SQ_INSTS_VALU but issue to the matrix pipe).  SQ_VALU_MFMA_COEXEC_CYCLES / SQ_WAVE_CYCLES is the measured overlap of
plain VALU work with the matrix pipe."""
from __future__ import annotations

import csv
import sys
from collections import defaultdict

from pmc_traffic import family

csv.field_size_limit(1 << 30)


def main():
    tot = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    with open(sys.argv[1], newline="") as f:
        for row in csv.DictReader(f):
            fam = family(row["Kernel_Name"])
            tot[fam][row["Counter_Name"]] += float(row["Counter_Value"])
            n[fam].add(row["Dispatch_Id"])
    hdr = ("family", "launches", "VALU/wave-cyc", "trans/VALU", "trans share", "cvt/VALU", "MFMA insts", "coexec/wave-cyc",
           "LDS insts/MFMA", "wait LDS")
    print("%-14s %8s %14s %11s %12s %9s %12s %16s %15s %9s" % hdr)
    for fam in ("gemm", "attn_spatial", "attn_temporal", "groupnorm", "layernorm", "other"):
        c = tot.get(fam)
        if not c:
            continue
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        valu, tr, mf = c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU_TRANS_F32"], c["SQ_INSTS_MFMA"]
        plain = max(valu - tr - mf, 0.0)
        print("%-14s %8d %14.3f %10.1f%% %11.1f%% %8.1f%% %12.3g %15.1f%% %15.2f %8.1f%%" % (
            fam, len(n[fam]), valu / wc, 100.0 * tr / max(valu, 1.0), 100.0 * 4 * tr / max(4 * tr + plain, 1.0),
            100.0 * c["SQ_INSTS_VALU_CVT"] / max(valu, 1.0), mf, 100.0 * c["SQ_VALU_MFMA_COEXEC_CYCLES"] / wc,
            c["SQ_INSTS_LDS"] / max(mf, 1.0), 100.0 * c["SQ_WAIT_INST_LDS"] / wc))


if __name__ == "__main__":
    main()
