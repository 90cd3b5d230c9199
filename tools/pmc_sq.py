"""tools/pmc_sq.py — SQ-side utilisation per kernel family from one rocprofv3 --pmc pass of bench.py:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
        SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
        --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o p -- \
        python bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python tools/pmc_sq.py gpurun_out/pmc_sq/p_counter_collection.csv

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): busy cycles are summed
over the 4 SIMDs of the 256 CUs (MI355X_MICROARCH.md, per-instruction constants) and rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs (check: the attention kernel's 947 TF/s x 36/32 MFMAs issued
per useful 32 = 51 % of the matrix rate at the ~2.0 GHz it clocks; this formula gives 47 %).  SQ_WAVE_CYCLES and the
SQ_WAIT_* / SQ_ACTIVE_INST_* buckets count quad-cycles per wave and are reported as fractions of
SQ_WAVE_CYCLES (WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall)."""
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
    hdr = ("family", "launches", "MFMA util", "wait_any", "wait_inst", "active", "valu", "lds", "LDS conflict / wave-cyc")
    print("%-14s %8s %10s %9s %10s %8s %7s %7s %24s" % hdr)
    for fam in ("gemm", "attn_spatial", "attn_temporal", "groupnorm", "layernorm", "other"):
        c = tot.get(fam)
        if not c:
            continue
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        gui = c["GRBM_GUI_ACTIVE"] or 1.0
        print("%-14s %8d %9.1f%% %8.1f%% %9.1f%% %7.1f%% %6.1f%% %6.1f%% %23.2f%%" % (
            fam, len(n[fam]), 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * 1024.0),
            100.0 * c["SQ_WAIT_ANY"] / wc, 100.0 * c["SQ_WAIT_INST_ANY"] / wc,
            100.0 * c["SQ_ACTIVE_INST_ANY"] / wc, 100.0 * c["SQ_ACTIVE_INST_VALU"] / wc,
            100.0 * c["SQ_ACTIVE_INST_LDS"] / wc, 100.0 * c["SQ_LDS_BANK_CONFLICT"] / (4.0 * wc)))


if __name__ == "__main__":
    main()
