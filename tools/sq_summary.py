"""Per-kernel averages of SQ / GRBM counters from `rocprofv3 --kernel-trace --pmc ... --output-format csv` passes
(one directory per pass; counters of all passes are merged by kernel name).

    python tools/sq_summary.py <out.txt> "header" <pass_dir> [<pass_dir> ...]
"""
import collections
import csv
import glob
import re
import sys


def main():
    out, header, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.Counter())
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
                tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[name][r["Counter_Name"]] += 1
        for f in sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out, "w") as o:
        o.write("# " + header + "\n")
        for name in sorted(tot, key=lambda k: -sum(dur.get(k, [0]))):
            if "wgrad_f16" not in name and "conv_gemm_f16" not in name and "gemm_hl" not in name and "wgrad_hl" not in name:
                continue
            d = dur.get(name, [])
            o.write("%s   (%d launches, %.1f us average while counting)\n" % (name, len(d), sum(d) / max(len(d), 1)))
            c = {k: tot[name][k] / cnt[name][k] for k in tot[name]}
            for k in sorted(c):
                o.write("    %-28s %16.0f per launch\n" % (k, c[k]))
            if "GRBM_GUI_ACTIVE" in c and d:
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0   # summed over the 8 XCDs
                o.write("    -> %.0f cycles per launch = %.2f GHz\n" % (cyc, cyc / (sum(d) / len(d)) / 1e3))
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                    o.write("    -> MFMA busy %.1f %% of the kernel's cycles (per SIMD: / 1024)\n" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc))
                if "SQ_LDS_IDX_ACTIVE" in c:
                    o.write("    -> LDS active %.1f %% (per CU: / 256)\n" % (100.0 * c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc))
            if "SQ_WAVE_CYCLES" in c:
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                    if k in c:
                        o.write("    -> %-20s %.1f %% of wave cycles\n" % (k, 100.0 * c[k] / c["SQ_WAVE_CYCLES"]))


if __name__ == "__main__":
    main()
