"""Condenses two `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv` passes of the same command
(separate passes, as MI355X_MICROARCH.md prescribes) into the table + json kept under profiles/ (bench.py reads the json
for `roofline.traffic`).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.txt> <out.json> "header line" ...
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    f = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True))[0]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


def main():
    fetch, n_f = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, n_w = per_kernel(sys.argv[2], "WRITE_SIZE")
    rows = sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0.0)))
    with open(sys.argv[3], "w") as o:
        for h in sys.argv[5:]:
            o.write("# " + h + "\n")
        o.write("# kernel | launches | FETCH_SIZE KiB / launch (raw) | WRITE_SIZE KiB / launch (raw)\n")
        shown = rows[:28] + [k for k in rows[28:] if "loss_" in k or "upsample" in k]   # the loss gather's kernels always
        for k in shown:
            o.write("%-78s | %5d | %12.1f | %12.1f\n" % (k[:78], n_f[k], fetch[k] / n_f[k], write.get(k, 0.0) / max(n_w.get(k, 0), 1)))
    # the gather-GEMM launches (forward + dgrad) = what bench.py's `roofline` covers: the hl32 kernel of the wide layers and the
    # fp32-operand kernel of the others
    gemm = [k for k in fetch if "conv_gemm_f16_kernel" in k or "conv_gemm_hl_kernel" in k]
    hl = [k for k in gemm if "conv_gemm_hl_kernel" in k]
    launches = sum(n_f[k] for k in gemm)
    f_raw = sum(fetch[k] for k in gemm) / launches
    w_raw = sum(write.get(k, 0.0) for k in gemm) / launches
    rec = {"workload": os.environ.get("DCN_PMC_WORKLOAD", "config2"), "conv_mode": "f16x3", "forward_calls": "pair",
           "kernel": "conv_gemm_hl_kernel + conv_gemm_f16_kernel (all variants)", "launches": launches,
           "fetch_kib_per_launch_raw": f_raw, "write_kib_per_launch_raw": w_raw,
           "hbm_bytes_per_launch": 1024.0 * (2.0 * f_raw + w_raw),
           "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE "
                         "raw (uncalibrated); fabric-side counters, Infinity-Cache hits included"}
    if hl:
        n = sum(n_f[k] for k in hl)
        fh, wh = sum(fetch[k] for k in hl) / n, sum(write.get(k, 0.0) for k in hl) / n
        rec["hl_kernel"] = {"launches": n, "fetch_kib_per_launch_raw": fh, "write_kib_per_launch_raw": wh,
                            "hbm_bytes_per_launch": 1024.0 * (2.0 * fh + wh)}
    json.dump(rec, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
