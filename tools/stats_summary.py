"""Condenses a `rocprofv3 --kernel-trace --stats --output-format csv` run into the text table kept under profiles/.

    python tools/stats_summary.py gpurun_out/prof_dir "header line 1" ["header line 2" ...] > profiles/rN_kernel_stats.txt
"""
import csv
import glob
import re
import sys


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for h in sys.argv[2:]:
        print("# " + h)
    print("# kernel | calls | total ms | avg us | %% of GPU kernel time   (total %.1f ms)" % (tot / 1e6))
    for r in rows:
        name = r["Name"].replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)[:84]
        pct = 100 * float(r["TotalDurationNs"]) / tot
        if pct < 0.05:
            continue
        print("%-84s | %6s | %9.3f | %9.1f | %5.1f" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, pct))


if __name__ == "__main__":
    main()
