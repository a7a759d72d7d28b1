"""Summarise rocprofv3 --pmc passes (one counter per pass, --output-format csv) per kernel.

    python tools/pmc_summary.py <dir_with_FETCH_SIZE_pass> <dir_with_WRITE_SIZE_pass> > profiles/rNN_pmc_per_kernel.json

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB; the factor 2 is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 128-byte requests as 64 bytes)."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                a = acc[(row["Kernel_Name"], row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    out = {}
    for d in sys.argv[1:]:
        for (kern, ctr), (tot, cnt) in collect(d).items():
            e = out.setdefault(kern, {})
            e[ctr + "_KB_avg"] = tot / cnt
            e["launches_" + ctr] = cnt
    for kern, e in out.items():
        if "FETCH_SIZE_KB_avg" in e and "WRITE_SIZE_KB_avg" in e:
            e["hbm_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024.0
    json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0.0) * kv[1].get("launches_FETCH_SIZE", 0))), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
