"""Summarise rocprofv3 --pmc passes (one counter per pass, --output-format csv) per kernel AND per launch shape.

    python tools/pmc_summary.py <dir_with_FETCH_SIZE_pass> <dir_with_WRITE_SIZE_pass> > profiles/rNN_pmc_per_kernel.json

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KB; the factor 2 is the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 128-byte requests as 64 bytes; confirmed on pure streams of
4 / 8 / 16 B per lane by tools/gpu/pmc_calib.hip, profiles/r06a_pmc_calibration.json).
A process launches the same kernel on meshes of different sizes (grid-sequencing levels of the primal, the section, the wing): the
launches are grouped by grid size; "largest_grid" is the group the bench workload belongs to, "all" the plain average (round 1-4 files)."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                grid = int(float(row.get("Grid_Size", row.get("Grid_Size_X", 0)) or 0))
                for key in ((row["Kernel_Name"], row["Counter_Name"], None), (row["Kernel_Name"], row["Counter_Name"], grid)):
                    a = acc[key]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    return acc


def main():
    out = {}
    for d in sys.argv[1:]:
        for (kern, ctr, grid), (tot, cnt) in collect(d).items():
            e = out.setdefault(kern, {}).setdefault("all" if grid is None else grid, {})
            e[ctr + "_KB_avg"] = tot / cnt
            e["launches_" + ctr] = cnt
    res = {}
    for kern, groups in out.items():
        for e in groups.values():
            if "FETCH_SIZE_KB_avg" in e and "WRITE_SIZE_KB_avg" in e:
                e["hbm_bytes_per_launch_corrected"] = (2.0 * e["FETCH_SIZE_KB_avg"] + e["WRITE_SIZE_KB_avg"]) * 1024.0
        grids = sorted(g for g in groups if g != "all")
        r = dict(groups["all"])
        if grids:
            r["largest_grid"] = dict(groups[grids[-1]], grid_size=grids[-1])
            r["n_grid_sizes"] = len(grids)
        res[kern] = r
    json.dump(dict(sorted(res.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch_corrected", 0.0) * kv[1].get("launches_FETCH_SIZE", 0))), sys.stdout, indent=1)


if __name__ == "__main__":
    main()
