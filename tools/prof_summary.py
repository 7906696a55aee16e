#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the small summaries committed under profiles/.

  prof_summary.py stats <dir> <out.md> "<command line profiled>"     # *_kernel_stats.csv -> markdown table
  prof_summary.py pmc <out.json> <dir> [<dir> ...]                   # *_counter_collection.csv -> per-kernel counter averages

The PMC passes are run one counter per pass (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`), never together with a trace domain
other than the kernel trace, as MI355X_MICROARCH.md prescribes; its gfx950 note (FETCH_SIZE counts 64-B units reported as KB
=> x2) is applied by the consumer (bench.py / profiles/traffic.json), not here.
"""
import csv, glob, json, sys
from collections import defaultdict


def short(name: str) -> str:
    return name if len(name) <= 110 else name[:110]


def stats(d, out, cmd):
    f = sorted(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True))
    if not f:
        sys.exit(f"no *_kernel_stats.csv under {d}")
    rows = list(csv.DictReader(open(f[0])))
    with open(out, "w") as o:
        o.write(f"{cmd}\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows:
            o.write(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.1f} | {r['Percentage']} |\n")
    print(f"{len(rows)} kernels -> {out}")


def pmc(out, dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0]
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"calls": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()} for k, cs in acc.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3:])
