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


def traffic(pmc_json, out):
    """profiles/traffic.json for bench.py's `roofline.traffic`: FETCH_SIZE x2 (gfx950 note, MI355X_MICROARCH.md §HBM) + WRITE_SIZE,
    KB -> bytes, per launch, for the two roofline kernels, stamped with the hash of the kernel sources they were measured on."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha16
    d = json.load(open(pmc_json))

    def pick(*subs):                                   # any of the spellings (rocprofv3 leaves some names mangled)
        c = [k for k in d if any(sub in k for sub in subs) and "FETCH_SIZE" in d[k] and "WRITE_SIZE" in d[k]]
        return max(c, key=lambda k: d[k]["FETCH_SIZE"]["avg"] + d[k]["WRITE_SIZE"]["avg"]) if c else None
    res = {"note": "L2-fill + write-back bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950 note in "
                   "MI355X_MICROARCH.md + WRITE_SIZE), KB*1024; FETCH_SIZE counts every L2 miss, also those the 256-MiB Infinity Cache serves",
           "source": pmc_json, "csrc_sha16": csrc_sha16()}
    fc1 = pick("gemm_8phase_persistent_kernel<4,")       # MODE 4 = EPI_LN_BIAS_GELU = FFN-1
    if fc1:
        f, w = d[fc1]["FETCH_SIZE"]["avg"] * 2 * 1024, d[fc1]["WRITE_SIZE"]["avg"] * 1024
        res.update(gemm_fc1_kernel=fc1, gemm_fc1_hbm_bytes_per_launch=int(f + w), gemm_fc1_fetch_bytes=int(f), gemm_fc1_write_bytes=int(w))
    sg = pick("search_groupmax_kernel<64", "search_groupmax_kernelILi64E")
    if sg:
        res.update(search_groupmax64_kernel=sg,
                   search_groupmax64_hbm_bytes_per_launch=int(d[sg]["FETCH_SIZE"]["avg"] * 2 * 1024 + d[sg]["WRITE_SIZE"]["avg"] * 1024))
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
