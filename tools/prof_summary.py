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


def mfma(pmc_json, stats_mds, out):
    """Per-kernel matrix-pipe utilisation from ONE SQ counter pass (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, summed
    over the chip's 1 024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; the wave-state counters count quad-cycles and are used as
    fractions of SQ_WAVE_CYCLES only):  MFMA busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8);
    clock under the profiler = GRBM_GUI_ACTIVE / 8 / the kernel's duration in the un-countered stats pass (indicative: profiled runs clock lower)."""
    import re
    d = json.load(open(pmc_json))
    dur = {}
    for md in stats_mds:
        for ln in open(md):
            m = re.match(r"\| (.+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", ln)
            if m:
                dur[m.group(1).split("(")[0].strip()] = float(m.group(3))
    rows = []
    for k, c in d.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"], c["GRBM_GUI_ACTIVE"]["avg"]
        if busy <= 0 or act <= 0:
            continue
        wc = c.get("SQ_WAVE_CYCLES", {}).get("avg", 0) or 1
        us = next((v for n, v in dur.items() if n.startswith(k[:60]) or k.startswith(n[:60])), None)
        rows.append((busy / 1024 / (act / 8), k, c["SQ_VALU_MFMA_BUSY_CYCLES"]["calls"], act / 8, us,
                     c.get("SQ_WAIT_ANY", {}).get("avg", 0) / wc, c.get("SQ_WAIT_INST_ANY", {}).get("avg", 0) / wc,
                     c.get("SQ_ACTIVE_INST_ANY", {}).get("avg", 0) / wc))
    rows.sort(reverse=True)
    with open(out, "w") as o:
        o.write("MFMA utilisation per kernel: one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE) over python3 bench.py (encode 5 steps + 10 M x 768 search, "
                "Qb 64 / 256); matrix pipe busy = (MFMA_BUSY / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs); wave states as fractions of SQ_WAVE_CYCLES\n\n"
                "| kernel | calls | matrix pipe busy | kernel cycles (GUI_ACTIVE / 8) | avg us (stats pass) | waves parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) |\n"
                "|---|---|---|---|---|---|---|---|\n")
        for u, k, n, cyc, us, wa, wi, ai in rows:
            o.write(f"| {short(k)} | {n} | {u:.3f} | {cyc:,.0f} | {'' if us is None else f'{us:.1f}'} | {wa:.2f} | {wi:.2f} | {ai:.2f} |\n")
    print(f"{len(rows)} kernels with matrix work -> {out}")


if __name__ == "__main__":
    if sys.argv[1] == "mfma":
        mfma(sys.argv[2], sys.argv[3:-1], sys.argv[-1])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3:])
