#!/usr/bin/env python3
"""Runs tools/micro/mfma_power for both MFMA shapes (random and zero operands) while sampling the card's hwmon clock / power."""
import glob, json, subprocess, sys, threading, time
from pathlib import Path
HERE = Path(__file__).resolve().parent


def sample(stop, out):
    files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    while not stop.is_set():
        best = None
        for f in files:                                  # the busy card = the one with the highest power
            try:
                mhz = int(open(f).read()) / 1e6; w = int(open(f.replace("freq1_input", "power1_input")).read()) / 1e6
            except Exception:
                continue
            if best is None or w > best[1]:
                best = (mhz, w)
        if best:
            out.append(best)
        stop.wait(0.25)


for mode in ([int(a) for a in sys.argv[1:]] or (0, 2, 3, 1, 0, 2, 3)):
    for zero in (0, 1):
        stop, samples = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, samples)); th.start()
        o = subprocess.run([str(HERE / "mfma_power"), str(mode), "4", str(zero)], capture_output=True, text=True)
        stop.set(); th.join()
        xs = sorted(s[0] for s in samples[4:]) or [0]; ws = sorted(s[1] for s in samples[4:]) or [0]
        line = json.loads(o.stdout.strip().splitlines()[-1]) if o.stdout.strip() else {"error": o.stderr[-200:]}
        line.update(clock_mhz_median=xs[len(xs) // 2], power_w_median=ws[len(ws) // 2])
        print(json.dumps(line), flush=True)
