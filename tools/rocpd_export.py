#!/usr/bin/env python3
"""Export the summaries the profiles/ directory keeps from rocprofv3's rocpd (sqlite) output.

    rocpd_export.py stats   <results.db>            -> kernel stats CSV on stdout (rocprofv3 --stats columns)
    rocpd_export.py pmc     <dir with *_results.db> -> pmc_traffic JSON on stdout (HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950)
    rocpd_export.py pmc-csv <results.db>            -> per-dispatch counter CSV on stdout
"""
import csv
import glob
import json
import math
import os
import sqlite3
import sys

mode, path = sys.argv[1], sys.argv[2]
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "silero_v5_step"

if mode == "stats":
    db = sqlite3.connect(path)
    rows = {}
    for name, dur in db.execute("select name, duration from kernels"):
        rows.setdefault(name, []).append(int(dur))
    total = sum(sum(v) for v in rows.values())
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        mean = sum(v) / len(v)
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / max(len(v) - 1, 1))
        w.writerow([name, len(v), sum(v), round(mean, 3), round(100.0 * sum(v) / total, 2), min(v), max(v), round(sd, 3)])
elif mode == "pmc-csv":
    db = sqlite3.connect(path)
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "DurationNs"])
    for r in db.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection order by dispatch_id"):
        w.writerow(list(r))
elif mode == "pmc":
    vals = {}
    for f in glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True):
        db = sqlite3.connect(f)
        for name, cname, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
            if KERNEL in name:
                vals.setdefault(cname, []).append(float(value))
    out = {"kernel": KERNEL, "counters": {k: {"dispatches": len(v), "mean": sum(v) / len(v)} for k, v in vals.items()}}
    f = out["counters"].get("FETCH_SIZE", {}).get("mean")
    wr = out["counters"].get("WRITE_SIZE", {}).get("mean")
    if f is not None and wr is not None:
        out["fetch_bytes_raw"] = f * 1024
        out["write_bytes"] = wr * 1024
        out["hbm_bytes_per_launch"] = (2 * f + wr) * 1024
        out["correction"] = "FETCH_SIZE x2 (gfx950, 16 B/lane coalesced reads), WRITE_SIZE x1; units KiB (MI355X_MICROARCH.md, HBM / rocprofv3)"
    print(json.dumps(out, indent=1))
