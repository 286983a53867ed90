#!/usr/bin/env python3
"""Timeline of one decode step's launches from a rocprofv3 --kernel-trace CSV: for every step of the chosen variant the
kernels between the aggregation pass and the compaction kernel, their durations and the gaps between them.
    rocprofv3 --kernel-trace -d DIR --output-format csv -- python tools/decode_step.py --batch 64
    python tools/s1_timeline.py DIR [aggregate_harvest_kernel]"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "aggregate_harvest_kernel"
    f = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    if not idx:
        raise SystemExit("no " + anchor)
    i0 = idx[-2] if len(idx) > 1 else idx[-1]                  # the last but one step: steady state
    t_end = int(rows[i0]["End_Timestamp"])
    print(f"{'kernel':70s} {'stream':>6s} {'start_us':>9s} {'dur_us':>7s} {'gap_us':>7s}")
    prev_end = t_end
    for r in rows[i0:i0 + 40]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("kvc::", "")[:70]
        print(f"{name:70s} {r.get('Stream_Id', r.get('Queue_Id', '')):>6s} {(s - t_end) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:7.1f}")
        prev_end = max(prev_end, e)
        if "compact_" in name and "plan" not in name:
            break


if __name__ == "__main__":
    main()
