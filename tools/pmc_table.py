#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes: python3 tools/pmc_table.py <kernel substring> <dir with */c_counter_collection.csv> ..."""
import collections
import csv
import glob
import os
import sys


def main():
    sub = sys.argv[1]
    for d in sys.argv[2:]:
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
            acc = collections.defaultdict(list)
            dur = []
            for r in csv.DictReader(open(f)):
                if sub in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            if acc:
                n = max(len(v) for v in acc.values())
                print("%s: %d launches, %.1f us" % (os.path.relpath(f), n, sum(dur) / len(dur) / 1e3))
                for k in sorted(acc):
                    v = sorted(acc[k])[len(acc[k]) // 4:]      # (the first launches run at another clock: upper three quarters)
                    print("    %-34s %16.0f" % (k, sum(v) / len(v)))


if __name__ == "__main__":
    main()
