#!/usr/bin/env python3
"""print name / calls / average ms of our kernels from a rocprofv3 kernel_stats.csv"""
import csv, sys, glob
for pat in sys.argv[1:]:
    for f in sorted(glob.glob(pat)):
        print("==", f)
        for r in csv.DictReader(open(f)):
            if "lh264" in r["Name"]:
                print("  %-34s calls %3s avg %9.3f ms" % (r["Name"].split("(")[0].replace("lh264::", ""), r["Calls"], float(r["AverageNs"]) / 1e6))
