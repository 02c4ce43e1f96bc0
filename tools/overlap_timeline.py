#!/usr/bin/env python3
"""start / end (ms, relative) of every kernel of the last step in a rocprofv3 kernel_trace.csv of bench.py: who runs beside whom"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lh264::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step that is not one of the serial, timed ones: find the last recon launch that overlaps a coder kernel
rec = [r for r in rows if "recon_chain" in r["Kernel_Name"]]
pick = None
for r in reversed(rec):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if any("coder_" in q["Kernel_Name"] and int(q["Start_Timestamp"]) < e and int(q["End_Timestamp"]) > s for q in rows):
        pick = r
        break
if pick is None:
    sys.exit("no overlapped step found")
s0 = int(pick["Start_Timestamp"]) - 12_000_000
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s0 <= s <= s0 + 45_000_000:
        print("%8.2f .. %8.2f  (%6.2f ms)  %s" % ((s - s0) / 1e6, (e - s0) / 1e6, (e - s) / 1e6, r["Kernel_Name"].split("(")[0].replace("lh264::", "")))
