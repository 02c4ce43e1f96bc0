#!/usr/bin/env python3
"""Summarise rocprofv3 counter-collection CSVs: mean counter value per dispatch for this repository's kernels.
usage: pmc_summary.py DIR [DIR ...]   (every *_counter_collection.csv below the directories is read)"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(list)
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "chain_kernel" in k or "ctx_" in k or "recon_" in k or "coder_" in k:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,dispatches,mean_per_dispatch")
for (k, c), v in sorted(acc.items()):
    print("%s,%s,%d,%.1f" % (k, c, len(v), sum(v) / len(v)))
