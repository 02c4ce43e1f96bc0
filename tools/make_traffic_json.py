#!/usr/bin/env python3
"""profiles/traffic.json from the per-config PMC summaries of tools/profile_round.sh:
  tools/make_traffic_json.py NAME [gpurun_out/NAME]   reads NAME_cfgC_pmc_hbm.csv + NAME_cfgC_build_id.txt + NAME_cfgC_streams.txt for C = 1..4
-> {"<config>": {"build_id": ..., "streams": ..., "source": ..., "kernels": {kernel: {"fetch_size_kb": ..., "write_size_kb": ...}}}}
bench.py reports roofline.traffic from it only when build id and batch size are the running ones."""
import csv, json, os, sys
name = sys.argv[1]
d = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", name)
out = {"_comment": "HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE passes over tools/one_launch.py --config C, collected by "
                   "tools/profile_round.sh); unit KB as reported; bench.py applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE x2) and x1024"}
for c in (1, 2, 3, 4):
    p = os.path.join(d, "%s_cfg%d_pmc_hbm.csv" % (name, c))
    if not os.path.exists(p):
        continue
    k = {}
    for r in csv.DictReader(open(p)):
        kn = r["kernel"].replace("lh264::", "")
        k.setdefault(kn, {})["fetch_size_kb" if r["counter"] == "FETCH_SIZE" else "write_size_kb"] = float(r["mean_per_dispatch"])
    out[str(c)] = {"build_id": open(os.path.join(d, "%s_cfg%d_build_id.txt" % (name, c))).read().strip(),
                   "streams": int(open(os.path.join(d, "%s_cfg%d_streams.txt" % (name, c))).read().strip()),
                   "source": "profiles/%s_cfg%d_pmc_hbm.csv" % (name, c), "kernels": k}
json.dump(out, open(os.path.join("profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
print({c: (v["build_id"], v["streams"], len(v["kernels"])) for c, v in out.items() if c != "_comment"})
