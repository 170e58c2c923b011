#!/usr/bin/env python
"""Per-kernel sums of the counters in rocprofv3 --pmc CSVs:  python tools/pmc_summary.py <dir> [name-substring]"""
import collections, csv, glob, os, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt in k:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("GRBM_GUI_ACTIVE", 0))):
    c = tot[k]; print(k[:110]); w = c.get("SQ_WAVE_CYCLES", 0)
    for name in sorted(c):
        print("   %-28s %16.0f  per launch %14.0f  %s" % (name, c[name], c[name] / max(1, n[k][name]),
              ("%.3f of WAVE_CYCLES" % (c[name] / w)) if w and name.startswith("SQ_") else ""))
