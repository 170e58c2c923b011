#!/usr/bin/env python
"""Turn the rocprofv3 CSVs a gpurun call left under gpurun_out/<dir>/ into the small tracked summaries
under profiles/ (kernel stats, HBM traffic per launch, bench line, per-launch table).

    python tools/summarize_profiles.py gpurun_out/prof_r1 r01

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of wide
coalesced streaming reads (MI355X_MICROARCH.md, HBM section), so hbm_read = 2 * FETCH_SIZE * 1024.
The PMC passes were separate rocprofv3 runs (FETCH_SIZE and WRITE_SIZE do not fit one pass).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def agg(pattern, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return tot, n


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    fa, fn = agg(os.path.join(src, "fetch", "*", "*counter_collection.csv"), "FETCH_SIZE")
    wa, wn = agg(os.path.join(src, "write", "*", "*counter_collection.csv"), "WRITE_SIZE")
    traffic = {}
    for k in sorted(set(fa) | set(wa)):
        rd = 2.0 * fa.get(k, 0.0) * 1024.0 / max(1, fn.get(k, 1))
        wr = wa.get(k, 0.0) * 1024.0 / max(1, wn.get(k, 1))
        traffic[k] = {"launches_fetch_pass": fn.get(k, 0), "launches_write_pass": wn.get(k, 0),
                      "hbm_read_bytes_per_launch": round(rd), "hbm_write_bytes_per_launch": round(wr),
                      "hbm_bytes_per_launch": round(rd + wr)}
    json.dump({"note": "hbm_read = 2 * FETCH_SIZE(KiB) * 1024 (gfx950 correction), hbm_write = WRITE_SIZE(KiB) * 1024; "
                       "separate --pmc passes of `bench.py --steps 3 --warmup 1 --inflight 1 --tiles <table of the timing run>`: the tile "
                       "table is installed, not measured, so both passes launch the same kernels the same number of times "
                       "(launches_fetch_pass == launches_write_pass) and no autotune launch is counted",
               "kernels": traffic}, open(os.path.join(out, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    tiles = os.path.join(src, "tiles.json")
    if os.path.exists(tiles):
        shutil.copy(tiles, os.path.join(out, f"{tag}_tiles.json"))
    for name in ("bench.log", "layers.json"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            if name == "bench.log":
                line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
                json.dump(json.loads(line), open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
            else:
                shutil.copy(p, os.path.join(out, f"{tag}_{name}"))
    for name, dst in (("bench_416.log", "bench_416.json"), ("bench_fp32.log", "bench_fp32.json")):
        p = os.path.join(src, name)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if lines:
                json.dump(json.loads(lines[-1]), open(os.path.join(out, f"{tag}_{dst}"), "w"), indent=1)
    if os.path.isdir(os.path.join(src, "sq")):                   # SQ counters of the band kernels (tools/pmc_summary.py)
        import subprocess
        txt = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_summary.py"), os.path.join(src, "sq"), "conv_band"],
                             capture_output=True, text=True).stdout
        if txt.strip():
            open(os.path.join(out, f"{tag}_sq_counters_band.txt"), "w").write(txt)
    final = os.path.join(os.path.dirname(src.rstrip("/")), f"final_{tag}", "pytest.log")
    if os.path.exists(final):
        shutil.copy(final, os.path.join(out, f"{tag}_gpu_pytest.log"))
    # traffic.json consumed by bench.py: keyed by the exact kernel names rocprofv3 reports
    json.dump({"source": f"profiles/{tag}_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py with the timing run's tile "
                         "table installed (2 * FETCH_SIZE + WRITE_SIZE per launch, equal launch counts in both passes); not measured in this run",
               "kernels": {k: v["hbm_bytes_per_launch"] for k, v in traffic.items() if v["launches_fetch_pass"] == v["launches_write_pass"]}},
              open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
