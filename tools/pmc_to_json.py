#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> the HBM traffic of the dominant kernel (k_update) per launch, as bench.py's
`roofline.traffic` reads it (profiles/pmc_k_update.json, one entry per workload).
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes and as calibrated on this pool in round 1
(profiles/r01_pmc_k_update_hbm_traffic.md): both counters count KiB; FETCH_SIZE x 2 on gfx950, WRITE_SIZE x 1.
Usage: pmc_to_json.py WORKLOAD FETCH_DIR WRITE_DIR OUT.json "source text" """
import csv
import glob
import json
import os
import sys


def total(d, counter):
    n, s = 0, 0.0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_update" in name and "k_update_reduce" not in name and r["Counter_Name"] == counter:
                n += 1; s += float(r["Counter_Value"])
    return n, s


def main():
    wl, fdir, wdir, out, src = sys.argv[1:6]
    nf, fetch = total(fdir, "FETCH_SIZE")
    nw, write = total(wdir, "WRITE_SIZE")
    if nf == 0 or nw == 0 or nf != nw:
        raise SystemExit(f"no / inconsistent k_update dispatches: FETCH {nf}, WRITE {nw}")
    traffic = fetch * 1024 * 2.0 + write * 1024 * 1.0
    entry = {"workload": wl, "kernel": "k_update", "launches_per_step": nf, "fetch_size_kib_sum": fetch, "write_size_kib_sum": write,
             "fetch_correction": 2.0, "write_correction": 1.0, "traffic_bytes_per_step": traffic,
             "traffic_bytes_per_launch": traffic / nf, "source": src}
    doc = {}
    if os.path.exists(out):
        try:
            doc = json.load(open(out))
        except Exception:
            doc = {}
    doc.setdefault("workloads", {})[wl] = entry
    if wl == "c4":                                   # top level = the bench workload (older readers)
        doc.update(entry)
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{wl}: {nf} launches, {traffic / nf / 1e9:.3f} GB per launch ({traffic / 1e9:.1f} GB per factorisation)")


if __name__ == "__main__":
    main()
