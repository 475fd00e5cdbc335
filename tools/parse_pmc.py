#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes into per-kernel HBM traffic per launch (profiles/traffic.json).

    python tools/parse_pmc.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> [out.json] [workload key]

With a workload key (bench.py prints it as roofline.traffic's lookup key, e.g.
"C3/50cams_20000pts_per_rank/ba_schur_mfma") the record is merged into out.json under that key, so one file holds
the traffic of several workloads and bench.py never attributes one workload's bytes to another.

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section) and
cdna_hip_programming.md section 7: FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC
slots), both are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read, so   bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   per launch.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short_name(kernel):
    m = re.search(r"sfm::(\w+?)(_kernel)?[<(]", kernel)
    return m.group(1) if m else kernel.split("(")[0][:60]


def per_kernel(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = short_name(row["Kernel_Name"])
                acc[name] += float(row["Counter_Value"])
                cnt[name] += 1
    return {k: acc[k] / cnt[k] for k in acc}, dict(cnt)


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic.json"
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _nw = per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res[k] = {"fetch_kib_raw": f, "write_kib": w, "launches": nf.get(k, 0),
                  "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
    flat = {k: v["hbm_bytes_per_launch"] for k, v in res.items()}
    flat["_detail"] = res
    flat["_note"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch; gfx950 FETCH_SIZE counts 64 B per 128 B request"
    if len(sys.argv) > 4:
        key = sys.argv[4]
        try:
            with open(out) as fh:
                doc = json.load(fh)
        except Exception:
            doc = {}
        doc[key] = flat
        flat = doc
    with open(out, "w") as fh:
        json.dump(flat, fh, indent=1, sort_keys=True)
    for k, v in res.items():
        print("%-28s fetch %10.1f KiB (raw)  write %10.1f KiB  -> %8.2f MB/launch" % (
            k, v["fetch_kib_raw"], v["write_kib"], v["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
