#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced stream, so the read side is doubled.
Usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [workgroup_size_filter]
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Workgroup_Size"], r["Grid_Size"])
        tot[key] += float(r["Counter_Value"])
        cnt[key] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count); WRITE_SIZE KiB x 1024",
           "kernels": []}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out["kernels"].append({"kernel": k[0], "workgroup": int(k[1]), "grid": int(k[2]), "launches": max(nf, nw),
                               "read_bytes": f * 1024 * 2, "read_bytes_raw": f * 1024, "write_bytes": w * 1024,
                               "hbm_bytes": f * 1024 * 2 + w * 1024})
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for e in out["kernels"]:
        if "bzh" in e["kernel"]:
            print("%-60s wg %4d grid %9d  x%-3d read %10.2f MB  write %9.2f MB" % (e["kernel"][:60], e["workgroup"], e["grid"], e["launches"],
                                                                                 e["read_bytes"] / 1e6, e["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
