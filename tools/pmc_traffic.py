"""Merge two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes) into the per-kernel HBM-traffic summary bench.py reads.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_<workload>_pmc_traffic.json

One entry per (kernel, workgroup size, grid size): mean bytes per launch.  FETCH_SIZE / WRITE_SIZE count KiB-like units of
1024 B on this stack; FETCH_SIZE under-counts coalesced 16-B-per-lane streams by 2 on gfx950 (guide's correction), so both
the corrected (read_bytes) and raw (read_bytes_raw) figures are kept.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(directory, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *counter_collection.csv under %s" % directory)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            m = re.match(r"(.*?[\w>])\(", name)      # up to the argument list ("(anonymous namespace)::k(...)" keeps its prefix)
            name = m.group(1) if m else name
            key = (name, int(row["Workgroup_Size"]), int(row["Grid_Size"]))
            a = acc[key]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = []
    for key in sorted(set(fetch) | set(write)):
        f, w = fetch.get(key, [0.0, 0]), write.get(key, [0.0, 0])
        n = max(f[1], w[1], 1)
        raw = f[0] / max(f[1], 1) * 1024.0
        wr = w[0] / max(w[1], 1) * 1024.0
        out.append({"kernel": key[0], "workgroup": key[1], "grid": key[2], "launches": n, "read_bytes": raw * 2.0, "read_bytes_raw": raw,
                    "write_bytes": wr, "hbm_bytes": raw * 2.0 + wr})
    json.dump({"units": "bytes per launch", "correction": "FETCH_SIZE x 1024 x 2 (gfx950 half-count); WRITE_SIZE x 1024", "kernels": out},
              sys.stdout, indent=1)


if __name__ == "__main__":
    main()
