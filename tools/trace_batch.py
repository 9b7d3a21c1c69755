#!/usr/bin/env python3
"""One proof batch out of a rocprofv3 --kernel-trace CSV, kernel by kernel: start offset, gap to the previous kernel's end,
duration, name.  The batch is cut between the last two launches of the quotient kernel (one per batch).

    rocprofv3 --kernel-trace --output-format csv -d out -o d -- python3 bench.py --no-cpu-baseline --batch 1 --concurrency 1 --steps 10
    python3 tools/trace_batch.py out [substring ...]        # only kernels whose name contains one of the substrings

DESIGN.md section 5 / 7 quote its output (the opening's ~365 us rounds of a single proof; the two-round 64-vector commit launches
and the one-workgroup-per-CU tail rounds of a batch of 64)."""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    want = sys.argv[2:]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % root)
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    q = [i for i, r in enumerate(rows) if "bzh_quotient" in r["Kernel_Name"] or "k_expr_vm2" in r["Kernel_Name"]]
    if len(q) < 2:
        sys.exit("fewer than two quotient launches in the trace")
    seq = rows[q[-2]:q[-1]]
    t0 = int(seq[0]["Start_Timestamp"])
    prev_end, busy = t0, 0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void bzh::", "").replace("bzh::", "")
        busy += e - s
        if not want or any(w in name for w in want):
            print("%10.1f us  gap %7.1f  dur %9.1f  grid %-9s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", "?"), name[:70]))
        prev_end = e
    print("%d kernels, span %.2f ms, busy %.2f ms" % (len(seq), (prev_end - t0) / 1e6, busy / 1e6))


if __name__ == "__main__":
    main()
