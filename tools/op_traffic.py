#!/usr/bin/env python3
"""profiles/op_traffic.json from two rocprofv3 --pmc passes of tools/op_traffic_run.py (FETCH_SIZE, WRITE_SIZE; KB units):
HBM bytes per launch of every OP (by name) = 2 * FETCH_SIZE + WRITE_SIZE summed over the op's dispatches - FETCH_SIZE doubled because
gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section). Dispatches are attributed by the marker
kernel op_traffic_run.py launches in front of every op.
    usage: op_traffic.py <fetch_dir> <write_dir> <op_order.json> <out.json>"""
import csv, glob, json, sys

def per_op(d, nops, reps):
    rows = []
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "op_marker_kernel" in r[1]]
    if len(marks) != nops * reps + 1:
        raise SystemExit(f"{d}: {len(marks)} markers, expected {nops * reps + 1}")
    acc = [0.0] * nops
    for k in range(nops * reps):
        acc[k % nops] += sum(r[2] for r in rows[marks[k] + 1:marks[k + 1]])
    return [v / reps for v in acc]

fetch_dir, write_dir, order, out = sys.argv[1:5]
o = json.load(open(order))
F, W = per_op(fetch_dir, len(o["ops"]), o["reps"]), per_op(write_dir, len(o["ops"]), o["reps"])
ops = {n: round((2.0 * f + w) * 1024) for n, k, f, w in zip(o["ops"], o["kernels"], F, W) if k != "-"}
json.dump({"unit": "HBM bytes per launch of the op (2*FETCH_SIZE + WRITE_SIZE, KB->B), YOLOv10-S 640x640 bs 32, eager launches in graph order",
           "ops": ops, "kernels": {n: k for n, k in zip(o["ops"], o["kernels"]) if k != "-"}}, open(out, "w"), indent=1)
print(len(ops), "ops")
