#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE calibration from the two rocprofv3 --pmc passes over profiles/ubench/stream8.bin.
usage: python profiles/calibrate_pmc.py <out.json> <fetch.db> <write.db>"""
import json
import sqlite3
import sys

N_BYTES = (1 << 29) * 8
KNOWN = {"read8": (N_BYTES, 0), "write8": (0, N_BYTES), "copy8": (N_BYTES, N_BYTES), "read16": (N_BYTES, 0)}


def last_values(path, counter):
    cur = sqlite3.connect(path).cursor()
    res = {}
    for name, disp, val in cur.execute(
            "select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = ? "
            "group by kernel_name, dispatch_id order by dispatch_id", (counter,)):
        res[name.split("(")[0]] = val  # last dispatch wins
    return res


def main():
    out, fdb, wdb = sys.argv[1:4]
    fe, wr = last_values(fdb, "FETCH_SIZE"), last_values(wdb, "WRITE_SIZE")
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB as reported) vs the known bytes of streaming kernels over "
                   "4 GiB (profiles/ubench/stream8.hip), second dispatch of each kernel; factor = true bytes / reported bytes",
           "kernels": {}}
    for k, (rb, wb) in KNOWN.items():
        f = next((v for n, v in fe.items() if n.startswith(k)), None)
        w = next((v for n, v in wr.items() if n.startswith(k)), None)
        e = {"true_read_bytes": rb, "true_write_bytes": wb, "fetch_kib": f, "write_kib": w}
        if f and rb:
            e["fetch_factor"] = rb / (f * 1024.0)
        if w and wb:
            e["write_factor"] = wb / (w * 1024.0)
        res["kernels"][k] = e
    with open(out, "w") as fo:
        json.dump(res, fo, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
