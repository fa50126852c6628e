#!/usr/bin/env python
"""Dump the hardware counters of the last dispatch of a kernel from rocprofv3 (rocpd sqlite) databases.

usage: python profiles/pmc_dump.py <out.json> <kernel_name> <note> <db> [<db> ...]
Each database comes from a separate `rocprofv3 --pmc ...` pass (no tracing domains)."""
import json
import sqlite3
import sys


def main():
    out_path, kernel, note = sys.argv[1], sys.argv[2], sys.argv[3]
    res = {"note": note}
    for path in sys.argv[4:]:
        cur = sqlite3.connect(path).cursor()
        rows = list(cur.execute(
            "select dispatch_id, counter_name, value, end - start, vgpr_count, accum_vgpr_count, lds_block_size, "
            "scratch_size, grid_size, workgroup_size from counters_collection where kernel_name = ? "
            "order by dispatch_id", (kernel,)))
        if not rows:
            continue
        last = max(r[0] for r in rows)
        for r in rows:
            if r[0] == last:
                res[r[1]] = res.get(r[1], 0.0) + r[2]
                res["duration_ms"] = r[3] / 1e6
                res["vgpr"], res["agpr"], res["lds_bytes"], res["scratch_bytes_per_lane"] = r[4], r[5], r[6], r[7]
                res["grid"], res["workgroup"] = r[8], r[9]
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
