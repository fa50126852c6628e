#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd sqlite) outputs into the small text/JSON files kept under profiles/.

usage: python profiles/summarize_rocprof.py <out_prefix> <ktrace.db> [<pmc_fetch.db> <pmc_write.db> [<pmc_bench.log> [<ktrace_bench.log>]]]
Writes <out_prefix>_kernel_stats.txt (per-kernel calls / total / avg / min / max, like
`rocprofv3 --kernel-trace --stats`) and, if counter databases are given, <out_prefix>_pmc.json with
the per-dispatch FETCH_SIZE / WRITE_SIZE of the stepper kernel (KiB, as reported) and the derived HBM
traffic per launch."""
import json
import sqlite3
import sys


def kernel_stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-64s %6s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for r in rows[:12]:
        lines.append("%-64s %6d %14d %14.0f %14d %14d %6.2f%%" % (r[0][:64], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
    # The stepper launches one by one: the first is the bench's warmup, the others are the timed region whose
    # average bench.py reports as roofline.kernel_ms_avg (HIP events around the same launches).
    per = [r[0] for r in cur.execute("select end - start from kernels where name = 'hy_taylor' order by start")]
    if len(per) > 1:
        timed = per[1:]
        lines.append("")
        lines.append("hy_taylor launches in order (ns): " + " ".join(str(x) for x in per))
        lines.append("hy_taylor timed launches (all but the warmup launch): n = %d, avg_ns = %.0f" % (len(timed), sum(timed) / len(timed)))
    return "\n".join(lines), rows


def counters(path, kernel="hy_taylor"):
    cur = sqlite3.connect(path).cursor()
    return list(cur.execute(
        "select dispatch_id, counter_name, value, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, "
        "scratch_size, grid_size, workgroup_size, end - start from counters_collection where kernel_name = ? "
        "order by dispatch_id", (kernel,)))


def main():
    prefix, ktrace = sys.argv[1], sys.argv[2]
    txt, rows = kernel_stats(ktrace)
    with open(prefix + "_kernel_stats.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats (rocpd database summarised by profiles/summarize_rocprof.py)\n")
        f.write(txt + "\n")
        if len(sys.argv) >= 7:
            # The bench line printed by the traced run itself: its kernel_ms_avg is measured with HIP events.
            for line in open(sys.argv[6]):
                if line.startswith("{"):
                    b = json.loads(line)
                    f.write("bench.py (same run) roofline.kernel_ms_avg = %.6f ms, value = %.6g %s\n"
                            % (b["roofline"]["kernel_ms_avg"], b["value"], b["unit"]))
    print(txt)
    if len(sys.argv) >= 5:
        fe, wr = counters(sys.argv[3]), counters(sys.argv[4])
        out = {
            "note": "FETCH_SIZE / WRITE_SIZE in KiB as reported by rocprofv3 (separate --pmc passes, no tracing "
                    "domains). Calibration on known-byte streams (profiles/r02_pmc_calibration.json, "
                    "profiles/ubench/stream8.hip): on gfx950 FETCH_SIZE reads 1/2 of the bytes for 8 B/lane as well as for "
                    "16 B/lane coalesced reads (factor 2.000), WRITE_SIZE is exact (factor 1.000): HBM traffic = "
                    "2 x FETCH_SIZE + WRITE_SIZE (traffic_bytes_fetch_x2). Infinity-Cache hits are counted.",
            "dispatches": [],
        }
        for a, b in zip(fe, wr):
            out["dispatches"].append({
                "dispatch_id": a[0], "fetch_kib": a[2], "write_kib": b[2], "vgpr": a[3], "agpr": a[4], "sgpr": a[5],
                "lds_bytes": a[6], "scratch_bytes_per_lane": a[7], "grid": a[8], "workgroup": a[9],
                "duration_ns_fetch_pass": a[10], "duration_ns_write_pass": b[10],
            })
        timed = out["dispatches"][1:] or out["dispatches"]
        n = len(timed)
        fetch = sum(d["fetch_kib"] for d in timed) / n * 1024.0
        write = sum(d["write_kib"] for d in timed) / n * 1024.0
        out["per_launch_avg"] = {
            "fetch_bytes_raw": fetch, "fetch_bytes_x2": 2 * fetch, "write_bytes": write,
            "traffic_bytes_raw": fetch + write, "traffic_bytes_fetch_x2": 2 * fetch + write,
            "kernel_ns": sum(d["duration_ns_fetch_pass"] for d in timed) / n,
        }
        if len(sys.argv) >= 6:
            # The bench JSON line printed by the profiled command identifies the kernel and the workload.
            for line in open(sys.argv[5]):
                if line.startswith("{"):
                    b = json.loads(line)
                    out["kernel_sha256"] = b["config"].get("kernel_sha256")
                    out["systems_per_gpu"] = b["config"].get("systems_per_gpu")
                    out["bench_line_of_profiled_run"] = b
                    # Registers / spills / scratch / LDS from the metadata notes of the code object (the per-dispatch
                    # "vgpr" / "agpr" columns of rocprofv3 above are what the tool reports: ArchVGPR granules, no AGPRs).
                    out["kernel_resources_from_code_object"] = b.get("roofline", {}).get("kernel_resources")
        with open(prefix + "_pmc.json", "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out["per_launch_avg"], indent=1))


if __name__ == "__main__":
    main()
