#!/usr/bin/env python
"""Cost of a lock-step step() with events on a large outer-SS ensemble: wave-cluster stepper + hy_ev_jets vs the
one-system-per-lane stepper with events vs an event-free step(). usage: events_scale.py [--systems N] [--steps K]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import heyoka_amd as hy
from heyoka_amd import configs

ap = argparse.ArgumentParser()
ap.add_argument("--systems", type=int, default=262144)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--skip-lane-stepper", action="store_true")
ap.add_argument("--d2", type=float, default=81.0, help="squared Jupiter - Saturn distance of the event")
ap.add_argument("--propagate", type=float, default=0.0, help="also time propagate_until(T)")
ap.add_argument("--event", default="d2", choices=["d2", "linear", "pairs"], help="d2: squared Jupiter - Saturn distance (three products); "
                "linear: Saturn crossing y = 0 (a state variable); pairs: the squared distances of all 15 pairs of bodies (close encounters)")
args = ap.parse_args()
M, G = configs.OUTER_SS_MASSES, configs.OUTER_SS_G
n = args.systems
st = configs.outer_ss_state(n, perturb=1e-6, seed=42)
sys_ = hy.model.nbody(6, masses=M, Gconst=G)


def events(log):
    x1, y1, z1, x2, y2, z2 = hy.make_vars("x_1", "y_1", "z_1", "x_2", "y_2", "z_2")
    d2 = (x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2) - args.d2
    if args.event == "pairs":
        def pos(b):
            return hy.make_vars("x_%d" % b, "y_%d" % b, "z_%d" % b)
        evs = []
        for a in range(6):
            for b in range(a + 1, 6):
                pa, pb = pos(a), pos(b)
                g = (pa[0] - pb[0]) * (pa[0] - pb[0]) + (pa[1] - pb[1]) * (pa[1] - pb[1]) + (pa[2] - pb[2]) * (pa[2] - pb[2]) - 1.0
                evs.append(hy.nt_event(g, lambda ta, t, d, i: log.append((i, t)), direction=hy.event_direction.negative))
        return evs
    if args.event == "linear":
        return [hy.nt_event(y2, lambda ta, t, d, i: log.append((i, t)), direction=hy.event_direction.positive)]
    return [hy.nt_event(d2, lambda ta, t, d, i: log.append((i, t)), direction=hy.event_direction.negative)]


def run(tag, **kw):
    log = []
    ta = hy.taylor_adaptive_batch(sys_, st, n, high_accuracy=True, **(dict(nt_events=events(log)) if kw.get("ev") else {}))
    ta.step()
    _ = ta.time
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta.step()
    _ = ta.time  # (step() without events returns before the kernel has finished: the getter synchronises)
    el = (time.perf_counter() - t0) / args.steps
    out = {"tag": tag, "systems": n, "s_per_step": el, "system_steps_per_s": n / el, "events_seen": len(log),
           "mode": ta.hip_source_mode[:70]}
    if args.propagate > 0:
        t0 = time.perf_counter()
        ta.propagate_until(float(ta.time[0]) + args.propagate)
        res = ta.propagate_res_arrays()
        el = time.perf_counter() - t0
        out.update({"propagate_s": el, "propagate_steps": float(res[3].sum()), "propagate_system_steps_per_s": float(res[3].sum()) / el,
                    "events_seen_total": len(log)})
    print(json.dumps(out), flush=True)


run("no events")
run("events on the cluster stepper", ev=True)
if not args.skip_lane_stepper:
    os.environ["HEYOKA_AMD_EVENTS_ON_CLUSTER"] = "0"
    run("events on the one-system-per-lane stepper", ev=True)
