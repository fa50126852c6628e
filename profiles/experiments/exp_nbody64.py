import sys, os, json, numpy as np
sys.path.insert(0,'/root/repo')
import heyoka_amd as hy
from heyoka_amd import configs
# fixed number of steps per system via max_steps: timing experiments with wrong numerics cannot run away
n=65536
st = configs.plummer_nbody_state(64, n, seed=1234+42)
for v in sys.argv[1:]:
    kv = dict(x.split("=",1) for x in v.split(",") if x)
    os.environ.update(kv)
    ta = hy.taylor_adaptive_batch(hy.model.nbody(64), st, n)
    for k in kv: del os.environ[k]
    rates=[]
    for r in range(3):
        ta.propagate_until(1e9, max_steps=10)
        ns = ta.propagate_res_arrays()[3]
        ms = list(ta.kernel_ms_history(1))[-1]
        rates.append(float(ns.sum())/(ms*1e-3))
        ta.set_time(0.0); ta.state = st
    print(json.dumps({"variant": v[-60:], "rates": ["%.4g"%x for x in rates]}), flush=True)
