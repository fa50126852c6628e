import sys, json; sys.path.insert(0,'/root/repo')
import torch, numpy as np, bench, heyoka_amd as hy
from heyoka_amd import configs
torch.cuda.set_device(0)
ctx=dict(torch=torch, hy=hy, configs=configs, dev_index=0)
print(json.dumps(bench.divergence_leg(ctx, 1048576), indent=1))
