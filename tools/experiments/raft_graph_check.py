import os, sys
sys.path.insert(0, "any-stereo_amd")
import torch
from anystereo.harness.synthetic import fill_module_deterministic
from anystereo.harness.train import Trainer, synthetic_train_batch
from anystereo.models import __models__, default_args
torch.backends.cudnn.deterministic = True
name = "continuous_RAFTStereo"
args = default_args(name)
def fresh(graph):
    m = __models__[name](args)
    fill_module_deterministic(m, base_seed=1)
    return Trainer(m.to("cuda:0"), lr=2e-4, num_steps=1000, train_iters=8, max_disp=args.max_disp, graph=graph)
batch = synthetic_train_batch(2, 160, 320, seed=0, device="cuda:0")
n = 8
e = fresh(False); le = [float(e.step(tuple(t.clone() for t in batch))[0]) for _ in range(n)]; del e
g = fresh(None); lg = [float(g.step(tuple(t.clone() for t in batch))[0]) for _ in range(n)]
print("graph used:", g.use_graph, g._graph is not None, "memsets", getattr(g, "graph_memsets", None))
for i, (a, b) in enumerate(zip(le, lg)):
    print(i, a, b, abs(a - b) / abs(a))
assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(le, lg))
print("RAFT graphed == eager OK")
