import sys, os
sys.path[:0]=['bess-kge_amd','tests','.']
import torch
from besskge import runtime
from test_checkpoint import _model, _batch
dev=torch.device('cuda',0)
variant=sys.argv[1]
n=1 if variant=="n1" else 2
def mmax(model):
    torch.cuda.synchronize()
    st = model._optimizer_state[model._local_table(0).data_ptr()]
    return f"{float(st['s'][0].abs().max()):.2e}"
model, sharding = _model(dev, n_shard=n)
batches=[_batch(sharding,n,16,6,s) for s in range(5)]
opt = runtime.SGD(lr=0.01, momentum=0.9) if variant=="sgdm" else runtime.Adam(lr=0.01)
if variant=="prewarm":  # an eager step of another model first
    m2,_=_model(dev, n_shard=n); r2=runtime.training_model(m2, runtime.Options(), runtime.Adam(lr=0.01), device=dev); r2(**batches[0]); torch.cuda.synchronize()
if variant=="prealloc":
    x=[torch.empty(1<<20, device=dev) for _ in range(64)]; del x
runner = runtime.training_model(model, runtime.Options(use_graphs=True), opt, device=dev)
line=[]
for i in (0,1,2,3):
    runner(**batches[i]); line.append(mmax(model))
print(variant, line, flush=True)
