import sys, tempfile, pathlib
sys.path[:0]=['bess-kge_amd','tests','.']
import torch
from besskge import checkpoint, runtime
from test_checkpoint import _model, _batch
dev=torch.device('cuda',0)
tmp=pathlib.Path(tempfile.mkdtemp())
snaps={}
for use_graphs in (False, True):
    model, sharding = _model(dev)
    batches=[_batch(sharding,2,16,6,s) for s in range(5)]
    runner = runtime.training_model(model, runtime.Options(use_graphs=use_graphs), runtime.Adam(lr=0.01), device=dev)
    log=[]
    def snap(tag):
        log.append((tag, model.score_fn.entity_embedding.detach().clone()))
    for b in batches[:2]: runner(**b)
    snap('after2')
    checkpoint.save_checkpoint(model, tmp/f'c{int(use_graphs)}', chunk_bytes=4096)
    for b in batches[2:4]: runner(**b)
    snap('after4')
    checkpoint.load_checkpoint(model, tmp/f'c{int(use_graphs)}', chunk_bytes=4096)
    snap('loaded')
    st = model._optimizer_state[model._local_table(0).data_ptr()]
    print(use_graphs, 'step', st['step'], st.get('step_dev'), [float(x.abs().sum()) for x in st['s']])
    for b in batches[2:4]: runner(**b)
    snap('again4')
    snaps[use_graphs]=log
for (t,a),(t2,b) in zip(snaps[False],snaps[True]):
    off=(a-b).abs()
    print(t, float(off.max()), float((off>1e-5).float().mean()))
a=dict(snaps[True]); print('graph: after4 vs again4', float((a['after4']-a['again4']).abs().max()))
a=dict(snaps[False]); print('eager: after4 vs again4', float((a['after4']-a['again4']).abs().max()))
print("---- second part: graph mode, repeated rollbacks")
model, sharding = _model(dev)
batches=[_batch(sharding,2,16,6,s) for s in range(5)]
runner = runtime.training_model(model, runtime.Options(use_graphs=True), runtime.Adam(lr=0.01), device=dev)
for b in batches[:2]: runner(**b)
checkpoint.save_checkpoint(model, tmp/'cc', chunk_bytes=4096)
outs=[]
for rep in range(3):
    checkpoint.load_checkpoint(model, tmp/'cc', chunk_bytes=4096)
    st = model._optimizer_state[model._local_table(0).data_ptr()]
    pre=[x.clone() for x in st['s']]
    runner(**batches[2])
    torch.cuda.synchronize()
    outs.append((model.score_fn.entity_embedding.detach().clone(), [x.clone() for x in st['s']], int(st['step_dev'])))
for i in (1,2):
    off=(outs[0][0]-outs[i][0]).abs()
    print('table diff', i, float(off.max()), float((off>1e-5).float().mean()), 'm diff', float((outs[0][1][0]-outs[i][1][0]).abs().max()), 'v diff', float((outs[0][1][1]-outs[i][1][1]).abs().max()), outs[i][2])


b=batches[2]
touched0=set(b['head'][0].flatten().tolist())|set(b['tail'][:,0].flatten().tolist())|set(b['negative'][0].flatten().tolist())
import numpy as np
f0=np.load(tmp/'cc'/'entity_shard0.state0.npy'); print('file m max', np.abs(f0).max())
for i in range(3):
    m,v=outs[i][1]
    big=(m.abs().max(dim=1).values>10).nonzero().flatten().tolist()
    print(i,'m max',float(m.abs().max()),'v max', float(v.max()), 'big rows', [(r, r in touched0) for r in big[:8]], 'table max', float(outs[i][0].abs().max()))
print('table0 vs table1 rows differing', ((outs[0][0]-outs[1][0]).abs().max(dim=-1).values>1e-5).nonzero().tolist()[:10])
