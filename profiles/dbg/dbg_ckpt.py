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
