import sys, os
sys.path[:0]=['bess-kge_amd','tests','.']
import torch
from besskge import runtime
from besskge import _native as nat
from test_checkpoint import _model, _batch
dev=torch.device('cuda',0)
keep=[]
orig=nat.coalesced_update
def spy(o, table, seg, grads, s1=None, s2=None, keep_=None, sum_only=False):
    keep.append((table, seg, list(grads), s1, s2))
    return orig(o, table, seg, grads, s1, s2, keep_, sum_only)
nat.coalesced_update=spy
keep2=[]
o2=nat.sparse_sgd_lists
def spy2(table, lists, lr):
    keep2.append(("sgd_lists", table, [(i,g) for i,g in lists], lr))
    return o2(table, lists, lr)
nat.sparse_sgd_lists=spy2
o3=nat.neg_score_pertriple_bwd
def spy3(d, query, neg, n_neg, go, want_d_neg=True):
    r=o3(d, query, neg, n_neg, go, want_d_neg=want_d_neg)
    keep2.append(("pt_bwd", query, neg.idx, go, r))
    return r
nat.neg_score_pertriple_bwd=spy3
import besskge.bess as B
x=[torch.empty(1<<20, device=dev) for _ in range(64)]; del x
model, sharding = _model(dev, n_shard=2)
batches=[_batch(sharding,2,16,6,s) for s in range(5)]
runner = runtime.training_model(model, runtime.Options(use_graphs=True), runtime.Adam(lr=0.01), device=dev)
for i in (0,1,2,3):
    n0=len(keep)
    runner(**batches[i]); torch.cuda.synchronize()
    st = model._optimizer_state[model._local_table(0).data_ptr()]
    print('step', i, 'm max', float(st['s'][0].abs().max()), 'new spy calls', len(keep)-n0)
    # the captured call records are the last ones appended during capture (first call): inspect those
    for j,(table, seg, grads, s1, s2) in enumerate(keep[-4:]):
        print('   call', j, 'table', tuple(table.shape), 'n_seg', int(seg.n_seg), 'n_refs', seg.n_refs, 'refs range', int(seg.refs.min()), int(seg.refs.max()),
              'grads', [(tuple(g.shape), f"{float(g.abs().max()):.2e}") for g in grads], 's1 max', f"{float(s1.abs().max()):.2e}" if s1 is not None else None)

    for rec in keep2[-4:]:
        if rec[0]=="sgd_lists":
            _, table, lists, lr = rec
            print('   sgd_lists dst', tuple(table.shape), f"{float(table.abs().max()):.2e}", 'lists', [(tuple(i.shape), int(i.min()), int(i.max()), tuple(g.shape), f"{float(g.abs().max()):.2e}") for i,g in lists], lr)
        else:
            _, q, idx, go, (dq, dn) = rec
            print('   pt_bwd q', f"{float(q.abs().max()):.2e}", 'idx', int(idx.min()), int(idx.max()), 'go', f"{float(go.abs().max()):.2e}", 'dq', f"{float(dq.abs().max()):.2e}", 'dn', tuple(dn.shape), f"{float(dn.abs().max()):.2e}")
