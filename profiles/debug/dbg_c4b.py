import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "bess-kge_amd"), REPO, os.path.join(REPO, "tests")]
import numpy as np, torch
from besskge import runtime, _native as nat
from besskge.batch_sampler import RandomShardedBatchSampler
from besskge.bess import EmbeddingMovingBessKGE
from besskge.dataset import KGDataset
from besskge.loss import SampledSoftmaxCrossEntropyLoss
from besskge.negative_sampler import RandomShardedNegativeSampler
from besskge.scoring import TransE
from besskge.sharding import PartitionedTripleSet, Sharding
from oracle import kge
dev = torch.device("cuda", 0)
n_entity,n_rel,n_triple,S,K,d=40_000,535,200_000,512,32,256
rng=np.random.default_rng(0)
triples=np.stack([rng.integers(n_entity,size=n_triple),rng.integers(n_rel,size=n_triple),rng.integers(n_entity,size=n_triple)],axis=1)
ds=KGDataset(n_entity=n_entity,n_relation_type=n_rel,triples={"train":triples},original_triple_ids={"train":np.arange(n_triple)})
sharding=Sharding.create(n_entity,1,seed=1234)
pts=PartitionedTripleSet.create_from_dataset(ds,"train",sharding)
ns=RandomShardedNegativeSampler(K,sharding,1234,"t",local_sampling=False,flat_negative_format=True)
bs=RandomShardedBatchSampler(pts,ns,S,1,1234)
torch.manual_seed(0)
ent=torch.randn(1,sharding.max_entity_per_shard,d).half().float(); rel=torch.randn(n_rel,d).half().float()
batch=bs[next(iter(bs.get_dataloader_sampler(shuffle=False)))]
keys=("head","relation","tail","negative")
spec=kge.StepSpec("TransE",1,True,"t",True,augment=True)
t0,r0=ent.clone().requires_grad_(True),rel.clone().requires_grad_(True)
with kge.half_queries():
    want=kge.bess_step(spec,"EmbeddingMoving",t0,r0,{k:batch[k][0] for k in keys},dict(kind="ssce",n_entity=n_entity))
    torch.stack(want["loss"]).sum().backward()
rec = {}
orig = nat.coalesced_update
def spy(o, table, seg, grads, s1=None, s2=None, keep=None, sum_only=False):
    sums = orig(None, table, seg, grads, sum_only=True)
    torch.cuda.synchronize()
    n_seg = int(seg.n_seg.item())
    rec.update(rows=seg.seg_rows[:n_seg].cpu().long(), sums=sums[:n_seg].cpu(), before=table.float().cpu().clone(), lr=float(o.lr), kind=int(o.kind),
               mom=float(o.momentum), wd=float(o.weight_decay), n_lists=len(grads))
    return orig(o, table, seg, grads, s1, s2, keep, sum_only)
nat.coalesced_update = spy
fn=TransE(True,1,sharding,n_rel,d,ent,rel)
m=EmbeddingMovingBessKGE(ns,fn,SampledSoftmaxCrossEntropyLoss(n_entity),return_scores=True,augment_negative=True)
lr=0.05
runner=runtime.training_model(m,optimizer=runtime.SGD(lr=lr),device=dev,dtype=torch.float16)
res=runner(**{k:batch[k].flatten(end_dim=1) for k in keys})
got=m.score_fn.entity_embedding.detach().float().cpu()[0]
want_ent=(ent-lr*t0.grad).half().float()[0]
ulp=torch.exp2(torch.floor(torch.log2(want_ent.abs().clamp(min=2.0**-14)))-10)
err=(got-want_ent).abs()/ulp
bad=(err>1.001).nonzero()
print("bad",len(bad),"max",float(err.max()),"opt",rec.get("kind"),rec.get("lr"),rec.get("mom"),rec.get("wd"),"lists",rec.get("n_lists"))
G=torch.zeros_like(ent[0]); G[rec["rows"]]=rec["sums"]
print("max |HIP summed grad - oracle grad|", float((G-t0.grad[0]).abs().max()))
for row,w in bad[:8].tolist():
    print(row,w,"before",float(rec["before"][row,w]),"ent",float(ent[0,row,w]),"G_hip",float(G[row,w]),"g_oracle",float(t0.grad[0,row,w]),
          "got",float(got[row,w]),"want",float(want_ent[row,w]),"ulp",float(ulp[row,w]), "fp32 result", float(ent[0,row,w]-lr*t0.grad[0,row,w]))
