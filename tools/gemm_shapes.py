"""every gpv_gemm call of one EAGER B = 32 training step of the bench model (GPV_TRAIN_GRAPHS=0), with its epilogue, as distinct
(M, N, K, layoutA, layoutB, batch, accumulate, res, mask, act, dropout, out dtype) rows + counts -> tools/gemm_shapes_step.json
usage (GPU box): GPV_TRAIN_GRAPHS=0 python tools/gemm_shapes.py [out.json]"""
import os, sys, json, collections
os.environ['GPV_TRAIN_GRAPHS'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gpv1_amd.hip as hip
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list
from gpv1_amd.train import FlatTrainer
from gpv1_amd.ops import RT

dev = torch.device('cuda:0')
hip.lib()
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
RT.manual_seed(1000)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, clip_max_norm=0.1, warmup_steps=100, t_total=1000)
images, mask, ids, attn, targets = bench.make_batch(0, bench.BATCH, dev)
samples = nested_tensor_from_tensor_list(images)
tr.train_step(samples, (ids, attn), [dict(t) for t in targets])
torch.cuda.synchronize()
seen = collections.Counter()
orig = hip.gemm


def rec(A, B, Cm, M, N, K, lda, ldb, ldc, layoutA=0, layoutB=0, batch=1, sA=0, sB=0, sC=0, alpha=1.0, rowscale=None, bias=None, res=None, ldr=0,
        sR=0, relu_mask=None, ldm=0, act=0, drop_p=0.0, seed=0, accumulate=False, split_k=1, a_rowsum=None, **kw):
    seen[(M, N, K, int(layoutA), int(layoutB), batch, int(bool(accumulate)), int(res is not None), int(relu_mask is not None), int(act),
          int(drop_p > 0), int(Cm.dtype == torch.float32), int(lda != (M if layoutA else K)), int(ldb != (N if layoutB else K)), int(ldc != N))] += 1
    return orig(A, B, Cm, M, N, K, lda, ldb, ldc, layoutA=layoutA, layoutB=layoutB, batch=batch, sA=sA, sB=sB, sC=sC, alpha=alpha, rowscale=rowscale,
                bias=bias, res=res, ldr=ldr, sR=sR, relu_mask=relu_mask, ldm=ldm, act=act, drop_p=drop_p, seed=seed, accumulate=accumulate,
                split_k=split_k, a_rowsum=a_rowsum, **kw)


hip.gemm = rec
tr.train_step(samples, (ids, attn), [dict(t) for t in targets])
torch.cuda.synchronize()
hip.gemm = orig
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_shapes_step.json')
rows = sorted(([list(k), v] for k, v in seen.items()), key=lambda r: r[0])
json.dump(rows, open(out, 'w'))
print('%d distinct gpv_gemm shapes, %d calls' % (len(rows), sum(seen.values())))
for k, v in rows:
    print(v, k)
