import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor
from gpv1_amd.train import FlatTrainer
import gpv1_amd.decode as dec
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5)
Bt = int(os.environ.get('BT', 2))
images, mask, ids, attn, targets = bench.make_batch(0, Bt, dev)
for _ in range(int(os.environ.get('STEPS', 1))):
    loss = tr.train_step(NestedTensor(images, mask), (ids, attn), [dict(t) for t in targets])
torch.cuda.synchronize(); print('trained', flush=True)
model.eval()
if os.environ.get('NOGC') == '1':
    import gc; gc.disable()
if os.environ.get('GPV_NO_GRAPHS') == '1': model.cfg['kv_graphs'] = False
if os.environ.get('NO_KV') == '1': model.cfg['kv_decode'] = False
orig = torch.cuda.CUDAGraph.replay
def replay(self):
    orig(self); torch.cuda.synchronize(); print('replayed', flush=True)
if os.environ.get('SYNC', '1') == '1': torch.cuda.CUDAGraph.replay = replay
for Bd in [int(b) for b in os.environ.get('BDS', '1,64').split(',')]:
    images, mask, ids, attn, _ = bench.make_batch(7, Bd, dev)
    with torch.no_grad():
        for it in range(int(os.environ.get('ITERS', 2))):
            o = model(NestedTensor(images, mask), (ids, attn), None, None)
            if os.environ.get('SYNC', '1') == '1' or os.environ.get('SYNC_DECODE') == '1': torch.cuda.synchronize()
            if os.environ.get('SYNC_STREAM') == '1': torch.cuda.current_stream().synchronize()
            k = int(os.environ.get('SYNC_EVERY', 0))
            if k and it % k == k - 1: torch.cuda.current_stream().synchronize()
            print('decode issued', Bd, it, flush=True)
        torch.cuda.synchronize(); print('decode ok', Bd, flush=True)
