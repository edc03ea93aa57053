"""does a training step leave reference cycles behind?  (memory growth and collectable objects with the cyclic GC off)"""
import os, sys, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list
from gpv1_amd.train import FlatTrainer
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5)
images, mask, ids, attn, targets = bench.make_batch(0, 32, dev)
samples = nested_tensor_from_tensor_list(images)
step = lambda: tr.train_step(samples, (ids, attn), [dict(t) for t in targets])
for _ in range(3): step()
torch.cuda.synchronize(); gc.collect(); gc.disable()
m0 = torch.cuda.memory_allocated()
for i in range(30):
    step()
    if i % 10 == 9:
        torch.cuda.synchronize()
        print('step', i + 1, 'allocated MB', (torch.cuda.memory_allocated() - m0) / 2**20, 'gc objects pending', gc.get_count())
n = gc.collect()
torch.cuda.synchronize()
print('collected', n, 'objects; allocated MB after collect', (torch.cuda.memory_allocated() - m0) / 2**20)
