"""which lines of the package issue torch-native GPU ops (copies, adds, cats...) during one training step
(python-level interception: counts calls on CUDA tensors by the innermost package frame)"""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor
from gpv1_amd.train import FlatTrainer
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5)
images, mask, ids, attn, targets = bench.make_batch(0, 8, dev)
step = lambda: tr.train_step(NestedTensor(images, mask), (ids, attn), [dict(t) for t in targets])
for _ in range(2): step()
torch.cuda.synchronize()
cnt = collections.Counter()
def where():
    for f in reversed(traceback.extract_stack(limit=14)[:-2]):
        if 'gpv-1_amd' in f.filename:
            return '%s:%d %s' % (os.path.basename(f.filename), f.lineno, (f.line or '')[:70])
    return 'outside package'
def wrap(owner, name):
    orig = getattr(owner, name)
    def inner(*a, **k):
        t = a[0] if a else None
        if isinstance(t, (list, tuple)) and t: t = t[0]
        if torch.is_tensor(t) and t.is_cuda:
            cnt[(name, where())] += 1
        return orig(*a, **k)
    setattr(owner, name, inner)
for n in ('copy_', 'contiguous', 'to', 'clone', '__add__', '__radd__', '__iadd__', 'add', 'add_', '__mul__', 'float', 'zero_', 'fill_', '__getitem__'):
    wrap(torch.Tensor, n)
for n in ('cat', 'stack', 'zeros', 'zeros_like'):
    wrap(torch, n)
step(); torch.cuda.synchronize()
for (name, frame), c in cnt.most_common(60):
    print('%4d  %-12s %s' % (c, name, frame))
