"""host<->device synchronisation points inside one training step (torch.cuda.set_sync_debug_mode)"""
import os, sys, warnings, traceback, torch
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
seen = []
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if 'gpv-1_amd' in f.filename]
    where = '%s:%d %s' % (os.path.basename(st[-1].filename), st[-1].lineno, st[-1].line) if st else '%s:%d' % (filename, lineno)
    seen.append(where)
warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step()
torch.cuda.set_sync_debug_mode('default')
import collections
for w, c in collections.Counter(seen).most_common(): print(c, w)
