import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
model.cfg['kv_graphs'] = False
Bd = int(os.environ.get('B', 1))
images, mask, ids, attn, _ = bench.make_batch(7, Bd, dev)
with torch.no_grad():
    o = model(NestedTensor(images, mask), (ids, attn), None, None)
torch.cuda.synchronize()
print('ok', o['answer_logits'].shape)
