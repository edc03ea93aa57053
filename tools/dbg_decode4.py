"""Which pointers do the captured decode graphs hold, and are they still live allocations afterwards?"""
import os, sys, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor
import gpv1_amd.hip as hip
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
rec = {}
orig_p = hip._p
import traceback
def p(t):
    if t is not None and torch.cuda.is_current_stream_capturing():
        st = traceback.extract_stack(limit=6)
        rec.setdefault(t.data_ptr(), (t.numel() * t.element_size(), ' <- '.join(f'{f.name}:{f.lineno}' for f in st[:-1][::-1][:4])))
    return orig_p(t)
hip._p = p
Bd = int(os.environ.get('B', 64))
images, mask, ids, attn, _ = bench.make_batch(7, Bd, dev)
with torch.no_grad():
    o = model(NestedTensor(images, mask), (ids, attn), None, None)
torch.cuda.synchronize()
del o
import gc; gc.collect()
snap = torch.cuda.memory_snapshot()
blocks = []
for seg in snap:
    a = seg['address']
    for b in seg['blocks']:
        blocks.append((a, a + b['size'], b['state'], seg.get('segment_pool_id', None)))
        a += b['size']
blocks.sort()
import bisect
starts = [b[0] for b in blocks]
bad = 0
stat = {}
for ptr, (nb, where) in rec.items():
    i = bisect.bisect_right(starts, ptr) - 1
    if i < 0 or not (blocks[i][0] <= ptr < blocks[i][1]):
        print('NOT IN ANY BLOCK', hex(ptr), nb, where); bad += 1; continue
    s, e, state, pool = blocks[i]
    key = (state, str(pool))
    stat[key] = stat.get(key, 0) + 1
    if state != 'active_allocated' and tuple(pool or (0, 0)) == (0, 0):
        print('STALE main-pool pointer', hex(ptr), nb, state, where); bad += 1
print('pointers recorded', len(rec), 'stats', stat, 'bad', bad)
