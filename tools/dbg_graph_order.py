"""Does work queued after a graph replay on the same stream wait for the graph?  (event after replay vs device sync)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
Bd = 64
images, mask, ids, attn, _ = bench.make_batch(7, Bd, dev)
with torch.no_grad():
    for it in range(4):
        t0 = time.perf_counter()
        o = model(NestedTensor(images, mask), (ids, attn), None, None)
        t1 = time.perf_counter()
        ev = torch.cuda.Event(); ev.record(); ev.synchronize()
        t2 = time.perf_counter()
        torch.cuda.current_stream().synchronize()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f'iter {it}: issue {1e3*(t1-t0):.1f} ms, event wait {1e3*(t2-t1):.1f} ms, then stream sync {1e3*(t3-t2):.1f} ms, then device sync {1e3*(t4-t3):.1f} ms', flush=True)
print('stream handle', torch.cuda.current_stream().cuda_stream)
