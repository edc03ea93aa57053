"""torch-native ops (not our C-ABI kernels) in one training step, by op and input shapes, with their GPU time"""
import os, sys, collections, torch
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
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    dt = getattr(ev, 'device_time_total', None) or getattr(ev, 'cuda_time_total', 0)
    if dt <= 0 or not ev.key.startswith('aten::'): continue
    rows.append((dt, ev.count, ev.key, str(ev.input_shapes)[:110]))
rows.sort(reverse=True)
for dt, c, k, sh in rows[:40]:
    print('%8.1f us  n=%4d  %-22s %s' % (dt, c, k, sh))
