"""host-only cost of a training step: every C-ABI call replaced by a no-op, so what remains is Python + torch dispatch"""
import os, sys, time, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor, nested_tensor_from_tensor_list
from gpv1_amd.train import FlatTrainer
import gpv1_amd.hip as hip
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5)
images, mask, ids, attn, targets = bench.make_batch(0, 32, dev)
samples = nested_tensor_from_tensor_list(images)
step = lambda: tr.train_step(samples, (ids, attn), [dict(t) for t in targets])
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print('real step        %.2f ms' % ((time.perf_counter() - t0) / 5 * 1e3))
class Stub:
    def __init__(self, real): self.real = real
    def __getattr__(self, n):
        if n in ('gpv_abi_version', 'gpv_set_option'): return getattr(self.real, n)
        return lambda *a: 0
real = hip.lib()
hip._LIB = Stub(real)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize()
print('kernels stubbed  %.2f ms host issue per step (no GPU work from our kernels)' % ((t1 - t0) / 5 * 1e3))
# split forward / backward
def fwd_only():
    model.train()
    tg = [dict(t) for t in targets]
    _, a = model.encode_answers(tg)
    for i, t in enumerate(tg): t['answer_token_ids'] = a[i, 1:]
    return model(samples, (ids, attn), a, tg)
t0 = time.perf_counter()
for _ in range(5): loss = fwd_only()
t1 = time.perf_counter()
print('  forward issue  %.2f ms' % ((t1 - t0) / 5 * 1e3))
loss = fwd_only(); t0 = time.perf_counter(); loss.backward(); t1 = time.perf_counter()
print('  backward issue %.2f ms' % ((t1 - t0) * 1e3))
