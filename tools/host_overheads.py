"""host-side cost of the per-step bookkeeping around the kernels (run on the GPU box)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import NestedTensor
from gpv1_amd.train import FlatTrainer
import gpv1_amd.hip as hip
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5)
images, mask, ids, attn, targets = bench.make_batch(0, 32, dev)
def T(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): tr.train_step(NestedTensor(images, mask), (ids, attn), [dict(t) for t in targets])
print('model.train()            %.3f ms' % T(lambda: model.train()))
print('encode_answers           %.3f ms' % T(lambda: model.encode_answers([dict(t) for t in targets])))
print('zero_grad                %.3f ms' % T(lambda: tr.zero_grad()))
print('optimizer step           %.3f ms' % T(lambda: tr.step()))
A = torch.randn(192, 768, device=dev).to(torch.bfloat16); B = torch.randn(768, 768, device=dev).to(torch.bfloat16); Cm = torch.empty(192, 768, device=dev, dtype=torch.bfloat16)
t0 = time.perf_counter()
for _ in range(2000): hip.gemm(A, B, Cm, 192, 768, 768, 768, 768, 768)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('hip.gemm host issue      %.1f us/call' % ((t1 - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000): torch.empty(192, 768, device=dev, dtype=torch.bfloat16)
print('torch.empty              %.1f us/call' % ((time.perf_counter() - t0) / 2000 * 1e6))
t0 = time.perf_counter()
for _ in range(2000): torch.cuda.current_stream().cuda_stream
print('current_stream().handle  %.1f us/call' % ((time.perf_counter() - t0) / 2000 * 1e6))
import gpv1_amd.ops as ops
from gpv1_amd.ops import W
lin = model.text_decoder.layers[0].linear1
x = torch.randn(640, 768, device=dev).to(torch.bfloat16)
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(1000): lin(x)
    t1 = time.perf_counter(); torch.cuda.synchronize()
print('LinearP fwd (no_grad)    %.1f us/call host' % ((t1 - t0) / 1000 * 1e6))
xg = x.clone().requires_grad_(True)
t0 = time.perf_counter()
for _ in range(1000): y = lin(xg)
t1 = time.perf_counter(); torch.cuda.synchronize()
print('LinearP fwd (autograd)   %.1f us/call host' % ((t1 - t0) / 1000 * 1e6))
