"""greedy decode at batch 1 (whole inference as one hipGraph): A/B of GreedyKVDecoder.embed_in_pick inside ONE process, alternating
(boxes and processes differ by +-1.5 %).  usage: python tools/ab_decode.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list
import gpv1_amd.decode as decode

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
images, mask, ids, attn, _ = bench.make_batch(7, 1, dev)
samples = nested_tensor_from_tensor_list(images)


def measure(flag, iters=20):
    os.environ['GPV_DECODE_EMBED'] = flag
    model._igraphs.clear(); model._kvdec.clear()
    with torch.no_grad():
        for _ in range(3):
            model(samples, (ids, attn), None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model(samples, (ids, attn), None, None)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for rnd in range(3):
    for flag in ('1', '0'):
        print('GPV_DECODE_EMBED=%s  %.3f ms per image' % (flag, measure(flag)), flush=True)
