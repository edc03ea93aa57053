"""Soak of the inference graphs: 600 + 200 greedy inferences at batch 1 and 120 at batch 64 replayed back to back, outputs compared with the
first one every 50 (the ROCm graph-launch fault of decode.py's comment needs ~140 queued launches; the stream is drained per call).   (GPU box)"""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
import gpv1_amd.hip as hip
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list
hip.lib(); dev = torch.device('cuda:0'); torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
with torch.no_grad():
    for B, n in ((1, 600), (64, 120), (1, 200)):
        images, mask, ids, attn, _ = bench.make_batch(7, B, dev)
        s = nested_tensor_from_tensor_list(images)
        ref = None
        t0 = time.perf_counter()
        for i in range(n):
            o = model(s, (ids, attn), None, None)
            if ref is None: ref = o['answer_logits'].clone()
            elif i % 50 == 0: assert torch.equal(ref, o['answer_logits']), i
        torch.cuda.synchronize()
        print('B=%d: %d inferences, %.3f ms each, outputs identical' % (B, n, (time.perf_counter() - t0) / n * 1e3), flush=True)
