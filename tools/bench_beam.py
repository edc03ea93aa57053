"""BASELINE configs[3]: beam_size 5 decode of a 480x640 batch of 64, KV-cached vs the reference's full-prefix schedule"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
B = int(os.environ.get('B', 64))
images, mask, ids, attn, _ = bench.make_batch(7, B, dev)
samples = nested_tensor_from_tensor_list(images)
with torch.no_grad():
    for kv in (True, False):
        model.cfg['kv_decode'] = kv
        for _ in range(2): out = model.forward_beam_search(samples, (ids, attn), beam_size=5)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out = model.forward_beam_search(samples, (ids, attn), beam_size=5)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print('beam 5, B=%d, %s: %.1f ms per batch, %.2f ms per image; first answer %s p=%.3g' % (B, 'KV-cached' if kv else 'full prefix', ms, ms / B, ' '.join(out['answers'][0][0][:4]), out['answer_probs'][0][0]))
