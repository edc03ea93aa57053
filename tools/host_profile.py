import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpv1_amd.gpv import GPV
dev = 'cuda:0'
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev)
lin = model.text_decoder.layers[0].linear1
x = torch.randn(640, 768, device=dev).to(torch.bfloat16).requires_grad_(True)
for _ in range(10): lin(x)
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): y = lin(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
