import os, sys, subprocess
if len(sys.argv) == 1:
    for t in (1, 2, 3):
        env = dict(os.environ, GPV_FORCE_TILE=str(t))
        print('== tile cfg', ['128x128', '128x64', '64x64'][t - 1]); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, 'run'], env=env)
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
for (M, N, K) in [(9600, 256, 256), (9600, 512, 256), (9600, 2048, 256), (9600, 256, 2048), (3200, 256, 256), (3200, 768, 768),
                  (3200, 3072, 768), (3200, 768, 3072), (3200, 768, 2304), (640, 768, 768), (640, 2304, 768), (608, 10000, 768), (10000, 768, 768)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.randn(N, device=dev)
    def run(): hip.gemm(a, w, c, M, N, K, K, K, N, bias=bias)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print('M=%5d N=%5d K=%4d  %6.1f us  %6.1f TF/s' % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
