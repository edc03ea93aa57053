"""linear weight-gradient GEMM (TRANS x TRANS, accumulate): tile config x split sweep on the step's shapes"""
import os, sys, subprocess
if len(sys.argv) == 1:
    for t in (1, 2, 3):
        print('== tile', ['128x128', '128x64', '64x64'][t - 1]); sys.stdout.flush()
        subprocess.run([sys.executable, __file__, 'run'], env=dict(os.environ, GPV_FORCE_TILE=str(t)))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
# (N_out, K_in, M_tokens, calls/step)
for (N, K, M, calls) in [(256, 256, 9600, 24), (768, 768, 3200, 12), (256, 256, 3200, 24), (256, 2048, 9600, 7), (2048, 256, 9600, 6), (512, 256, 9600, 6),
                         (1536, 768, 3392, 3), (768, 3072, 3200, 3), (3072, 768, 3200, 3), (768, 768, 640, 10), (256, 2048, 3200, 6), (2048, 256, 3200, 6), (768, 768, 192, 13), (768, 10000, 640, 1)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    res = []
    for split in (1, 2, 4, 8, 16, 32):
        if split > max(1, M // 64): continue
        def run():
            hip.gemm(dy, x, dw, N, K, M, N, K, K, layoutA=hip.TRANS, layoutB=hip.TRANS, accumulate=True, split_k=split, a_rowsum=db)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append((split, e0.elapsed_time(e1) * 50))
    print('out %5dx%5d red %5d x%2d: ' % (N, K, M, calls) + '  '.join('s%d:%.1f' % r for r in res))
