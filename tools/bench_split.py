import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
for (M, N, K) in [(9600, 512, 256), (9600, 256, 256), (3200, 256, 256), (3200, 768, 768), (9600, 2048, 256), (640, 768, 768), (640, 2304, 768), (192, 768, 768)]:
    dy = torch.randn(M, N, device=dev).to(torch.bfloat16); x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    res = []
    for split in (1, 2, 4, 8, 16, 32, 64):
        if split > (M + 31) // 32: continue
        def run():
            hip.gemm(dy, x, dw, N, K, M, N, K, K, layoutA=hip.TRANS, layoutB=hip.TRANS, accumulate=True, split_k=split)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        res.append((split, e0.elapsed_time(e1) * 50))
    print('out %5dx%4d red %5d: ' % (N, K, M) + '  '.join('s%d:%.1fus' % r for r in res))
