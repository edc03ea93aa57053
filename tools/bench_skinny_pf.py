"""small-M kernel (gemm_skinny.hip): prefetch depth A/B on the shapes it runs (run with GPV_TUNING_LIB=1 GPV_SKINNY_PF=1|2|3|4)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpv1_amd.hip as hip
from bench_attn import timeit
dev = 'cuda'
print('GPV_SKINNY_PF', os.environ.get('GPV_SKINNY_PF'))
for (M, N, K, tr, what) in [(300, 256, 2048, 0, 'batch-1 DETR ffn2'), (100, 768, 3072, 0, 'batch-1 coatt ffn2'), (100, 768, 768, 0, 'batch-1 coatt proj'), (300, 256, 256, 0, 'batch-1 enc proj'),
                            (640, 768, 768, 0, 'text out/q'), (640, 2304, 768, 0, 'text qkv'), (640, 2048, 768, 0, 'text ffn1'), (640, 768, 2048, 0, 'text ffn2'),
                            (192, 768, 768, 0, 'bert proj'), (192, 3072, 768, 0, 'bert ffn1'), (192, 768, 3072, 0, 'bert ffn2'),
                            (640, 768, 2048, 1, 'text ffn1 dgrad'), (640, 768, 2304, 1, 'text qkv dgrad'), (3200, 256, 2048, 1, 'dec ffn1 dgrad')]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tr else (N, K), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16); b = torch.randn(N, device=dev)
    if tr:
        f = lambda: hip.gemm(A, B, C, M, N, K, K, N, N, layoutB=hip.TRANS)
    else:
        f = lambda: hip.gemm(A, B, C, M, N, K, K, K, N, bias=b)
    hip.set_option(hip.OPT_SKINNY_LAUNCHES, 0) if hasattr(hip, 'OPT_SKINNY_LAUNCHES') else None
    t = timeit(f)
    print('%-20s M=%5d N=%5d K=%5d %s  %6.1f us  %5.0f TF/s' % (what, M, N, K, 'T' if tr else 'K', t, 2.0 * M * N * K / t / 1e6), flush=True)
