"""the DETR feed-forward GEMM 9600 x 256 -> 2048 (bias + ReLU + dropout) under rocprofv3 --pmc: streaming kernel (default) or GPV_C1S_LINEAR=0 tile kernel; 4 launches"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'
M, F, D = 9600, 2048, 256
x = torch.randn(M, D, device=dev).to(torch.bfloat16); w1 = (torch.randn(F, D, device=dev) / 16).to(torch.bfloat16); b1 = torch.randn(F, device=dev)
h = torch.empty(M, F, device=dev, dtype=torch.bfloat16)
for _ in range(4):
    hip.gemm(x, w1, h, M, F, D, D, D, F, bias=b1, act=hip.ACT_RELU, drop_p=0.1, seed=3)
torch.cuda.synchronize()
