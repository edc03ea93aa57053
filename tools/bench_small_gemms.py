import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import gpv1_amd.hip as hip
from bench_attn import timeit
dev='cuda'
for (M,N,K,what) in [(640,2304,768,'text self qkv'),(640,768,768,'text out/q'),(640,2048,768,'text ffn1'),(640,768,2048,'text ffn2'),(640,10000,768,'logits'),
                     (192,768,768,'coatt text'),(192,2304,768,'coatt text qkv'),(192,3072,768,'text ffn1 bert-like'),(3392,1536,768,'kv hoist text'),(3200,768,768,'coatt vis'),(3200,2304,768,'coatt vis qkv'),(3200,3072,768,'coatt ffn1'),(3200,768,3072,'coatt ffn2'),
                     (9600,256,256,'enc out'),(9600,512,256,'enc qk'),(3200,256,256,'dec'),(9600,256,2048,'ffn2')]:
    A=torch.randn(M,K,device=dev).to(torch.bfloat16); B=torch.randn(N,K,device=dev).to(torch.bfloat16); C=torch.empty(M,N,device=dev,dtype=torch.bfloat16); b=torch.randn(N,device=dev)
    t=timeit(lambda: hip.gemm(A,B,C,M,N,K,K,K,N,bias=b))
    print('%-22s M=%5d N=%5d K=%5d  %6.1f us  %5.0f TF/s' % (what,M,N,K,t,2.0*M*N*K/t/1e6), flush=True)
