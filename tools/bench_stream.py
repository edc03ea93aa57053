import torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
from bench_attn import timeit
for M, C in ((614400, 256), (153600, 512), (614400, 64)):
    a = torch.randn(M, C, device='cuda').to(torch.bfloat16); b = torch.randn_like(a); c = torch.empty_like(a)
    t1 = timeit(lambda: torch.add(a, b, out=c), 30); t2 = timeit(lambda: c.copy_(a), 30); t3 = timeit(lambda: torch.relu_(c), 30)
    mb = M * C * 2 / 1e6
    print('%d x %d (%.0f MB): add(2R+1W) %.1f us %.2f TB/s | copy(1R+1W) %.1f us %.2f TB/s | relu_ inplace(1R+1W) %.1f us %.2f TB/s' % (M, C, mb, t1, 3 * mb / t1, t2, 2 * mb / t2, t3, 2 * mb / t3))
