"""attention core timings (forward, backward) on the model's shapes at B=32: encoder self (300x300), decoder cross (100x300),
decoder self (100x100), co-attention (100x24 / 24x100), text (20x20 causal).  usage: python tools/bench_attn.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpv1_amd.hip as hip
dev = 'cuda'


def timeit(run, n=50):
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


def case(name, B, H, Sq, Sk, dh, p, causal=False, kpm=False):
    D = H * dh
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * Sq, D, generator=g).to(dev).to(torch.bfloat16)
    k = torch.randn(B * Sk, D, generator=g).to(dev).to(torch.bfloat16)
    v = torch.randn(B * Sk, D, generator=g).to(dev).to(torch.bfloat16)
    do = torch.randn(B * Sq, D, generator=g).to(dev).to(torch.bfloat16)
    o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    lse = torch.empty(B, H, Sq, device=dev, dtype=torch.float32)
    m = torch.zeros(B, Sk, dtype=torch.uint8, device=dev) if kpm else None
    st = ((Sq * D, D), (Sk * D, D), (Sk * D, D), (Sq * D, D))
    sc = 1 / math.sqrt(dh)
    f = lambda: hip.attention_fwd(q, k, v, o, st, B, H, Sq, Sk, dh, sc, kpm=m, causal=causal, drop_p=p, seed=11, lse=lse)
    b = lambda: hip.attention_bwd(q, k, v, o, do, dq, dk, dv, st, (Sq * D, D), B, H, Sq, Sk, dh, sc, kpm=m, causal=causal, drop_p=p, seed=11, lse=lse)
    tf, tb = timeit(f), timeit(b)
    fl = 4.0 * B * H * Sq * Sk * dh
    print('%-28s fwd %6.1f us (%5.1f TF)  bwd %6.1f us (%5.1f TF)' % (name, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6), flush=True)


def main():
    import sys as _s
    if len(_s.argv) > 1:
        case('enc self 300x300 dh32', 32, 8, 300, 300, 32, 0.1); _s.exit(0)
    case('enc self 300x300 dh32', 32, 8, 300, 300, 32, 0.1)
    case('enc self 300x300 kpm', 32, 8, 300, 300, 32, 0.1, kpm=True)
    case('enc self 300x300 p=0', 32, 8, 300, 300, 32, 0.0)
    case('dec cross 100x300 dh32', 32, 8, 100, 300, 32, 0.1)
    case('dec self 100x100 dh32', 32, 8, 100, 100, 32, 0.1)
    case('coatt 24x100 dh96', 32, 8, 24, 100, 96, 0.1)
    case('coatt 100x24 dh96', 32, 8, 100, 24, 96, 0.1)
    case('text cross 20x124 dh64', 32, 12, 20, 124, 64, 0.1)
    case('text self 20x20 causal dh64', 32, 12, 20, 20, 64, 0.1, causal=True)
    case('bert 24x24 kpm dh64', 32, 12, 24, 24, 64, 0.1, kpm=True)


if __name__ == '__main__':
    main()
