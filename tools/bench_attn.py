"""attention core timings (forward, backward) on the model's shapes at B=32: encoder self (300x300, 8 x 32), decoder cross (100x300) /
self (100x100), co-attention (16 heads x 48: 100x6 / 6x100), text decoder (8 x 96: 20x20 causal, 20x106), BERT (12 x 64).  usage: python tools/bench_attn.py"""
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
    tf = timeit(f)
    prev = hip.set_option(hip.OPT_ATTN_BWD1, 0)
    tb2 = timeit(b)                                   # dQ + dK/dV launches
    hip.set_option(hip.OPT_ATTN_BWD1, 2)
    hip.set_option(hip.OPT_ATTN_BWD1_LAUNCHES, 0)
    tb = timeit(b)                                    # single launch (where the shape is instantiated)
    n1 = hip.set_option(hip.OPT_ATTN_BWD1_LAUNCHES, 0)
    hip.set_option(hip.OPT_ATTN_BWD1, prev)
    fl = 4.0 * B * H * Sq * Sk * dh
    print('%-28s fwd %6.1f us (%5.1f TF)  bwd %6.1f us (%5.1f TF)  [two launches %6.1f us%s]' %
          (name, tf, fl / tf / 1e6, tb, 2.5 * fl / tb / 1e6, tb2, '' if n1 else '; single launch not taken'), flush=True)


def main():
    import sys as _s
    if len(_s.argv) > 1:
        case('enc self 300x300 dh32', 32, 8, 300, 300, 32, 0.1); _s.exit(0)
    case('enc self 300x300 dh32', 32, 8, 300, 300, 32, 0.1)
    case('enc self 300x300 kpm', 32, 8, 300, 300, 32, 0.1, kpm=True)
    case('enc self 300x300 p=0', 32, 8, 300, 300, 32, 0.0)
    case('dec cross 100x300 dh32', 32, 8, 100, 300, 32, 0.1)
    case('dec self 100x100 dh32', 32, 8, 100, 100, 32, 0.1)
    # co-attention: 16 heads x dh 48 (configs/exp/gpv.yaml:70 bi_num_attention_heads 16, bi_hidden 768); T_l = 6 query tokens (bench), 16 (longest class)
    case('coatt vis->lang 100x6 dh48', 32, 16, 100, 6, 48, 0.1)
    case('coatt lang->vis 6x100 dh48', 32, 16, 6, 100, 48, 0.1)
    case('coatt vis->lang 100x16 dh48 kpm', 32, 16, 100, 16, 48, 0.1, kpm=True)
    # text decoder: 8 heads x dh 96 (gpv.yaml text_decoder.nheads 8, hidden 768); S = 20 tokens, memory = 100 + T_l
    case('text cross 20x106 dh96', 32, 8, 20, 106, 96, 0.1)
    case('text self 20x20 causal dh96', 32, 8, 20, 20, 96, 0.1, causal=True)
    case('bert 6x6 kpm dh64', 32, 12, 6, 6, 64, 0.1, kpm=True)

if __name__ == '__main__':
    main()
