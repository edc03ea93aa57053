"""Batch-1 greedy inference, everything BEFORE the token loop (backbone, DETR, RoI head, co-attention: ~250 dependent nodes of the
inference graph, ~2.3 of the 4.4 ms per image): every distinct gpv_gemm shape under every kernel family, timed AS NODES OF A hipGraph
CHAIN (stream launches from Python would time the host; a node's floor here is 1.7 us -- tools/probe_decode_nodes.py), and every
gpv_conv2d call with its default kernel.  usage (GPU box): python tools/tune_gemms_bs1.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gpv1_amd.hip as hip
from gpv1_amd.gpv import GPV
from gpv1_amd.misc import nested_tensor_from_tensor_list

dev = torch.device('cuda:0')
hip.lib()
torch.manual_seed(0)
model = GPV(bench.make_cfg()).to(dev).eval()
images, mask, ids, attn, _ = bench.make_batch(7, 1, dev)
s = nested_tensor_from_tensor_list(images)
gemms, convs = collections.OrderedDict(), collections.OrderedDict()
og, oc = hip.gemm, hip.conv2d


def rec_g(A, B, Cm, M, N, K, lda, ldb, ldc, **kw):
    if M > 8:
        key = (M, N, K, int(kw.get('layoutA', 0)), int(kw.get('layoutB', 0)), kw.get('batch', 1), int(kw.get('res') is not None), int(kw.get('act', 0)),
               int(Cm.dtype == torch.float32), int(kw.get('bias') is not None), lda, ldb, ldc)
        gemms[key] = gemms.get(key, 0) + 1
    return og(A, B, Cm, M, N, K, lda, ldb, ldc, **kw)


def rec_c(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, **kw):
    key = (mode, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, int(kw.get('res') is not None), int(kw.get('act', 0)))
    convs[key] = convs.get(key, 0) + 1
    return oc(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, **kw)


with torch.no_grad(), hip.gemm_flags(hip.GEMM_NO_PIPE_SMALL):
    model._forward_impl(s, (ids, attn), None, None, None, kv_graphs=False)          # warm-up (weight copies)
    torch.cuda.synchronize()
    hip.gemm, hip.conv2d = rec_g, rec_c
    model._forward_impl(s, (ids, attn), None, None, None, kv_graphs=False)
    torch.cuda.synchronize()
    hip.gemm, hip.conv2d = og, oc


def chain(f, n=100):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        f(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(n):
                f()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); gr.replay(); e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1000.0 / n


FAM = [('pipe off', hip.OPT_PIPE, 0)] + [('pipe cfg %d' % i, hip.OPT_PIPE, 100 + i) for i in range(8)] + \
      [('glds off', hip.OPT_GLDS, 0), ('glds 8-wave', hip.OPT_GLDS, 2), ('glds 4-wave', hip.OPT_GLDS, 3), ('skinny off', hip.OPT_SKINNY, 0), ('skinny all', hip.OPT_SKINNY, 2)]
bf = torch.bfloat16
print('# gpv_gemm shapes with M > 8 of one batch-1 inference before the token loop: us per graph node')
tot_d = tot_b = 0.0
for (M, N, K, la, lb, batch, has_res, act, f32o, has_bias, lda, ldb, ldc), cnt in ({} if os.environ.get('TUNE_SKIP_GEMM') else gemms).items():
    if batch != 1:
        print('%3d x M=%5d N=%5d K=%5d batch %d  (batched: skipped)' % (cnt, M, N, K, batch)); continue
    A = torch.randn(K * lda if la else M * lda, device=dev).to(bf)
    B = (torch.randn(K * ldb if lb else N * ldb, device=dev) / K ** 0.5).to(bf)
    Cm = torch.empty(M * ldc, device=dev, dtype=torch.float32 if f32o else bf)
    kw = dict(layoutA=la, layoutB=lb, act=act)
    if has_bias: kw['bias'] = torch.randn(N, device=dev)
    if has_res: kw.update(res=torch.randn(M * ldc, device=dev).to(Cm.dtype), ldr=ldc)
    run = lambda: hip.gemm(A, B, Cm, M, N, K, lda, ldb, ldc, **kw)
    with hip.gemm_flags(hip.GEMM_NO_PIPE_SMALL):
        t0 = chain(run)
        best, bname = t0, 'default'
        for name, opt, val in FAM:
            prev = hip.set_option(opt, val)
            try:
                t = chain(run)
            except RuntimeError:
                t = float('inf')
            hip.set_option(opt, prev)
            if t < best * 0.95:
                best, bname = t, name
    with hip.option(hip.OPT_PIPE_SMALL, 1):
        ts = chain(run)
    if ts < best * 0.95:
        best, bname = ts, 'pipe small-M configurations'
    tot_d += t0 * cnt; tot_b += best * cnt
    flag = '' if bname == 'default' else '   <-- %s %.2f us (x%d: %.0f us per image)' % (bname, best, cnt, (t0 - best) * cnt)
    print('%3d x M=%5d N=%5d K=%5d %s%s res%d act%d f32out%d ld(%d,%d,%d)  default %6.2f us%s' % (cnt, M, N, K, 'T' if la else 'K', 'T' if lb else 'K', has_res, act, f32o, lda, ldb, ldc, t0, flag), flush=True)
print('gemm total per image: default %.0f us, best-of-families %.0f us' % (tot_d, tot_b))
print('# gpv_conv2d calls (default kernels): us per graph node')
tot = 0.0
for (mode, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, has_res, act), cnt in convs.items():
    x = torch.randn(B * IH * IW * Cs, device=dev).to(bf)
    w = (torch.randn(Cout * KH * KW * Cin, device=dev) / (KH * KW * Cin) ** 0.5).to(bf)
    y = torch.empty(B * OH * OW * Cout, device=dev, dtype=bf)
    kw = dict(bias=torch.zeros(Cout, device=dev), act=act)          # as backbone._conv_fwd calls it: the BatchNorm scale is in the weight copy (rowscale is per output PIXEL in modes 0 / 1)
    if has_res: kw['res'] = torch.randn(B * OH * OW * Cout, device=dev).to(bf)
    run = lambda: hip.conv2d(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, **kw)
    print('  -> mode %d B %d in %dx%dx%d(stride %d) out %dx%dx%d k %dx%d s %dx%d p %dx%d res%d act%d' % (mode, B, IH, IW, Cin, Cs, OH, OW, Cout, KH, KW, SH, SW, PH, PW, has_res, act), flush=True)
    try:
        t = chain(run)
    except RuntimeError as e:
        print('conv', (IH, IW, Cin, Cout, KH, SH), 'failed', str(e)[:60]); continue
    tot += t * cnt
    gf = 2.0 * B * OH * OW * Cout * KH * KW * Cin / 1e9
    print('%3d x %3dx%3d %4d -> %4d %dx%d/%d res%d act%d  %6.2f us  (%.2f GFLOP: %.0f TF/s)' % (cnt, IH, IW, Cin, Cout, KH, KW, SH, has_res, act, t, gf, gf / t * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 if False else gf / t / 1e-6 / 1e12 * 1e9), flush=True)
print('conv2d total per image: %.0f us' % tot)
