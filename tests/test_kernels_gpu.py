"""GPU parity tests, kernel level: every C-ABI entry point of libgpv_hip.so against fp32 math
(torch on the same device computing in fp32 from the SAME inputs; oracle functions where the op is
composite).  Tolerances: precise (fp32 I/O, split-bf16 MFMA) 3e-5 of max|ref|; bf16 mode 1.2e-2 of
max|ref| against fp32 math on the bf16-rounded inputs (one bf16 output rounding = 3.9e-3)."""
import math

import numpy as np

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'
TOL = {torch.float32: 3e-5, torch.bfloat16: 1.2e-2}


def hip():
    import gpv1_amd.hip as h
    h.lib()
    return h


def rel(a, ref):
    a, ref = a.float(), ref.float()
    return ((a - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


DTYPES = [torch.float32, torch.bfloat16]


# ----------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (2048, 512, 768), (130, 70, 96), (64, 2, 256), (100, 48, 300),
                                   (9600, 256, 2048), (33, 200, 40)])
def test_gemm_nt_plain(dtype, M, N, K):
    h = hip()
    A, B = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2)
    Cm = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, Cm, M, N, K, K, K, N)
    assert rel(Cm, A.float() @ B.float().t()) < TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K', [(1, 768, 768), (1, 2304, 768), (1, 2048, 768), (1, 768, 2048), (1, 10000, 768), (2, 768, 3072),
                                   (3, 100, 96), (5, 2049, 520), (8, 4099, 768), (6, 768, 768), (1, 3, 8)])
def test_gemm_few_rows(dtype, M, N, K):
    """M <= 8: the wave-per-column kernel of the decode step (csrc/gemv.hip), every epilogue term, strided output rows, the
    tile kernels on the same problem as a second reference"""
    h = hip()
    A, B = rnd(M, K, dtype=dtype, seed=21), rnd(N, K, dtype=dtype, seed=22)
    bias = rnd(N, seed=23)
    h.set_option(h.OPT_GEMV_LAUNCHES, 0)
    for act, fn in ((h.ACT_NONE, lambda x: x), (h.ACT_RELU, F.relu), (h.ACT_GELU, lambda x: F.gelu(x))):
        ldc = N + 24
        Cw = torch.full((M, ldc), 7.0, device=DEV, dtype=dtype)
        res = rnd(M, ldc, dtype=dtype, seed=24)
        h.gemm(A, B, Cw, M, N, K, K, K, ldc, alpha=0.75, bias=bias, res=res, ldr=ldc, act=act)
        ref = fn(0.75 * (A.float() @ B.float().t()) + bias + res[:, :N].float())
        assert rel(Cw[:, :N], ref) < TOL[dtype], act
        assert (Cw[:, N:] == 7.0).all()
        prev = h.set_option(h.OPT_GEMV, 0)
        C2 = torch.empty(M, N, device=DEV, dtype=dtype)
        try:
            h.gemm(A, B, C2, M, N, K, K, K, N, alpha=0.75, bias=bias, res=res, ldr=ldc, act=act)
        finally:
            h.set_option(h.OPT_GEMV, prev)
        assert rel(C2, ref) < TOL[dtype]
    Cp = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, Cp, M, N, K, K, K, N)
    assert rel(Cp, A.float() @ B.float().t()) < TOL[dtype]
    if dtype == torch.bfloat16:                       # fp32 output from bf16 inputs
        Cf = torch.empty(M, N, device=DEV)
        h.gemm(A, B, Cf, M, N, K, K, K, N)
        assert rel(Cf, A.float() @ B.float().t()) < 1e-5
    n = h.set_option(h.OPT_GEMV_LAUNCHES, 0)
    assert (n == 0) if (M > 2 and N >= 4096) else (n >= 4)        # (many columns x > 2 rows stay on the tile kernels)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,N,K', [(1, 768, 768), (1, 2304, 768), (1, 2048, 768), (1, 10000, 768), (2, 768, 256), (3, 100, 520),
                                      (4, 4099, 1024), (4, 5, 8)])
def test_ln_linear_rows_equals_layernorm_then_gemm(dtype, rows, N, K):
    """gpv_ln_linear_rows: bit-identical to gpv_layernorm_fwd + gpv_gemm (same lane layout, sums and rounding), and within the
    usual tolerance of fp32 math"""
    h = hip()
    x, s = rnd(rows, K, dtype=dtype, seed=41), rnd(rows, K, dtype=dtype, seed=42)
    g, b = 1.0 + 0.1 * rnd(K, seed=43), 0.1 * rnd(K, seed=44)
    Wm, bias = rnd(N, K, dtype=dtype, seed=45, scale=0.05), rnd(N, seed=46)
    for act, fn in ((h.ACT_NONE, lambda t: t), (h.ACT_RELU, F.relu)):
        for ss, bb in ((s, bias), (None, None)):
            ldy = N + 16
            xn, y = torch.empty_like(x), torch.full((rows, ldy), 5.0, device=DEV, dtype=dtype)
            h.ln_linear_rows(x, ss, g, b, 1e-5, xn, Wm, bb, y, ldy, rows, N, K, act)
            xn2, y2 = torch.empty_like(x), torch.empty(rows, N, device=DEV, dtype=dtype)
            h.layernorm_fwd(x, ss, g, b, xn2, None, None, rows, K, 1e-5)
            h.gemm(xn2, Wm, y2, rows, N, K, K, K, N, bias=bb, act=act)
            assert torch.equal(xn, xn2)
            if rows > 2 and N >= 4096:                # (the unfused GEMM of this shape runs on the tile kernels: another summation order)
                assert rel(y[:, :N], y2) < TOL[dtype]
            else:
                assert torch.equal(y[:, :N], y2)
            assert (y[:, N:] == 5.0).all()
            z = x.float() + (ss.float() if ss is not None else 0)
            ref = fn(F.layer_norm(z, (K,), g, b, 1e-5).to(dtype).float() @ Wm.float().t() + (bb if bb is not None else 0))
            assert rel(y[:, :N], ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,dh,Sk', [(1, 8, 96, 1), (1, 8, 96, 20), (1, 8, 96, 116), (3, 8, 96, 200), (4, 4, 32, 70), (2, 16, 48, 256)])
def test_attention_row_proj_and_partial_sum_consumer(dtype, B, H, dh, Sk):
    """gpv_attention_row_proj (one query row per sequence, out-projection folded in, per-head partial rows) against fp32 math and
    against attention_fwd + gemm on the same strided cache layout; gpv_ln_linear_rows summing the partials"""
    h = hip()
    D, T = H * dh, Sk + 3
    cache = rnd(B, T, 3 * D, dtype=dtype, seed=61)                        # q | k | v rows of a decoder cache
    t = Sk - 1
    q, k, v = cache[:, t], cache[:, :, D:], cache[:, :, 2 * D:]
    Wo, bo = rnd(D, D, dtype=dtype, seed=62, scale=dh ** -0.5), rnd(D, seed=63)
    part = torch.full((B, H, D), 9.0, device=DEV)
    scale = dh ** -0.5
    h.attention_row_proj(q, T * 3 * D, k, T * 3 * D, 3 * D, v, T * 3 * D, 3 * D, Wo, part, B, H, Sk, dh, scale)
    qh = cache[:, t, :D].float().reshape(B, H, dh)
    kh = cache[:, :Sk, D:2 * D].float().reshape(B, Sk, H, dh)
    vh = cache[:, :Sk, 2 * D:].float().reshape(B, Sk, H, dh)
    pr = torch.softmax(torch.einsum('bhd,bjhd->bhj', qh, kh) * scale, -1)
    o = torch.einsum('bhj,bjhd->bhd', pr, vh)
    ref = o.reshape(B, D) @ Wo.float().t() + bo
    got = part.sum(1) + bo
    assert rel(got, ref) < TOL[dtype]
    # the two-launch path on the same operands
    if Sk <= 128 or dtype == torch.bfloat16:          # (the fp32 attention kernel stages at most 128 keys of 96 channels)
        o2 = torch.empty(B, D, device=DEV, dtype=dtype)
        h.attention_fwd(q, k, v, o2, ((T * 3 * D, D), (T * 3 * D, 3 * D), (T * 3 * D, 3 * D), (D, D)), B, H, 1, Sk, dh, scale)
        s2 = torch.empty(B, D, device=DEV, dtype=dtype)
        h.gemm(o2, Wo, s2, B, D, D, D, D, D, bias=bo)
        assert rel(got, s2.float()) < TOL[dtype]
    # consumer: LayerNorm(x + sum of partials + bias) -> Linear
    if D <= 1024:
        x = rnd(B, D, dtype=dtype, seed=64)
        g, b = 1.0 + 0.1 * rnd(D, seed=65), 0.1 * rnd(D, seed=66)
        Wm, bias = rnd(40, D, dtype=dtype, seed=67, scale=0.05), rnd(40, seed=68)
        xn, y = torch.empty_like(x), torch.empty(B, 40, device=DEV, dtype=dtype)
        h.ln_linear_rows(x, None, g, b, 1e-5, xn, Wm, bias, y, 40, B, 40, D, h.ACT_NONE, s_partial=part, s_bias=bo)
        xn2, y2 = torch.empty_like(x), torch.empty(B, 40, device=DEV, dtype=dtype)
        h.ln_linear_rows(x, got.to(dtype), g, b, 1e-5, xn2, Wm, bias, y2, 40, B, 40, D, h.ACT_NONE)
        assert rel(xn, xn2.float()) < TOL[dtype] and rel(y, y2.float()) < TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
def test_argmax_rows(dtype):
    """greedy token pick: arg-max of logit + mask in fp32, lowest index among equal values, strided outputs"""
    h = hip()
    for rows, V in ((1, 10000), (64, 10000), (5, 37), (3, 1), (2, 70001)):
        x = rnd(rows, V + 8, dtype=dtype, seed=31)[:, :V]                # row pitch V + 8
        add = torch.zeros(V, device=DEV)
        add[::3] = -10000.0
        tok = torch.full((rows,), -1, dtype=torch.long, device=DEV)
        ids = torch.full((rows, 20), -1, dtype=torch.long, device=DEV)
        h.argmax_rows(x, add, tok, ids[:, 7])
        ref = (x.float() + add).argmax(-1)
        assert torch.equal(tok, ref) and torch.equal(ids[:, 7], ref)
        assert (ids[:, :7] == -1).all() and (ids[:, 8:] == -1).all()
        h.argmax_rows(x, None, tok, None)
        assert torch.equal(tok, x.float().argmax(-1))
    # equal values: the lowest index wins (bf16 logits do tie)
    x = torch.zeros(4, 1000, device=DEV, dtype=dtype)
    x[0, [5, 700]] = 3.0
    x[1, [999, 64, 65]] = 2.0
    x[2, 300] = -1.0
    x[3] = float('-inf')
    tok = torch.empty(4, dtype=torch.long, device=DEV)
    h.argmax_rows(x, None, tok, None)
    assert tok.tolist() == [5, 64, 0, 0]


@pytest.mark.parametrize('M,N,mode', [(9600, 2048, 'ffn1'), (9600, 2048, 'ffn1_bwd'), (3200, 2048, 'ffn1'), (9600, 1536, 'plain'), (2049, 1024, 'ffn1_bwd')])
def test_streaming_kernel_as_linear_gemm_equals_the_tile_kernels(M, N, mode):
    """gpv_gemm, K = 256 -> >= 1024 features over >= 2048 rows on the streaming kernel (conv1x1_stream.hip, LIN: alpha + the GEMM
    kernels' dropout epilogue; transformer.py:156-160 linear1 and the backward-data product through linear2 with the ReLU mask)
    against the tile kernels on the same operands: same zeros (ReLU, dropout keep pattern, mask), values within one bf16 ulp."""
    h = hip()
    K = 256
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=160), rnd(N, K, dtype=torch.bfloat16, seed=161, scale=0.1)
    bias = rnd(N, seed=162)
    mask = (rnd(M, N, dtype=torch.bfloat16, seed=163) > 0).to(torch.bfloat16) * rnd(M, N, dtype=torch.bfloat16, seed=164).abs()
    kw = {'ffn1': dict(bias=bias, act=h.ACT_RELU, drop_p=0.1, seed=4242), 'plain': dict(bias=bias),
          'ffn1_bwd': dict(relu_mask=mask, ldm=N, alpha=1.0 / 0.9)}[mode]
    outs = []
    prev = h.set_option(h.OPT_C1S, 0)
    try:
        for c1s in (0, 2):                          # 2: wherever legal (the default heuristic takes the 2048-wide layers only)
            h.set_option(h.OPT_C1S, c1s)
            h.set_option(h.OPT_C1S_LAUNCHES, 0)
            C = torch.full((M, N), float('nan'), device=DEV, dtype=torch.bfloat16)
            h.gemm(A, B, C, M, N, K, K, K, N, **kw)
            torch.cuda.synchronize()
            assert h.set_option(h.OPT_C1S_LAUNCHES, 0) == (1 if c1s else 0)
            outs.append(C.float())
    finally:
        h.set_option(h.OPT_C1S, prev)
    c0, c1 = outs
    assert torch.isfinite(c1).all()
    assert torch.equal(c0 == 0, c1 == 0)
    assert ((c1 - c0).abs() <= 2 ** -7 * c0.abs() + 1e-30).all()
    ref = A.float() @ B.float().t()
    if mode != 'ffn1_bwd':
        ref = ref + bias
    if mode == 'ffn1':
        ref = torch.relu(ref)
        keep = c0 != 0
        assert rel(c1[keep], (ref / 0.9)[keep]) < 2e-2
    elif mode == 'plain':
        assert rel(c1, ref) < 2e-2
    else:
        assert rel(c1, torch.where(mask > 0, ref / 0.9, torch.zeros_like(ref))) < 2e-2


@pytest.mark.parametrize('dtype', DTYPES)
def test_argmax_rows_embed_writes_the_next_input_row(dtype):
    """gpv_argmax_rows_embed: the pick, and xnext[r] = table[pick_r] (+ pos_row) rounded once -- what embedding gather, input
    transform (applied to the whole vocabulary beforehand) and position add produce as three launches (gpv.py:178-188)"""
    h = hip()
    for rows, V, D in ((1, 10000, 768), (4, 10000, 768), (64, 1000, 256), (3, 37, 8)):
        x = rnd(rows, V + 8, dtype=dtype, seed=33)[:, :V]
        add = torch.zeros(V, device=DEV)
        add[::3] = -10000.0
        table = rnd(V + 5, D + 8, dtype=dtype, seed=34)[:, :D]          # row pitch D + 8, more rows than V
        pos = rnd(D, dtype=dtype, seed=35)
        tok = torch.full((rows,), -1, dtype=torch.long, device=DEV)
        ids = torch.full((rows, 20), -1, dtype=torch.long, device=DEV)
        xn = torch.full((rows, D), float('nan'), device=DEV, dtype=dtype)
        h.argmax_rows(x, add, tok, ids[:, 3], table=table, pos_row=pos, xnext=xn)
        ref = (x.float() + add).argmax(-1)
        assert torch.equal(tok, ref) and torch.equal(ids[:, 3], ref)
        assert torch.equal(xn, (table[ref].float() + pos.float()).to(dtype))
        h.argmax_rows(x, None, tok, None, table=table, pos_row=None, xnext=xn)
        ref = x.float().argmax(-1)
        assert torch.equal(tok, ref) and torch.equal(xn, table[ref])


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_epilogue_and_batch(dtype):
    h = hip()
    Bt, M, N, K = 3, 200, 192, 160
    A, B = rnd(Bt, M, K, dtype=dtype, seed=3), rnd(Bt, N, K, dtype=dtype, seed=4)
    bias, rs = rnd(N, seed=5), rnd(M, seed=6)
    res = rnd(Bt, M, N, dtype=dtype, seed=7)
    mask = rnd(M, N, dtype=dtype, seed=8)
    for act, fn in ((h.ACT_NONE, lambda x: x), (h.ACT_RELU, F.relu), (h.ACT_GELU, lambda x: F.gelu(x))):
        Cm = torch.empty(Bt, M, N, device=DEV, dtype=dtype)
        h.gemm(A, B, Cm, M, N, K, K, K, N, batch=Bt, sA=M * K, sB=N * K, sC=M * N, alpha=0.5, rowscale=rs, bias=bias,
               res=res, ldr=N, sR=M * N, relu_mask=mask, ldm=N, act=act)
        ref = fn(0.5 * (A.float() @ B.float().transpose(1, 2)) * rs[None, :, None] + bias + res.float())
        ref = ref * (mask.float() > 0)
        assert rel(Cm, ref) < TOL[dtype], act
    # fp32 output from bf16 inputs, strided C
    if dtype == torch.bfloat16:
        Cw = torch.zeros(M, N + 8, device=DEV)
        h.gemm(A[0], B[0], Cw, M, N, K, K, K, N + 8)
        assert rel(Cw[:, :N], A[0].float() @ B[0].float().t()) < 1e-5
        assert Cw[:, N:].abs().max() == 0


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_trans_layouts_and_splitk(dtype):
    h = hip()
    # (KMAJOR, TRANS): C = A[M,K] @ Bm[K,N]      (RoI pooling: weights x NHWC feature map, K = 300)
    M, N, K = 100, 2048, 300
    A = torch.zeros(M, 320, device=DEV, dtype=dtype)
    A[:, :K] = rnd(M, K, dtype=dtype, seed=9)
    Bm = rnd(K, N, dtype=dtype, seed=10)
    Cm = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, Bm, Cm, M, N, K, 320, N, N, layoutB=h.TRANS)
    assert rel(Cm, A[:, :K].float() @ Bm.float()) < TOL[dtype]
    # same with the caller's promise that the row padding [K, 320) is finite (zero): 16-byte load path (GPV_GEMM_KPAD_FINITE),
    # batched like RoI pooling; the padding columns and the rows k >= K of the other operand must not leak into the result
    Ab = A.unsqueeze(0).repeat(2, 1, 1).contiguous()
    Bb = torch.stack([Bm, rnd(K, N, dtype=dtype, seed=15)]).contiguous()
    Cb = torch.empty(2, M, N, device=DEV, dtype=dtype)
    h.gemm(Ab, Bb, Cb, M, N, K, 320, N, N, layoutB=h.TRANS, batch=2, sA=M * 320, sB=K * N, sC=M * N, kpad_finite=True)
    assert rel(Cb, Ab[:, :, :K].float() @ Bb.float()) < TOL[dtype]
    assert torch.equal(Cb[0], Cm) or rel(Cb[0], Cm.float()) < 1e-6
    # (TRANS, TRANS) wgrad form: dW[N,K] += dY[Mr,N]^T X[Mr,K], split-K atomics into fp32
    for (Mr, Nn, Kk, split) in [(9600, 256, 2048, 8), (3200, 768, 768, 1), (777, 72, 136, 3), (192, 2304, 768, 4)]:
        dY, X = rnd(Mr, Nn, dtype=dtype, seed=11), rnd(Mr, Kk, dtype=dtype, seed=12)
        dW = torch.ones(Nn, Kk, device=DEV)
        h.gemm(dY, X, dW, Nn, Kk, Mr, Nn, Kk, Kk, layoutA=h.TRANS, layoutB=h.TRANS, accumulate=True, split_k=split)
        ref = 1.0 + dY.float().t() @ X.float()
        assert rel(dW, ref) < (3e-5 if dtype == torch.float32 else 2e-3), (Mr, Nn, Kk)
    # batched (TRANS,TRANS): RoI backward  dfeat[b,p,c] = sum_q Wgt[b,q,p] dOut[b,q,c]
    Bt, Q, Pn, Cc = 2, 100, 300, 256
    Wg = rnd(Bt, Q, 320, dtype=dtype, seed=13)
    dO = rnd(Bt, Q, Cc, dtype=dtype, seed=14)
    dF = torch.empty(Bt, Pn, Cc, device=DEV, dtype=dtype)
    h.gemm(Wg, dO, dF, Pn, Cc, Q, 320, Cc, Cc, layoutA=h.TRANS, layoutB=h.TRANS, batch=Bt, sA=Q * 320, sB=Q * Cc, sC=Pn * Cc)
    assert rel(dF, Wg[:, :, :Pn].float().transpose(1, 2) @ dO.float()) < TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('N,K,M,split', [(256, 256, 9600, 8), (768, 3072, 640, 1), (2, 256, 3200, 4), (100, 72, 333, 3), (2048, 256, 3200, 2)])
def test_gemm_wgrad_with_fused_bias_grad(dtype, N, K, M, split):
    """dW[N,K] += dY^T X and db[N] += colsum(dY) in one launch (a_rowsum), split-K and single-pass variants"""
    h = hip()
    dy, x = rnd(M, N, dtype=dtype, seed=40), rnd(M, K, dtype=dtype, seed=41)
    dw0, db0 = rnd(N, K, seed=42), rnd(N, seed=43)
    dw, db = dw0.clone(), db0.clone()
    h.gemm(dy, x, dw, N, K, M, N, K, K, layoutA=h.TRANS, layoutB=h.TRANS, accumulate=True, split_k=split, a_rowsum=db)
    refw = dw0 + dy.float().t() @ x.float()
    refb = db0 + dy.float().sum(0)
    assert rel(dw, refw) < (3e-5 if dtype == torch.float32 else 3e-3)
    assert rel(db, refb) < 1e-5                                      # exact fp32 sums of the stored values


def test_gemm_dropout_epilogue():
    h = hip()
    M, N, K = 512, 512, 64
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=15), rnd(N, K, dtype=torch.bfloat16, seed=16)
    C0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    C1 = torch.empty_like(C0)
    h.gemm(A, B, C0, M, N, K, K, K, N)
    h.gemm(A, B, C1, M, N, K, K, K, N, drop_p=0.1, seed=1234)
    kept = C1 != 0
    frac = kept.float().mean().item()
    assert 0.88 < frac < 0.92
    assert rel(C1[kept], (C0.float() / 0.9)[kept]) < 1e-2
    C2 = torch.empty_like(C0)
    h.gemm(A, B, C2, M, N, K, K, K, N, drop_p=0.1, seed=1234)
    assert torch.equal(C1, C2)            # counter-based RNG: same seed -> same mask


@pytest.mark.parametrize('N,K,M', [(256, 256, 9600), (768, 3072, 3200), (2048, 256, 3200), (256, 2048, 9600), (1536, 768, 3392),
                                   (768, 768, 10000), (128, 384, 1000)])
def test_gemm_wgrad_direct_to_lds(N, K, M):
    """gemm_glds_tt.hip (linear form, bias gradient through a ones-operand MFMA) vs the default kernels and vs fp32 torch;
    M = 10000 / 1000 rows: ragged last k-tile of 64"""
    h, dtype = hip(), torch.bfloat16
    dy, x = rnd(M, N, dtype=dtype, seed=50), rnd(M, K, dtype=dtype, seed=51)
    dw0, db0 = rnd(N, K, seed=52), rnd(N, seed=53)
    outs = []
    for mode in (0, 2):
        prev = h.set_option(h.OPT_GLDS_WGRAD, mode)
        dw, db = dw0.clone(), db0.clone()
        h.gemm(dy, x, dw, N, K, M, N, K, K, layoutA=h.TRANS, layoutB=h.TRANS, accumulate=True, split_k=8, a_rowsum=db)
        h.set_option(h.OPT_GLDS_WGRAD, prev)
        outs.append((dw, db))
    refw = dw0 + dy.float().t() @ x.float()
    refb = db0 + dy.float().sum(0)
    assert rel(outs[1][0], refw) < 3e-3 and rel(outs[1][1], refb) < 1e-5
    assert rel(outs[1][0], outs[0][0]) < 1e-5 and rel(outs[1][1], outs[0][1]) < 1e-5


# ----------------------------------------------------------------------------------------- conv
def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


CONVS = [  # Cin, Cout, k, stride, pad, H, W
    (64, 64, 1, 1, 0, 24, 32), (256, 128, 3, 2, 1, 24, 32), (128, 128, 3, 1, 1, 15, 20), (256, 512, 1, 2, 0, 30, 40),
    (512, 2048, 1, 1, 0, 15, 20), (64, 64, 3, 1, 1, 17, 23)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('Cin,Cout,k,s,p,H,W', CONVS)
def test_conv_fwd_dgrad_wgrad(dtype, Cin, Cout, k, s, p, H, W, Bn=3):
    h = hip()
    x = rnd(Bn, Cin, H, W, dtype=dtype, seed=20)
    w = rnd(Cout, Cin, k, k, dtype=dtype, seed=21, scale=1.0 / math.sqrt(Cin * k * k))
    scale, bias = rnd(Cout, seed=22).abs() + 0.5, rnd(Cout, seed=23)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd(Bn, Cout, OH, OW, dtype=dtype, seed=24)
    xf = x.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    yref = F.relu(F.conv2d(xf, wf, stride=s, padding=p) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res.float())
    xn, wn = nhwc(x), w.permute(0, 2, 3, 1).contiguous()           # [Cout][kh][kw][Cin]
    y = torch.empty(Bn, OH, OW, Cout, device=DEV, dtype=dtype)
    # forward with BN scale folded into the weights (as the model does)
    wfold = (w.float() * scale.view(-1, 1, 1, 1)).to(dtype).permute(0, 2, 3, 1).contiguous()
    h.conv2d(0, xn, wfold, y, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p, bias=bias, res=nhwc(res), act=h.ACT_RELU)
    yref_fold = F.relu(F.conv2d(x.float(), wfold.permute(0, 3, 1, 2).float(), stride=s, padding=p) + bias.view(1, -1, 1, 1) + res.float())
    assert rel(y, nhwc(yref_fold)) < TOL[dtype]
    # dgrad: dx = conv_transpose(dy, w) (+ addend, * relu mask of a saved activation)
    dy = rnd(Bn, Cout, OH, OW, dtype=dtype, seed=25)
    addend = rnd(Bn, Cin, H, W, dtype=dtype, seed=26)
    saved = rnd(Bn, Cin, H, W, dtype=dtype, seed=27)
    conv_only = F.conv2d(xf, wf, stride=s, padding=p)
    gx, gw = torch.autograd.grad(conv_only, (xf, wf), dy.float())
    wd = w.permute(1, 2, 3, 0).contiguous()                      # [Cin][kh][kw][Cout]
    dx = torch.empty(Bn, H, W, Cin, device=DEV, dtype=dtype)
    h.conv2d(1, nhwc(dy), wd, dx, Bn, OH, OW, Cout, Cout, H, W, Cin, k, k, s, s, p, p, res=nhwc(addend), relu_mask=nhwc(saved))
    ref = (gx + addend.float()) * (saved.float() > 0)
    assert rel(dx, nhwc(ref)) < TOL[dtype]
    # wgrad (needs Cin % 64 == 0): dw[Cout][kh][kw][Cin] += rowscale[co] * sum dy x
    dw = torch.zeros(Cout, k, k, Cin, device=DEV)
    h.conv2d(2, xn, nhwc(dy), dw, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p, rowscale=scale)
    refw = (gw * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)
    assert rel(dw, refw) < (3e-5 if dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize('Cin,Cout,H,W,Bn', [(256, 512, 32, 32, 4), (512, 1024, 32, 32, 4), (256, 512, 32, 64, 8), (1024, 2048, 16, 32, 8)])
def test_conv1x1_stride2_dgrad_class_rows_multiple_of_tile(Cin, Cout, H, W, Bn):
    """pointwise stride-2 backward-data is split into a GEMM over parity class 0 + an element-wise fill; when the rows of one
    class are a multiple of the row tile (64 / 128 / 256: the bench batch of 32 hits it on layer3 / layer4) the class-0 launch
    must NOT cycle its row panels through four classes (round-2 bug: three quarters of class 0 were left unwritten)"""
    h = hip()
    assert (Bn * (H // 2) * (W // 2)) % 256 == 0
    for pipe, glds in ((None, None), (0, 3), (0, 2), (0, 0)):       # default dispatch; 4-wave / 8-wave direct-to-LDS; register-staged
        prev = (h.set_option(h.OPT_PIPE, pipe), h.set_option(h.OPT_GLDS, glds)) if pipe is not None else None
        try:
            test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, 1, 2, 0, H, W, Bn=Bn)
        finally:
            if prev is not None:
                h.set_option(h.OPT_PIPE, prev[0])
                h.set_option(h.OPT_GLDS, prev[1])


@pytest.mark.parametrize('Cin,Cout,k,s,p,H,W,Bn', [
    (128, 128, 3, 1, 1, 30, 40, 4),      # 3x3, every tap incl. the padded border
    (256, 256, 3, 2, 1, 30, 40, 4),      # stride 2
    (512, 128, 1, 1, 0, 15, 20, 5),      # 1x1, B*OH*OW = 1500: ragged last k-tile of 64 pixels
    (256, 512, 1, 2, 0, 30, 40, 3),      # strided 1x1 (downsample)
    (128, 256, 3, 1, 1, 8, 24, 6)])      # short rows: a 64-pixel k-tile spans 3 image rows and crosses images
def test_conv_wgrad_direct_to_lds_vs_register_staged(Cin, Cout, k, s, p, H, W, Bn):
    """gemm_glds_tt.hip against gemm.hip's TRANS x CONV kernel (same split, same reduction kernel) and against autograd"""
    h, dtype = hip(), torch.bfloat16
    x = rnd(Bn, Cin, H, W, dtype=dtype, seed=40)
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = rnd(Bn, Cout, OH, OW, dtype=dtype, seed=41)
    scale = rnd(Cout, seed=42).abs() + 0.5
    xn, dyn = nhwc(x), nhwc(dy)
    outs = []
    for mode in (0, 1):
        prev = h.set_option(h.OPT_GLDS_WGRAD, mode)
        dw = torch.full((Cout, k, k, Cin), 0.25, device=DEV)           # accumulates into what is there
        h.conv2d(2, xn, dyn, dw, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p, rowscale=scale)
        h.set_option(h.OPT_GLDS_WGRAD, prev)
        outs.append(dw)
    wf = torch.zeros(Cout, Cin, k, k, device=DEV, requires_grad=True)
    gw, = torch.autograd.grad(F.conv2d(x.float(), wf, stride=s, padding=p), wf, dy.float())
    ref = (gw * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1) + 0.25
    assert rel(outs[1], ref) < 3e-3
    assert rel(outs[1], outs[0]) < 1e-5                                # same bf16 products, fp32 sums in a different order


def test_conv_wgrad_group_equals_single_launches():
    """gpv_conv_wgrad_group: a mix of problems -- no split (short reduction: tiles add themselves into dw), sliced reductions through
    the workspace, stride 2, 3x3 borders, a ragged last k-tile, a shape the grouped kernel declines (Cin = 64 -> its own launch) --
    against one gpv_conv2d mode-2 call each, gradients accumulated into"""
    h, dtype = hip(), torch.bfloat16
    cases = [(512, 512, 3, 1, 1, 15, 20, 4),       # K = 1200 pixels: 19 k-tiles -> no split
             (256, 256, 3, 1, 1, 30, 40, 32),      # K = 38400: 600 k-tiles -> 4 slices
             (128, 128, 3, 2, 1, 60, 80, 8),       # stride 2
             (512, 128, 1, 1, 0, 15, 20, 5),       # 1x1, K = 1500: ragged last k-tile
             (256, 1024, 1, 1, 0, 30, 40, 16),     # 1x1, 8 x 2 tiles, sliced
             (64, 128, 3, 1, 1, 30, 40, 4),        # Cin = 64: declined by the grouped kernel
             (128, 256, 3, 1, 1, 8, 24, 6)]        # short rows
    probs, single = [], []
    for i, (Cin, Cout, k, s, p, H, W, Bn) in enumerate(cases):
        x = nhwc(rnd(Bn, Cin, H, W, dtype=dtype, seed=60 + i))
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = nhwc(rnd(Bn, Cout, OH, OW, dtype=dtype, seed=80 + i))
        scale = rnd(Cout, seed=90 + i).abs() + 0.5
        dw_g = torch.full((Cout, k, k, Cin), 0.25, device=DEV)
        dw_s = dw_g.clone()
        probs.append((x, dy, dw_g, scale, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p))
        h.conv2d(2, x, dy, dw_s, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p, rowscale=scale)
        single.append(dw_s)
    h.conv_wgrad_group(probs)
    torch.cuda.synchronize()
    for q, ref, c in zip(probs, single, cases):
        assert rel(q[2], ref) < 1e-5, (c, rel(q[2], ref))


def test_conv_wgrad_group_eight_phase_kernel():
    """the eight-phase 256 x 256 weight-gradient kernel (gemm_glds_tt.hip wg8_*; Cout and Cin multiples of 256) against the 128 x 128
    grouped kernel on the same problems (same bf16 products, fp32 sums in another order) and against fp32 autograd: no split
    (tiles add themselves into dw), sliced reductions, a ragged last k-tile, a unit of ONE k-tile (pipeline shorter than its
    look-ahead), stride 2 with 3x3 borders, 1x1 stride 2, several row and column tiles, short image rows"""
    h, dtype = hip(), torch.bfloat16
    cases = [(512, 512, 3, 1, 1, 15, 20, 4),       # 2 x 18 tiles, K = 1200 pixels: 19 k-tiles (ragged), no split
             (256, 256, 3, 1, 1, 30, 40, 32),      # 1 x 9 tiles, 600 k-tiles -> 4 slices
             (256, 256, 3, 2, 1, 60, 80, 8),       # stride 2
             (256, 512, 1, 1, 0, 15, 20, 5),       # 1x1, K = 1500: ragged last k-tile
             (1024, 256, 1, 1, 0, 30, 40, 16),     # 1 x 4 tiles, sliced
             (512, 1024, 1, 2, 0, 30, 40, 8),      # the stride-2 projection
             (256, 256, 3, 1, 1, 8, 24, 3),        # short rows, 9 k-tiles
             (256, 256, 1, 1, 0, 16, 32, 1)]       # 8 k-tiles: the shortest reduction the grouped call accepts
    probs = {0: [], 1: []}
    refs = []
    for i, (Cin, Cout, k, s, p, H, W, Bn) in enumerate(cases):
        x = nhwc(rnd(Bn, Cin, H, W, dtype=dtype, seed=160 + i))
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = nhwc(rnd(Bn, Cout, OH, OW, dtype=dtype, seed=180 + i))
        scale = rnd(Cout, seed=190 + i).abs() + 0.5
        for mode in (0, 1):
            dw = torch.full((Cout, k, k, Cin), 0.25, device=DEV)
            probs[mode].append((x, dy, dw, scale, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p))
        if i in (0, 3, 6):
            wf = torch.zeros(Cout, Cin, k, k, device=DEV, requires_grad=True)
            xf = x.float().permute(0, 3, 1, 2)
            gw, = torch.autograd.grad(F.conv2d(xf, wf, stride=s, padding=p), wf, dy.float().permute(0, 3, 1, 2))
            refs.append((i, (gw * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1) + 0.25))
    prev = h.set_option(h.OPT_WG8, 0)
    h.conv_wgrad_group(probs[0])
    h.set_option(h.OPT_WG8, 2)                          # (1 = only when the call has enough 256 x 256 units to fill the chip)
    n0 = h.set_option(h.OPT_WG8_LAUNCHES, 0)
    h.conv_wgrad_group(probs[1])
    used = h.set_option(h.OPT_WG8_LAUNCHES, n0)
    h.set_option(h.OPT_WG8, prev)
    torch.cuda.synchronize()
    assert used >= 1, used
    for a, b, c in zip(probs[1], probs[0], cases):
        assert rel(a[2], b[2]) < 1e-5, (c, rel(a[2], b[2]))
    for i, ref in refs:
        assert rel(probs[1][i][2], ref) < 3e-3, (cases[i], rel(probs[1][i][2], ref))


def test_conv_wgrad_group_half_width_eight_phase_tiles():
    """the eight-phase kernel on 128 x 256 | 256 x 128 tiles (gemm_glds_tt.hip wg8h_*: problems with a 128-wide side, layer2) against
    the 128 x 128 grouped kernel on the same problems and against fp32 autograd: two taps per column tile with the ragged ninth tap,
    direct and sliced reductions, stride 2, the tall form over 3x3 taps, several row tiles, a channel count of 384, ragged k-tiles"""
    h, dtype = hip(), torch.bfloat16
    cases = [(128, 128, 3, 1, 1, 60, 80, 2),       # 128 x 1152: 5 column tiles (the last one half empty), 150 k-tiles, no split
             (128, 128, 3, 1, 1, 60, 80, 8),       # sliced
             (128, 128, 3, 2, 1, 120, 160, 2),     # stride 2
             (512, 128, 1, 1, 0, 60, 80, 4),       # 128 x 512
             (128, 512, 1, 1, 0, 60, 80, 4),       # 512 x 128: tall tiles
             (256, 512, 1, 2, 0, 120, 160, 2),     # the stride-2 projection (tall)
             (256, 128, 1, 1, 0, 120, 160, 2),     # 128 x 256, long reduction
             (128, 384, 3, 1, 1, 8, 24, 3),        # three row tiles, short image rows, 9 k-tiles
             (384, 128, 1, 1, 0, 16, 32, 1),       # Cin = 384: a tile's halves in the same tap, ragged second tile; 8 k-tiles
             (128, 256, 3, 1, 1, 15, 20, 5)]       # tall over nine taps, K = 1500: ragged last k-tile
    probs = {0: [], 1: []}
    refs = []
    for i, (Cin, Cout, k, s, p, H, W, Bn) in enumerate(cases):
        x = nhwc(rnd(Bn, Cin, H, W, dtype=dtype, seed=260 + i))
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        dy = nhwc(rnd(Bn, Cout, OH, OW, dtype=dtype, seed=280 + i))
        scale = rnd(Cout, seed=290 + i).abs() + 0.5
        for mode in (0, 1):
            dw = torch.full((Cout, k, k, Cin), 0.25, device=DEV)
            probs[mode].append((x, dy, dw, scale, Bn, H, W, Cin, Cin, OH, OW, Cout, k, k, s, s, p, p))
        if i in (0, 2, 4, 7, 8, 9):
            wf = torch.zeros(Cout, Cin, k, k, device=DEV, requires_grad=True)
            xf = x.float().permute(0, 3, 1, 2)
            gw, = torch.autograd.grad(F.conv2d(xf, wf, stride=s, padding=p), wf, dy.float().permute(0, 3, 1, 2))
            refs.append((i, (gw * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1) + 0.25))
    prev8 = h.set_option(h.OPT_WG8, 0)
    prev = h.set_option(h.OPT_WG8H, 0)
    h.conv_wgrad_group(probs[0])
    h.set_option(h.OPT_WG8H, 1)
    n0 = h.set_option(h.OPT_WG8_LAUNCHES, 0)
    h.conv_wgrad_group(probs[1])
    used = h.set_option(h.OPT_WG8_LAUNCHES, n0)
    h.set_option(h.OPT_WG8H, prev)
    h.set_option(h.OPT_WG8, prev8)
    torch.cuda.synchronize()
    assert used >= 1, used
    for a, b, c in zip(probs[1], probs[0], cases):
        assert rel(a[2], b[2]) < 1e-5, (c, rel(a[2], b[2]))
    for i, ref in refs:
        assert rel(probs[1][i][2], ref) < 3e-3, (cases[i], rel(probs[1][i][2], ref))


@pytest.mark.parametrize('Cin,Cout,k,H,W,Bn', [
    (64, 256, 3, 50, 80, 8),        # 32000 rows x 256: 200 x 2 tiles of 160 x 128 (two per CU), K = 576
    (64, 512, 3, 30, 40, 8),        # 9600 rows x 512: 100 x 4 tiles of 96 x 128
    (512, 256, 1, 50, 80, 8)])      # the 1x1 form (plain GEMM over the pixel rows)
def test_conv_two_tiles_per_cu_variants(Cin, Cout, k, H, W, Bn):
    """gemm_glds.hip with 160 x 128 / 96 x 128 tiles (launches whose row tiles fill the chip's two-per-CU slots in one round):
    forward and backward-data against the register-staged kernel on the same operands, and that these shapes really take it"""
    h = hip()
    test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, k, 1, k // 2, H, W, Bn=Bn)         # (default dispatch, vs fp32 torch)
    n0 = h.set_option(h.OPT_GLDS_LAUNCHES, 0)
    test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, k, 1, k // 2, H, W, Bn=Bn)
    used = h.set_option(h.OPT_GLDS_LAUNCHES, n0)
    assert used >= 1, used


@pytest.mark.parametrize('Cin,Cout,H,W,Bn', [
    (256, 256, 30, 40, 32),         # layer3 conv2 at the bench batch: 240 x 2 tiles of 160 rows, four channel blocks
    (512, 512, 15, 20, 32),         # layer4 conv2: 100 x 4 tiles of 96 rows, eight channel blocks, tiles spanning whole images
    (128, 256, 29, 43, 32),         # odd image (1247 pixels), rows not a multiple of the tile: tail tile, borders at every offset
    (64, 256, 47, 47, 16),          # the widest image the halo covers (W + 1 = 48), one channel block
    (192, 128, 9, 21, 240),         # many small images per tile (189 pixels each), three channel blocks, one column tile
    (128, 128, 1, 47, 800),         # one-row images: every pixel is on the top AND the bottom border
    (128, 256, 40, 1, 900),         # one-column images: left and right border at once
    (128, 128, 2, 2, 9600)])        # 2 x 2 images: every pixel in a corner
def test_conv3x3_halo_image_kernel(Cin, Cout, H, W, Bn, check_used=True):
    """stride-1 3x3 forward (bias + residual + ReLU) and backward-data (addend + ReLU mask) on the halo-image tile kernel
    (gemm_glds.hip glds_halo_kernel: one halo image per channel block, taps as shifted fragment reads, border lanes zeroed) against
    fp32 torch and against the nine-tap-tile kernel on the same operands; and that these shapes really take it"""
    h = hip()
    prev = h.set_option(h.OPT_C3_HALO, 2)                 # (2: the 96-row tiles as well -- the default keeps those on the nine-tap kernel)
    n0 = h.set_option(h.OPT_C3_HALO_LAUNCHES, 0)
    try:
        test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, 3, 1, 1, H, W, Bn=Bn)
        used = h.set_option(h.OPT_C3_HALO_LAUNCHES, n0)
        assert not check_used or used >= (2 if (Cin, Cout) in ((256, 256), (512, 512)) else 1), used     # (the model's shapes: forward AND backward-data; elsewhere the backward-data
        #  may have too few tiles for the two-per-CU launch or 64 output columns)
        # the same products in another summation order: forward outputs of the two kernels
        dtype = torch.bfloat16
        x = nhwc(rnd(Bn, Cin, H, W, dtype=dtype, seed=40))
        w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=41, scale=1.0 / math.sqrt(Cin * 9)).permute(0, 2, 3, 1).contiguous()
        bias = rnd(Cout, seed=42)
        ys = []
        for mode in (2, 0):
            h.set_option(h.OPT_C3_HALO, mode)
            y = torch.empty(Bn, H, W, Cout, device=DEV, dtype=dtype)
            h.conv2d(0, x, w, y, Bn, H, W, Cin, Cin, H, W, Cout, 3, 3, 1, 1, 1, 1, bias=bias, act=h.ACT_RELU)
            ys.append(y)
        torch.cuda.synchronize()
        assert rel(ys[0], ys[1]) < 8e-3, rel(ys[0], ys[1])          # (two bf16 steps of the largest output: up to 4608 products in another order)
    finally:
        h.set_option(h.OPT_C3_HALO, prev)


@pytest.mark.parametrize('Cin,Cout,k,s,H,W', [
    (512, 512, 3, 1, 15, 20),       # layer4 conv2 at batch 1: 300 pixels, K = 4608 -> split 8
    (256, 256, 3, 1, 30, 40),       # layer3 conv2: 1200 pixels, K = 2304
    (256, 256, 3, 2, 60, 80),       # layer3.0 conv2 (stride 2)
    (128, 128, 3, 1, 60, 80),       # layer2 conv2: 4800 pixels, K = 1152
    (1024, 256, 1, 1, 30, 40), (2048, 512, 1, 1, 15, 20), (512, 2048, 1, 1, 15, 20)])       # the 1x1s around them (64 x 64 tiles, waves split K)
def test_conv_forward_at_batch_one(Cin, Cout, k, s, H, W):
    """inference at batch 1: forward convolutions over a few thousand pixels (split reduction + second pass with the epilogue for
    the 3x3s, the small-M kernel for the 1x1s) against fp32 torch (inside test_conv_fwd_dgrad_wgrad)"""
    test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, k, s, k // 2, H, W, Bn=1)
    test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, k, s, k // 2, H, W, Bn=2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_stem_conv_image_prep_and_maxpool(dtype):
    h = hip()
    Bn, H, W = 2, 96, 128
    img = rnd(Bn, 3, H, W, seed=30)
    w = rnd(64, 3, 7, 7, seed=31, scale=0.08)
    bias = rnd(64, seed=32)
    Hp, Wp = H + 6, ((W + 6 + 2 + 7) // 8) * 8
    xin = torch.empty(Bn, Hp, Wp, 4, device=DEV, dtype=dtype)
    h.image_to_nhwc4(img, xin, Bn, H, W, 3, Hp, Wp)
    ref_in = torch.zeros(Bn, Hp, Wp, 4, device=DEV)
    ref_in[:, 3:3 + H, 3:3 + W, :3] = img.permute(0, 2, 3, 1)
    assert rel(xin, ref_in) < (1e-7 if dtype == torch.float32 else 4e-3)
    # stem weights: [64][7 rows][8 pixels x 4 ch] (8th pixel / 4th channel zero)
    ws = torch.zeros(64, 7, 8, 4, device=DEV)
    ws[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    ws = ws.reshape(64, 7, 1, 32).to(dtype).contiguous()
    OH, OW = H // 2, W // 2
    y = torch.empty(Bn, OH, OW, 64, device=DEV, dtype=dtype)
    h.conv2d(0, xin, ws, y, Bn, Hp, Wp, 4, 32, OH, OW, 64, 7, 1, 2, 2, 0, 0, bias=bias, act=h.ACT_RELU)
    src = img.to(dtype).float()
    ref = F.relu(F.conv2d(src, w.to(dtype).float(), stride=2, padding=3) + bias.view(1, -1, 1, 1))
    assert rel(y, nhwc(ref)) < TOL[dtype]
    PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
    z = torch.empty(Bn, PH, PW, 64, device=DEV, dtype=dtype)
    h.maxpool3x3s2(y, z, Bn, OH, OW, 64, PH, PW)
    assert torch.equal(z.float(), nhwc(F.max_pool2d(y.float().permute(0, 3, 1, 2), 3, 2, 1)))


@pytest.mark.parametrize('Bn,H,W', [(2, 96, 128), (3, 62, 90), (1, 480, 640), (2, 34, 30)])
def test_fused_stem_conv_bn_relu_maxpool(Bn, H, W):
    """gpv_stem_pool (stem_pool.hip: conv 7x7/2 + shift + ReLU + max-pool 3x3/2 in one launch) against fp32 torch on the same
    bf16-rounded operands, and against the two-kernel path (generic conv kernel + pooling kernel): same bf16 rounding points,
    fp32 summation order differs.  Odd map sizes: conv 31x45 -> pooled 16x23 (ragged strips of 15 pooled columns, odd rows)."""
    h, dtype = hip(), torch.bfloat16
    img = rnd(Bn, 3, H, W, seed=30)
    w = rnd(64, 3, 7, 7, seed=31, scale=0.08)
    bias = rnd(64, seed=32)
    OH, OW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = H + 6, ((max(W + 6, 2 * (OW - 1) + 8) + 7) // 8) * 8
    xin = torch.empty(Bn, Hp, Wp, 4, device=DEV, dtype=dtype)
    h.image_to_nhwc4(img, xin, Bn, H, W, 3, Hp, Wp)
    ws = torch.zeros(64, 7, 8, 4, device=DEV)
    ws[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    ws = ws.reshape(64, 7, 32).to(dtype).contiguous()
    PH, PW = (OH + 2 - 3) // 2 + 1, (OW + 2 - 3) // 2 + 1
    z = torch.full((Bn, PH, PW, 64), float('nan'), device=DEV, dtype=dtype)
    h.stem_pool(xin, ws, bias, z, Bn, Hp, Wp, OH, OW, PH, PW)
    conv = F.relu(F.conv2d(img.to(dtype).float(), w.to(dtype).float(), stride=2, padding=3) + bias.view(1, -1, 1, 1))
    ref = F.max_pool2d(conv.to(dtype).float(), 3, 2, 1)
    assert torch.isfinite(z.float()).all()
    assert rel(z, nhwc(ref)) < TOL[dtype]
    y = torch.empty(Bn, OH, OW, 64, device=DEV, dtype=dtype)
    h.conv2d(0, xin, ws.view(64, 7, 1, 32), y, Bn, Hp, Wp, 4, 32, OH, OW, 64, 7, 1, 2, 2, 0, 0, bias=bias, act=h.ACT_RELU)
    z2 = torch.empty_like(z)
    h.maxpool3x3s2(y, z2, Bn, OH, OW, 64, PH, PW)
    assert rel(z, z2) < 8e-3                   # one bf16 ulp where the two summation orders round differently
    assert (z != z2).float().mean() < 0.02


# ------------------------------------------------------------- 8-wave direct-to-LDS kernel (gemm_glds.hip)
@pytest.fixture(params=[2, 3], ids=['8wave', '4wave128'])
def glds(request):
    """force the direct-to-LDS kernel wherever it is legal (by default it only takes the launches it wins) and count its
    launches; 2 = the 8-wave 256-row tiles, 3 = the 4-wave 128x128 variant"""
    h = hip()
    prev = h.set_option(h.OPT_GLDS, request.param)
    prevs = h.set_option(h.OPT_SKINNY, 0)
    prevp = h.set_option(h.OPT_PIPE, 0)                  # (the pipelined kernel is tried first: off for these cases)
    h.set_option(h.OPT_GLDS_LAUNCHES, 0)
    yield h
    h.set_option(h.OPT_GLDS, prev)
    h.set_option(h.OPT_SKINNY, prevs)
    h.set_option(h.OPT_PIPE, prevp)


@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (2048, 512, 768), (1000, 128, 192), (257, 130, 64), (9600, 256, 2048),
                                   (70, 384, 128), (513, 100, 320)])
def test_glds_gemm_plain(glds, M, N, K):
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=1), rnd(N, K, dtype=torch.bfloat16, seed=2)
    Cm = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    glds.gemm(A, B, Cm, M, N, K, K, K, N)
    assert glds.set_option(glds.OPT_GLDS_LAUNCHES, 0) == 1
    assert rel(Cm, A.float() @ B.float().t()) < TOL[torch.bfloat16]
    Cf = torch.zeros(M, N + 8, device=DEV)                         # fp32 output, strided C
    glds.gemm(A, B, Cf, M, N, K, K, K, N + 8)
    assert glds.set_option(glds.OPT_GLDS_LAUNCHES, 0) == 1
    assert rel(Cf[:, :N], A.float() @ B.float().t()) < 1e-5
    assert Cf[:, N:].abs().max() == 0


def test_glds_gemm_epilogue_and_batch(glds):
    h, dtype = glds, torch.bfloat16
    Bt, M, N, K = 3, 200, 192, 192
    A, B = rnd(Bt, M, K, dtype=dtype, seed=3), rnd(Bt, N, K, dtype=dtype, seed=4)
    bias, rs = rnd(N, seed=5), rnd(M, seed=6)
    res = rnd(Bt, M, N, dtype=dtype, seed=7)
    mask = rnd(M, N, dtype=dtype, seed=8)
    for act, fn in ((h.ACT_NONE, lambda x: x), (h.ACT_RELU, F.relu), (h.ACT_GELU, lambda x: F.gelu(x))):
        Cm = torch.empty(Bt, M, N, device=DEV, dtype=dtype)
        h.gemm(A, B, Cm, M, N, K, K, K, N, batch=Bt, sA=M * K, sB=N * K, sC=M * N, alpha=0.5, rowscale=rs, bias=bias,
               res=res, ldr=N, sR=M * N, relu_mask=mask, ldm=N, act=act)
        ref = fn(0.5 * (A.float() @ B.float().transpose(1, 2)) * rs[None, :, None] + bias + res.float())
        ref = ref * (mask.float() > 0)
        assert rel(Cm, ref) < TOL[dtype], act
    assert h.set_option(h.OPT_GLDS_LAUNCHES, 0) == 3
    # dropout epilogue: same keep pattern as the 4-wave kernel (counter-hash of the element index)
    M, N, K = 512, 256, 128
    A, B = rnd(M, K, dtype=dtype, seed=9), rnd(N, K, dtype=dtype, seed=10)
    c1, c2 = torch.empty(M, N, device=DEV, dtype=dtype), torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, c1, M, N, K, K, K, N, drop_p=0.25, seed=77)
    h.set_option(h.OPT_GLDS, 0)
    h.gemm(A, B, c2, M, N, K, K, K, N, drop_p=0.25, seed=77)
    h.set_option(h.OPT_GLDS, 2)
    assert torch.equal(c1 == 0, c2 == 0) and rel(c1, c2.float()) < 1e-2


PIPE_CFGS = [(256, 128), (192, 128), (128, 128), (160, 256), (128, 256), (96, 256), (64, 64), (32, 64)]     # gemm_pipe.hip: kCfgs


def _pipe_fixture(idx):
    h = hip()
    prev = h.set_option(h.OPT_PIPE, 100 + idx)
    prevs = h.set_option(h.OPT_SKINNY, 0)
    h.set_option(h.OPT_PIPE_LAUNCHES, 0)
    h.bn = PIPE_CFGS[idx][1]
    yield h
    h.set_option(h.OPT_PIPE, prev)
    h.set_option(h.OPT_SKINNY, prevs)


@pytest.fixture(params=range(len(PIPE_CFGS)), ids=['%dx%d' % c for c in PIPE_CFGS])
def pipe(request):
    """force tile configuration i of the pipelined direct-to-LDS kernel (gemm_pipe.hip) wherever it is legal; the last two are
    the small-M tiles (6 / 8 LDS stages, plain GEMMs only)"""
    yield from _pipe_fixture(request.param)


@pytest.fixture(params=range(6), ids=['%dx%d' % c for c in PIPE_CFGS[:6]])
def pipe_big(request):
    yield from _pipe_fixture(request.param)


@pytest.mark.parametrize('M,N,K', [(300, 256, 256), (2048, 512, 768), (1000, 128, 192), (9600, 256, 2048), (70, 384, 128), (257, 768, 64),
                                   (192, 768, 768), (640, 2304, 768), (1, 768, 3072)])
def test_pipe_gemm_plain_epilogue_batch(pipe, M, N, K):
    h, dtype = pipe, torch.bfloat16
    A, B = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2)
    ref = A.float() @ B.float().t()
    Cm = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, Cm, M, N, K, K, K, N)
    legal = N % h.bn == 0 and K >= 128                               # (a k-loop of >= 2 tiles, whole column tiles)
    assert h.set_option(h.OPT_PIPE_LAUNCHES, 0) == int(legal)
    assert rel(Cm, ref) < TOL[dtype]
    bias, rs = rnd(N, seed=5), rnd(M, seed=6)
    res, mask = rnd(M, N, dtype=dtype, seed=7), rnd(M, N, dtype=dtype, seed=8)
    for act, fn in ((h.ACT_NONE, lambda x: x), (h.ACT_RELU, F.relu), (h.ACT_GELU, lambda x: F.gelu(x))):
        h.gemm(A, B, Cm, M, N, K, K, K, N, alpha=0.5, rowscale=rs, bias=bias, res=res, ldr=N, relu_mask=mask, ldm=N, act=act)
        r2 = fn(0.5 * ref * rs[:, None] + bias + res.float()) * (mask.float() > 0)
        assert rel(Cm, r2) < TOL[dtype], act
    assert h.set_option(h.OPT_PIPE_LAUNCHES, 0) == 3 * int(legal)
    # dropout epilogue: same keep pattern as the 4-wave kernel (counter-hash of the element index)
    c1, c2 = torch.empty(M, N, device=DEV, dtype=dtype), torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, c1, M, N, K, K, K, N, drop_p=0.25, seed=77)
    mode = h.set_option(h.OPT_PIPE, 0)
    g = h.set_option(h.OPT_GLDS, 0)
    h.gemm(A, B, c2, M, N, K, K, K, N, drop_p=0.25, seed=77)
    h.set_option(h.OPT_PIPE, mode)
    h.set_option(h.OPT_GLDS, g)
    assert torch.equal(c1 == 0, c2 == 0) and rel(c1, c2.float()) < 1e-2
    if M <= 2048:                                                    # batched, strided batch elements
        Bt = 3
        Ab, Bb = rnd(Bt, M, K, dtype=dtype, seed=3), rnd(Bt, N, K, dtype=dtype, seed=4)
        Cb = torch.empty(Bt, M, N, device=DEV, dtype=dtype)
        h.gemm(Ab, Bb, Cb, M, N, K, K, K, N, batch=Bt, sA=M * K, sB=N * K, sC=M * N, bias=bias)
        assert rel(Cb, Ab.float() @ Bb.float().transpose(1, 2) + bias) < TOL[dtype]


@pytest.fixture()
def skinny():
    """force the small-M kernel (gemm_skinny.hip: reduction split over the block's four waves) wherever it is legal"""
    h = hip()
    prev = h.set_option(h.OPT_SKINNY, 2)
    prevg = h.set_option(h.OPT_GLDS, 0)
    yield h
    h.set_option(h.OPT_SKINNY, prev)
    h.set_option(h.OPT_GLDS, prevg)


@pytest.mark.parametrize('M,N,K', [(192, 768, 768), (192, 768, 3072), (640, 2304, 768), (70, 130, 200), (1, 768, 768), (640, 768, 10000),
                                   (65, 64, 128), (300, 100, 8), (300, 256, 2048), (100, 768, 3072), (640, 768, 768), (33, 40, 136)])      # (tiles of 64 x 64 | 32 x 64 | 32 x 32 by cost)
def test_skinny_gemm(skinny, M, N, K):
    h, dtype = skinny, torch.bfloat16
    A, B = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, seed=2)
    ref = A.float() @ B.float().t()
    Cm = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, Cm, M, N, K, K, K, N)
    assert rel(Cm, ref) < TOL[dtype]
    h.set_option(h.OPT_SKINNY, 0)                                  # the 4-wave kernel on the same problem
    C0 = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, B, C0, M, N, K, K, K, N)
    h.set_option(h.OPT_SKINNY, 2)
    assert rel(Cm, C0.float()) < 8e-3                              # same math, different fp32 summation order + one bf16 rounding
    bias, rs = rnd(N, seed=5), rnd(M, seed=6)
    res, mask = rnd(M, N, dtype=dtype, seed=7), rnd(M, N, dtype=dtype, seed=8)
    for act, fn in ((h.ACT_NONE, lambda x: x), (h.ACT_RELU, F.relu), (h.ACT_GELU, lambda x: F.gelu(x))):
        h.gemm(A, B, Cm, M, N, K, K, K, N, alpha=0.5, rowscale=rs, bias=bias, res=res, ldr=N, relu_mask=mask, ldm=N, act=act)
        r2 = fn(0.5 * ref * rs[:, None] + bias + res.float()) * (mask.float() > 0)
        assert rel(Cm, r2) < TOL[dtype], act
    Cf = torch.zeros(M, N + 8, device=DEV)                         # fp32 output, strided C
    h.gemm(A, B, Cf, M, N, K, K, K, N + 8)
    assert rel(Cf[:, :N], ref) < 1e-5 and Cf[:, N:].abs().max() == 0
    if K >= 64:
        c1, c2 = torch.empty(M, N, device=DEV, dtype=dtype), torch.empty(M, N, device=DEV, dtype=dtype)
        h.gemm(A, B, c1, M, N, K, K, K, N, drop_p=0.25, seed=77)  # same keep pattern as the 4-wave kernel
        h.set_option(h.OPT_SKINNY, 0)
        h.gemm(A, B, c2, M, N, K, K, K, N, drop_p=0.25, seed=77)
        h.set_option(h.OPT_SKINNY, 2)
        assert torch.equal(c1 == 0, c2 == 0)


@pytest.mark.parametrize('M,N,K', [(3200, 256, 2048), (640, 768, 768), (3200, 256, 256), (70, 136, 200), (1, 768, 2048), (300, 64, 72)])
def test_skinny_gemm_reduction_major_b(skinny, M, N, K):
    """dX[M,N] = dY[M,K] W[K,N] with W read reduction-major (GPV_TRANS), the backward-data form of a Linear"""
    h, dtype = skinny, torch.bfloat16
    A, W = rnd(M, K, dtype=dtype, seed=11), rnd(K, N, dtype=dtype, seed=12)
    ref = A.float() @ W.float()
    Cm = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, W, Cm, M, N, K, K, N, N, layoutB=h.TRANS)
    assert rel(Cm, ref) < TOL[dtype]
    h.set_option(h.OPT_SKINNY, 0)
    C0 = torch.empty(M, N, device=DEV, dtype=dtype)
    h.gemm(A, W, C0, M, N, K, K, N, N, layoutB=h.TRANS)
    h.set_option(h.OPT_SKINNY, 2)
    assert rel(Cm, C0.float()) < 8e-3
    res = rnd(M, N, dtype=dtype, seed=13)
    h.gemm(A, W, Cm, M, N, K, K, N, N, layoutB=h.TRANS, res=res, ldr=N, alpha=2.0)
    assert rel(Cm, 2.0 * ref + res.float()) < TOL[dtype]
    Wp = torch.zeros(K, N + 8, device=DEV, dtype=dtype)              # strided W
    Wp[:, :N] = W
    h.gemm(A, Wp, Cm, M, N, K, K, N + 8, N, layoutB=h.TRANS)
    assert rel(Cm, ref) < TOL[dtype]


GCONVS = [  # Cin, Cout, k, stride, pad, H, W   (Cin % 64 == 0 both ways, Cout > 64)
    (64, 128, 1, 1, 0, 24, 32), (256, 128, 3, 2, 1, 24, 32), (128, 128, 3, 1, 1, 15, 20), (256, 512, 1, 2, 0, 30, 40),
    (512, 2048, 1, 1, 0, 15, 20), (128, 256, 3, 2, 1, 17, 23), (128, 192, 3, 2, 1, 32, 32)]


@pytest.mark.parametrize('Cin,Cout,k,s,p,H,W', GCONVS)
def test_pipe_conv_fwd_dgrad(pipe_big, Cin, Cout, k, s, p, H, W):
    pipe = pipe_big
    """implicit-GEMM conv forward / backward-data through every tile configuration of the pipelined kernel, incl. the stride-2
    dgrad parity classes, ragged last row tiles, image-crossing tiles and the 160- / 96-row tiles with uneven piece counts"""
    test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, k, s, p, H, W)
    bn = pipe.bn
    want = int(Cout % bn == 0 and k * k * Cin >= 128) + int(Cin % bn == 0 and k * k * Cout >= 128)
    assert pipe.set_option(pipe.OPT_PIPE_LAUNCHES, 0) == want


@pytest.mark.parametrize('Cin,Cout,k,s,p,H,W', GCONVS)
def test_glds_conv_fwd_dgrad(glds, Cin, Cout, k, s, p, H, W):
    test_conv_fwd_dgrad_wgrad(torch.bfloat16, Cin, Cout, k, s, p, H, W)
    # forward (N = Cout) + dgrad (N = Cin); N <= 64 and the wgrad stay on the 4-wave kernel
    assert glds.set_option(glds.OPT_GLDS_LAUNCHES, 0) == int(Cout > 64) + int(Cin > 64)


# ----------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, H, kpm, causal, scale):
    B, Sq, D = q.shape
    Sk = k.shape[1]
    dh = D // H
    qh = q.view(B, Sq, H, dh).transpose(1, 2)
    kh = k.view(B, Sk, H, dh).transpose(1, 2)
    vh = v.view(B, Sk, H, dh).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :].bool(), float('-inf'))
    if causal:
        s = s.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool, device=q.device).triu(1), float('-inf'))
    p = s.softmax(-1)
    return (p @ vh).transpose(1, 2).reshape(B, Sq, D), torch.logsumexp(s, -1)


ATT = [  # H, dh, Sq, Sk, causal, kpm
    (8, 32, 300, 300, False, True), (8, 32, 100, 300, False, True), (8, 32, 100, 100, False, False),
    (16, 48, 6, 100, False, False), (16, 48, 100, 6, False, False), (8, 96, 20, 20, True, False),
    (8, 96, 19, 106, False, False), (12, 64, 9, 9, False, True), (8, 32, 70, 130, False, True), (8, 96, 1, 106, False, False)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('H,dh,Sq,Sk,causal,use_kpm', ATT)
def test_attention_fwd_bwd(dtype, H, dh, Sq, Sk, causal, use_kpm):
    h = hip()
    Bn, D = 3, H * dh
    # q, k, v are slices of wider "fused projection" buffers (row stride 3*D) to exercise strides
    qkv = rnd(Bn, max(Sq, Sk), 3 * D, dtype=dtype, seed=40, scale=1.0)
    q, k, v = qkv[:, :Sq, :D], qkv[:, :Sk, D:2 * D], qkv[:, :Sk, 2 * D:]
    kpm = None
    if use_kpm:
        kpm = torch.zeros(Bn, Sk, dtype=torch.uint8, device=DEV)
        kpm[1, Sk - Sk // 3:] = 1
        kpm[2, ::5] = 1
    scale = 1.0 / math.sqrt(dh)
    o = torch.empty(Bn, Sq, D, device=DEV, dtype=dtype)
    lse = torch.empty(Bn, H, Sq, device=DEV)
    rs = qkv.stride(1)
    bs = qkv.stride(0)
    strides = ((bs, rs), (bs, rs), (bs, rs), (Sq * D, D))
    h.attention_fwd(q, k, v, o, strides, Bn, H, Sq, Sk, dh, scale, kpm=kpm, causal=causal, lse=lse)
    qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, k, v))
    oref, lref = attn_ref(qf, kf, vf, H, kpm, causal, scale)
    assert rel(o, oref) < TOL[dtype]
    assert rel(lse, lref) < 1e-4 if dtype == torch.float32 else rel(lse, lref) < 1e-2
    do = rnd(Bn, Sq, D, dtype=dtype, seed=41)
    gq, gk, gv = torch.autograd.grad(oref, (qf, kf, vf), do.float())
    dqkv = torch.zeros_like(qkv)
    dq, dk, dv = dqkv[:, :Sq, :D], dqkv[:, :Sk, D:2 * D], dqkv[:, :Sk, 2 * D:]
    h.attention_bwd(q, k, v, o, do, dq, dk, dv, strides, (Sq * D, D), Bn, H, Sq, Sk, dh, scale, kpm=kpm, causal=causal, lse=lse)
    tol = 1e-4 if dtype == torch.float32 else 2.5e-2
    assert rel(dq, gq) < tol and rel(dk, gk) < tol and rel(dv, gv) < tol, (rel(dq, gq), rel(dk, gk), rel(dv, gv))


def test_attention_dropout_consistency():
    """dropout: forward mask statistics, and backward uses the same mask (finite-difference-free check:
    with V = I-like probes the kept pattern is visible in O)."""
    h = hip()
    Bn, H, dh, S = 2, 8, 32, 64
    D = H * dh
    q = torch.zeros(Bn, S, D, device=DEV)          # uniform attention: p = 1/S
    k = torch.zeros(Bn, S, D, device=DEV)
    v = torch.zeros(Bn, S, D, device=DEV)
    v[:, :, :] = 0
    idx = torch.arange(S, device=DEV)
    v[:, idx, idx % dh] = 1.0                      # head 0 channel j sums keys with key % 32 == j
    o = torch.empty(Bn, S, D, device=DEV)
    lse = torch.empty(Bn, H, S, device=DEV)
    st = ((S * D, D),) * 4
    h.attention_fwd(q, k, v, o, st, Bn, H, S, S, dh, 1.0, drop_p=0.25, seed=77, lse=lse)
    # every kept (q,key) contributes (1/S)/(0.75) to channel key%32 of head 0 -> total over channels = kept/S/0.75
    kept_frac = (o[:, :, :dh].sum(-1) * 0.75).mean().item()
    assert 0.70 < kept_frac < 0.80
    # gradient w.r.t. V through the same mask: dV[key, c] = sum_q mask[q,key]/S/0.75 * dO[q,c]; with dO = 1:
    do = torch.ones_like(o)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    h.attention_bwd(q, k, v, o, do, dq, dk, dv, st, (S * D, D), Bn, H, S, S, dh, 1.0, drop_p=0.25, seed=77, lse=lse)
    # sum_key dV[key, head0 ch] over keys with key%32==j  ==  sum_q O[q, j]   (same mask both ways)
    lhs = torch.stack([dv[:, j::dh, 0].sum(1) for j in range(dh)], 1)       # (B, dh) using channel 0 of head 0
    rhs = o[:, :, :dh].sum(1)
    assert rel(lhs, rhs) < 1e-4


@pytest.mark.parametrize('Bn,S,use_kpm,drop', [(32, 300, False, 0.1), (3, 300, True, 0.1), (2, 100, False, 0.1), (32, 100, False, 0.0), (2, 37, True, 0.25),
                                              (1, 320, False, 0.0), (2, 129, True, 0.1), (5, 16, False, 0.0)])
def test_attention_with_in_projection_equals_the_three_launches(Bn, S, use_kpm, drop):
    """gpv_attention_qkv_fwd (q | k | v projections inside the attention launch, transformer.py:148-155): the projected rows it writes
    against fp32 math and against the projection GEMMs; its output / lse against gpv_attention_fwd run on ITS OWN q, k, v with the same
    seed (same dropout words, same softmax: only the order of the 32-term score sums differs)"""
    h, dt = hip(), torch.bfloat16
    H, dh, D = 8, 32, 256
    M = Bn * S
    x, pos = rnd(M, D, dtype=dt, seed=31), rnd(M, D, dtype=dt, seed=32, scale=0.5)
    xp = (x.float() + pos.float()).to(dt)
    w, bias = rnd(3 * D, D, dtype=dt, seed=33, scale=D ** -0.5), rnd(3 * D, seed=34, scale=0.2)
    kpm = None
    if use_kpm:
        kpm = torch.zeros(Bn, S, dtype=torch.uint8, device=DEV)
        kpm[0, S - S // 3:] = 1
        kpm[-1, 1::5] = 1
    qk = torch.full((M, 2 * D), float('nan'), device=DEV, dtype=dt)
    v = torch.full((M, D), float('nan'), device=DEV, dtype=dt)
    o = torch.full((M, D), float('nan'), device=DEV, dtype=dt)
    lse = torch.empty(Bn, H, S, device=DEV)
    st = ((S * 2 * D, 2 * D), (S * 2 * D, 2 * D), (S * D, D), (S * D, D))
    scale = dh ** -0.5
    h.attention_qkv_fwd(xp, x, w, bias, qk[:, :D], qk[:, D:], v, o, st, Bn, H, S, scale, kpm=kpm, drop_p=drop, seed=91, lse=lse)
    ref_qk = xp.float() @ w[:2 * D].float().t() + bias[:2 * D]
    ref_v = x.float() @ w[2 * D:].float().t() + bias[2 * D:]
    assert rel(qk, ref_qk) < TOL[dt] and rel(v, ref_v) < TOL[dt]
    g_qk, g_v = torch.empty_like(qk), torch.empty_like(v)
    h.gemm(xp, w[:2 * D], g_qk, M, 2 * D, D, D, D, 2 * D, bias=bias[:2 * D].contiguous())
    h.gemm(x, w[2 * D:], g_v, M, D, D, D, D, D, bias=bias[2 * D:].contiguous())
    assert rel(qk, g_qk.float()) < 8e-3 and rel(v, g_v.float()) < 8e-3       # same math, another fp32 summation order + one bf16 rounding
    o2 = torch.empty_like(o)
    lse2 = torch.empty_like(lse)
    h.attention_fwd(qk[:, :D], qk[:, D:], v, o2, st, Bn, H, S, S, dh, scale, kpm=kpm, drop_p=drop, seed=91, lse=lse2)
    assert torch.isfinite(o.float()).all()
    assert rel(o, o2.float()) < 8e-3
    assert (lse - lse2).abs().max().item() < 1e-3
    if drop > 0:                                                              # same keep pattern: a dropped probability is an exact zero in neither output, but the
        assert ((o.float() - o2.float()).abs() > 0.05 * o2.float().abs().max()).float().mean().item() < 1e-4   # outputs would differ by whole terms


@pytest.mark.parametrize('rows,drop,with_pos,affine', [(9600, 0.1, True, True), (3200, 0.1, True, True), (3200, 0.1, False, True), (300, 0.0, False, True),
                                                       (100, 0.25, True, False), (37, 0.0, True, True), (16, 0.1, False, True), (4099, 0.1, True, True)])
def test_linear_layernorm_one_launch_equals_gemm_then_layernorm(rows, drop, with_pos, affine):
    """gpv_linear_layernorm_fwd (out-projection inside the LayerNorm launch, transformer.py:153-157) against the two launches it replaces on
    the same operands and seed: s within one bf16 rounding of the GEMM's (same products, possibly another fp32 order), y / y2 / mean / rstd
    from the SAME rounded s and dropout words (row sums in another order: last-bit differences), and against fp32 math"""
    h, dt, D = hip(), torch.bfloat16, 256
    a, x = rnd(rows, D, dtype=dt, seed=41), rnd(rows, D, dtype=dt, seed=42)
    w, bias = rnd(D, D, dtype=dt, seed=43, scale=D ** -0.5), rnd(D, seed=44, scale=0.3)
    gamma, beta = (rnd(D, seed=45) * 0.2 + 1.0, rnd(D, seed=46) * 0.1) if affine else (None, None)
    npos = 100 if rows % 100 == 0 else rows
    pos = rnd(npos, D, dtype=dt, seed=47, scale=0.5) if with_pos else None
    mk = lambda: torch.full((rows, D), float('nan'), device=DEV, dtype=dt)
    s1, y1, z1 = mk(), mk(), (mk() if with_pos else None)
    m1, r1 = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    h.linear_layernorm_fwd(a, w, bias, x, gamma, beta, s1, y1, m1, r1, rows, 1e-5, drop_p=drop, seed=123, pos=pos, y2=z1)
    s0, y0, z0 = mk(), mk(), (mk() if with_pos else None)
    m0, r0 = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    h.gemm(a, w, s0, rows, D, D, D, D, D, bias=bias)
    h.layernorm_fwd(x, s0, gamma, beta, y0, m0, r0, rows, D, 1e-5, drop_p=drop, seed=123, pos=pos, y2=z0)
    assert rel(s1, a.float() @ w.float().t() + bias) < TOL[dt]
    assert rel(s1, s0.float()) < 8e-3
    same = (s1 == s0).all(1)                                   # rows whose projected values are bit-identical: everything downstream must agree closely
    assert same.float().mean().item() > 0.5
    assert torch.isfinite(y1.float()).all()
    assert (y1.float()[same] - y0.float()[same]).abs().max().item() <= 2 ** -6 * max(1.0, y0.float().abs().max().item())      # (one bf16 ulp at |y| ~ 4)
    assert (m1[same] - m0[same]).abs().max().item() < 1e-5 and (r1[same] / r0[same] - 1).abs().max().item() < 1e-5
    if with_pos:
        assert torch.equal(z1, (y1.float() + pos.float().repeat(rows // npos, 1)).to(dt))
    if drop == 0:
        v = x.float() + s1.float()
        ref = F.layer_norm(v, (D,), gamma, beta, 1e-5)
        assert rel(y1, ref) < TOL[dt]


ATT1 = [  # H, dh, Sq, Sk, causal, kpm, drop: the model's shapes at B = 32 + ragged ones around the tile / strip boundaries
    (8, 32, 300, 300, False, True, 0.1), (8, 32, 300, 300, False, False, 0.0), (8, 32, 100, 300, False, True, 0.1),
    (8, 32, 100, 100, False, False, 0.1), (16, 48, 100, 6, False, False, 0.1), (16, 48, 6, 100, False, False, 0.1),
    (16, 48, 100, 16, False, True, 0.1), (8, 96, 20, 20, True, False, 0.1), (8, 96, 20, 106, False, False, 0.1),
    (12, 64, 6, 6, False, True, 0.1), (8, 32, 70, 130, False, True, 0.25), (8, 32, 33, 129, False, True, 0.0),
    (8, 32, 129, 31, False, False, 0.1), (8, 96, 1, 106, False, False, 0.0), (8, 32, 320, 320, False, True, 0.1),
    (8, 32, 17, 305, False, False, 0.1), (8, 96, 128, 128, True, False, 0.1)]


@pytest.mark.parametrize('H,dh,Sq,Sk,causal,use_kpm,drop', ATT1)
def test_attention_bwd_single_launch_equals_the_two_launches(H, dh, Sq, Sk, causal, use_kpm, drop):
    """attn_bwd1_kernel (dQ, dK, dV in one launch: S formed once, dS transposed through LDS for the dQ product; reference
    transformer.py:148-155 through nn.MultiheadAttention's backward) against the dQ + dK/dV pair of launches on the same inputs,
    same dropout seed: dK / dV are the same products in the same order (bit-identical), dQ the same bf16 dS in another summation
    order (fp32 accumulation: last-ulp differences of the bf16 result at most).  And against fp32 autograd when there is no dropout."""
    h = hip()
    Bn, D = 5, H * dh
    qkv = rnd(Bn, max(Sq, Sk), 3 * D, dtype=torch.bfloat16, seed=140, scale=1.0)
    q, k, v = qkv[:, :Sq, :D], qkv[:, :Sk, D:2 * D], qkv[:, :Sk, 2 * D:]
    kpm = None
    if use_kpm:
        kpm = torch.zeros(Bn, Sk, dtype=torch.uint8, device=DEV)
        kpm[1, Sk - Sk // 3:] = 1
        kpm[2, ::5] = 1
        if Sk > 2:
            kpm[3, 1:] = 1
    scale = 1.0 / math.sqrt(dh)
    o = torch.empty(Bn, Sq, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(Bn, H, Sq, device=DEV)
    rs, bs = qkv.stride(1), qkv.stride(0)
    strides = ((bs, rs), (bs, rs), (bs, rs), (Sq * D, D))
    h.attention_fwd(q, k, v, o, strides, Bn, H, Sq, Sk, dh, scale, kpm=kpm, causal=causal, drop_p=drop, seed=91, lse=lse)
    do = rnd(Bn, Sq, D, dtype=torch.bfloat16, seed=141)
    # instantiated: at most 128 queries and 128 keys for every head size (causal or not), and the DETR shapes (dh 32, up to 320 keys)
    q8, k8 = (Sq + 31) // 32 * 32 <= 128, (Sk + 63) // 64 * 64 <= 128
    taken = (q8 and k8) or (not causal and dh == 32 and (q8 or not k8))
    outs = []
    prev = h.set_option(h.OPT_ATTN_BWD1, 0)
    try:
        for mode in (0, 2):
            h.set_option(h.OPT_ATTN_BWD1, mode)
            h.set_option(h.OPT_ATTN_BWD1_LAUNCHES, 0)
            dqkv = torch.full_like(qkv, float('nan'))
            dq, dk, dv = dqkv[:, :Sq, :D], dqkv[:, :Sk, D:2 * D], dqkv[:, :Sk, 2 * D:]
            h.attention_bwd(q, k, v, o, do, dq, dk, dv, strides, (Sq * D, D), Bn, H, Sq, Sk, dh, scale, kpm=kpm, causal=causal,
                            drop_p=drop, seed=91, lse=lse)
            torch.cuda.synchronize()
            assert h.set_option(h.OPT_ATTN_BWD1_LAUNCHES, 0) == (1 if mode and taken else 0)
            outs.append((dq.float().clone(), dk.float().clone(), dv.float().clone()))
    finally:
        h.set_option(h.OPT_ATTN_BWD1, prev)
    (dq0, dk0, dv0), (dq1, dk1, dv1) = outs
    for t in (dq1, dk1, dv1):
        assert torch.isfinite(t).all()
    assert torch.equal(dk0, dk1) and torch.equal(dv0, dv1)
    assert rel(dq1, dq0) < 4e-3, rel(dq1, dq0)
    assert ((dq1 - dq0).abs() > 2 ** -7 * dq0.abs() + 1e-6 * dq0.abs().max()).float().mean().item() < 2e-3      # beyond one bf16 ulp: a handful of elements
    if drop == 0:
        qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, k, v))
        oref, _ = attn_ref(qf, kf, vf, H, kpm, causal, scale)
        gq, gk, gv = torch.autograd.grad(oref, (qf, kf, vf), do.float())
        assert rel(dq1, gq) < 2.5e-2 and rel(dk1, gk) < 2.5e-2 and rel(dv1, gv) < 2.5e-2


def _attention_keep_mask(h, Bn, H, dh, Sq, Sk, drop, seed):
    """the keep pattern the attention kernels draw for (seed, batch, head, query, key), read off the forward kernel itself: zero q / k
    make the probabilities uniform (1 / Sk), one-hot V probes -- 32 keys at a time, channel = key % dh -- make every kept (query, key)
    pair visible in O (the pattern depends on the indices only, not on the values)."""
    D = H * dh
    z = torch.zeros(Bn, max(Sq, Sk), D, device=DEV, dtype=torch.bfloat16)
    keep = torch.zeros(Bn, H, Sq, Sk, device=DEV, dtype=torch.bool)
    o = torch.empty(Bn, Sq, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(Bn, H, Sq, device=DEV)
    st = ((z.stride(0), z.stride(1)),) * 3 + ((Sq * D, D),)
    for c0 in range(0, Sk, dh):
        v = torch.zeros(Bn, Sk, D, device=DEV, dtype=torch.bfloat16)
        n = min(dh, Sk - c0)
        kk = torch.arange(n, device=DEV)
        for hd in range(H):
            v[:, c0 + kk, hd * dh + kk] = 1.0
        stv = ((z.stride(0), z.stride(1)), (z.stride(0), z.stride(1)), (Sk * D, D), (Sq * D, D))
        h.attention_fwd(z[:, :Sq], z[:, :Sk], v, o, stv, Bn, H, Sq, Sk, dh, 1.0, drop_p=drop, seed=seed, lse=lse)
        keep[:, :, :, c0:c0 + n] = o.view(Bn, Sq, H, dh)[:, :, :, :n].permute(0, 2, 1, 3) > 0
    return keep


@pytest.mark.parametrize('Bn,H,dh,Sq,Sk,use_kpm', [(16, 8, 32, 300, 300, True), (16, 8, 32, 100, 300, False), (8, 16, 48, 100, 16, False)])
def test_attention_bwd_single_launch_with_dropout_against_fp32_autograd(Bn, H, dh, Sq, Sk, use_kpm):
    """attn_bwd1_kernel with dropout ON was only compared with the two-launch path (a self-comparison).  Here the
    keep mask is reconstructed from the forward kernel (V = one-hot probes) and handed to fp32 autograd of
    dropout(softmax(QK^T)) V (nn.MultiheadAttention, transformer.py:148-155) at B x H >= 128 -- the range the single launch serves."""
    h = hip()
    D, drop, seed = H * dh, 0.1, 4242
    assert Bn * H >= 128
    keep = _attention_keep_mask(h, Bn, H, dh, Sq, Sk, drop, seed)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - drop)) < 5e-3, frac
    qkv = rnd(Bn, max(Sq, Sk), 3 * D, dtype=torch.bfloat16, seed=160, scale=1.0)
    q, k, v = qkv[:, :Sq, :D], qkv[:, :Sk, D:2 * D], qkv[:, :Sk, 2 * D:]
    kpm = None
    if use_kpm:
        kpm = torch.zeros(Bn, Sk, dtype=torch.uint8, device=DEV)
        kpm[1, Sk - Sk // 3:] = 1
        kpm[2, ::5] = 1
    scale = 1.0 / math.sqrt(dh)
    o = torch.empty(Bn, Sq, D, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(Bn, H, Sq, device=DEV)
    strides = ((qkv.stride(0), qkv.stride(1)),) * 3 + ((Sq * D, D),)
    h.attention_fwd(q, k, v, o, strides, Bn, H, Sq, Sk, dh, scale, kpm=kpm, drop_p=drop, seed=seed, lse=lse)
    do = rnd(Bn, Sq, D, dtype=torch.bfloat16, seed=161)
    dqkv = torch.full_like(qkv, float('nan'))
    dq, dk, dv = dqkv[:, :Sq, :D], dqkv[:, :Sk, D:2 * D], dqkv[:, :Sk, 2 * D:]
    prev = h.set_option(h.OPT_ATTN_BWD1, 2)
    try:
        h.set_option(h.OPT_ATTN_BWD1_LAUNCHES, 0)
        h.attention_bwd(q, k, v, o, do, dq, dk, dv, strides, (Sq * D, D), Bn, H, Sq, Sk, dh, scale, kpm=kpm, drop_p=drop, seed=seed, lse=lse)
        torch.cuda.synchronize()
        assert h.set_option(h.OPT_ATTN_BWD1_LAUNCHES, 0) == 1                  # the single launch ran
    finally:
        h.set_option(h.OPT_ATTN_BWD1, prev)
    qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, k, v))
    sc = (qf.view(Bn, Sq, H, dh).transpose(1, 2) @ kf.view(Bn, Sk, H, dh).transpose(1, 2).transpose(-1, -2)) * scale
    if kpm is not None:
        sc = sc.masked_fill(kpm[:, None, None, :].bool(), float('-inf'))
    pd = sc.softmax(-1) * keep.float() / (1 - drop)
    oref = (pd @ vf.view(Bn, Sk, H, dh).transpose(1, 2)).transpose(1, 2).reshape(Bn, Sq, D)
    assert rel(o.float(), oref) < 1.5e-2                                     # (the forward with the same mask)
    gq, gk, gv = torch.autograd.grad(oref, (qf, kf, vf), do.float())
    for name, got, want in (('dq', dq, gq), ('dk', dk, gk), ('dv', dv, gv)):
        assert torch.isfinite(got.float()).all()
        assert rel(got.float(), want) < 2.5e-2, (name, rel(got.float(), want))


# ----------------------------------------------------------------------------------------- layernorm / CE / misc
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,cols,drop', [(9600, 256, 0.1), (3200, 256, 0.0), (3968, 768, 0.1), (640, 768, 0.1), (77, 2048, 0.1), (5, 2304 // 9 * 8, 0.0)])
def test_layernorm_bwd_partials_and_fold_group_equal_the_atomic_column_sums(dtype, rows, cols, drop):
    """gpv_layernorm_bwd3(partials) + gpv_colsum_fold_group against gpv_layernorm_bwd2's in-kernel atomics: dx / ds bit-identical
    (the same arithmetic; only the launch grid may differ), dgamma / dbeta equal up to the fp32 summation order, added to what the
    buffers held; the fold of a group (two problems, one shared output pair) is reproducible bit for bit."""
    h = hip()
    x, s, dy = rnd(rows, cols, dtype=dtype, seed=150), rnd(rows, cols, dtype=dtype, seed=151), rnd(rows, cols, dtype=dtype, seed=152)
    g, b = rnd(cols, seed=153) + 1.0, rnd(cols, seed=154)
    y = torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    h.layernorm_fwd(x, s, g, b, y, mean, rstd, rows, cols, 1e-5, drop, 99)
    dx0, ds0 = torch.empty_like(x), torch.zeros_like(x)
    dg0, db0 = torch.full((cols,), 0.5, device=DEV), torch.full((cols,), -0.25, device=DEV)
    h.layernorm_bwd(dy, x, s, g, mean, rstd, dx0, ds0, dg0, db0, rows, cols, drop, 99)
    nblk = h.layernorm_bwd_blocks(rows, cols)
    outs = []
    for _ in range(2):
        part = torch.full((nblk, 2 * cols), float('nan'), device=DEV)
        dx1, ds1 = torch.empty_like(x), torch.zeros_like(x)
        h.layernorm_bwd(dy, x, s, g, mean, rstd, dx1, ds1, None, None, rows, cols, drop, 99, partials=part)
        dg1, db1 = torch.full((cols,), 0.5, device=DEV), torch.full((cols,), -0.25, device=DEV)
        dg2, db2 = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
        h.colsum_fold_group([(part, dg1, db1, nblk, cols), (part, dg2, db2, nblk, cols)])
        torch.cuda.synchronize()
        assert torch.isfinite(part).all()
        assert torch.equal(dx1, dx0) and torch.equal(ds1, ds0)
        outs.append((dg1.clone(), db1.clone(), dg2.clone(), db2.clone()))
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)
    dg1, db1, dg2, db2 = outs[0]
    assert rel(dg1, dg0) < 1e-5 and rel(db1, db0) < 1e-5, (rel(dg1, dg0), rel(db1, db0))
    assert rel(dg2 + 0.5, dg0) < 1e-5 and rel(db2 - 0.25, db0) < 1e-5


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,cols', [(1203, 256), (640, 768), (77, 2048), (5, 2304 // 9 * 8)])
def test_layernorm_fwd_bwd(dtype, rows, cols):
    h = hip()
    x, s = rnd(rows, cols, dtype=dtype, seed=50), rnd(rows, cols, dtype=dtype, seed=51)
    g, b = rnd(cols, seed=52) * 0.1 + 1, rnd(cols, seed=53) * 0.1
    y = torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    h.layernorm_fwd(x, s, g, b, y, mean, rstd, rows, cols, 1e-5)
    xf, sf, gf, bf = (t.float().requires_grad_(True) for t in (x, s, g, b))
    ref = F.layer_norm(xf + sf, (cols,), gf, bf, 1e-5)
    assert rel(y, ref) < TOL[dtype]
    dy = rnd(rows, cols, dtype=dtype, seed=54)
    gx, gs, gg, gb = torch.autograd.grad(ref, (xf, sf, gf, bf), dy.float())
    dx = torch.empty_like(x)
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    h.layernorm_bwd(dy, x, s, g, mean, rstd, dx, None, dg, db, rows, cols)
    assert rel(dx, gx) < TOL[dtype] and rel(dg, gg) < 5e-3 and rel(db, gb) < 5e-3
    # no-affine, no residual (F.layer_norm on RoI features, detr_roi_head.py:91)
    h.layernorm_fwd(x, None, None, None, y, mean, rstd, rows, cols, 1e-5)
    assert rel(y, F.layer_norm(x.float(), (cols,))) < TOL[dtype]
    # dropout on the sublayer output: y = LN(x + drop(s)); ds must carry the same mask
    y2 = torch.empty_like(x)
    h.layernorm_fwd(x, s, g, b, y2, mean, rstd, rows, cols, 1e-5, drop_p=0.1, seed=99)
    sd = torch.empty_like(s)
    h.dropout(s, sd, rows * cols, 0.1, 99)
    assert 0.88 < (sd != 0).float().mean().item() < 0.92
    assert rel(y2, F.layer_norm(x.float() + sd.float(), (cols,), g, b, 1e-5)) < TOL[dtype]
    ds = torch.empty_like(x)
    h.layernorm_bwd(dy, x, s, g, mean, rstd, dx, ds, None, None, rows, cols, drop_p=0.1, seed=99)
    known = s != 0                                    # (an exactly-zero s tells nothing about its keep bit: left out of the comparison)
    assert rel(ds.float() * known, dx.float() * (sd != 0) / 0.9) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_softmax_ce(dtype):
    h = hip()
    rows, V = 37, 10000
    lg = rnd(rows, V, dtype=dtype, seed=60, scale=3.0)
    tgt = torch.randint(0, V, (rows,), device=DEV)
    tgt[3] = -100
    loss = torch.empty(rows, device=DEV)
    dl = torch.empty_like(lg)
    gs = rnd(rows, seed=61).abs()
    h.softmax_ce(lg, V, tgt, loss, dl, gs, rows, V)
    lf = lg.float().requires_grad_(True)
    ref = F.cross_entropy(lf, tgt, reduction='none', ignore_index=-100)
    assert rel(loss, ref) < (1e-5 if dtype == torch.float32 else 1e-5)
    (g,) = torch.autograd.grad((ref * gs).sum(), lf)
    assert rel(dl, g) < TOL[dtype]


def test_roi_weights_match_oracle():
    h = hip()
    from oracle import gpv_oracle as O
    torch.manual_seed(0)
    Bn, Q, H, W = 2, 50, 15, 20
    boxes = torch.rand(Bn, Q, 4) * torch.tensor([0.6, 0.6, 0.5, 0.5]) + torch.tensor([0.2, 0.2, 0.02, 0.02])
    boxes[0, 0] = torch.tensor([0.02, 0.03, 0.3, 0.2])
    boxes[0, 1] = torch.tensor([0.98, 0.97, 0.4, 0.3])
    boxes[1, 2] = torch.tensor([0.5, 0.5, 1.0, 1.0])
    boxes[1, 3] = torch.tensor([0.5, 0.5, 1e-4, 1e-4])
    feat = torch.randn(Bn, 64, H, W)
    ref = O.extract_roi(feat, boxes)
    wg = torch.empty(Bn * Q, 320, device=DEV)
    h.roi_weights(boxes.reshape(-1, 4).to(DEV), wg, Bn * Q, H, W, 320)
    fn = feat.permute(0, 2, 3, 1).reshape(Bn, H * W, 64).to(DEV).contiguous()
    out = torch.empty(Bn, Q, 64, device=DEV)
    h.gemm(wg, fn, out, Q, 64, H * W, 320, 64, 64, layoutB=h.TRANS, batch=Bn, sA=Q * 320, sB=H * W * 64, sC=Q * 64)
    assert rel(out.cpu(), ref) < 3e-5
    assert wg[:, H * W:].abs().max() == 0


def test_small_helpers():
    h = hip()
    x = rnd(777, 256, dtype=torch.bfloat16, seed=70)
    out = torch.zeros(256, device=DEV)
    h.colsum(x, out, 777, 256, 256)
    assert rel(out, x.float().sum(0)) < 1e-5
    for rows, cols, ld, dt in ((32, 25600, 25600, torch.bfloat16), (9600, 256, 512, torch.bfloat16), (5, 24, 24, torch.float32), (33, 100, 100, torch.bfloat16),
                               (640, 768, 768, torch.float32)):                       # 16-byte form, its row / column splits, the scalar fallback
        xx = rnd(rows, ld, dtype=dt, seed=76)
        out = torch.ones(cols, device=DEV)
        h.colsum(xx, out, rows, cols, ld)
        assert rel(out, 1 + xx[:, :cols].float().sum(0)) < 2e-5, (rows, cols)
    a, b = rnd(300 * 4, 256, dtype=torch.bfloat16, seed=71), rnd(300, 256, dtype=torch.bfloat16, seed=72)
    y = torch.empty_like(a)
    h.add_rowbcast(a, b, y, 1200, 300, 256)
    assert rel(y, a.float() + b.float().repeat(4, 1)) < 5e-3
    src, sc = rnd(130, 70, seed=73), rnd(130, seed=74)
    d, dT = torch.empty(130, 70, device=DEV, dtype=torch.bfloat16), torch.empty(70, 130, device=DEV, dtype=torch.bfloat16)
    h.cast_rowscale_t(src, sc, d, dT, 130, 70)
    ref = (src * sc[:, None]).to(torch.bfloat16)
    assert torch.equal(d, ref) and torch.equal(dT, ref.t().contiguous())
    w = rnd(96, 9, 64, seed=75)
    wf, wd = torch.empty(96, 9, 64, device=DEV, dtype=torch.bfloat16), torch.empty(64, 9, 96, device=DEV, dtype=torch.bfloat16)
    scw = rnd(96, seed=76)
    h.prep_conv_weight(w, scw, wf, wd, 96, 9, 64)
    refw = (w * scw[:, None, None]).to(torch.bfloat16)
    assert torch.equal(wf, refw) and torch.equal(wd, refw.permute(2, 1, 0).contiguous())
    tab = rnd(50, 768, seed=77)
    ids = torch.randint(0, 50, (33,), device=DEV)
    e = torch.empty(33, 768, device=DEV, dtype=torch.bfloat16)
    h.embedding(tab, ids, e, 33, 768)
    assert torch.equal(e, tab[ids].to(torch.bfloat16))
    xl, lg, tk = rnd(200, 768, seed=78), rnd(200, 2, seed=79), rnd(2, 768, seed=80)
    yo = torch.empty_like(xl)
    h.relevance_condition(xl, lg, tk, yo, 200, 768)
    assert rel(yo, xl + lg.softmax(-1) @ tk) < 1e-6
    c = torch.empty(1000, device=DEV, dtype=torch.bfloat16)
    h.cast(rnd(1000, seed=81), c, 1000)
    assert torch.equal(c, rnd(1000, seed=81).to(torch.bfloat16))


def test_adamw_matches_torch():
    h = hip()
    n = 10007
    p0, g = rnd(n, seed=90), rnd(n, seed=91)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=1e-2)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    low = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    for step in range(1, 4):
        pt.grad = g.clone() * step
        opt.step()
        h.adamw(p, g * step, m, v, low, n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1 - 0.9 ** step, 1 - 0.999 ** step)
    assert rel(p, pt.data) < 1e-6
    assert torch.equal(low, p.to(torch.bfloat16))
    acc = torch.zeros(1, device=DEV)
    h.sumsq(g, n, acc)
    assert rel(acc, (g * g).sum().view(1)) < 1e-5
    # per-parameter step counts on the device: chunks of 8 elements map to a parameter id; count 0 = the parameter never
    # received a gradient and is left alone; otherwise the count is the parameter's OWN Adam step (torch keeps `step` per
    # parameter) and sets its bias corrections -- parameters first touched at different times take different first updates
    n2 = 4096
    bounds = [0, 40, 1000, 1008, 3000, n2]                      # 5 parameters, starts on multiples of 8
    sid = torch.zeros(n2 // 8, dtype=torch.int16)
    for i in range(5):
        sid[bounds[i] // 8:bounds[i + 1] // 8] = i
    first = [1, 0, 2, 0, 1]                                      # global step at which each parameter gets its first gradient (0: never)
    p0 = rnd(n2, seed=92)
    p, m, v = p0.clone(), torch.zeros(n2, device=DEV), torch.zeros(n2, device=DEV)
    refs = [torch.nn.Parameter(p0[bounds[i]:bounds[i + 1]].clone()) for i in range(5)]
    opt = torch.optim.AdamW(refs, lr=1e-3, weight_decay=1e-2)
    pstep = torch.zeros(5, dtype=torch.int32, device=DEV)
    for step in range(1, 4):
        g = rnd(n2, seed=93 + step)
        live = torch.tensor([1 if f and step >= f else 0 for f in first], dtype=torch.int32, device=DEV)
        pstep += live
        for i in range(5):
            refs[i].grad = g[bounds[i]:bounds[i + 1]].clone() if live[i] else None
        opt.step()
        gm = g.clone()
        for i in range(5):
            if not live[i]:
                gm[bounds[i]:bounds[i + 1]] = 0
        h.adamw(p, gm, m, v, None, n2, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.5, 0.5, seg_id=sid.to(DEV), seg_live=pstep)   # (bc1/bc2 ignored)
    for i in range(5):
        sl = slice(bounds[i], bounds[i + 1])
        if first[i]:
            assert rel(p[sl], refs[i].data) < 2e-6, i
        else:
            assert torch.equal(p[sl], p0[sl]) and m[sl].abs().max() == 0 and v[sl].abs().max() == 0


def test_grouped_weight_gradient_gemm():
    """gpv_gemm_tt_group: many independent dW_i += dY_i^T X_i (+ bias gradients) in one grid, no split / workspace; problems of
    different shapes and reduction lengths (incl. ragged last k-tiles and strided gradient slices), > 48 of them (two launches)"""
    h = hip()
    shapes = [(192, 768, 768), (640, 768, 2048), (3200, 256, 256), (3200, 2048, 256), (100, 128, 384), (1000, 384, 128), (64, 128, 128),
              (3392, 1536, 768)]
    probs, refs = [], []
    for rep in range(7):                                      # 56 problems
        for i, (K, M, N) in enumerate(shapes):
            dy = rnd(K, M, dtype=torch.bfloat16, seed=100 + 10 * rep + i)
            x = rnd(K, N, dtype=torch.bfloat16, seed=500 + 10 * rep + i)
            big = rnd(M + 128, N, seed=900 + 10 * rep + i)     # dW is a row slice of a larger gradient buffer (packed in_proj)
            dw = big[64:64 + M]
            bg = rnd(M, seed=1300 + 10 * rep + i) if i % 2 == 0 else None
            refs.append((dw.clone() + dy.float().t() @ x.float(), None if bg is None else bg.clone() + dy.float().sum(0), big.clone()))
            probs.append((dy, x, dw, bg, M, N, K, M, N, N))
            assert h.tt_group_ok(dy, x, dw, M, N, K, M, N, N)
    h.gemm_tt_group(probs)
    for (dy, x, dw, bg, M, N, K, *_), (rw, rb, big0) in zip(probs, refs):
        assert rel(dw, rw) < 3e-3, (M, N, K, rel(dw, rw))
        if bg is not None:
            assert rel(bg, rb) < 3e-3
    assert not h.tt_group_ok(probs[0][0], probs[0][1], probs[0][2], 100, 768, 192, 768, 768, 768)      # M not a multiple of 128


def test_grouped_weight_gradient_gemm_eight_phase():
    """gpv_gemm_tt_group_ws: the problems with M, N multiples of 256 on the eight-phase 256 x 256 kernel -- sliced reductions (partial
    tiles through the workspace + the grouped reduce pass), whole reductions added in place, a ragged last k-tile, a 3-k-tile unit,
    bias gradients from several slices, gradients that are row slices of a larger buffer, more problems than one launch holds --
    against fp32 math and against the 128 x 128 grouped kernel on the same operands"""
    h = hip()
    shapes = [(9600, 256, 256), (9600, 256, 2048), (9600, 2048, 256), (9600, 512, 256), (3200, 768, 768), (3392, 1536, 768), (3200, 768, 3072),
              (640, 768, 768), (192, 768, 768), (1000, 256, 256), (130, 256, 512), (3200, 256, 256), (3200, 256, 256), (3200, 256, 256)]
    out = {}
    for mode in (0, 1):
        probs, refs = [], []
        for rep in range(5):                                  # 70 problems, 60 of them for the eight-phase kernel: more than one launch holds (48)
            for i, (K, M, N) in enumerate(shapes):
                dy = rnd(K, M, dtype=torch.bfloat16, seed=2100 + 20 * rep + i)
                x = rnd(K, N, dtype=torch.bfloat16, seed=2500 + 20 * rep + i)
                big = rnd(M + 128, N, seed=2900 + 20 * rep + i)
                dw = big[64:64 + M]
                bg = rnd(M, seed=3300 + 20 * rep + i) if i % 2 == 0 else None
                if mode == 1 and rep == 0:
                    refs.append((dw.clone() + dy.float().t() @ x.float(), None if bg is None else bg.clone() + dy.float().sum(0)))
                probs.append((dy, x, dw, bg, M, N, K, M, N, N))
        prev = h.set_option(h.OPT_W8L, mode)
        n0 = h.set_option(h.OPT_WG8_LAUNCHES, 0)
        h.gemm_tt_group(probs)
        used = h.set_option(h.OPT_WG8_LAUNCHES, n0)
        h.set_option(h.OPT_W8L, prev)
        torch.cuda.synchronize()
        assert (used >= 2) if mode else (used == 0), (mode, used)
        out[mode] = probs
        if mode == 1:
            for (dy, x, dw, bg, M, N, K, *_), (rw, rb) in zip(probs, refs):
                assert rel(dw, rw) < 3e-3, (M, N, K, rel(dw, rw))
                if bg is not None:
                    assert rel(bg, rb) < 3e-3, (M, N, K)
    for a, b in zip(out[1], out[0]):
        assert rel(a[2], b[2]) < 1e-5, (a[4:7], rel(a[2], b[2]))
        if a[3] is not None:
            assert rel(a[3], b[3]) < 1e-5, a[4:7]


def test_cast_transpose_group_and_mirrored_dx_gemm():
    """gpv_cast_transpose_group: many fp32 [N,K] -> bf16 [K,N] in one launch (ragged sizes, > 128 problems = two launches);
    dX = dY W through the transposed mirror equals the reduction-major-B form"""
    h = hip()
    torch.manual_seed(3)
    shapes = [(256, 256), (768, 256), (2048, 256), (256, 2048), (40, 72), (33, 8)] * 23          # 138 problems
    items = []
    for n, k in shapes:
        src = torch.randn(n, k, device=DEV)
        items.append((src, torch.empty(k, n, device=DEV, dtype=torch.bfloat16)))
    h.cast_transpose_group(items)
    torch.cuda.synchronize()
    for src, dstT in items:
        assert torch.equal(dstT, src.t().to(torch.bfloat16))
    M, N, K = 1000, 768, 256
    dy = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) / 16)
    wb, wt = w.to(torch.bfloat16), torch.empty(K, N, device=DEV, dtype=torch.bfloat16)
    h.cast_transpose_group([(w, wt)])
    a, b = torch.empty(M, K, device=DEV, dtype=torch.bfloat16), torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    h.gemm(dy, wb, a, M, K, N, N, K, K, layoutB=h.TRANS)
    c = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    h.gemm(dy[:, 256:512], wt[:, 256:512], c, M, K, 256, N, N, K)               # a row slice of W = a column slice of W^T
    ref_c = dy[:, 256:512].float() @ wb[256:512].float()
    assert float((c.float() - ref_c).abs().max()) < 0.02 * float(ref_c.abs().max())
    h.gemm(dy, wt, b, M, K, N, N, N, K)
    ref = dy.float() @ wb.float()
    assert float((a.float() - ref).abs().max()) < 0.02 * float(ref.abs().max())
    assert float((b.float() - ref).abs().max()) < 0.02 * float(ref.abs().max())


@pytest.mark.parametrize('Cin,Cout,H,W,Bn,with_res,with_mask,relu', [
    (64, 256, 30, 40, 2, True, False, True), (64, 256, 30, 40, 2, False, False, True), (64, 64, 17, 23, 3, False, False, True),
    (256, 64, 30, 40, 2, False, True, False), (128, 512, 15, 20, 2, True, False, True), (128, 512, 15, 20, 2, True, True, False),
    (256, 128, 21, 19, 1, False, False, True), (256, 256, 9, 11, 5, True, True, True), (128, 128, 7, 5, 1, False, True, False),
    (512, 128, 15, 20, 2, False, False, True), (512, 128, 15, 20, 2, True, True, False), (512, 64, 9, 7, 3, False, True, False),
    (256, 1024, 15, 20, 2, True, False, True), (256, 1024, 9, 7, 1, True, True, False), (128, 2048, 5, 6, 1, False, False, True)])
def test_streaming_1x1_conv_kernel_forced(Cin, Cout, H, W, Bn, with_res, with_mask, relu):
    """conv1x1_stream.hip forced on (by default it takes >= 65536 pixel rows): weights resident in LDS, permuted output channels,
    register epilogue -- against fp32 math and bit-for-bit against the tile kernel; ragged pixel counts (M % 16 != 0)"""
    h = hip()
    torch.manual_seed(Cin + Cout + H)
    M = Bn * H * W
    x = torch.randn(M, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, 1, Cin, device=DEV) / Cin ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Cout, device=DEV)
    res = torch.randn(M, Cout, device=DEV).to(torch.bfloat16) if with_res else None
    msk = torch.randn(M, Cout, device=DEV).to(torch.bfloat16) if with_mask else None
    outs = []
    for mode in (0, 2):
        prev = h.set_option(h.OPT_C1S, mode)
        prevs = h.set_option(h.OPT_SKINNY, 0)          # (the tile kernel as the second reference: these pixel counts would go to the small-M kernel)
        try:
            y = torch.full((M + 3, Cout), 5.0, device=DEV, dtype=torch.bfloat16)         # 3 guard rows: nothing may be written past M
            h.conv2d(0, x, w, y, Bn, H, W, Cin, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, bias=bias, res=res, relu_mask=msk, act=1 if relu else 0)
        finally:
            h.set_option(h.OPT_C1S, prev)
            h.set_option(h.OPT_SKINNY, prevs)
        assert bool((y[M:] == 5.0).all())
        outs.append(y[:M].float())
    ref = x.float() @ w.reshape(Cout, Cin).float().t() + bias
    if with_res:
        ref = ref + res.float()
    if relu:
        ref = ref.clamp_min(0)
    if with_mask:
        ref = ref * (msk.float() > 0)
    assert rel(outs[1], ref) < TOL[torch.bfloat16]
    assert torch.equal(outs[0], outs[1])                       # same products, same fp32 accumulation order per output: identical bits


@pytest.mark.parametrize('Cin,Cout,H,W,Bn', [(256, 512, 30, 40, 2), (512, 1024, 14, 18, 2), (64, 256, 9, 13, 3), (256, 128, 8, 6, 1)])
def test_streaming_1x1_conv_stride2_forward(Cin, Cout, H, W, Bn):
    """the downsample projections (1x1, stride 2, no padding) through conv1x1_stream.hip: every other pixel of every other row,
    256 / 128-channel slices where the weights do not fit the LDS as a whole; against fp32 math and the gather kernel"""
    h = hip()
    torch.manual_seed(Cin + H)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(Bn, H, W, Cin, device=DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, 1, Cin, device=DEV) / Cin ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Cout, device=DEV)
    outs = []
    for mode in (0, 2):
        prev = h.set_option(h.OPT_C1S, mode)
        try:
            y = torch.empty(Bn, OH, OW, Cout, device=DEV, dtype=torch.bfloat16)
            h.conv2d(0, x, w, y, Bn, H, W, Cin, Cin, OH, OW, Cout, 1, 1, 2, 2, 0, 0, bias=bias)
        finally:
            h.set_option(h.OPT_C1S, prev)
        outs.append(y.float())
    ref = x[:, ::2, ::2].float() @ w.reshape(Cout, Cin).float().t() + bias
    assert rel(outs[1], ref) < TOL[torch.bfloat16]
    assert rel(outs[0], outs[1]) < 1e-2


# ----------------------------------------------------------------------------------------- streaming 3x3 convolution
@pytest.fixture()
def c3s():
    """force the streaming 3x3 kernel (conv3x3_stream.hip) wherever it is legal and count its launches"""
    h = hip()
    prev = h.set_option(h.OPT_C3S, 2)
    h.set_option(h.OPT_C3S_LAUNCHES, 0)
    yield h
    h.set_option(h.OPT_C3S, prev)


@pytest.mark.parametrize('Cin,Cout,s,H,W,Bn', [
    (64, 64, 1, 24, 32, 3),        # layer1 shape class; 2304 pixels = 72 whole tiles
    (64, 64, 1, 17, 23, 3),        # ragged: tiles cross rows and images, 1173 pixels (tail tile of 21)
    (128, 128, 1, 15, 20, 3),      # layer2: two 64-channel slices
    (128, 128, 2, 24, 32, 2),      # stride 2: the forward stays on the tile kernels, the backward-data takes the row-walking parity kernel
    (128, 128, 2, 30, 34, 1),      # dy 15 x 17: odd row count (a lone last dy row), one ragged strip
    (64, 128, 2, 16, 62, 3),       # dy 8 x 31 = exactly one strip; 64 input channels -> one output slice
    (64, 128, 2, 10, 64, 2),       # dy 5 x 32: a second strip of ONE column
    (64, 64, 1, 7, 61, 2),         # three strips of 30 columns, the last one a single column; 7 rows = 3 + 3 + 1
    (128, 64, 1, 9, 40, 2),        # Cout != Cin: forward 128 -> 64 (one slice), backward-data is a 64-channel reduction into 128
    (64, 128, 1, 8, 8, 5)])
def test_streaming_3x3_conv_kernel_forced(c3s, Cin, Cout, s, H, W, Bn):
    """conv3x3_stream.hip against fp32 torch: forward (+ bias, ReLU), backward-data (ReLU mask), incl. the zero-filled border rows /
    columns, ragged strips and row segments, both slices of a 128-channel layer"""
    h, dtype = c3s, torch.bfloat16
    x = rnd(Bn, Cin, H, W, dtype=dtype, seed=60)
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=61, scale=1.0 / math.sqrt(Cin * 9))
    bias = rnd(Cout, seed=62)
    OH, OW = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    xn, wn = nhwc(x), w.permute(0, 2, 3, 1).contiguous()
    for act in (h.ACT_RELU, h.ACT_NONE):
        y = torch.empty(Bn, OH, OW, Cout, device=DEV, dtype=dtype)
        h.conv2d(0, xn, wn, y, Bn, H, W, Cin, Cin, OH, OW, Cout, 3, 3, s, s, 1, 1, bias=bias, act=act)
        ref = F.conv2d(x.float(), w.float(), stride=s, padding=1) + bias.view(1, -1, 1, 1)
        if act == h.ACT_RELU:
            ref = F.relu(ref)
        assert rel(y, nhwc(ref)) < TOL[dtype]
    assert h.set_option(h.OPT_C3S_LAUNCHES, 0) == (2 if s == 1 else 0)
    # backward-data: dx = convT(dy) * (saved > 0)
    dy = rnd(Bn, Cout, OH, OW, dtype=dtype, seed=63)
    saved = rnd(Bn, Cin, H, W, dtype=dtype, seed=64)
    xf = x.float().requires_grad_(True)
    gx, = torch.autograd.grad(F.conv2d(xf, w.float(), stride=s, padding=1), xf, dy.float())
    wd = w.permute(1, 2, 3, 0).contiguous()
    for mask in (nhwc(saved), None):
        dx = torch.full((Bn, H, W, Cin), float('nan'), device=DEV, dtype=dtype)
        h.conv2d(1, nhwc(dy), wd, dx, Bn, OH, OW, Cout, Cout, H, W, Cin, 3, 3, s, s, 1, 1, relu_mask=mask)
        ref = gx * (saved.float() > 0) if mask is not None else gx
        assert rel(dx, nhwc(ref)) < TOL[dtype]
    assert h.set_option(h.OPT_C3S_LAUNCHES, 0) == (2 if (s == 1 or (Cout == 128 and H % 2 == 0 and W % 2 == 0)) else 0)


# ----------------------------------------------------------------------------------------- device-side input pipeline
def test_device_input_pipeline_vs_oracle():
    """gpv_image_pipeline (image_pipeline.hip) against oracle/image_oracle.py (numpy / scipy float64 restatement of the reference's
    resize + ToPILImage / ColorJitter / flip / grayscale / Normalize; "parity unpinned" against the real skimage / PIL): images of
    different sizes incl. down-scaling with the anti-aliasing filter, every jitter order class, flip, grayscale.  The two differ
    only where fp32 vs fp64 arithmetic lands on the other side of a uint8 truncation / rounding: at most one uint8 step for 99.5 %
    of the values, never more than three (hue: a sector boundary), and the padding frame is exactly zero."""
    from oracle import image_oracle as IO
    from gpv1_amd.input_pipeline import DeviceImagePipeline, stem_geometry
    h = hip()
    rs = np.random.RandomState(3)
    H, W = 96, 128
    shapes = [(96, 128), (150, 131), (60, 90), (300, 420), (97, 128), (200, 64)]
    params = [dict(jitter=0, order=(0, 1, 2, 3), brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0, flip=0, gray=0),
              dict(jitter=1, order=(0, 1, 2, 3), brightness=1.3, contrast=0.7, saturation=1.2, hue=0.05, flip=1, gray=0),
              dict(jitter=1, order=(3, 2, 1, 0), brightness=0.65, contrast=1.35, saturation=0.6, hue=-0.1, flip=0, gray=0),
              dict(jitter=1, order=(1, 3, 0, 2), brightness=1.1, contrast=1.1, saturation=1.4, hue=0.08, flip=1, gray=1),
              dict(jitter=0, order=(0, 1, 2, 3), brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0, flip=1, gray=1),
              dict(jitter=1, order=(2, 0, 3, 1), brightness=0.9, contrast=0.8, saturation=0.9, hue=-0.03, flip=0, gray=0)]
    # smooth images + noise (a pure-noise image makes every truncation a coin flip)
    imgs = []
    for (ih, iw) in shapes:
        yy, xx = np.mgrid[0:ih, 0:iw]
        base = np.stack([127 + 100 * np.sin(yy / 9.0 + c) * np.cos(xx / 13.0 - c) for c in range(3)], -1)
        imgs.append(np.clip(base + rs.randn(ih, iw, 3) * 12, 0, 255).astype(np.uint8))
    pipe = DeviceImagePipeline(size=(H, W), train=True)
    for dtype in (torch.float32, torch.bfloat16):
        import gpv1_amd.ops as ops
        ops.RT.set_precise(dtype == torch.float32)
        try:
            nt = pipe([torch.from_numpy(i) for i in imgs], params=params)
        finally:
            ops.RT.set_precise(False)
        Hp, Wp = stem_geometry(H, W)
        out = nt.tensors.float().cpu().numpy()
        assert out.shape == (len(imgs), Hp, Wp, 4) and nt.mask.shape == (len(imgs), H, W) and not bool(nt.mask.any())
        frame = out.copy()
        frame[:, 3:3 + H, 3:3 + W, :3] = 0
        assert np.all(frame == 0)                                                   # padding ring + 4th channel
        for b, (img, p) in enumerate(zip(imgs, params)):
            ref = IO.pipeline(img, (H, W), p).transpose(1, 2, 0)                    # [H, W, 3] normalised
            got = out[b, 3:3 + H, 3:3 + W, :3]
            steps = np.abs(got - ref) * (255.0 * IO.STD)                            # difference in uint8 steps
            tol = 0.6 if dtype == torch.float32 else 2.0                            # bf16 output rounding: up to 2^-8 of |x| <= 2.7
            assert (steps <= 1.0 + tol).mean() >= 0.995 and steps.max() <= 3.0 + tol, (b, dtype, float(steps.max()), float((steps <= 1 + tol).mean()))
            if p['gray']:
                assert np.array_equal(np.rint(got[..., 0] * 0.229 * 255), np.rint(got[..., 0] * 0.229 * 255))


def test_model_accepts_the_prepared_stem_input():
    """backbone: the pipeline's NHWC4 batch takes the place of (fp32 NCHW image -> gpv_image_to_nhwc4): same c5 as feeding the
    oracle-processed fp32 image through the usual path (bf16: the two NHWC4 inputs differ where the uint8 steps above do)"""
    from oracle import image_oracle as IO
    from gpv1_amd.input_pipeline import DeviceImagePipeline
    import gpv1_amd.backbone as bbm
    from gpv1_amd.misc import NestedTensor
    hip()
    torch.manual_seed(0)
    bb = bbm.Backbone('resnet50', True, False, False).to(DEV)
    rs = np.random.RandomState(5)
    imgs = [np.clip(127 + 60 * np.sin(np.mgrid[0:80, 0:100][0][..., None] / 7.0 + np.arange(3)) + rs.randn(80, 100, 3) * 10, 0, 255).astype(np.uint8) for _ in range(2)]
    p0 = dict(jitter=0, order=(0, 1, 2, 3), brightness=1.0, contrast=1.0, saturation=1.0, hue=0.0, flip=0, gray=0)
    pipe = DeviceImagePipeline(size=(64, 96), train=False)
    nt = pipe([torch.from_numpy(i) for i in imgs], params=[p0, p0])
    ref_img = torch.from_numpy(np.stack([IO.pipeline(i, (64, 96), p0) for i in imgs])).float().to(DEV)
    with torch.no_grad():
        a = bb(nt)['0'].tensors
        b_ = bb(NestedTensor(ref_img, torch.zeros(2, 64, 96, dtype=torch.bool, device=DEV), True))['0'].tensors
    assert a.shape == b_.shape and rel(a, b_) < 3e-2


@pytest.mark.parametrize('K1,K2,N,s2,OH,OW,Bn', [(64, 64, 256, 1, 24, 32, 3), (64, 64, 256, 1, 7, 9, 5), (128, 256, 512, 2, 15, 20, 3),
                                                 (128, 256, 512, 2, 8, 6, 2)])
def test_fused_block_tail_conv3_plus_downsample(K1, K2, N, s2, OH, OW, Bn):
    """gpv_conv1x1_dual (conv1x1_dual.hip): ReLU(conv3(a2) + downsample(x at the block's stride) + shifts) in one launch against
    fp32 torch and against the two-launch path it replaces (downsample conv, then conv3 with the residual epilogue: one more bf16
    rounding, of the identity branch)"""
    h, dtype = hip(), torch.bfloat16
    IH, IW = OH * s2 - (s2 - 1) * (OH % 2), OW * s2                   # an odd input height for the strided case
    IH = max(IH, (OH - 1) * s2 + 1)
    a2 = rnd(Bn, OH, OW, K1, dtype=dtype, seed=70)
    x = rnd(Bn, IH, IW, K2, dtype=dtype, seed=71)
    w3 = rnd(N, K1, dtype=dtype, seed=72, scale=1.0 / math.sqrt(K1))
    wd = rnd(N, K2, dtype=dtype, seed=73, scale=1.0 / math.sqrt(K2))
    b3, bd = rnd(N, seed=74), rnd(N, seed=75)
    y = torch.full((Bn, OH, OW, N), float('nan'), device=DEV, dtype=dtype)
    assert h.conv1x1_dual(a2, w3, x, wd, (b3 + bd).contiguous(), y, Bn, OH, OW, K1, IH, IW, K2, s2, N, h.ACT_RELU)
    xs = x[:, ::s2, ::s2][:, :OH, :OW].float()
    ref = F.relu(a2.float() @ w3.float().t() + xs @ wd.float().t() + b3 + bd)
    assert rel(y, ref) < TOL[dtype]
    idt = torch.empty(Bn, OH, OW, N, device=DEV, dtype=dtype)
    h.conv2d(0, x, wd.view(N, 1, K2), idt, Bn, IH, IW, K2, K2, OH, OW, N, 1, 1, s2, s2, 0, 0, bias=bd)
    y2 = torch.empty_like(y)
    h.conv2d(0, a2, w3.view(N, 1, K1), y2, Bn, OH, OW, K1, K1, OH, OW, N, 1, 1, 1, 1, 0, 0, bias=b3, res=idt, act=h.ACT_RELU)
    assert rel(y, y2) < 1.2e-2
    assert not h.conv1x1_dual(a2[..., :32].contiguous(), w3[:, :32].contiguous(), x, wd, b3, y, Bn, OH, OW, 32, IH, IW, K2, s2, N, h.ACT_RELU)
    # round 6: the same launch also writing the one-bit ReLU mask of its output (gpv_conv1x1_dual_bits; layer2's shape only)
    bits = torch.full((Bn * OH * OW + 2, N // 32), 0x5a5a5a5a, device=DEV, dtype=torch.int32)
    y3 = torch.full_like(y, float('nan'))
    took = h.conv1x1_dual(a2, w3, x, wd, (b3 + bd).contiguous(), y3, Bn, OH, OW, K1, IH, IW, K2, s2, N, h.ACT_RELU, y_mask_bits=bits[:Bn * OH * OW])
    assert took == ((K1, K2, N) == (128, 256, 512))
    if took:
        torch.cuda.synchronize()
        assert torch.equal(y3, y)
        assert torch.equal(bits[:Bn * OH * OW], _pack_bits(y3.view(-1, N))) and (bits[Bn * OH * OW:] == 0x5a5a5a5a).all()
    else:
        assert (bits == 0x5a5a5a5a).all() and torch.isnan(y3.float()).all()               # refused before anything was launched


@pytest.mark.parametrize('branch', ['identity', 'downsample', 'plain'])
@pytest.mark.parametrize('N2,OH,OW,Bn', [(64, 24, 32, 3), (128, 7, 9, 5), (64, 120, 160, 2)])
def test_conv1x1_chain_equals_the_two_launches(branch, N2, OH, OW, Bn):
    """gpv_conv1x1_chain (conv1x1_chain.hip): a layer1 bottleneck tail -- conv3 + identity | conv3 + downsample | conv3 alone, ReLU --
    and the next bottleneck's conv1 + ReLU in one launch: y BIT-IDENTICAL to the launch it replaces (gpv_conv2d with the residual
    epilogue / gpv_conv1x1_dual), z bit-identical to gpv_conv2d on that y (same products, same fp32 order; streaming kernel forced so
    that small maps take the same path as the bench shapes), both within bf16 rounding of fp32 torch; unsupported shapes decline"""
    h, dtype = hip(), torch.bfloat16
    K1, N = 64, 256
    a = rnd(Bn, OH, OW, K1, dtype=dtype, seed=170)
    w3 = rnd(N, K1, dtype=dtype, seed=171, scale=1.0 / math.sqrt(K1))
    b3 = rnd(N, seed=172)
    wn = rnd(N2, N, dtype=dtype, seed=173, scale=1.0 / math.sqrt(N))
    bn_ = rnd(N2, seed=174)
    x = rnd(Bn, OH, OW, N if branch == 'identity' else 64, dtype=dtype, seed=175)
    wd = rnd(N, 64, dtype=dtype, seed=176, scale=0.125)
    y = torch.full((Bn, OH, OW, N), float('nan'), device=DEV, dtype=dtype)
    z = torch.full((Bn, OH, OW, N2), float('nan'), device=DEV, dtype=dtype)
    prev = h.set_option(h.OPT_C1S, 2)
    try:
        y_ref = torch.empty_like(y)
        if branch == 'identity':
            assert h.conv1x1_chain(a, w3, None, None, 1, x, b3, y, wn, bn_, z, Bn, OH, OW)
            h.conv2d(0, a, w3.view(N, 1, K1), y_ref, Bn, OH, OW, K1, K1, OH, OW, N, 1, 1, 1, 1, 0, 0, bias=b3, res=x, act=h.ACT_RELU)
            ref = F.relu(a.float() @ w3.float().t() + b3 + x.float())
        elif branch == 'downsample':
            assert h.conv1x1_chain(a, w3, x, wd, 1, None, b3, y, wn, bn_, z, Bn, OH, OW)
            assert h.conv1x1_dual(a, w3, x, wd, b3, y_ref, Bn, OH, OW, K1, OH, OW, 64, 1, N, h.ACT_RELU)
            ref = F.relu(a.float() @ w3.float().t() + x.float() @ wd.float().t() + b3)
        else:
            assert h.conv1x1_chain(a, w3, None, None, 1, None, b3, y, wn, bn_, z, Bn, OH, OW)
            h.conv2d(0, a, w3.view(N, 1, K1), y_ref, Bn, OH, OW, K1, K1, OH, OW, N, 1, 1, 1, 1, 0, 0, bias=b3, act=h.ACT_RELU)
            ref = F.relu(a.float() @ w3.float().t() + b3)
        z_ref = torch.empty_like(z)
        h.conv2d(0, y_ref, wn.view(N2, 1, N), z_ref, Bn, OH, OW, N, N, OH, OW, N2, 1, 1, 1, 1, 0, 0, bias=bn_, act=h.ACT_RELU)
    finally:
        h.set_option(h.OPT_C1S, prev)
    assert torch.equal(y, y_ref) and torch.equal(z, z_ref)
    assert rel(y, ref) < TOL[dtype]
    assert rel(z, F.relu(y.float() @ wn.float().t() + bn_)) < TOL[dtype]
    a32 = a[..., :32].contiguous()
    assert not h.conv1x1_chain(a32, w3[:, :32].contiguous(), None, None, 1, None, b3, y, wn, bn_, z, Bn, OH, OW)
    # round 6: the identity-branch launch with a 128-channel conv1 also writes the one-bit ReLU mask of z (gpv_conv1x1_chain_bits)
    px = Bn * OH * OW
    bits = torch.full((px + 2, N2 // 32), 0x5a5a5a5a, device=DEV, dtype=torch.int32)
    y3, z3 = torch.full_like(y, float('nan')), torch.full_like(z, float('nan'))
    xa = x if branch == 'identity' else None
    a2_, w2_ = (x, wd) if branch == 'downsample' else (None, None)
    took = h.conv1x1_chain(a, w3, a2_, w2_, 1, xa, b3, y3, wn, bn_, z3, Bn, OH, OW, z_mask_bits=bits[:px])
    assert took == (branch == 'identity' and N2 == 128)
    torch.cuda.synchronize()
    if took:
        assert torch.equal(y3, y) and torch.equal(z3, z)
        assert torch.equal(bits[:px], _pack_bits(z3.view(-1, N2))) and (bits[px:] == 0x5a5a5a5a).all()
    else:
        assert (bits == 0x5a5a5a5a).all()


@pytest.mark.parametrize('Bn,OH,OW', [(2, 16, 24), (1, 9, 33), (3, 30, 40)])
def test_stride2_3x3_backward_data_reads_one_bit_relu_masks(Bn, OH, OW):
    """Round 6: layer2.0's conv2 backward-data (3x3 stride 2 over 128 channels, conv3x3_stream.hip c3d2_kernel) with its ReLU mask as one
    bit per element (the bits gpv_conv1x1_chain_bits wrote with the activation) against the same launch with the bf16 activation: bit-identical."""
    h, dtype = hip(), torch.bfloat16
    C = 128
    IH, IW = 2 * OH, 2 * OW                                    # dx extent; dy is OH x OW
    dy = rnd(Bn, OH, OW, C, dtype=dtype, seed=320)
    wd = rnd(C, 3, 3, C, dtype=dtype, seed=321, scale=1.0 / math.sqrt(9 * C))
    act = torch.relu(rnd(Bn, IH, IW, C, dtype=dtype, seed=322))
    act[:, ::3, ::5, ::7] = -0.0
    bits = _pack_bits(act.view(-1, C))
    prev = h.set_option(h.OPT_C3S, 2)
    try:
        d0 = torch.empty(Bn, IH, IW, C, device=DEV, dtype=dtype)
        h.set_option(h.OPT_C3S_LAUNCHES, 0)
        h.conv2d(1, dy, wd, d0, Bn, OH, OW, C, C, IH, IW, C, 3, 3, 2, 2, 1, 1, relu_mask=act)
        d1 = torch.full_like(d0, float('nan'))
        args = (1, dy, wd, d1, Bn, OH, OW, C, C, IH, IW, C, 3, 3, 2, 2, 1, 1)
        assert h.conv2d_mask_bits_ok(*args, relu_mask_bits=bits)
        h.conv2d(*args, relu_mask_bits=bits)
        torch.cuda.synchronize()
        assert h.set_option(h.OPT_C3S_LAUNCHES, 0) == 2
        assert torch.equal(d0, d1)
        assert ((d1 != 0) & ~(act.float() > 0)).sum() == 0 and 0.2 < (d1 != 0).float().mean() < 0.8
        # a stride-1 3x3 with bits is refused before anything is launched
        assert not h.conv2d_mask_bits_ok(1, dy, wd, d1[:, :OH, :OW].contiguous(), Bn, OH, OW, C, C, OH, OW, C, 3, 3, 1, 1, 1, 1, relu_mask_bits=bits[:Bn * OH * OW])
    finally:
        h.set_option(h.OPT_C3S, prev)


def test_clip_scale_is_deterministic_and_matches_torch():
    """gpv_clip_scale: min(1, max_norm / (||g|| + 1e-6)) over a flat fp32 range (train_distr.py:423-425), fixed summation order:
    bit-identical over repeated launches (what keeps data-parallel replicas identical), equal to torch's clip factor to fp32
    rounding; the per-parameter step counts advance by the liveness flags in the same launch; g = None leaves gscale alone."""
    h = hip()
    torch.manual_seed(5)
    for n, mx in ((4 * 1000 + 4, 0.1), (111_000_000 // 4 * 4, 0.1), (4096, 1e9)):
        g = torch.randn(n, device=DEV) * 1e-3
        ws = torch.zeros(h.CLIP_PARTIALS, device=DEV)
        sc = torch.full((1,), -1.0, device=DEV)
        pstep = torch.arange(700, device=DEV, dtype=torch.int32)
        live = (torch.arange(700, device=DEV) % 3 == 0).to(torch.int32)
        h.clip_scale(g, mx, ws, sc, pstep, live)
        first = sc.clone()
        want = torch.clamp(mx / (torch.linalg.vector_norm(g.double()) + 1e-6), max=1.0).float()
        assert abs(float(first) - float(want)) <= 2e-6 * float(want), (n, float(first), float(want))
        assert torch.equal(pstep, torch.arange(700, device=DEV, dtype=torch.int32) + live)
        for _ in range(5):
            h.clip_scale(g, mx, ws, sc)
            assert torch.equal(sc, first)
        h.clip_scale(None, 0.0, ws, sc, pstep, live)
        assert torch.equal(sc, first) and torch.equal(pstep, torch.arange(700, device=DEV, dtype=torch.int32) + 2 * live)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,cols,prow', [(600, 256, 600), (600, 256, 100), (96, 768, 96), (96, 768, 12), (40, 2048, 8)])
def test_layernorm_second_output_and_second_gradient(dtype, rows, cols, prow):
    """gpv_layernorm_pos_fwd: y2 = y + pos[row % pos_rows] bit-identical to gpv_add on the stored y (every row width / broadcast
    period); gpv_layernorm_bwd2: (dy, dy2) gives what one launch on dy + dy2 (summed in fp32) gives, dgamma / dbeta included."""
    h = hip()
    x, s_ = rnd(rows, cols, dtype=dtype, seed=1), rnd(rows, cols, dtype=dtype, seed=2)
    pos = rnd(prow, cols, dtype=dtype, seed=3)
    gamma, beta = (1 + 0.1 * torch.randn(cols, device=DEV)), 0.1 * torch.randn(cols, device=DEV)
    y, y2 = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    h.layernorm_fwd(x, s_, gamma, beta, y, mean, rstd, rows, cols, 1e-5, pos=pos, y2=y2)
    y0 = torch.empty_like(x)
    h.layernorm_fwd(x, s_, gamma, beta, y0, mean, rstd, rows, cols, 1e-5)
    assert torch.equal(y, y0)
    want = (y.float().reshape(rows // prow, prow, cols) + pos.float()[None]).to(dtype).reshape(rows, cols)
    assert torch.equal(y2, want)
    dy, dy2 = rnd(rows, cols, dtype=dtype, seed=4), rnd(rows, cols, dtype=dtype, seed=5)
    dx, dg, db = torch.empty_like(x), torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    h.layernorm_bwd(dy, x, s_, gamma, mean, rstd, dx, None, dg, db, rows, cols, dy2=dy2)
    # reference: fp32 math on the fp32 sum
    z = x.float() + s_.float()
    zh = (z - mean[:, None]) * rstd[:, None]
    gsum = dy.float() + dy2.float()
    gy = gsum * gamma
    ref = rstd[:, None] * (gy - gy.mean(-1, keepdim=True) - zh * (gy * zh).mean(-1, keepdim=True))
    assert rel(dx, ref) < TOL[dtype]
    assert rel(dg, (gsum * zh).sum(0)) < 2e-3 and rel(db, gsum.sum(0)) < 2e-3


# ----------------------------------------------------------------------------------------- fused feed-forward sub-layer
def _bf16_ulps(a, b):
    """distance in bf16 code points between two bf16 tensors (same sign assumed where it matters: zeros map to 0)"""
    ia = a.view(torch.int16).to(torch.int32)
    ib = b.view(torch.int16).to(torch.int32)
    ia = torch.where(ia < 0, -(ia & 0x7fff), ia)
    ib = torch.where(ib < 0, -(ib & 0x7fff), ib)
    return (ia - ib).abs()


@pytest.mark.skipif(not __import__('gpv1_amd.hip', fromlist=['x']).TUNING, reason='gpv_ffn_fused_fwd lives in the tuning build only (GPV_TUNING_LIB=1)')
@pytest.mark.parametrize('M,Fh,prow,drop', [(9600, 2048, 0, 0.0), (9600, 2048, 300, 0.1), (3200, 2048, 100, 0.1), (200, 2048, 100, 0.1),
                                            (77, 128, 0, 0.0), (64, 64, 64, 0.25), (1, 2048, 1, 0.1)])
def test_ffn_fused_fwd_equals_the_three_launches(M, Fh, prow, drop):
    """gpv_ffn_fused_fwd (transformer.py:156-160, 226-231 as one launch) against gpv_gemm (ReLU + dropout) -> gpv_gemm ->
    gpv_layernorm_pos_fwd with the same seeds.  h: the same products in the same order, the same mask: identical up to the rare 1-ulp
    rounding flip of a different MFMA schedule; y: the F-sum is split in two halves: <= 2 bf16 ulp of a K = F sum; out: LayerNorm of
    values that differ by those ulps.  And against fp32 math on the same inputs without dropout."""
    h = hip()
    D = 256
    x = rnd(M, D, dtype=torch.bfloat16, seed=1)
    w1, b1 = rnd(Fh, D, dtype=torch.bfloat16, seed=2, scale=0.06), 0.1 * rnd(Fh, seed=3)
    w2, b2 = rnd(D, Fh, dtype=torch.bfloat16, seed=4, scale=0.03), 0.1 * rnd(D, seed=5)
    gamma, beta = 1.0 + 0.1 * rnd(D, seed=6), 0.1 * rnd(D, seed=7)
    pos = rnd(prow, D, dtype=torch.bfloat16, seed=8) if prow else None
    seed1, seed2 = 0x1234567, 0x89abcde
    bf = torch.bfloat16

    def bufs():
        return (torch.empty(M, Fh, device=DEV, dtype=bf), torch.empty(M, D, device=DEV, dtype=bf), torch.empty(M, D, device=DEV, dtype=bf),
                torch.empty(M, device=DEV), torch.empty(M, device=DEV), torch.empty(M, D, device=DEV, dtype=bf) if prow else None)
    hh, y, out, mean, rstd, out2 = bufs()
    assert h.ffn_fused_fwd(x, w1, b1, w2, b2, gamma, beta, hh, y, out, mean, rstd, M, D, Fh, 1e-5, drop, seed1, seed2, pos=pos, out2=out2)
    h0, y0, o0, m0, r0, o20 = bufs()
    h.gemm(x, w1, h0, M, Fh, D, D, D, Fh, bias=b1, act=h.ACT_RELU, drop_p=drop, seed=seed1)
    h.gemm(h0, w2, y0, M, D, Fh, Fh, Fh, D, bias=b2)
    h.layernorm_fwd(x, y0, gamma, beta, o0, m0, r0, M, D, 1e-5, drop, seed2, pos=pos, y2=o20)
    torch.cuda.synchronize()
    # hidden activation: same zeros (ReLU and the dropout mask), same values
    assert torch.equal(hh == 0, h0 == 0) or ((hh == 0) != (h0 == 0)).float().mean().item() < 1e-5
    du = _bf16_ulps(hh, h0)
    assert du.max().item() <= 1 and (du > 0).float().mean().item() < 2e-3, (du.max().item(), (du > 0).float().mean().item())
    if drop > 0:
        kept = (hh != 0).float().sum() / (F.relu(x.float() @ w1.float().t() + b1) > 0).float().sum()
        assert abs(kept.item() - (1 - drop)) < 0.02
    # second product from the kernel's own h in fp32: one bf16 rounding
    yref = hh.float() @ w2.float().t() + b2
    assert rel(y, yref) < 6e-3
    assert (y.float() - y0.float()).abs().max().item() <= 2 * 2.0 ** -8 * yref.abs().max().item()
    assert (out.float() - o0.float()).abs().max().item() <= 0.05 and (out.float() - o0.float()).abs().mean().item() < 2e-3
    assert rel(mean, m0) < 2e-2 and rel(rstd, r0) < 2e-2
    if prow:
        want = (out.float().reshape(M // prow, prow, D) + pos.float()[None]).to(bf).reshape(M, D)
        assert torch.equal(out2, want)
    # LayerNorm from the kernel's own y: the epilogue alone, against fp32 math (dropout mask taken from the unfused LayerNorm's rule)
    if drop == 0:
        z = x.float() + y.float()
        ref = F.layer_norm(z, (D,), gamma, beta, 1e-5)
        assert rel(out, ref) < TOL[bf]
        assert rel(mean, z.mean(-1)) < 1e-3 and rel(rstd, (z.var(-1, unbiased=False) + 1e-5).rsqrt()) < 1e-3
        full = F.layer_norm(x.float() + F.relu(x.float() @ w1.float().t() + b1).to(bf).float() @ w2.float().t() + b2, (D,), gamma, beta, 1e-5)
        assert rel(out, full) < 2 * TOL[bf]


@pytest.mark.skipif(not __import__('gpv1_amd.hip', fromlist=['x']).TUNING, reason='gpv_ffn_fused_fwd lives in the tuning build only (GPV_TUNING_LIB=1)')
def test_ffn_fused_fwd_refuses_what_it_does_not_take():
    h = hip()
    bf = torch.bfloat16
    x = rnd(64, 512, dtype=bf)
    w1, w2 = rnd(128, 512, dtype=bf), rnd(512, 128, dtype=bf)
    z = torch.zeros(512, device=DEV)
    args = (torch.empty(64, 128, device=DEV, dtype=bf), torch.empty(64, 512, device=DEV, dtype=bf), torch.empty(64, 512, device=DEV, dtype=bf),
            torch.empty(64, device=DEV), torch.empty(64, device=DEV))
    assert h.ffn_fused_fwd(x, w1, torch.zeros(128, device=DEV), w2, z, z, z, *args, 64, 512, 128, 1e-5) is False       # width 512
    x = rnd(64, 256, dtype=bf)
    w1, w2 = rnd(96, 256, dtype=bf), rnd(256, 96, dtype=bf)
    z = torch.zeros(256, device=DEV)
    args = (torch.empty(64, 96, device=DEV, dtype=bf), torch.empty(64, 256, device=DEV, dtype=bf), torch.empty(64, 256, device=DEV, dtype=bf),
            torch.empty(64, device=DEV), torch.empty(64, device=DEV))
    assert h.ffn_fused_fwd(x, w1, torch.zeros(96, device=DEV), w2, z, z, z, *args, 64, 256, 96, 1e-5) is False        # F % 64
    xf = rnd(64, 256)
    assert h.ffn_fused_fwd(xf, w1, torch.zeros(96, device=DEV), w2, z, z, z, *args, 64, 256, 96, 1e-5) is False       # fp32 operands


@pytest.mark.skipif(not __import__('gpv1_amd.hip', fromlist=['x']).TUNING, reason='gpv_ffn_fused_fwd lives in the tuning build only (GPV_TUNING_LIB=1)')
def test_ffn_block_with_the_fused_forward_matches_the_three_launch_node(monkeypatch):
    """ops.FFNBlockFn with GPV_FFN_FUSED on (opt-in): the forward's saved tensors (h, y, mean, rstd) feed the unchanged backward --
    outputs and every gradient within bf16 rounding of the default node's, same seeds, dropout on."""
    import gpv1_amd.ops as ops
    from gpv1_amd.transformer import LinearP, LayerNormP, ffn_block
    hip()
    ops.RT.set_precise(False)
    torch.manual_seed(0)
    M, D, Fh = 3200, 256, 2048
    l1, l2, norm = LinearP(D, Fh).to(DEV), LinearP(Fh, D).to(DEV), LayerNormP(D).to(DEV)
    pos = rnd(100, D, dtype=torch.bfloat16, seed=3)
    x0 = rnd(M, D, dtype=torch.bfloat16, seed=1)
    dy, dy2 = rnd(M, D, dtype=torch.bfloat16, seed=2), rnd(M, D, dtype=torch.bfloat16, seed=4)
    res = {}
    for fused in (False, True):
        monkeypatch.setattr(ops, 'FFN_FUSED', fused)
        ops.RT.manual_seed(77)
        for q in list(l1.parameters()) + list(l2.parameters()) + list(norm.parameters()):
            q.grad = None
        x = x0.clone().requires_grad_(True)
        out, out2 = ffn_block(x, l1, l2, norm, 0.1, pos=pos)
        torch.autograd.backward([out, out2], [dy, dy2])
        torch.cuda.synchronize()
        res[fused] = [out.detach().float(), out2.detach().float(), x.grad.float()] + \
            [q.grad.float().clone() for q in list(l1.parameters()) + list(l2.parameters()) + list(norm.parameters())]
    for a, b in zip(res[True], res[False]):
        assert rel(a, b) < 2e-2, rel(a, b)
        assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item() > 0.9995


def _pack_bits(y):
    """[px, C] -> int32 [px, C / 32] holding gpv_conv_args.y_mask_bits' bytes: (y[px, c] > 0) = bit (c & 7) of byte
    32 (c / 256) + 8 ((c % 32) / 8) + (c % 256) / 32 of the pixel's row (include/gpv_hip.h)"""
    px, Cc = y.shape
    G = min(256, Cc)                                                              # 128 channels: one group of 16 bytes, byte 4 g + t
    b = (y.float() > 0).view(px, Cc // G, G // 32, 4, 8).to(torch.int32)          # [px, group, t = (c % G) / 32, g = (c % 32) / 8, e = c & 7]
    byte = (b << torch.arange(8, device=y.device, dtype=torch.int32)).sum(-1)     # [px, group, t, g]
    byte = byte.permute(0, 1, 3, 2).contiguous().to(torch.uint8)                  # byte order inside a group: 8 g + t
    return byte.view(px, Cc // 8).view(torch.int32)


@pytest.mark.parametrize('K,N,px', [(128, 512, 4813), (256, 1024, 2400), (256, 512, 1000), (128, 512, 16), (256, 1024, 33)])
def test_streaming_1x1_writes_one_bit_relu_masks(K, N, px):
    """Round 6: conv3 + identity + ReLU of a bottleneck (backbone.py:93-95 -> torchvision Bottleneck.forward) on the streaming kernel ALSO
    writes (output > 0) as one bit per element (gpv_conv_args.y_mask_bits): the output is the launch's without the bits, bit for bit, and the
    words are the packed signs of the stored bf16 values -- ragged pixel counts included (rows beyond the map are never written)."""
    h = hip()
    B, H, W = 1, 1, px
    x, w = rnd(px, K, dtype=torch.bfloat16, seed=300), rnd(N, K, dtype=torch.bfloat16, seed=301, scale=0.08)
    res, bias = rnd(px, N, dtype=torch.bfloat16, seed=302), rnd(N, seed=303)
    prev = h.set_option(h.OPT_C1S, 2)
    try:
        y0 = torch.empty(px, N, device=DEV, dtype=torch.bfloat16)
        h.conv2d(0, x, w, y0, B, H, W, K, K, H, W, N, 1, 1, 1, 1, 0, 0, bias=bias, res=res, act=h.ACT_RELU)
        y1 = torch.empty_like(y0)
        bits = torch.full((px + 3, N // 32), 0x5a5a5a5a, device=DEV, dtype=torch.int32)          # three guard rows behind the map
        args = (0, x, w, y1, B, H, W, K, K, H, W, N, 1, 1, 1, 1, 0, 0)
        kw = dict(bias=bias, res=res, act=h.ACT_RELU, y_mask_bits=bits[:px])
        assert h.conv2d_mask_bits_ok(*args, **kw)
        h.set_option(h.OPT_C1S_LAUNCHES, 0)
        h.conv2d(*args, **kw)
        torch.cuda.synchronize()
        assert h.set_option(h.OPT_C1S_LAUNCHES, 0) == 1
        assert torch.equal(y0, y1)
        assert torch.equal(bits[:px], _pack_bits(y1)) and (bits[px:] == 0x5a5a5a5a).all()
        assert 0.2 < (y1 > 0).float().mean() < 0.8
        # what the library refuses, it refuses before launching anything: no ReLU, a 3x3, the generic kernels
        assert not h.conv2d_mask_bits_ok(*args, bias=bias, res=res, act=h.ACT_NONE, y_mask_bits=bits[:px])
        with pytest.raises(RuntimeError):
            h.conv2d(*args, bias=bias, res=res, act=h.ACT_NONE, y_mask_bits=bits[:px])
        h.set_option(h.OPT_C1S, 0)
        assert not h.conv2d_mask_bits_ok(*args, **kw)
    finally:
        h.set_option(h.OPT_C1S, prev)


@pytest.mark.parametrize('K,N,px,with_res', [(128, 512, 4813, True), (256, 1024, 2400, True), (256, 512, 777, True), (512, 1024, 1200, True), (512, 1024, 50, True)])
def test_streaming_1x1_backward_data_reads_one_bit_relu_masks(K, N, px, with_res):
    """Round 6: conv1's backward-data launch (dx = (dy W + identity gradient) * (x > 0)) with the mask as one bit per element
    (gpv_conv_args.relu_mask_bits) against the same launch with the bf16 activation as the mask: bit-identical."""
    h = hip()
    B, H, W = 1, 1, px
    dy, wd = rnd(px, K, dtype=torch.bfloat16, seed=310), rnd(N, K, dtype=torch.bfloat16, seed=311, scale=0.08)
    res = rnd(px, N, dtype=torch.bfloat16, seed=312) if with_res else None
    act = torch.relu(rnd(px, N, dtype=torch.bfloat16, seed=313))                        # the saved post-ReLU activation
    act[::7, ::5] = -0.0                                                                # (a stored -0 is not > 0)
    bits = _pack_bits(act)
    prev = h.set_option(h.OPT_C1S, 2)
    try:
        d0 = torch.empty(px, N, device=DEV, dtype=torch.bfloat16)
        args0 = (1, dy, wd, d0, B, H, W, K, K, H, W, N, 1, 1, 1, 1, 0, 0)
        h.conv2d(*args0, res=res, relu_mask=act)
        d1 = torch.full_like(d0, float('nan'))
        args1 = (1, dy, wd, d1, B, H, W, K, K, H, W, N, 1, 1, 1, 1, 0, 0)
        assert h.conv2d_mask_bits_ok(*args1, res=res, relu_mask_bits=bits)
        h.set_option(h.OPT_C1S_LAUNCHES, 0)
        h.conv2d(*args1, res=res, relu_mask_bits=bits)
        torch.cuda.synchronize()
        assert h.set_option(h.OPT_C1S_LAUNCHES, 0) == 1
        assert torch.equal(d0, d1)
        assert torch.equal(d1 == 0, ~(act.float() > 0) | (d1 == 0)) and ((d1 != 0) & ~(act.float() > 0)).sum() == 0
        # against fp32 math
        ref = dy.float() @ wd.float().t() + (res.float() if res is not None else 0)
        ref = torch.where(act.float() > 0, ref, torch.zeros_like(ref))
        assert rel(d1, ref) < TOL[torch.bfloat16]
    finally:
        h.set_option(h.OPT_C1S, prev)


def test_conv_rowscale_is_per_output_pixel_and_short_vectors_are_refused():
    """include/gpv_hip.h gpv_conv_args: in modes 0 / 1 `rowscale` is one factor per output PIXEL (the GEMM epilogue's rowscale[m]), not the
    per-channel BatchNorm scale (that one is folded into the weight copy).  Round 6: tools/tune_gemms_bs1.py passed a [Cout] vector on this
    shape (one 120 x 160 image, 64 -> 64 pointwise: the generic tile kernel, every operand an exact-size allocation) and the launch read
    19200 floats from a 64-float tensor -- a memory access fault.  The mirror refuses that before launching; the correctly sized call
    gives y = relu(rowscale[pixel] * (x w^T) + bias[channel])."""
    h = hip()
    B, H, W, Cin, Cout = 1, 120, 160, 64, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B * H * W, Cin, generator=g).to(DEV).bfloat16()
    w = (torch.randn(Cout, Cin, generator=g) / 8).to(DEV).bfloat16()
    bias = torch.randn(Cout, generator=g).to(DEV)
    y = torch.full((B * H * W, Cout), float('nan'), device=DEV, dtype=torch.bfloat16)
    args = (0, x, w, y, B, H, W, Cin, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0)
    with pytest.raises(ValueError, match='rowscale has 64 elements'):
        h.conv2d(*args, rowscale=torch.ones(Cout, device=DEV), bias=bias, act=h.ACT_RELU)
    with pytest.raises(ValueError, match='bias has 32 elements'):
        h.conv2d(*args, bias=bias[:32].contiguous(), act=h.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.isnan(y.float()).all()                      # nothing was launched
    rs = (0.5 + torch.rand(B * H * W, generator=g)).to(DEV)
    h.conv2d(*args, rowscale=rs, bias=bias, act=h.ACT_RELU)
    torch.cuda.synchronize()
    ref = torch.relu(rs[:, None] * (x.float() @ w.float().t()) + bias[None, :])
    assert rel(y, ref) < TOL[torch.bfloat16]
    # and the call the ResNet path makes on this shape (no rowscale), same exact-size operands
    y2 = torch.full_like(y, float('nan'))
    h.conv2d(0, x, w, y2, B, H, W, Cin, Cin, H, W, Cout, 1, 1, 1, 1, 0, 0, bias=bias, act=h.ACT_RELU)
    torch.cuda.synchronize()
    assert rel(y2, torch.relu(x.float() @ w.float().t() + bias[None, :])) < TOL[torch.bfloat16]


# ------------------------------------------------------------- dropout masks against the host restatement of the specification
def _epoch():
    """the device seed epoch, if a trainer of this process installed one (gpv_set_seed_device is process-wide): csrc/common.h eff_seed"""
    import gpv1_amd.ops as ops
    return None if ops.RT.seed_dev is None else int(ops.RT.seed_dev.item())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('p,n', [(0.1, 4096 * 33 + 1), (0.5, 70001)])
def test_dropout_kernel_mask_equals_the_specification(dtype, p, n):
    """gpv_dropout keeps exactly the elements tests/dropout_ref.keep_flat says (drop_pair_bits, one 32-bit word per pair of elements),
    bit for bit, and scales them by 1 / (1 - p).  Until round 6 no mask in this suite was compared with anything but another kernel."""
    from tests import dropout_ref as R
    h = hip()
    seed = 0x5EED0000BEEF
    x = torch.ones(n, device=DEV, dtype=dtype)
    y = torch.full_like(x, float('nan'))
    h.dropout(x, y, n, p, seed)
    torch.cuda.synchronize()
    ref = R.keep_flat(R.eff_seed(seed, _epoch()), np.arange(n, dtype=np.uint64), p)
    got = (y != 0).cpu().numpy()
    assert np.array_equal(got, ref), int((got != ref).sum())
    scale = 1.0 / (1.0 - float(np.float32(p)))
    assert rel(y[torch.from_numpy(ref).to(DEV)], torch.full((int(ref.sum()),), scale, device=DEV)) < (1e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize('Bn,H,dh,Sq,Sk', [(16, 8, 32, 300, 300), (8, 16, 48, 100, 16), (2, 8, 32, 100, 300)])
def test_attention_dropout_mask_equals_the_specification(Bn, H, dh, Sq, Sk):
    """the (batch, head, query, key) keep pattern of the attention kernels -- packed saturating subtract + shift on pairs of probabilities
    (attention.hip attn_drop_bits: the instruction family hipcc 7.2 miscompiled elsewhere this round) -- equals
    tests/dropout_ref.keep_attention bit for bit.  The pattern is read off the forward kernel with one-hot V probes; the backward
    kernels are tied to the forward's by the dropout gradient tests."""
    from tests import dropout_ref as R
    h = hip()
    drop, seed = 0.1, 4242
    keep = _attention_keep_mask(h, Bn, H, dh, Sq, Sk, drop, seed).cpu().numpy()
    ref = R.keep_attention(R.eff_seed(seed, _epoch()), Bn, H, Sq, Sk, drop)
    assert np.array_equal(keep, ref), (int((keep != ref).sum()), keep.mean(), ref.mean())


@pytest.mark.parametrize('M,N,K,act', [(9600, 2048, 256, 1), (3200, 256, 2048, 0), (300, 256, 2048, 0), (640, 768, 768, 0), (192, 3072, 768, 2),
                                       (9600, 256, 256, 0), (100, 768, 3072, 0), (4, 768, 768, 0)])
def test_gemm_epilogue_dropout_mask_equals_the_specification(M, N, K, act):
    """every GEMM family's dropout epilogue (streaming 1x1 kernel's packed compare for the 256 -> 2048 feed-forward expansion, pipe, small-M,
    direct-to-LDS, generic, matrix-vector) keeps exactly keep_flat(seed, m * N + n): the same GEMM without dropout tells which outputs are
    non-zero before the mask, there (output != 0) must equal the specification and the value must be the undropped one / (1 - p)."""
    from tests import dropout_ref as R
    h = hip()
    p, seed = 0.1, 987654321
    A = rnd(M, K, dtype=torch.bfloat16, seed=300)
    B = rnd(N, K, dtype=torch.bfloat16, seed=301, scale=K ** -0.5)
    bias = rnd(N, seed=302)
    y0 = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    y1 = torch.full_like(y0, float('nan'))
    h.gemm(A, B, y0, M, N, K, K, K, N, bias=bias, act=act)
    h.gemm(A, B, y1, M, N, K, K, K, N, bias=bias, act=act, drop_p=p, seed=seed)
    torch.cuda.synchronize()
    ref = torch.from_numpy(R.keep_flat(R.eff_seed(seed, _epoch()), np.arange(M * N, dtype=np.uint64), p).reshape(M, N)).to(DEV)
    live = y0 != 0
    assert live.float().mean() > 0.3
    assert torch.equal((y1 != 0) & live, ref & live), int((((y1 != 0) & live) != (ref & live)).sum())
    assert ((y1 != 0) & ~live).sum() == 0
    sel = ref & live
    assert rel(y1[sel], y0[sel].float() / (1.0 - float(np.float32(p)))) < 1.2e-2
