"""TEST-ONLY emulation of the libgpv_hip.so entry points with plain torch on CPU.

Purpose: let the `-m "not gpu"` suite exercise the HOST logic of the product package (module tree,
state-dict keys, autograd wiring, layouts/strides passed to the kernels, criterion, trainer,
multi-process gradient exchange over gloo) in a container without a GPU.  It is installed by
monkeypatching ``gpv1_amd.hip`` inside tests only; the product never imports this file and has no
CPU path of its own (gpv1_amd.hip raises without the library / with CPU tensors).
Semantics follow include/gpv_hip.h; dropout is only supported with p = 0.
"""
import math

import torch
import torch.nn.functional as F


def _sv(t, shape, strides, extra_off=0):
    return torch.as_strided(t, shape, strides, t.storage_offset() + extra_off)


def _mat(t, rows, cols, ld, layout, batch=1, bs=0):
    """logical [batch, rows, cols] view of an operand stored KMAJOR ([r*ld + c]) or TRANS ([c*ld + r])"""
    if layout == 0:
        return _sv(t, (batch, rows, cols), (bs, ld, 1))
    return _sv(t, (batch, rows, cols), (bs, 1, ld))


def _act(x, act):
    if act == 1:
        return F.relu(x)
    if act == 2:
        return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    return x


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, layoutA=0, layoutB=0, batch=1, sA=0, sB=0, sC=0, alpha=1.0, rowscale=None,
         bias=None, res=None, ldr=0, sR=0, relu_mask=None, ldm=0, act=0, drop_p=0.0, seed=0, accumulate=False, split_k=1,
         a_rowsum=None, kpad_finite=False):
    assert drop_p == 0.0, 'cpu shim: dropout unsupported'
    a = _mat(A, M, K, lda, layoutA, batch, sA).float()
    if a_rowsum is not None:
        assert layoutA == 1 and layoutB == 1 and batch == 1
        a_rowsum[:M] += a[0].sum(1)
    b = _mat(B, N, K, ldb, layoutB, batch, sB).float()
    y = alpha * (a @ b.transpose(1, 2))
    if rowscale is not None:
        y = y * rowscale[:M].view(1, M, 1)
    if bias is not None:
        y = y + bias[:N]
    if res is not None:
        y = y + _sv(res, (batch, M, N), (sR, ldr, 1)).float()
    y = _act(y, act)
    if relu_mask is not None:
        y = y * (_sv(relu_mask, (1, M, N), (0, ldm, 1)).float() > 0)
    out = _sv(Cm, (batch, M, N), (sC, ldc, 1))
    if accumulate:
        out += y.to(out.dtype)
    else:
        out.copy_(y.to(out.dtype))


def conv2d_mask_bits_ok(*a, **k):
    return False                      # (the one-bit ReLU masks are a GPU-kernel matter: the emulation keeps the bf16 masks)


def conv2d(mode, x, w, y, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, rowscale=None, bias=None, res=None,
           relu_mask=None, act=0, split_k=0, y_mask_bits=None, relu_mask_bits=None):
    assert y_mask_bits is None and relu_mask_bits is None
    if Cs != Cin:
        # stem trick: a tap reads Cin contiguous elements starting at the pixel -> unfold explicitly
        cols = []
        for r in range(KH):
            rows = _sv(x, (B, OH, OW, Cin), (IH * IW * Cs, SH * IW * Cs, SW * Cs, 1), (r - PH) * IW * Cs).float()
            cols.append(rows)
        a = torch.cat(cols, -1).reshape(B * OH * OW, KH * Cin)
        out = a @ w.reshape(Cout, KH * KW * Cin).float().t()
        out = out.view(B, OH, OW, Cout)
        if bias is not None:
            out = out + bias
        y.copy_(_act(out, act).to(y.dtype))
        return
    xg = _sv(x, (B, IH, IW, Cin), (IH * IW * Cs, IW * Cs, Cs, 1)).float()
    if mode == 0:
        wt = w.reshape(Cout, KH, KW, Cin).permute(0, 3, 1, 2).float()
        out = F.conv2d(xg.permute(0, 3, 1, 2), wt, stride=(SH, SW), padding=(PH, PW)).permute(0, 2, 3, 1)
        if rowscale is not None:
            out = out * rowscale
        if bias is not None:
            out = out + bias
        if res is not None:
            out = out + res.float()
        out = _act(out, act)
        if relu_mask is not None:
            out = out * (relu_mask.float() > 0)
        y.copy_(out.to(y.dtype))
    elif mode == 1:
        # gathered tensor = dy [B,IH,IW,Cin(=fwd Cout)], output dx [B,OH,OW,Cout(=fwd Cin)], w = wd [fwdCin][T][fwdCout]
        wt = w.reshape(Cout, KH, KW, Cin).permute(3, 0, 1, 2).float()                # [fwdCout, fwdCin, kh, kw]
        oph = OH - ((IH - 1) * SH - 2 * PH + KH)
        opw = OW - ((IW - 1) * SW - 2 * PW + KW)
        out = F.conv_transpose2d(xg.permute(0, 3, 1, 2), wt, stride=(SH, SW), padding=(PH, PW),
                                 output_padding=(oph, opw)).permute(0, 2, 3, 1)
        if res is not None:
            out = out + res.float()
        if relu_mask is not None:
            out = out * (relu_mask.float() > 0)
        y.copy_(out.to(y.dtype))
    else:
        # wgrad: x gathered fwd input, w = dy [B,OH,OW,Cout], y = dw [Cout][T][Cin] fp32 (+=)
        xi = xg.permute(0, 3, 1, 2).requires_grad_(False)
        dy = w.reshape(B, OH, OW, Cout).permute(0, 3, 1, 2).float()
        wz = torch.zeros(Cout, Cin, KH, KW, requires_grad=True, device=x.device)
        with torch.enable_grad():
            o = F.conv2d(xi, wz, stride=(SH, SW), padding=(PH, PW))
        (gw,) = torch.autograd.grad(o, wz, dy)
        if rowscale is not None:
            gw = gw * rowscale.view(-1, 1, 1, 1)
        y += gw.permute(0, 2, 3, 1).reshape(y.shape)


def _heads(t, bs, rs, B, S, H, dh):
    return _sv(t, (B, H, S, dh), (bs, dh, rs, 1)).float()


def _attn_core(q, k, v, scale, kpm, causal):
    s = q @ k.transpose(-1, -2) * scale
    if kpm is not None:
        s = s.masked_fill(kpm.bool()[:, None, None, :], float('-inf'))
    if causal:
        Sq, Sk = s.shape[-2:]
        s = s.masked_fill(torch.ones(Sq, Sk, dtype=torch.bool, device=s.device).triu(1), float('-inf'))
    return s.softmax(-1) @ v, torch.logsumexp(s, -1)


def attention_fwd(q, k, v, o, strides, B, H, Sq, Sk, dh, scale, kpm=None, causal=False, drop_p=0.0, seed=0, lse=None):
    assert drop_p == 0.0
    (qb, qr), (kb, kr), (vb, vr), (ob, orr) = strides
    out, l = _attn_core(_heads(q, qb, qr, B, Sq, H, dh), _heads(k, kb, kr, B, Sk, H, dh), _heads(v, vb, vr, B, Sk, H, dh),
                        scale, kpm, causal)
    _sv(o, (B, H, Sq, dh), (ob, dh, orr, 1)).copy_(out.to(o.dtype))
    if lse is not None:
        lse.copy_(l)


def attention_qkv_fwd(xp, x, w, bias, q, k, v, o, strides, B, H, S, scale, kpm=None, drop_p=0.0, seed=0, lse=None):
    """gpv_attention_qkv_fwd: the in-projection (rounded to the buffers' dtype, as the launch stores it) + the core"""
    assert drop_p == 0.0
    E = w.shape[1]
    (qb, qr), (kb, kr), (vb, vr), _ = strides
    bb = bias.float() if bias is not None else torch.zeros(3 * E)
    yqk = xp.float() @ w[:2 * E].float().t() + bb[:2 * E]
    yv = x.float() @ w[2 * E:].float().t() + bb[2 * E:]
    _sv(q, (B * S, E), (qr, 1)).copy_(yqk[:, :E].to(q.dtype))
    _sv(k, (B * S, E), (kr, 1)).copy_(yqk[:, E:].to(k.dtype))
    _sv(v, (B * S, E), (vr, 1)).copy_(yv.to(v.dtype))
    attention_fwd(q, k, v, o, strides, B, H, S, S, E // H, scale, kpm=kpm, lse=lse)


def attention_bwd(q, k, v, o, dout, dq, dk, dv, strides, do_strides, B, H, Sq, Sk, dh, scale, kpm=None, causal=False,
                  drop_p=0.0, seed=0, lse=None):
    assert drop_p == 0.0
    (qb, qr), (kb, kr), (vb, vr), _ = strides
    qq = _heads(q, qb, qr, B, Sq, H, dh).requires_grad_(True)
    kk = _heads(k, kb, kr, B, Sk, H, dh).requires_grad_(True)
    vv = _heads(v, vb, vr, B, Sk, H, dh).requires_grad_(True)
    with torch.enable_grad():
        out, _ = _attn_core(qq, kk, vv, scale, kpm, causal)
    g = _heads(dout, do_strides[0], do_strides[1], B, Sq, H, dh)
    gq, gk, gv = torch.autograd.grad(out, (qq, kk, vv), g)
    _sv(dq, (B, H, Sq, dh), (qb, dh, qr, 1)).copy_(gq.to(dq.dtype))
    _sv(dk, (B, H, Sk, dh), (kb, dh, kr, 1)).copy_(gk.to(dk.dtype))
    _sv(dv, (B, H, Sk, dh), (vb, dh, vr, 1)).copy_(gv.to(dv.dtype))


def layernorm_fwd(x, s, gamma, beta, y, mean, rstd, rows, cols, eps, drop_p=0.0, seed=0, pos=None, y2=None):
    assert drop_p == 0.0
    z = x.float().reshape(rows, cols) + (0 if s is None else s.float().reshape(rows, cols))
    mu = z.mean(-1)
    var = ((z - mu[:, None]) ** 2).mean(-1)
    rs = (var + eps).rsqrt()
    n = (z - mu[:, None]) * rs[:, None]
    if gamma is not None:
        n = n * gamma + beta
    y.reshape(rows, cols).copy_(n.to(y.dtype))
    if y2 is not None:
        pr = pos.numel() // cols
        y2.reshape(rows // pr, pr, cols).copy_((y.reshape(rows // pr, pr, cols).float() + pos.reshape(1, pr, cols).float()).to(y2.dtype))
    mean.copy_(mu)
    rstd.copy_(rs)


def linear_layernorm_fwd(a, w, bias, x, gamma, beta, s, y, mean, rstd, rows, eps, drop_p=0.0, seed=0, pos=None, y2=None):
    """gpv_linear_layernorm_fwd: the projection (rounded to the activation dtype, as the launch stores it), then the LayerNorm"""
    sv = a.float() @ w.float().t()
    if bias is not None:
        sv = sv + bias.float()
    s.copy_(sv.to(s.dtype))
    layernorm_fwd(x, s, gamma, beta, y, mean, rstd, rows, x.shape[-1], eps, drop_p, seed, pos=pos, y2=y2)


def layernorm_bwd(dy, x, s, gamma, mean, rstd, dx, ds, dgamma, dbeta, rows, cols, drop_p=0.0, seed=0, dy2=None):
    assert drop_p == 0.0
    z = x.float().reshape(rows, cols) + (0 if s is None else s.float().reshape(rows, cols))
    zh = (z - mean[:, None]) * rstd[:, None]
    g = dy.float().reshape(rows, cols)
    if dy2 is not None:
        g = g + dy2.float().reshape(rows, cols)
    gy = g * (gamma if gamma is not None else 1.0)
    dz = rstd[:, None] * (gy - gy.mean(-1, keepdim=True) - zh * (gy * zh).mean(-1, keepdim=True))
    dx.reshape(rows, cols).copy_(dz.to(dx.dtype))
    if dgamma is not None:
        dgamma += (g * zh).sum(0)
        dbeta += g.sum(0)


def softmax_ce(logits, ld, target, loss, dlogits, gscale, rows, V):
    lg = _sv(logits, (rows, V), (ld, 1)).float()
    valid = (target >= 0) & (target < V)
    t = target.clamp(0, V - 1)
    lp = F.log_softmax(lg, -1)
    loss.copy_(torch.where(valid, -lp.gather(1, t[:, None])[:, 0], torch.zeros(rows, device=lg.device)))
    if dlogits is not None:
        gs = (gscale if gscale is not None else torch.ones(rows, device=lg.device)) * valid
        d = (lp.exp() - F.one_hot(t, V).float()) * gs[:, None]
        _sv(dlogits, (rows, V), (ld, 1)).copy_(d.to(dlogits.dtype))


def image_to_nhwc4(img, out, B, H, W, pad, Hp, Wp):
    out.zero_()
    out[:, pad:pad + H, pad:pad + W, :3] = img.permute(0, 2, 3, 1).to(out.dtype)


def maxpool3x3s2(x, y, B, H, W, Cc, OH, OW):
    y.copy_(F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).to(y.dtype))


def image_pipeline(descs_dev, B, scratch, grey_sum, out, OH, OW, pad, Hp, Wp):
    raise RuntimeError('cpu_shim: the device input pipeline has no CPU emulation (oracle/image_oracle.py is its checker)')


def conv1x1_dual(a1, w1, a2, w2, bias, y, B, OH, OW, K1, IH2, IW2, K2, s2, N, act=1, y_mask_bits=None):
    return False                       # (bf16 kernel only: the CPU emulation runs the two convolutions)


def conv1x1_chain(a1, w1, a2, w2, s2, res, bias, y, wn, bias_n, z, B, OH, OW, z_mask_bits=None):
    return False                       # (bf16 kernel only: the CPU emulation runs the convolutions one by one)


def ffn_fused_fwd(*a, **k):
    return False                       # (bf16 kernel only: the CPU emulation runs gemm, gemm, layernorm_fwd)


def conv_wgrad_group(problems):
    for x, dy, dw, scale, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW in problems:
        conv2d(2, x, dy, dw, B, IH, IW, Cs, Cin, OH, OW, Cout, KH, KW, SH, SW, PH, PW, rowscale=scale)


def stem_pool(x, w, shift, y, B, Hp, Wp, CH, CW, PH, PW):
    # x [B,Hp,Wp,4] padded NHWC4, w [64][7][32] = [co][r][8 px][4 ch]
    wk = w.float().view(64, 7, 8, 4).permute(0, 3, 1, 2)                       # [co][ch][r][8]
    c = F.conv2d(x.float().permute(0, 3, 1, 2), wk, stride=2)[:, :, :CH, :CW] + shift.view(1, -1, 1, 1)
    c = F.relu(c).to(y.dtype).float()
    y.copy_(F.max_pool2d(c, 3, 2, 1).permute(0, 2, 3, 1).to(y.dtype))


def roi_weights(boxes, wgt, n_roi, H, W, ldw):
    from oracle import gpv_oracle as O
    x1 = W * (boxes[:, 0] - 0.5 * boxes[:, 2]) - 0.5
    x2 = W * (boxes[:, 0] + 0.5 * boxes[:, 2]) - 0.5
    y1 = H * (boxes[:, 1] - 0.5 * boxes[:, 3]) - 0.5
    y2 = H * (boxes[:, 1] + 0.5 * boxes[:, 3]) - 0.5
    ay = O.roi_axis_weights(y1.cpu(), (y2 - y1).cpu(), H).to(wgt.device)
    ax = O.roi_axis_weights(x1.cpu(), (x2 - x1).cpu(), W).to(wgt.device)
    wgt.zero_()
    wgt[:, :H * W] = (ay[:, :, None] * ax[:, None, :]).reshape(n_roi, H * W).to(wgt.dtype)


def add(a, b, y, n):
    y.copy_((a.float() + b.float()).to(y.dtype))


def add_rowbcast(a, b, y, rows_total, rows_b, cols):
    rep = rows_total // rows_b
    y.reshape(rep, rows_b * cols).copy_((a.float().reshape(rep, rows_b * cols) + b.float().reshape(1, rows_b * cols)).to(y.dtype))


def colsum(x, out, rows, cols, ld):
    out += _sv(x, (rows, cols), (ld, 1)).float().sum(0)


def cast(src, dst, n):
    dst.reshape(-1).copy_(src.reshape(-1).to(dst.dtype))


def cast_rowscale_t(src, scale, dst, dstT, rows, cols):
    v = src.reshape(rows, cols) * (scale[:, None] if scale is not None else 1.0)
    if dst is not None:
        dst.copy_(v.to(dst.dtype))
    if dstT is not None:
        dstT.copy_(v.t().to(dstT.dtype))


def cast_transpose_group(items):
    for src, dstT in items:
        dstT.copy_(src.t().to(dstT.dtype))


def prep_conv_weight(src, scale, wf, wd, Cout, T, Cin):
    v = src.reshape(Cout, T, Cin) * (scale[:, None, None] if scale is not None else 1.0)
    if wf is not None:
        wf.copy_(v.to(wf.dtype))
    if wd is not None:
        wd.copy_(v.permute(2, 1, 0).to(wd.dtype))


def embedding(table, ids, out, n_ids, dim):
    out.copy_(table[ids.reshape(-1)].to(out.dtype))


def dropout(x, y, n, p, seed):
    raise NotImplementedError('cpu shim: dropout unsupported')


def relevance_condition(x, logits, tokens, y, rows, dim):
    y.copy_((x.float() + logits.softmax(-1) @ tokens).to(y.dtype))


def adamw(p, g, m, v, p_lowp, n, lr, beta1, beta2, eps, wd, bc1, bc2, gscale=None, seg_id=None, seg_live=None):
    gi = g * (gscale if gscale is not None else 1.0)
    p2 = p * (1 - lr * wd)
    m2 = m * beta1 + gi * (1 - beta1)
    v2 = v * beta2 + gi * gi * (1 - beta2)
    if seg_id is not None:           # seg_live = per-parameter Adam step count (0 = never touched): per-parameter bias corrections
        t = seg_live[seg_id[:(n + 7) // 8].long()].repeat_interleave(8)[:n].double()
        live = t > 0
        c1 = (1 - beta1 ** t.clamp(min=1)).float()
        c2 = (1 - beta2 ** t.clamp(min=1)).float()
        p2 = p2 - (lr / c1) * m2 / (v2.sqrt() / c2.sqrt() + eps)
        p2, m2, v2 = torch.where(live, p2, p), torch.where(live, m2, m), torch.where(live, v2, v)
    else:
        p2 = p2 - (lr / bc1) * m2 / (v2.sqrt() / math.sqrt(bc2) + eps)
    p.copy_(p2); m.copy_(m2); v.copy_(v2)
    if p_lowp is not None:
        p_lowp.copy_(p.to(p_lowp.dtype))


def clip_scale(g, max_norm, partial, gscale, pstep=None, live=None):
    if g is not None:
        norm = g.double().pow(2).sum().sqrt().float()
        gscale.copy_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
    if pstep is not None:
        pstep.add_(live)


def sumsq(x, n, out):
    out += (x * x).sum()


def act_fwd(x, y, n, act):
    y.copy_(_act(x.float(), act).to(y.dtype))


def argmax_rows(x, addend, out0=None, out1=None, table=None, pos_row=None, xnext=None):
    f = x.float() if addend is None else x.float() + addend
    idx = f.argmax(-1)                       # (first maximum)
    for o in (out0, out1):
        if o is not None:
            o.copy_(idx)
    if table is not None:                    # gpv_argmax_rows_embed: the next input rows
        r = table[idx].float()
        xnext.copy_((r if pos_row is None else r + pos_row.float()).to(xnext.dtype))


def attention_row_proj(q, q_bs, k, k_bs, k_rs, v, v_bs, v_rs, Wo, partial, B, H, Sk, dh, scale):
    D = H * dh
    qh = _sv(q, (B, H, dh), (q_bs, dh, 1)).float()
    kh = _sv(k, (B, Sk, H, dh), (k_bs, k_rs, dh, 1)).float()
    vh = _sv(v, (B, Sk, H, dh), (v_bs, v_rs, dh, 1)).float()
    p = torch.softmax(torch.einsum('bhd,bjhd->bhj', qh, kh) * scale, -1)
    o = torch.einsum('bhj,bjhd->bhd', p, vh).to(q.dtype).float()
    w = Wo.float().reshape(D, H, dh)
    partial.copy_(torch.einsum('bhd,nhd->bhn', o, w))


def ln_linear_rows(x, s, gamma, beta, eps, xn, Wm, bias, y, ldy, rows, N, K, act=0, s_partial=None, s_bias=None):
    if s_partial is not None:
        s = (s_partial.sum(1) + (0 if s_bias is None else s_bias.float())).to(x.dtype)
    z = x.float().reshape(rows, K) + (0 if s is None else s.float().reshape(rows, K))
    n = F.layer_norm(z, (K,), None if gamma is None else gamma.float(), None if beta is None else beta.float(), eps).to(xn.dtype)
    xn.reshape(rows, K).copy_(n)
    o = n.float() @ Wm.float().t()
    if bias is not None:
        o = o + bias.float()
    _sv(y, (rows, N), (ldy, 1)).copy_(_act(o, act).to(y.dtype))


def act_bwd(dy, ref, dx, n, act, alpha=1.0):
    g, r = dy.float(), ref.float()
    if act == 1:
        out = torch.where(r > 0, g * alpha, torch.zeros_like(g))
    else:
        cdf = 0.5 * (1 + torch.erf(r / math.sqrt(2.0)))
        pdf = torch.exp(-0.5 * r * r) / math.sqrt(2 * math.pi)
        out = g * (cdf + r * pdf) * alpha
    dx.copy_(out.to(dx.dtype))


def install(only=None):
    """monkeypatch gpv1_amd.hip with the emulations above; returns an uninstall callable.
    `only`: optional list of entry-point names (GPU bisecting: swap single kernels for torch math)."""
    import gpv1_amd.hip as h
    names = only or ['gemm', 'conv2d', 'conv2d_mask_bits_ok', 'attention_fwd', 'attention_bwd', 'attention_qkv_fwd', 'layernorm_fwd', 'layernorm_bwd', 'linear_layernorm_fwd', 'softmax_ce',
             'image_to_nhwc4', 'maxpool3x3s2', 'stem_pool', 'conv1x1_dual', 'conv1x1_chain', 'ffn_fused_fwd', 'conv_wgrad_group', 'roi_weights', 'add', 'add_rowbcast', 'colsum', 'cast', 'cast_rowscale_t',
             'prep_conv_weight', 'embedding', 'dropout', 'relevance_condition', 'adamw', 'sumsq', 'clip_scale', 'act_fwd', 'act_bwd',
             'cast_transpose_group', 'argmax_rows', 'ln_linear_rows', 'attention_row_proj']
    saved = {n: getattr(h, n) for n in names}
    for n in names:
        setattr(h, n, globals()[n])

    def undo():
        for n, f in saved.items():
            setattr(h, n, f)
    return undo
