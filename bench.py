"""bench.py -- GPV-1 train-step throughput on MI355X (BASELINE.json metric: images/sec/node, train
step, 480x640, bs32/GPU; workload = configs[1]: CocoCaptioning-only train step, bf16, one rank per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W]        (N>1: launched by torch.distributed.run)

A "step" is one full iteration of the reference hot loop (exp/gpv/train_distr.py:399-428):
encode_answers -> GPV.forward (ResNet-50 -> DETR 6+6 -> RoI head -> BERT -> co-attention x3 -> text
decoder x3 -> vocabulary logits V=10000 -> caption CE) -> backward -> gradient all-reduce (N>1) ->
clip_grad_norm_(DETR, 0.1) -> AdamW.  Dropout 0.1 is ON (train mode).  Synthetic data (seeded),
random-init weights; inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract fields + "roofline" (dominant kernel: the implicit-GEMM
convolution of the backbone, timed live with HIP events on the launch stream inside the timed region)
+ "roofline_attention" (encoder attention core against the dense bf16 MFMA peak) + "cpu_baseline" (the CPU
oracle's forward+backward and greedy decode on this host, bounded sample, N=1 only).
--gpus N without a launcher starts the N ranks itself (torch.distributed.run on 127.0.0.1).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V = 10000
BATCH = 32
IMG = (480, 640)
TL = 6
CAP_WORDS = 18           # + __cls__ + __stop__ = 20 tokens = max_text_len


def live_conv_traffic():
    """roofline.traffic measured by the run that prints the line (a committed figure cannot notice a regression):
    two child runs of this file under `rocprofv3 --pmc` -- FETCH_SIZE, then WRITE_SIZE, each counter in its own pass and with no trace
    domain beside it, exactly as the guide's HBM section prescribes -- after the timed region, with the GPU otherwise idle.
    -> tools/pmc_traffic.py's dict, {'error': ...} when a pass failed, None when rocprofv3 is not there."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location('pmc_traffic', os.path.join(here, 'tools', 'pmc_traffic.py'))
    pt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pt)
    tmp = tempfile.mkdtemp(prefix='gpv_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    dirs = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = dirs[counter] = os.path.join(tmp, counter)
            cmd = [exe, '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 'r', '--', sys.executable, os.path.abspath(__file__), '--no-cpu-baseline',
                   '--no-decode', '--no-ragged', '--no-extra', '--no-traffic', '--steps', '4', '--warmup', '3']
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                return {'error': '%s pass: rc %d: %s' % (counter, r.returncode, (r.stderr or r.stdout)[-300:])}
        res = pt.compute(dirs['FETCH_SIZE'], dirs['WRITE_SIZE'])
        passes = res['passes_profiled']
        if not (passes >= 1 and passes == int(passes) and res['conv_launches_fetch_pass'] == res['conv_launches_write_pass']
                and res['conv_launches_fetch_pass'] % int(passes) == 0):
            return {'error': 'the two passes do not hold the same whole passes of the body: %r' % {k: res[k] for k in ('passes_profiled', 'conv_launches_fetch_pass', 'conv_launches_write_pass')}}
        res.pop('by_kernel_KB', None)
        return res
    except Exception as e:                                  # (a profiler hiccup must not cost the run its line)
        return {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _note_sync():
    """a synchronous collective just ran on the compute stream: the next capture waits for RCCL's watchdog once (misc.CollectiveClock)"""
    from gpv1_amd.misc import note_sync_collective
    note_sync_collective()


def _cc():
    from gpv1_amd.misc import CollectiveClock
    return CollectiveClock


def make_cfg():
    from gpv1_amd import synthetic
    g = torch.Generator().manual_seed(0)
    return synthetic.model_cfg(vocab=synthetic.make_vocab(V), vocab_embed=0.1 * torch.randn(V, 768, generator=g))


def make_batch(rank, B, dev):
    g = torch.Generator().manual_seed(1234 + rank)
    images = torch.randn(B, 3, *IMG, generator=g).to(dev)
    mask = torch.zeros(B, *IMG, dtype=torch.bool, device=dev)
    ids = torch.randint(1000, 30000, (B, TL), generator=g).to(dev)
    attn = torch.ones(B, TL, dtype=torch.long, device=dev)
    words = torch.randint(0, V - 4, (B, CAP_WORDS), generator=g)
    targets = [{'task': 'CocoCaptioning', 'answer': ' '.join(f'w{int(w)}' for w in row)} for row in words]
    return images, mask, ids, attn, targets


def conv_algorithmic(model, B):
    """algorithmic bytes / flops of the backbone convolutions for one step (fwd, and bwd of the
    trainable layers), bf16 activations+weights, fp32 weight gradients -- DESIGN.md 'roofline'."""
    H, W = IMG
    body = model.detr.backbone[0].body
    oh, ow = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    fb = B * (H + 6) * (W + 8) * 4 * 2 + B * oh * ow * 64 * 2 + 64 * 7 * 32 * 2
    ff = 2.0 * B * oh * ow * 64 * 7 * 32
    n_f, n_b = 1, 0
    bb = bf = 0.0
    bound = [max(fb / 8e12, ff / 2.5e15)]        # per launch: max(bytes / HBM peak, flops / dense bf16 MFMA peak), seconds
    h, w = (oh + 2 - 3) // 2 + 1, (ow + 2 - 3) // 2 + 1
    seen = False
    for blk in body.blocks():
        tr = blk.trainable()
        hin, win = h, w
        for conv, _ in blk.convs():
            ih, iw = (hin, win) if conv is not blk.conv3 else (h, w)
            if conv is blk.conv1 or (blk.downsample is not None and conv is blk.downsample[0]):
                ih, iw = hin, win
            o_h, o_w = (ih + 2 * conv.pad - conv.k) // conv.stride + 1, (iw + 2 * conv.pad - conv.k) // conv.stride + 1
            x_b, y_b = B * ih * iw * conv.cin * 2, B * o_h * o_w * conv.cout * 2
            w_b = conv.cout * conv.k * conv.k * conv.cin * 2
            fl = 2.0 * B * o_h * o_w * conv.cout * conv.k * conv.k * conv.cin
            fb += x_b + y_b + w_b
            ff += fl
            n_f += 1
            bound.append(max((x_b + y_b + w_b) / 8e12, fl / 2.5e15))
            if tr:
                bb += x_b + y_b + 2 * w_b                      # wgrad: read x, dy ; write dW (fp32)
                bf += fl
                n_b += 1
                bound.append(max((x_b + y_b + 2 * w_b) / 8e12, fl / 2.5e15))
                needs_dx = not (conv is blk.conv1 or (blk.downsample is not None and conv is blk.downsample[0])) or seen
                if needs_dx:
                    bb += x_b + y_b + w_b                      # dgrad: read dy, W ; write dx
                    bf += fl
                    n_b += 1
                    bound.append(max((x_b + y_b + w_b) / 8e12, fl / 2.5e15))
            if conv is blk.conv2:
                h, w = o_h, o_w
        if tr:
            seen = True
    return {'fwd_bytes': fb, 'fwd_flops': ff, 'bwd_bytes': bb, 'bwd_flops': bf, 'fwd_launches': n_f, 'bwd_launches': n_b,
            'mixed_bound_s': sum(bound)}


def cpu_baseline(model, seconds_budget=40.0):
    """CPU oracle (oracle/gpv_oracle.py, the pinned restatement of the reference) on the host cores, SURVEY 8(d) protocol:
    forward+loss+backward at B=4, full size, median of 5 after 2 warm-ups; greedy decode at B=1, median of 3 after 1 warm-up.
    A reported baseline, not the target.  (tools/time_reference_cpu.py times the imported reference itself in the build
    container: BASELINE.md.)"""
    from oracle import gpv_oracle as O
    # 256 OpenMP threads on ~1e4 small ops is far slower than 32 (barrier cost): use min(cores, 32) and say so
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    Bc = 4
    cfg = make_cfg()
    cfg['_cls_id'] = V - 3
    Pm = {k: v.detach().float().cpu().contiguous() for k, v in model.state_dict().items()}
    images, mask, ids, attn, targets = make_batch(0, Bc, 'cpu')
    w2i = {w: i for i, w in enumerate(cfg['vocab'])}
    _, tok = O.encode_answers(targets, w2i, cfg['max_text_len'])
    for i, t in enumerate(targets):
        t['answer_token_ids'] = tok[i, 1:]
    train_keys = [n for n, p in model.named_parameters() if p.requires_grad and not n.startswith('bert.')]
    times = []
    t_start = time.time()
    warm = 2
    for it in range(warm + 5):
        leaves = {k: Pm[k].clone().requires_grad_(True) for k in train_keys}
        Pg = dict(Pm)
        Pg.update(leaves)
        t0 = time.time()
        out = O.gpv_forward(Pg, cfg, images, mask, ids, attn, tok, training=True)
        loss, _ = O.gpv_criterion(out, targets, cfg['losses'])
        loss.backward()
        if it >= warm:
            times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget and len(times) >= 1:
            break
    t = sorted(times)[len(times) // 2]
    gt = []
    with torch.no_grad():
        for it in range(4):
            t0 = time.time()
            O.gpv_forward(Pm, cfg, images[:1], mask[:1], ids[:1], attn[:1], None, training=False)
            if it >= 1:
                gt.append(time.time() - t0)
            if time.time() - t_start > 2 * seconds_budget and gt:
                break
    g = sorted(gt)[len(gt) // 2]
    return {'value': Bc / t, 'unit': 'images/sec', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'oracle fwd+loss+bwd (no optimizer), B={Bc}, 480x640, V={V}, fp32, median of {len(times)} after {warm} warm-ups; '
                      f'greedy decode B=1: median of {len(gt)} after 1 warm-up',
            'greedy_ms_per_image': g * 1e3}


def attention_roofline(dev, B):
    """north_star's second roofline: MFMA utilisation on the encoder attention path.  The attention core of one DETR encoder
    layer at the workload shape (B x 8 heads, 300 x 300 scores, dh = 32, dropout 0.1, operands = column slices of the fused
    q|k and v projection buffers exactly as the model passes them), timed live with HIP events on the launch stream.
    Algorithmic flops of the fused core = 4*B*h*S^2*dh forward (QK^T + PV), x2.5 backward."""
    import math
    import gpv1_amd.hip as hip
    H, S, dh = 8, 300, 32
    D = H * dh
    g = torch.Generator().manual_seed(5)
    qk = torch.randn(B * S, 2 * D, generator=g).to(dev).to(torch.bfloat16)
    v = torch.randn(B * S, D, generator=g).to(dev).to(torch.bfloat16)
    do = torch.randn(B * S, D, generator=g).to(dev).to(torch.bfloat16)
    o = torch.empty(B * S, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
    dqk, dv = torch.empty_like(qk), torch.empty_like(v)
    st = ((S * 2 * D, 2 * D), (S * 2 * D, 2 * D), (S * D, D), (S * D, D))
    scale = 1.0 / math.sqrt(dh)

    def fwd():
        hip.attention_fwd(qk[:, :D], qk[:, D:], v, o, st, B, H, S, S, dh, scale, drop_p=0.1, seed=11, lse=lse)

    def bwd():
        hip.attention_bwd(qk[:, :D], qk[:, D:], v, o, do, dqk[:, :D], dqk[:, D:], dv, st, (S * D, D), B, H, S, S, dh, scale,
                          drop_p=0.1, seed=11, lse=lse)
    out = {}
    for name, fn in (('fwd', fwd), ('bwd', bwd)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * H * S * S * dh
    tf = fl / (out['fwd'] * 1e-6) / 1e12
    # the launch the encoder layers actually run since round 5: q | k | v projections + core in ONE launch (gpv_attention_qkv_fwd);
    # flops = projections (2 M 256 768) + core ("with projections")
    g2 = torch.Generator().manual_seed(6)
    x = torch.randn(B * S, D, generator=g2).to(dev).to(torch.bfloat16)
    xp = (x.float() + torch.randn(B * S, D, generator=g2).to(dev)).to(torch.bfloat16)
    w = (torch.randn(3 * D, D, generator=g2) / 16).to(dev).to(torch.bfloat16)
    bias = torch.randn(3 * D, generator=g2).to(dev)

    def fused():
        hip.attention_qkv_fwd(xp, x, w, bias, qk[:, :D], qk[:, D:], v, o, st, B, H, S, scale, drop_p=0.1, seed=11, lse=lse)
    for _ in range(3):
        fused()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fused()
    e1.record()
    torch.cuda.synchronize()
    fused_us = e0.elapsed_time(e1) / 20 * 1e3
    fl_proj = 2.0 * B * S * D * 3 * D
    fused_tf = (fl + fl_proj) / (fused_us * 1e-6) / 1e12
    # The VALU ceiling the core sits under: instruction count x issue rate.  SQ_INSTS_VALU of the forward launch at
    # this shape = 4.579e6 wave-instructions (profiles/r03_pmc_attention.txt: rocprofv3 --pmc, per launch, B = 32); a wave64 VALU
    # instruction occupies its SIMD16 for 4 cycles, v_exp_f32 (one per score and lane: B h S^2 / 64 = 3.6e5 of them) for 16; the
    # chip has 256 CUs x 4 SIMDs at 2.4 GHz.  Scaled with B h S^2 for other batch sizes.
    valu_insts = 4.579e6 * (B * H * S * S) / (32.0 * 8 * 300 * 300)
    exp_insts = B * H * S * S / 64.0
    valu_cycles = (valu_insts * 4.0 + exp_insts * 12.0) / (256 * 4)
    valu_us = valu_cycles / 2400.0
    valu_tf = fl / (valu_us * 1e-6) / 1e12
    return {'bound': 'mfma', 'valu_ceiling_tflops': valu_tf, 'valu_ceiling_us': valu_us, 'frac_of_valu_ceiling': tf / valu_tf,
            'valu_ceiling_how': 'SQ_INSTS_VALU per launch (profiles/r03_pmc_attention.txt) x 4 cycles per wave64 instruction (+12 for each '
                                'quarter-rate v_exp_f32) / 1024 SIMDs / 2.4 GHz: the time the launch needs for its vector instructions alone', 'achieved': tf, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': tf / 2500.0,
            'kernel': 'attn_q_kernel<bf16,32,32,20,0> (DETR encoder self-attention core, forward; B x 8 heads, 300x300, dh 32, dropout on)',
            'flops_per_launch': fl, 'avg_launch_us': out['fwd'], 'bwd_us': out['bwd'],
            'bwd_kernel': 'attn_bwd1_kernel<32,32,20,20> (dQ, dK, dV in ONE launch: S formed once, dS transposed through LDS for the dQ product; '
                          'round 3: attn_q_kernel<..,1> + attn_kv2_kernel, 55 us)',
            'bwd_tflops': 2.5 * fl / (out['bwd'] * 1e-6) / 1e12,
            'with_projections': {'kernel': 'attn_qkv_kernel<20> (q | k | v in-projection + core in one launch: what an encoder layer runs)',
                                 'flops_per_launch': fl + fl_proj, 'avg_launch_us': fused_us, 'achieved': fused_tf, 'frac': fused_tf / 2500.0},
            'note': 'dh = 32: 2 MFMAs per 256 scores against ~12 VALU instructions per score -- the core is VALU-bound, see DESIGN.md'}


def self_launch(args):
    """python bench.py --gpus N without a launcher: start N ranks on this node (the reference spawns its own ranks too,
    exp/gpv/train_distr.py:488-493) through torch.distributed.run and pass its exit code on"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def greedy_decode_bench(model, dev):
    """second half of BASELINE.json's metric: greedy-decode ms/image (whole inference: backbone -> DETR -> BERT ->
    co-attention -> 20-token KV-cached decode, one hipGraph per step), eval mode, bf16, inputs resident in HBM."""
    from gpv1_amd.misc import nested_tensor_from_tensor_list
    model.eval()
    if os.environ.get('GPV_DEBUG_SYNC') == '1' or os.environ.get('GPV_NO_GRAPHS') == '1':
        model.cfg['kv_graphs'] = model.cfg['graph_inference'] = False   # debug aids (hip.py): no capture while synchronising per call
    res = {}
    with torch.no_grad():
        for Bd, iters in ((1, 10), (64, 5)):
            images, mask, ids, attn, _ = make_batch(7, Bd, dev)
            samples_d = nested_tensor_from_tensor_list(images)
            for _ in range(2):                                   # warm-up: captures the 20 step graphs for this batch size
                model(samples_d, (ids, attn), None, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                o = model(samples_d, (ids, attn), None, None)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            res[f'bs{Bd}'] = {'ms_per_batch': ms, 'ms_per_image': ms / Bd}
    res['what'] = 'GPV.forward(images, queries, None): max_text_len=20 greedy, KV cache, whole inference replayed as one hipGraph, bf16'
    model.train()
    return res


def ragged_bench(model, tr, dev, rank, steps=36):
    """the reference's own API and batch shapes (train_distr.py:399-428): `queries` as strings of 4..14 words (6..16 WordPiece
    tokens), caption answers of 1..18 words, a fresh combination every step; the trainer tokenises on the host, pads to size classes
    and replays captured hipGraphs (train.FlatTrainer._classed).  Reported next to the fixed-shape number."""
    import tempfile
    from gpv1_amd import synthetic
    from gpv1_amd.bert import WordPieceTokenizer
    from gpv1_amd.misc import nested_tensor_from_tensor_list
    with tempfile.TemporaryDirectory() as td:
        words = synthetic.write_wordpiece_vocab(os.path.join(td, 'vocab.txt'))
        model.bert.tokenizer = WordPieceTokenizer(os.path.join(td, 'vocab.txt'))
    g = torch.Generator().manual_seed(4321 + rank)
    images = torch.randn(BATCH, 3, *IMG, generator=g).to(dev)
    samples = nested_tensor_from_tensor_list(images)

    def make(it):
        nq = 4 + (5 * it) % 11                                            # longest query of the batch: 4..14 words
        na = 1 + (7 * it) % 18                                            # longest caption: 1..18 words
        qs = [' '.join(words[int(j)] for j in torch.randint(0, len(words), (max(2, nq - i % 3),), generator=g)) for i in range(BATCH)]
        tg = [{'task': 'CocoCaptioning', 'answer': ' '.join(f'w{int(j)}' for j in torch.randint(0, V - 4, (max(1, na - i % 4),), generator=g))}
              for i in range(BATCH)]
        return qs, tg
    batches = [make(it) for it in range(steps)]
    g0, e0 = tr.graph_steps, tr.eager_steps
    for qs, tg in batches[:18]:                                           # warm-up: every size class is seen twice (eager, capture)
        tr.train_step(samples, list(qs), [dict(t) for t in tg])
    for qs, tg in batches[:18]:
        tr.train_step(samples, list(qs), [dict(t) for t in tg])
    torch.cuda.synchronize()
    g1, e1 = tr.graph_steps, tr.eager_steps
    t0 = time.perf_counter()
    for qs, tg in batches:
        tr.train_step(samples, list(qs), [dict(t) for t in tg])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'ms_per_step': dt / steps * 1e3, 'images_per_sec': BATCH * steps / dt, 'steps': steps,
            'graph_steps_timed': tr.graph_steps - g1, 'eager_steps_timed': tr.eager_steps - e1,
            'captured_bodies': len(tr._bodies),
            'what': 'string queries of 6..16 WordPiece tokens, captions of 1..18 words, a different (T_l, S) every step; host tokenisation + '
                    'size-class padding inside the timed region'}


def extra_configs(model, tr, dev, rank):
    """BASELINE.json configs[3] (beam_size 5 decode, 64 images, one hipGraph) and configs[4] (CocoDetection-only train step, 64
    images per GPU: Hungarian matcher + set criterion on 100 queries) -- single-GPU figures."""
    from gpv1_amd.misc import nested_tensor_from_tensor_list
    out = {}
    images, mask, ids, attn, _ = make_batch(11 + rank, 64, dev)
    samples = nested_tensor_from_tensor_list(images)
    g = torch.Generator().manual_seed(99 + rank)
    tg = []
    for i in range(64):
        n = 1 + int(torch.randint(0, 10, (1,), generator=g))
        cxcy = 0.25 + 0.5 * torch.rand(n, 2, generator=g)
        wh = 0.05 + 0.3 * torch.rand(n, 2, generator=g)
        tg.append({'task': 'CocoDetection', 'boxes': torch.cat((cxcy, wh), 1).to(dev), 'labels': torch.zeros(n, dtype=torch.long, device=dev)})
    for _ in range(3):
        tr.train_step(samples, (ids, attn), [dict(t) for t in tg])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_it = 6
    for _ in range(n_it):
        tr.train_step(samples, (ids, attn), [dict(t) for t in tg])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    out['detection_only_bs64_train_step'] = {'ms_per_step': dt * 1e3, 'images_per_sec': 64 / dt,
                                             'what': 'configs[4]: CocoDetection-only targets (1..10 boxes), B = 64, matcher + set criterion on the host, AdamW'}
    model.eval()
    with torch.no_grad():
        for _ in range(2):
            model.forward_beam_search(samples, (ids, attn), beam_size=5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.forward_beam_search(samples, (ids, attn), beam_size=5)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    model.train()
    out['beam5_bs64'] = {'ms_per_batch': dt * 1e3, 'ms_per_image': dt * 1e3 / 64,
                         'what': 'configs[3]: forward_beam_search(beam_size=5), 64 images, KV cache, the whole search one hipGraph, incl. host detokenisation'}
    return out


def precise_bench(model, tr, dev, rank):
    """the mode that meets north_star's 1e-3 parity bar (fp32 storage, every MFMA product as hi*hi + lo*hi + hi*lo on split bf16
    operands: tests/test_model_gpu.py precise cases) on the SAME workload and trainer: eager steps (the hipGraph path is bf16 only)"""
    from gpv1_amd.ops import RT
    from gpv1_amd.misc import nested_tensor_from_tensor_list
    images, mask, ids, attn, targets = make_batch(rank, BATCH, dev)
    samples = nested_tensor_from_tensor_list(images)
    RT.set_precise(True)
    try:
        for _ in range(2):
            tr.train_step(samples, (ids, attn), [dict(t) for t in targets])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_it = 3
        for _ in range(n_it):
            loss = tr.train_step(samples, (ids, attn), [dict(t) for t in targets])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_it
    finally:
        RT.set_precise(False)
    return {'ms_per_step': dt * 1e3, 'images_per_sec': BATCH / dt, 'finite': bool(torch.isfinite(loss.detach())),
            'what': 'configs[1] workload in precise mode (fp32 storage, 3-MFMA split-bf16 products; the mode the 1e-3 parity bar is checked in), eager launches'}


def input_pipeline_bench(dev):
    """32 COCO-sized JPEG files -> host entropy decoding (8 threads) -> GPU (IDCT, upsampling, colour, anti-aliased
    resize to 480x640, ColorJitter / flip / grayscale, normalise) -> the stem's NHWC4 input.  Needs Pillow to WRITE the synthetic
    files; reported beside the train step because this is what has to keep up with it."""
    try:
        from PIL import Image
    except ImportError:
        return {'skipped': 'Pillow not importable (it only writes the synthetic test files)'}
    import io
    import numpy as np
    from gpv1_amd.jpeg import DeviceJpegDecoder
    from gpv1_amd.input_pipeline import DeviceImagePipeline
    r = np.random.RandomState(0)
    files = []
    for i in range(BATCH):
        h, w = ((480, 640), (427, 640), (640, 480), (500, 375))[i % 4]
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 100 * np.sin(xx / (11.0 + i)) * np.cos(yy / 23.0), xx * 255.0 / w, yy * 255.0 / h], -1) + r.randn(h, w, 3) * 14
        buf = io.BytesIO()
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(buf, 'JPEG', quality=90, subsampling=2)
        files.append(buf.getvalue())
    dec, pipe = DeviceJpegDecoder(device=dev, threads=8), DeviceImagePipeline(size=IMG, train=True, device=dev)
    tasks = ['CocoClassification'] * BATCH
    for _ in range(2):
        pipe(dec(files), tasks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_it = 10
    for _ in range(n_it):
        out = pipe(dec(files), tasks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    return {'ms_per_batch': dt * 1e3, 'images_per_sec': BATCH / dt, 'jpeg_mbytes_per_batch': sum(len(f) for f in files) / 1e6,
            'host_threads': 8, 'output': list(out.tensors.shape),
            'what': '32 JPEG files (4:2:0, q90, 480x640 / 427x640 / 640x480 / 500x375) -> host Huffman decoding -> device IDCT + upsampling + '
                    'colour + resize + augmentation + normalise -> NHWC4 bf16 stem input; wall clock incl. the uploads'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--no-ragged', action='store_true', help='skip the string-query / ragged-length run reported next to the fixed-shape number')
    ap.add_argument('--no-extra', action='store_true', help='skip BASELINE configs[3] (beam 5, bs64) and configs[4] (detection-only bs64)')
    ap.add_argument('--no-traffic', action='store_true', help='do not run the two rocprofv3 --pmc passes of this command (roofline.traffic then comes from the committed profile)')
    ap.add_argument('--soak', type=int, default=0, help='after the timed region: this many more graphed steps (soak of the hipGraph replay path)')
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU '
                         f'(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('GPV_GRAPHS_STRICT', '1')       # a failed hipGraph capture must fail the bench, not degrade it to eager steps
    local = local % max(torch.cuda.device_count(), 1)   # (only differs on a box with fewer GPUs than ranks: the gloo dry run below)
    torch.cuda.set_device(local)
    dev = f'cuda:{local}'
    # GPV_FORCE_COMM=1 at --gpus 1: a process group of ONE rank under the real backend -- the whole N > 1 path (RCCL init, bucketed
    # asynchronous all-reduces between the stage graphs of the backward pass, the gloo agreement channel) on a single-GPU box
    multi = world > 1 or os.environ.get('GPV_FORCE_COMM', '0') == '1'
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        backend = os.environ.get('GPV_DIST_BACKEND', 'nccl')     # 'gloo': dry run of the N > 1 path with ranks sharing one GPU
        from gpv1_amd.train import init_process_group
        init_process_group(rank, world, dev, backend)           # RCCL: high-priority stream, communicator bound to this rank's GPU

    import gpv1_amd.hip as hip
    import gpv1_amd.backbone as bbm
    from gpv1_amd.gpv import GPV
    from gpv1_amd.misc import nested_tensor_from_tensor_list
    from gpv1_amd.train import FlatTrainer
    from gpv1_amd.ops import RT
    hip.lib()
    torch.manual_seed(0)
    model = GPV(make_cfg())
    for n, buf in model.named_buffers():
        if n.endswith('running_var'):
            buf.uniform_(0.5, 1.5)
    model.to(dev)
    RT.manual_seed(1000 + rank)
    total_steps = 1000
    tr = FlatTrainer(model, lr=1e-4, lr_backbone=1e-5, weight_decay=1e-4, clip_max_norm=0.1,
                     warmup_steps=int(0.1 * total_steps), t_total=total_steps)
    images, mask, ids, attn, targets = make_batch(rank, args.batch, dev)
    samples = nested_tensor_from_tensor_list(images)           # the reference's collate (detr_misc.py:267-299): same-size images, no padding

    def step():
        tg = [dict(t) for t in targets]
        return tr.train_step(samples, (ids, attn), tg)

    if os.environ.get('GPV_BENCH_INJECT_CAPTURE_FAILURE') == '1':            # exercises the fallback below (tests / dry runs only)
        import gpv1_amd.train as _trm

        def _boom(self, *a, **k):
            raise RuntimeError('injected capture failure (GPV_BENCH_INJECT_CAPTURE_FAILURE)')
        _trm.GraphedBody.__init__ = _boom
    # warm-up (captures the hipGraphs on its second step).  GPV_GRAPHS_STRICT=1: a failed capture raises instead of silently
    # degrading a rank to eager steps.  On one GPU that ends the bench (it is a bug).  With several ranks -- RCCL + capture is the
    # one path no single-GPU box could exercise -- the ranks agree on the failure, ALL switch the graphs off, warm up again, and the
    # line says so (`graphs.enabled: false` + the error): a labelled eager number instead of none.
    graphs_note = None
    try:
        for _ in range(args.warmup):
            loss = step()
        torch.cuda.synchronize()
        ok = 1
    except RuntimeError as err:
        if not multi:
            raise
        ok, graphs_note = 0, '%s: %s' % (type(err).__name__, (str(err).splitlines() or [''])[0])
    if multi:
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        _note_sync()
        if int(flag) == 0:
            os.environ['GPV_GRAPHS_STRICT'] = '0'
            tr.disable_graphs(graphs_note or 'another rank failed to capture')
            graphs_note = graphs_note or 'another rank failed to capture'
            torch.cuda.synchronize()
            for _ in range(args.warmup):
                loss = step()
            torch.cuda.synchronize()
        dist.barrier()
        _note_sync()
    torch.cuda.synchronize()
    bbm.PROF = []
    if multi:
        tr.comm_prof = []
    import gpv1_amd.train as trm
    trm.HOST_PROF = {}
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - h0              # host time inside train_step (no sync): how far the host runs ahead
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
        _note_sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    host_prof, trm.HOST_PROF = trm.HOST_PROF, None
    trm.HOST_PROF = host_prof if False else None
    prof_timed, bbm.PROF = bbm.PROF, None
    # Convolution time for the roofline.  In the timed region the backbone graphs also carry other kernels as parallel branches
    # (train.GraphedBody: the frozen BERT beside the forward convolutions, the model body's weight-gradient GEMMs beside the
    # backward ones), so their brackets are conv + 1-3 ms of other kernels' work sharing the CUs (reported as
    # timed_region_brackets_ms).  The conv kernels' own duration is taken right after the timed region, in this same process: the
    # same 135 launches (forward_nhwc + backward_nhwc on the workload batch), alone on the stream, HIP events around them.
    if prof_timed is not None and any(True for _ in prof_timed):
        body = model.detr.backbone[0].body
        dc5 = None
        prof = []
        with torch.no_grad():
            for it in range(args.steps + 2):
                keep = []
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                c5 = body.forward_nhwc(images, keep)
                e1.record()
                if dc5 is None:
                    dc5 = torch.randn(c5.shape, device=dev).to(c5.dtype)
                body.backward_nhwc(keep, dc5)
                e2.record()
                if it >= 2:
                    prof += [('conv_fwd', e0, e1), ('conv_bwd', e1, e2)]
        torch.cuda.synchronize()
    else:
        prof = prof_timed
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        _note_sync()
        elapsed = float(t)
    comm = None
    if multi:
        torch.cuda.synchronize()
        exposed = [a.elapsed_time(b) for a, b in (tr.comm_prof or [])]
        tr.comm_prof = None
        comm = {'backend': dist.get_backend(), 'rccl_ranks': world if dist.get_backend() == 'nccl' else 0, 'ranks': world,
                'grad_comm_dtype': str(tr.grad_comm_dtype).replace('torch.', ''), 'bytes_per_rank_per_step': tr.comm_bytes_per_step(),
                'buckets': len(tr.buckets), 'exposed_ms_per_step': sum(exposed) / max(len(exposed), 1),
                'graph_steps': tr.graph_steps, 'eager_steps': tr.eager_steps,
                'capture_quiesce': {'calls': _cc().calls, 'sleeps': _cc().sleeps, 'mode': _cc().MODE},
                'milestones_last_step': [m for m, _ in tr.milestone_log], 'left_after_backward_bytes': getattr(tr, 'left_after_backward', None),
                'what': 'exposed = GPU time between the first bucket wait and the last bucket back on the compute stream (what the overlap did not hide), rank 0'}
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    soak = None
    if args.soak > 0 and world == 1:
        t0s = time.perf_counter()
        for _ in range(args.soak):
            ls = step()
        torch.cuda.synchronize()
        soak = {'steps': args.soak, 'ms_per_step': (time.perf_counter() - t0s) / args.soak * 1e3, 'final_loss': float(ls.detach()),
                'finite': bool(torch.isfinite(ls.detach()))}
    # ---- roofline of the dominant kernel (implicit-GEMM conv, backbone fwd+bwd), live HIP events ----
    alg = conv_algorithmic(model, args.batch)
    ms = {'conv_fwd': 0.0, 'conv_bwd': 0.0}
    for tag, a, b in prof:
        ms[tag] += a.elapsed_time(b)
    conv_ms = (ms['conv_fwd'] + ms['conv_bwd']) / args.steps
    launches = alg['fwd_launches'] + alg['bwd_launches']
    bytes_step = alg['fwd_bytes'] + alg['bwd_bytes']
    flops_step = alg['fwd_flops'] + alg['bwd_flops']
    ach_gbs = bytes_step / (conv_ms * 1e-3) / 1e9
    # HBM traffic per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md): cannot be collected
    # inside a timed run, so it is the committed measurement of this same command (tools/pmc_traffic.py ->
    # profiles/rNN_pmc_conv_traffic.json, per-launch = a pass's bytes / the reference's 135 convolutions); null if that file is
    # missing or does not hold whole passes.
    traffic = None
    traffic_source = None
    traffic_live = None
    if not (args.no_traffic or args.no_extra) and world == 1 and args.batch == BATCH:
        traffic_live = live_conv_traffic()
        if traffic_live is not None and 'traffic_bytes_per_launch' in traffic_live:
            traffic = traffic_live['traffic_bytes_per_launch']
            traffic_source = ('live: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE) of `python bench.py --steps 4 --warmup 3` run by THIS process after the timed region '
                              '(%d whole passes of the body; FETCH_SIZE x 2 on gfx950, units and corrections as /opt/skills/guides/MI355X_MICROARCH.md; tools/pmc_traffic.py)'
                              % int(traffic_live['passes_profiled']))
    try:
        if traffic is not None:
            raise OSError('live')
        import glob
        latest = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_pmc_conv_traffic.json')))[-1]
        pm = json.load(open(latest))
        passes = pm['passes_profiled']                # whole forward + backward passes of the body in the profiled run (stem launches)
        if passes >= 1 and passes == int(passes) and pm['conv_launches_fetch_pass'] % int(passes) == 0 \
                and pm['conv_launches_fetch_pass'] == pm['conv_launches_write_pass'] and args.batch == BATCH:
            traffic = pm['traffic_bytes_per_launch']
            traffic_source = 'profiles/%s (committed PMC measurement of this command, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: not collected by this run)' % os.path.basename(latest)
    except (OSError, KeyError, ValueError, IndexError):
        pass
    roof = {'bound': 'hbm', 'achieved': ach_gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach_gbs / 8000.0, 'traffic': traffic,
            'traffic_source': traffic_source,
            'traffic_over_algorithmic': None if traffic is None else traffic / (bytes_step / launches),
            'traffic_live_error': None if (traffic_live is None or 'error' not in traffic_live) else traffic_live['error'],
            'frac_mixed_per_conv_bound': alg['mixed_bound_s'] / (conv_ms * 1e-3),      # sum over launches of max(bytes / 8 TB/s, flops / 2.5 PF) / measured time
            'kernel': 'c1s_kernel (streaming 1x1) / c3r_kernel (streaming 3x3) / stem_pool_kernel / c1d_kernel / glds_wgrad_group_kernel + wg8_group_kernel + wg8h_group_kernel (the 42 weight gradients as one grouped call per stage) / glds_kernel<OP_CONV> / pipe_kernel<OP_CONV> / pipe_conv1x1_kernel / conv1x1_kernel / gemm_kernel<OP_CONV> (NHWC conv fwd/dgrad/wgrad, ResNet-50 body)',
            'launches_per_step': launches, 'avg_launch_us': conv_ms * 1e3 / launches,
            'algorithmic_bytes_per_launch': bytes_step / launches,
            'fwd_ms': ms['conv_fwd'] / args.steps, 'bwd_ms': ms['conv_bwd'] / args.steps,
            'timed_region_brackets_ms': {'backbone_fwd_graph_with_bert_branch': sum(a.elapsed_time(b) for t_, a, b in prof_timed if t_ == 'conv_fwd') / args.steps,
                                         'backbone_bwd_graph_with_wgrad_branch': sum(a.elapsed_time(b) for t_, a, b in prof_timed if t_ == 'conv_bwd') / args.steps,
                                         'graph_f2_after_backbone_to_criterion_inputs': sum(a.elapsed_time(b) for t_, a, b in prof_timed if t_ == 'graph_f2') / args.steps,
                                         'graph_b1_backward_down_to_backbone': sum(a.elapsed_time(b) for t_, a, b in prof_timed if t_ == 'graph_b1') / args.steps},
            'mfma_tflops': flops_step / (conv_ms * 1e-3) / 1e12, 'mfma_frac_of_2500': flops_step / (conv_ms * 1e-3) / 2.5e15}
    out = {'metric': 'images/sec/node (train step, 480x640, bs32/GPU)', 'value': world * args.batch * args.steps / elapsed,
           'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': elapsed / args.steps * 1e3, 'host_ms_per_step': host_s / args.steps * 1e3,
           'host_phases_ms': {k: v / max(host_prof.get('steps', 1), 1) * 1e3 for k, v in host_prof.items() if k != 'steps'}, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': 'GPV-1 (ResNet-50 + 6+6 DETR layers, 100 queries, RoI head, BERT-base, 3 co-attention, '
                                  '3 text-decoder layers, V=10000) CocoCaptioning-only train step, dropout 0.1, AdamW',
                      'global_batch': world * args.batch, 'image': '480x640', 'caption_tokens': 20,
                      'parallelism': f'dp{world}', 'final_loss': float(loss.detach())},
           'roofline': roof}
    out['graphs'] = {'enabled': bool(tr.graphs), 'graph_steps': tr.graph_steps, 'eager_steps': tr.eager_steps}
    if graphs_note:
        out['graphs']['error'] = graphs_note
    out['roofline_attention'] = attention_roofline(dev, args.batch)
    if comm is not None:
        out['comm'] = comm
    if soak is not None:
        out['soak'] = soak
    if world == 1 and not args.no_ragged and args.batch == BATCH:
        out['ragged'] = ragged_bench(model, tr, dev, rank)
    if world == 1 and not args.no_extra:
        out['extra'] = extra_configs(model, tr, dev, rank)
        out['extra']['input_pipeline_bs32'] = input_pipeline_bench(dev)
    if world == 1 and not args.no_decode:
        out['greedy_decode'] = greedy_decode_bench(model, dev)
    if world == 1 and not args.no_extra and args.batch == BATCH:
        out['extra']['precise_train_step_ms'] = precise_bench(model, tr, dev, rank)       # (last: it retires every captured graph)
    if world == 1 and not args.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline(model)
        except Exception as e:                                     # the baseline must never sink the bench line
            out['cpu_baseline'] = {'value': None, 'unit': 'images/sec', 'cores': os.cpu_count(), 'kind': 'port', 'sample': f'failed: {e}'}
    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
