"""Deterministic synthetic configs / weights / inputs shared by the golden generator
(tools/gen_golden.py, run in the build container against the real reference), the CPU
oracle tests and the GPU parity tests.  Nothing here reads /root/reference.

Weights are a pure function of (parameter name, shape): ``torch.Generator`` seeded with
crc32(name), so a fixture only has to store the name->shape manifest and the reference's
OUTPUTS; the inputs/weights are regenerated bit-identically wherever the test runs.
"""
import zlib

import torch


from gpv1_amd.synthetic import model_cfg, make_vocab      # noqa: E402,F401  (one definition: the bench workload's)


def small_cfg(dropout=0.0, **over):
    """Reduced problem used by the committed goldens: real widths/head sizes (dh 32/48/96), but
    10 queries, 2+2 DETR layers, 2 co-attention layers, 2 text-decoder layers, max_text_len 6."""
    c = model_cfg(
        max_text_len=6,
        detr={'num_queries': 10, 'num_encoder_layers': 2, 'num_decoder_layers': 2, 'dropout': dropout},
        text_decoder={'num_layers': 2, 'dropout': dropout},
        co_att={'num_layers': 2, 'attention_probs_dropout_prob': dropout, 'hidden_dropout_prob': dropout,
                'v_attention_probs_dropout_prob': dropout, 'v_hidden_dropout_prob': dropout})
    for k, v in over.items():
        c[k] = v
    return c


def synth_tensor(name, shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    shape = tuple(shape)
    if dtype in (torch.int64, torch.long):          # e.g. bert position_ids buffers
        n = 1
        for s in shape:
            n *= s
        return torch.arange(shape[-1]).expand(shape).clone() if shape else torch.zeros((), dtype=torch.long)
    last = name.rsplit('.', 1)[-1]
    if last == 'empty_weight':
        return None                                   # criterion buffer [1, eos_coef]: derived from cfg
    if last == 'running_var':
        return torch.rand(shape, generator=g) + 0.5
    if last == 'running_mean':
        return 0.1 * torch.randn(shape, generator=g)
    if last == 'num_batches_tracked':
        return torch.zeros(shape)
    if len(shape) <= 1:
        low = name.lower()
        if last == 'weight' and ('norm' in low or '.bn' in low or 'downsample.1' in low):
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        if last in ('vision_token', 'lang_token'):
            return 0.1 * torch.randn(shape, generator=g)
        return 0.02 * torch.randn(shape, generator=g)
    if name == 'pos_enc':
        return None                                   # deterministic buffer, kept from the model
    if len(shape) == 2 and ('vocab_embed' in name or 'embedding_layer' in name):
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) == 2 and ('embeddings.' in name or 'query_embed' in name):      # lookup tables (BERT, object queries)
        return 0.5 * torch.randn(shape, generator=g)
    if name == 'relevance_tokens':
        return 0.1 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    # He gain for the ReLU convs; the block-closing 1x1 (conv3) is damped so that the 16 residual adds do not
    # blow the activations up (c5 stays O(1-10) like a trained, FrozenBN-normalised ResNet; with gain 1.4 everywhere
    # c5 reached 1e4 and encoder attention scores 4e7, an ill-conditioned regime no real checkpoint is in)
    gain = 0.35 if 'conv3' in name else (1.4 if ('conv' in name or 'downsample.0' in name) else 1.0)
    return gain * torch.randn(shape, generator=g) / fan_in ** 0.5


def synth_state(manifest):
    """manifest: {name: [shape, dtype_str]} -> {name: tensor} (skips entries synth returns None for)."""
    out = {}
    for name, (shape, dt) in manifest.items():
        dtype = getattr(torch, dt.replace('torch.', ''))
        t = synth_tensor(name, shape, dtype)
        if t is not None:
            out[name] = t.to(dtype)
    # the two frozen copies of the vocabulary embedding are one tensor in the reference
    if 'answer_head.vocab_embed' in out and 'answer_input_embedings.embedding_layer.weight' in out:
        out['answer_input_embedings.embedding_layer.weight'] = out['answer_head.vocab_embed'].clone()
    return out


def synth_batch(B, H, W, Tl, V, seed=1234, pad_to=None):
    """images N(0,1) NCHW fp32, all-False padding mask (or ragged sizes padded, if pad_to),
    BERT token ids in [1000,30000) with a ragged attention mask."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(B, 3, H, W, generator=g)
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    if pad_to is not None:                       # ragged: image i is valid on [:h_i,:w_i]
        for i, (h, w) in enumerate(pad_to):
            mask[i, h:, :] = True
            mask[i, :, w:] = True
            images[i, :, h:, :] = 0
            images[i, :, :, w:] = 0
    ids = torch.randint(1000, 30000, (B, Tl), generator=g)
    attn = torch.ones(B, Tl, dtype=torch.long)
    for i in range(B):
        n = Tl - (i % 3)
        attn[i, n:] = 0
        ids[i, n:] = 0
    return images, mask, ids, attn


def synth_targets(B, V, S, seed=99, tasks=('CocoCaptioning', 'CocoVqa', 'CocoClassification', 'CocoDetection')):
    """one target dict per sample, cycling through the four task types
    (schema: datasets/coco_generic_dataset.py:98-114 + train_distr.py:410-412)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(B):
        task = tasks[i % len(tasks)]
        t = {'task': task}
        if task == 'CocoDetection':
            n = int(torch.randint(1, 5, (1,), generator=g))
            cxcy = 0.25 + 0.5 * torch.rand(n, 2, generator=g)
            wh = 0.05 + 0.3 * torch.rand(n, 2, generator=g)
            t['boxes'] = torch.cat((cxcy, wh), 1)
            t['labels'] = torch.zeros(n, dtype=torch.long)
        else:
            n = {'CocoCaptioning': S - 2, 'CocoVqa': 1, 'CocoClassification': 1}[task]
            t['answer'] = ' '.join(f'w{int(j)}' for j in torch.randint(0, V - 4, (n,), generator=g))
        out.append(t)
    return out
