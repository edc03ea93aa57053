"""Synthetic workload of the benchmark / smoke entry points: the reference's model + loss configuration tree
(configs/exp/gpv.yaml:27-117) as a plain dict, and a synthetic answer vocabulary with the reference's layout
(exp/gpv/compute_vocab_bert.py:35-41: the four specials are the LAST entries).  No dataset, no network: BASELINE.json's
metric is quoted on synthetic 480x640 COCO-shaped images + random token sequences."""
import copy


def model_cfg(**over):
    """Mirror of configs/exp/gpv.yaml:27-117 (model + losses) as a plain dict; keyword arguments are merged over it."""
    cfg = {
        'pretr_detr': None, 'vocab': None, 'vocab_embed': None,
        'max_pos_enc_len': 30, 'max_text_len': 20, 'answer_head': None,
        'answering_type': 'generation', 'hidden_dim': 768, 'roi_head': True,
        'relevance_conditioning': True,
        'detr': {'num_queries': 100, 'num_classes': 1, 'hidden_dim': 256, 'nheads': 8,
                 'num_encoder_layers': 6, 'num_decoder_layers': 6, 'backbone': 'resnet50',
                 'lr_backbone': 1e-5, 'position_embedding': 'sine', 'masks': False,
                 'dilation': False, 'dropout': 0.1, 'dim_feedforward': 2048, 'pre_norm': False,
                 'aux_loss': False, 'frozenbatchnorm': True, 'last_layer_only': True},
        'detr_joiner': {'detr_dim': 2304, 'out_dim': 768},
        'bert_joiner': {'bert_dim': 768, 'out_dim': 768},
        'text_decoder': {'hidden_dim': 768, 'dropout': 0.1, 'nheads': 8, 'pos_enc': False,
                         'num_layers': 3},
        'co_att': {'visualization': False, 'bi_num_attention_heads': 16, 'bi_hidden_size': 768,
                   'hidden_size': 768, 'intermediate_size': 3072, 'output_size': 768,
                   'attention_probs_dropout_prob': 0.1, 'hidden_dropout_prob': 0.1,
                   'hidden_act': 'gelu', 'v_hidden_size': 768, 'v_intermediate_size': 3072,
                   'v_output_size': 768, 'v_attention_probs_dropout_prob': 0.1,
                   'v_hidden_dropout_prob': 0.1, 'v_hidden_act': 'gelu', 'num_layers': 3},
        'losses': {
            'CaptionLoss': {'name': 'caption_criterion', 'pad_idx': None, 'loss_wts': {'loss_caption': 5e-2}},
            'VqaLoss': {'name': 'vqa_criterion', 'pad_idx': None, 'loss_wts': {'loss_vqa': 1}},
            'ClsLoss': {'name': 'cls_criterion', 'pad_idx': None, 'loss_wts': {'loss_cls': 1}},
            'Localization': {'name': 'localization_criterion',
                             'cost_wts': {'ce': 1, 'bbox': 5, 'giou': 2},
                             'loss_wts': {'loss_ce': 1, 'loss_bbox': 5, 'loss_giou': 2},
                             'eos_coef': 0.1, 'num_classes': 1},
        },
    }
    cfg = copy.deepcopy(cfg)

    def merge(d, o):
        for k, v in o.items():
            if isinstance(v, dict) and isinstance(d.get(k), dict):
                merge(d[k], v)
            else:
                d[k] = v
    merge(cfg, over)
    return cfg


def make_vocab(V):
    """specials are the LAST four entries (exp/gpv/compute_vocab_bert.py:35-41)."""
    words = [f'w{i}' for i in range(V - 4)]
    return words + ['__pad__', '__cls__', '__stop__', '__unk__']


def write_wordpiece_vocab(path, n_words=400):
    """a WordPiece vocabulary file in bert-base-uncased's layout (specials first) with n_words synthetic whole-word tokens
    (`q0`, `q1`, ...): bench.py's string-query mode needs a tokenizer, the real vocab.txt is not available offline"""
    toks = ['[PAD]'] + [f'[unused{i}]' for i in range(99)] + ['[UNK]', '[CLS]', '[SEP]', '[MASK]']
    toks += list('abcdefghijklmnopqrstuvwxyz0123456789') + ['##' + c for c in 'abcdefghijklmnopqrstuvwxyz0123456789']
    words = [f'q{i}' for i in range(n_words)]
    with open(path, 'w') as f:
        f.write('\n'.join(toks + words) + '\n')
    return words
