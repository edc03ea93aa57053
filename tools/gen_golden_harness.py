"""Golden fixtures for the inference / prediction harness from the REFERENCE's own functions.  Build container only:

    python tools/gen_golden_harness.py        ->  tests/golden/harness.npz, tests/golden/harness.json

The reference's harness modules (inference.py, exp/gpv/compute_predictions.py) import hydra, skimage, h5py, imagesize, nltk and
the dataset code at module level, none of which the image has; what is pinned here are three FUNCTIONS of theirs, executed from
the source files where they lie (parsed out of /root/reference with `ast`, compiled and run here -- nothing is copied into the
repository):

  * inference.decode_outputs                                  (inference.py:24-49)     boxes sorted by softmax(relevance)[:, :, 0],
                                                                                       top-1 tokens cut at __stop__ / __pad__
  * compute_predictions.create_coco_vocab_mask                (compute_predictions.py:88-109) with the real data/coco/synonyms.py table
  * compute_predictions.make_predictions                      (:30-85)                 predictions JSON + boxes HDF5 group layout

run on the small synthetic fixture through the real reference model (tools/ref_harness.py).  Stand-ins, stated in the fixture:
nltk's word_tokenize / TreebankWordDetokenizer (nltk is absent: the oracle's regex split and the product's punctuation-attachment
join -- detokenisation therefore stays "parity unpinned"), h5py (a recorder with the h5py calls the function makes), tqdm (identity)."""
import ast
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import ref_harness as RH                                     # noqa: E402
from tests import synth                                       # noqa: E402
import gen_golden as GG                                       # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = RH.REF


def ref_functions(path, names, namespace):
    """compile the named top-level functions of a reference source file into `namespace` (executed in place, not copied)"""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, os.path.join(REF, path), 'exec'), namespace)
    return namespace


def ref_literal(path, name):
    """a module-level literal (dict) of a reference source file"""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    for n in tree.body:
        if isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id == name for t in n.targets):
            return ast.literal_eval(n.value)
    raise KeyError(name)


class H5Recorder:
    """records the h5py calls make_predictions makes: File(path, 'w') -> create_group(name) -> create_dataset(name, data=...)"""
    files = {}

    class _Group:
        def __init__(self, store):
            self.store = store

        def create_dataset(self, name, data=None):
            self.store[name] = np.asarray(data)

    class File:
        def __init__(self, path, mode):
            assert mode == 'w'
            self.path, self.groups = path, {}
            H5Recorder.files[path] = self.groups

        def create_group(self, name):
            assert name not in self.groups
            self.groups[name] = {}
            return H5Recorder._Group(self.groups[name])

        def close(self):
            pass


def main():
    from gpv1_amd.inference import detokenize
    from oracle.gpv_oracle import simple_word_tokenize
    torch.set_num_threads(8)
    V, B, H, W, Tl = 40, 4, 96, 128, 5
    cfg = synth.small_cfg(dropout=0.0)
    G, model, manifest, vocab = GG.build_reference(cfg, V, bert_layers=2)
    images, mask, ids, attn = synth.synth_batch(B, H, W, Tl, V, pad_to=[(96, 128), (96, 128), (64, 96), (96, 100)])
    model.eval()

    class Detok:
        def detokenize(self, toks):
            return detokenize(list(toks))
    io_mod = types.SimpleNamespace(dump_json_object=lambda obj, path: json.dump(obj, open(path, 'w'), indent=4, sort_keys=True))
    synonyms = ref_literal('data/coco/synonyms.py', 'SYNONYMS')
    task_to_id = ref_literal('exp/gpv/evaluators.py', 'task_to_id')
    ns = {'np': np, 'torch': torch, 'os': os, 'TreebankWordDetokenizer': Detok, 'word_tokenize': simple_word_tokenize,
          'SYNONYMS': synonyms, 'h5py': H5Recorder, 'tqdm': (lambda x: x), 'io': io_mod,
          'evaluators': types.SimpleNamespace(task_to_id=task_to_id)}
    ref_functions('inference.py', ['decode_outputs'], ns)
    decode_outputs = ns['decode_outputs']
    ref_functions('exp/gpv/compute_predictions.py', ['make_predictions', 'create_coco_vocab_mask'], ns)

    out, meta = {}, {'stand_ins': ['nltk.word_tokenize -> oracle.simple_word_tokenize', 'TreebankWordDetokenizer -> gpv1_amd.inference.detokenize',
                                   'h5py -> call recorder', 'tqdm -> identity']}
    with torch.no_grad():
        o = model(GG.nested(images, mask), (ids, attn), None)
        dec = decode_outputs(o, model)
    out.update({'dec_in_relevance_logits': o['pred_relevance_logits'], 'dec_in_boxes': o['pred_boxes'],
                'dec_in_top1': o['answer_logits'][-1].topk(1, -1).indices[..., 0]})
    out['dec_boxes'] = np.stack([d['boxes'] for d in dec])
    out['dec_relevance'] = np.stack([d['relevance'] for d in dec])
    meta['dec_answers'] = [d['answer'] for d in dec]
    # a tie case for the sort: two queries with identical relevance logits and distinct boxes (stable: first stays first)
    o2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}
    o2['pred_relevance_logits'][0, 7] = o2['pred_relevance_logits'][0, 2]
    dec2 = decode_outputs(o2, model)
    out['tie_in_relevance_logits'] = o2['pred_relevance_logits']
    out['tie_boxes0'] = dec2[0]['boxes']

    # ---- vocabulary mask: the reference function on a vocabulary that holds some class / synonym words ----
    words = []
    for cls in list(synonyms)[:40]:
        for syn in synonyms[cls][:3]:
            for t in simple_word_tokenize(syn):
                if t not in words:
                    words.append(t)
    mvocab = words[::2] + [f'filler{i}' for i in range(30)] + ['__pad__', '__cls__', '__stop__', '__unk__']
    fake = types.SimpleNamespace(vocab=mvocab, word_to_idx={w: i for i, w in enumerate(mvocab)})
    toks, m = ns['create_coco_vocab_mask'](fake, use_syns=False)
    toks_s, m_s = ns['create_coco_vocab_mask'](fake, use_syns=True)
    sub = {c: synonyms[c] for c in list(synonyms)[:40]}
    meta.update({'mask_vocab': mvocab, 'mask_classes': list(synonyms), 'mask_tokens': toks, 'mask_tokens_syn': toks_s,
                 'mask_synonyms_subset': sub})
    out['mask'] = m
    # use_syns=True restricted to the first 40 classes' synonyms is an INPUT of the test: the expected mask for that input
    ns_sub = dict(ns, SYNONYMS=sub)
    ref_functions('exp/gpv/compute_predictions.py', ['create_coco_vocab_mask'], ns_sub)
    toks_sub, m_sub = ns_sub['create_coco_vocab_mask'](fake, use_syns=True)
    meta['mask_tokens_syn_subset'] = toks_sub
    out['mask_syn_subset'] = m_sub

    # ---- make_predictions: two batches of two samples, classification (vocabulary mask on) and VQA ----
    from utils.detr_misc import NestedTensor
    for task in ('CocoClassification', 'CocoVqa'):
        H5Recorder.files.clear()
        tmp = tempfile.mkdtemp()
        os.makedirs(os.path.join(tmp, 'eval'))
        idn = task_to_id[task]
        samples = [{idn: 100 + 7 * i} for i in range(B)]
        loader = [(NestedTensor(images[i:i + 2], mask[i:i + 2]), (ids[i:i + 2], attn[i:i + 2]), [{}, {}]) for i in (0, 2)]
        pcfg = RH.AttrDict.wrap({'eval': {'task': task, 'subset': 'val', 'num_eval_batches': None}, 'exp_dir': tmp, 'gpu': 'cpu',
                                 'task_configs': {'data_split': 'original_split'}})
        fake_model = model
        if task == 'CocoClassification':
            # the mask is built from model.vocab: the small fixture's vocabulary has no class words -> only __stop__ / __pad__ stay open
            pass
        with torch.no_grad():
            ns['make_predictions'](fake_model, loader, samples, pcfg)
        (h5path, groups), = H5Recorder.files.items()
        jfiles = [f for f in os.listdir(os.path.join(tmp, 'eval')) if f.endswith('.json')]
        assert len(jfiles) == 1
        preds = json.load(open(os.path.join(tmp, 'eval', jfiles[0])))
        meta['pred_' + task] = {'json_name': jfiles[0], 'h5_name': os.path.basename(h5path), 'predictions': preds,
                                'groups': {g: {k: [list(v.shape), str(v.dtype)] for k, v in d.items()} for g, d in groups.items()},
                                'sample_ids': [s[idn] for s in samples], 'id_field': idn}
        for gname, d in groups.items():
            for k, v in d.items():
                out[f'pred_{task}/{gname}/{k}'] = v
    # ---- the task mixes `learning_datasets=<name>` selects (configs/learning_datasets/*.yaml, scripts/train.sh:14-34) ----
    import glob
    import yaml
    meta['learning_datasets'] = {os.path.basename(f)[:-5]: [[k, v] for k, v in yaml.safe_load(open(f)).items()]
                                 for f in sorted(glob.glob(os.path.join(REF, 'configs', 'learning_datasets', '*.yaml')))}
    np.savez_compressed(os.path.join(GOLD, 'harness.npz'), **GG.to_np(out))
    json.dump(meta, open(os.path.join(GOLD, 'harness.json'), 'w'), indent=1, sort_keys=True)
    print('answers', meta['dec_answers'])
    print('mask open entries', int((m == 0).sum()), 'with synonyms', int((m_s == 0).sum()), 'files', meta['pred_CocoVqa']['json_name'], meta['pred_CocoVqa']['h5_name'])


if __name__ == '__main__':
    main()
