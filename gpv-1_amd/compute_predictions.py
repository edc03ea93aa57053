"""Batched prediction files for the reference's evaluators.

Reference: exp/gpv/compute_predictions.py -- ``make_predictions`` :30-85, ``create_coco_vocab_mask`` :88-109.
Reproduced:
  * the classification vocabulary mask (:88-109): -10000 on every vocabulary entry, 0 on the word tokens of the class
    names (optionally of their synonyms) that are in the vocabulary and on ``__stop__`` / ``__pad__``; passed to
    ``model(imgs, queries, None, vocab_mask=...)`` only for ``CocoClassification`` (:31-34);
  * the loop (:43-80): greedy forward, boxes sorted by ``softmax(relevance)[..., 0]`` descending (stable, all
    ``num_queries`` boxes kept), answer = top-1 tokens up to the first ``__stop__`` / ``__pad__``, detokenised;
    ``num_eval_batches`` stops AFTER batch index ``num_eval_batches`` like the reference's ``i > num_eval_batches``;
  * the two files (:36-38,72-85): ``<task>_<data_split>_<subset>_predictions.json`` = {sample_id: {'answer': str}} and the
    boxes file with one group per sample id holding ``boxes`` [Q,4] float32 and ``relevance`` [Q] float32.
Differences: the boxes file is HDF5 (``<task>_<subset>_boxes.h5py``, the reference's name) when ``h5py`` is importable and
an ``.npz`` with the keys ``'<sample_id>/boxes'`` / ``'<sample_id>/relevance'`` otherwise (this image has no h5py;
``boxes_to_h5py`` converts on a machine that has); dataset construction, image decoding and the evaluators themselves are
out of scope -- ``batches`` is any iterable of ``(NestedTensor | list of CHW tensors, queries, sample_ids)``.
The 80 class names are the public COCO detection categories (the reference reads them from data/coco/synonyms.py).
"""
import json
import os
import re

import numpy as np
import torch

from .inference import decode_outputs
from .misc import NestedTensor, nested_tensor_from_tensor_list

TASK_TO_ID = {'CocoVqa': 'question_id', 'CocoCaptioning': 'cap_id', 'CocoClassification': 'id', 'CocoDetection': 'id',
              'RefCocop': 'sent_id'}                                   # exp/gpv/evaluators.py task_to_id

COCO_CLASSES = (
    'person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light', 'fire hydrant',
    'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra',
    'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite',
    'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle', 'wine glass', 'cup', 'fork',
    'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut',
    'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard',
    'cell phone', 'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear',
    'hair drier', 'toothbrush')

COCO_CLASSES = tuple(sorted(COCO_CLASSES))      # the reference iterates data/coco/synonyms.py's table, whose keys are in alphabetical order

_WORD = re.compile(r"\w+|[^\w\s]")


def word_tokenize(text):
    """nltk.word_tokenize on class names: words and single punctuation marks"""
    return _WORD.findall(text)


def create_vocab_mask(model, classes=COCO_CLASSES, synonyms=None, use_syns=False):
    """compute_predictions.py:88-109 -> (tokens, float32 mask [V])"""
    mask = -10000.0 * np.ones([len(model.vocab)], dtype=np.float32)
    tokens = []
    for cls in classes:
        names = [cls]
        if use_syns and synonyms is not None:
            if cls not in synonyms:
                continue
            names = synonyms[cls]
        for name in names:
            for token in word_tokenize(name):
                if token in model.word_to_idx:
                    mask[model.word_to_idx[token]] = 0
                    tokens.append(token)
    for token in ('__stop__', '__pad__'):
        mask[model.word_to_idx[token]] = 0
        tokens.append(token)
    return tokens, mask


class BoxesWriter:
    """one group per sample id: 'boxes' [Q,4] float32, 'relevance' [Q] float32"""

    def __init__(self, path_h5py):
        try:
            import h5py
            self.h5, self.path = h5py.File(path_h5py, 'w'), path_h5py
            self.arrays = None
        except ImportError:
            self.h5, self.path = None, os.path.splitext(path_h5py)[0] + '.npz'
            self.arrays = {}

    def add(self, sample_id, boxes, relevance):
        if self.h5 is not None:
            grp = self.h5.create_group(str(sample_id))
            grp.create_dataset('boxes', data=boxes)
            grp.create_dataset('relevance', data=relevance)
        else:
            self.arrays[f'{sample_id}/boxes'] = boxes
            self.arrays[f'{sample_id}/relevance'] = relevance

    def close(self):
        if self.h5 is not None:
            self.h5.close()
        else:
            np.savez(self.path, **self.arrays)
        return self.path


def boxes_to_h5py(npz_path, h5py_path):
    """the reference's boxes layout from the .npz fallback (needs h5py)"""
    import h5py
    z = np.load(npz_path)
    with h5py.File(h5py_path, 'w') as f:
        for sid in sorted({k.rsplit('/', 1)[0] for k in z.files}):
            grp = f.create_group(sid)
            grp.create_dataset('boxes', data=z[sid + '/boxes'])
            grp.create_dataset('relevance', data=z[sid + '/relevance'])


@torch.no_grad()
def make_predictions(model, batches, eval_dir, task, subset='val', data_split='original_split', num_eval_batches=None,
                     vocab_mask=None):
    """compute_predictions.py:30-85; returns (predictions dict, json path, boxes path)"""
    dev = model.vision_token.device
    if vocab_mask is None and task == 'CocoClassification':
        vocab_mask = create_vocab_mask(model)[1]
    if vocab_mask is not None:
        vocab_mask = torch.as_tensor(vocab_mask, dtype=torch.float32, device=dev)
    os.makedirs(eval_dir, exist_ok=True)
    boxes_file = BoxesWriter(os.path.join(eval_dir, f'{task}_{subset}_boxes.h5py'))
    predictions = {}
    model.eval()
    for i, (imgs, queries, sample_ids) in enumerate(batches):
        if num_eval_batches is not None and i > num_eval_batches:
            break
        if not isinstance(imgs, NestedTensor):
            imgs = nested_tensor_from_tensor_list([x.to(dev) for x in imgs])
        outputs = model(imgs, queries, None, vocab_mask=vocab_mask)
        for sid, d in zip(sample_ids, decode_outputs(outputs, model, num_output_boxes=None)):
            predictions[sid] = {'answer': d['answer']}
            boxes_file.add(sid, d['boxes'], d['relevance'])
    boxes_path = boxes_file.close()
    json_path = os.path.join(eval_dir, f'{task}_{data_split}_{subset}_predictions.json')
    with open(json_path, 'w') as f:
        json.dump(predictions, f, indent=4, sort_keys=True)            # utils/io.py dump_json_object
    return predictions, json_path, boxes_path
