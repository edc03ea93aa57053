"""Single image + query inference harness.

Reference: inference.py:24-83 (greedy), inference_beam_search.py:23-75 (beam), inference_util.py (image read).
Reproduced: checkpoint load with the ``module.`` prefix (inference.py:56-62), ImageNet normalisation (:64-67),
``model(images, queries, None)`` / ``forward_beam_search(images, queries, beam_size)``, boxes sorted by
``softmax(relevance)[..., 0]`` descending and cut to ``num_output_boxes`` (:33-37,77-78), answer = tokens up to the first
``__stop__`` / ``__pad__`` (:40-44), detokenised.
Out of scope: image decoding (skimage / cv2 in the reference).  The image is a ``.npy`` array -- HxWx3 uint8 RGB, HxWx3
float in [0,1], or an already normalised 3xHxW float32 tensor; nltk's Treebank detokenizer is replaced by its
punctuation-attachment rules (a join that glues ``, . ! ? ; : ' n't 's %`` to the previous token).

usage: python -m gpv1_amd.inference [--config some.yaml] ckpt=... inputs.img=img.npy inputs.query="what is this?"
                                      [beam_size=5] [num_output_boxes=5]
"""
import argparse
import os
import re
import sys

import numpy as np
import torch

from .config import from_dict, load_config
from .default_config import default_tree
from .gpv import GPV
from .misc import nested_tensor_from_tensor_list

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def load_model_state(model, path, map_location='cpu'):
    """inference.py:56-62: every key of the model must be present as ``module.<key>`` (a DDP checkpoint);
    unprefixed checkpoints are accepted as well."""
    loaded = torch.load(path, map_location=map_location, weights_only=False)['model']
    sd = model.state_dict()
    for k in sd:
        src = loaded.get('module.' + k, loaded.get(k))
        if src is None:
            raise KeyError(f'checkpoint {path} has no entry for {k}')
        sd[k] = src
    model.load_state_dict(sd)
    return model


def resize_image(img, size=(480, 640)):
    """``skimage.transform.resize(img, (imh, imw), anti_aliasing=True)`` of datasets/coco_datasets.py:174 (the reference's
    images are resized to ``task_configs.image_size`` = 480x640 before normalisation): float image in [0,1], Gaussian
    pre-filter with sigma = (scale - 1) / 2 on the axes that shrink, then order-1 interpolation on the pixel-area grid,
    mirrored borders.  Restated from scikit-image's published algorithm with scipy.ndimage (scikit-image is not in this
    image, so this step is NOT pinned against it); HxWx3 uint8 or float array in, float32 HxWx3 in [0,1] out."""
    from scipy import ndimage as ndi
    a = np.asarray(img)
    a = a.astype(np.float64) / (255.0 if a.dtype == np.uint8 else 1.0)
    factors = np.array([a.shape[0] / size[0], a.shape[1] / size[1], 1.0])
    sigma = np.maximum(0.0, (factors - 1.0) / 2.0)
    if sigma.max() > 0:
        a = ndi.gaussian_filter(a, sigma, mode='mirror')
    out = ndi.zoom(a, 1.0 / factors, order=1, mode='mirror', grid_mode=True)
    return np.clip(out, 0.0, 1.0).astype(np.float32)


def preprocess_image(img):
    """ToPILImage -> ToTensor -> Normalize of inference.py:64-67 for the array forms documented above"""
    if torch.is_tensor(img) and img.dim() == 3 and img.shape[0] == 3 and img.is_floating_point():
        return img.float()                                            # already CHW, normalised by the caller
    a = np.asarray(img)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f'expected an HxWx3 image, got {a.shape}')
    x = torch.from_numpy(a.astype(np.float32) / (255.0 if a.dtype == np.uint8 else 1.0)).permute(2, 0, 1)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (x - mean) / std


_ATTACH_LEFT = re.compile(r" (n't|'s|'re|'ve|'m|'ll|'d|[,.!?;:%)\]}])")


def detokenize(tokens):
    """TreebankWordDetokenizer, reduced to the cases a lower-cased word-level vocabulary produces"""
    s = ' '.join(tokens)
    s = _ATTACH_LEFT.sub(r'\1', s)
    return re.sub(r'([(\[{$]) ', r'\1', s)


def decode_outputs(outputs, model, num_output_boxes=None):
    """inference.py:24-49 (greedy) / inference_beam_search.py:23-50 (beam: best hypothesis = first in ``answers``)"""
    relevance = outputs['pred_relevance_logits'].float().softmax(-1).detach().cpu().numpy()
    pred_boxes = outputs['pred_boxes'].float().detach().cpu().numpy()
    if 'answers' in outputs:
        pred_answers = [hyps[0] for hyps in outputs['answers']]
    else:
        top1 = torch.topk(outputs['answer_logits'][-1].float(), k=1, dim=-1).indices[..., 0].detach().cpu().numpy()
        pred_answers = model.token_ids_to_words(top1)
    decoded = []
    for b in range(len(pred_answers)):
        order = sorted(range(relevance.shape[1]), key=lambda i: relevance[b, i, 0], reverse=True)   # stable, like the reference's sort on score
        scores = np.asarray([relevance[b, i, 0] for i in order], dtype=np.float32)
        boxes = np.asarray([pred_boxes[b, i] for i in order], dtype=np.float32)
        answer = []
        for token in pred_answers[b]:
            if token in ('__stop__', '__pad__'):
                break
            answer.append(token)
        if answer and answer[0] == '__cls__':
            answer = answer[1:]
        d = {'answer': detokenize(answer), 'boxes': boxes[:num_output_boxes], 'relevance': scores[:num_output_boxes]}
        if 'answer_probs' in outputs:
            d['answer_prob'] = outputs['answer_probs'][b][0]
        decoded.append(d)
    return decoded


@torch.no_grad()
def predict(model, images, queries, beam_size=None, num_output_boxes=None, size=None):
    """images: list of arrays/tensors (see preprocess_image); queries: list[str] or (ids, mask) tensors;
    size: (H, W) to resize HxWx3 arrays to first (the data loader's 480x640), None = as they are (inference.py)"""
    dev = model.vision_token.device
    if size is not None:
        images = [resize_image(i, size) if not torch.is_tensor(i) else i for i in images]
    imgs = nested_tensor_from_tensor_list([preprocess_image(i).to(dev) for i in images])
    if beam_size:
        out = model.forward_beam_search(imgs, queries, beam_size=beam_size)
    else:
        out = model(imgs, queries, None)
    return decode_outputs(out, model, num_output_boxes)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--config', default=None, help='YAML file (e.g. the reference configs/exp/gpv.yaml); default: gpv1_amd.default_config')
    ap.add_argument('overrides', nargs='*')
    args = ap.parse_args(argv)
    # ckpt= inputs.img= inputs.query= beam_size= are added keys
    cfg = load_config(args.config, args.overrides, strict=False) if args.config else from_dict(default_tree(), args.overrides, strict=False)
    model = GPV(cfg.model).cuda().eval()
    load_model_state(model, cfg.get('ckpt', cfg.eval.ckpt), map_location='cuda:0')
    img = np.load(cfg.inputs.img)
    pred = predict(model, [img], [cfg.inputs.query], beam_size=cfg.get('beam_size'), num_output_boxes=cfg.get('num_output_boxes', 5))[0]
    for k, v in pred.items():
        print('-' * 80)
        print(k)
        print('-' * 80)
        print(v)


if __name__ == '__main__':
    main(sys.argv[1:])
