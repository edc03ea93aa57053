"""Hydra-compatible configuration loading without Hydra (reference: ``@hydra.main(config_path=
'../../configs', config_name='exp/gpv')`` at exp/gpv/train_distr.py:478 and the YAML it names, configs/exp/gpv.yaml).

What the reference's drivers use from Hydra/OmegaConf and what is reproduced here:
  * one YAML file with ``${dotted.path}`` interpolation (configs/exp/gpv.yaml:3-6,15-16,28-30,...): resolved lazily,
    recursively, against the root;
  * ``key=value`` command-line overrides with dotted keys (scripts/train.sh passes ``exp_name=... training.freeze=True``):
    values are parsed as YAML scalars (``True``/``null``/``1e-4``/``[10,15]``), new keys may be added with ``+key=value``;
  * attribute access, ``.items()``, real bools (the model reads ``cfg.roi_head is True`` style flags).
The ``defaults:`` list (dataset / task groups) selects data-loading YAMLs, which are out of scope here (BASELINE uses
synthetic COCO-shaped tensors); it is kept in the tree untouched.
This loads the reference's own ``configs/exp/gpv.yaml``; the drivers' built-in tree is gpv1_amd/default_config.py.
"""
import re

import yaml

from .misc import AttrDict

_INTERP = re.compile(r'\$\{([^${}]+)\}')

# PyYAML's default resolver reads "1e-4" as a string (YAML 1.1 floats need a dot); OmegaConf reads it as a float.
_FLOAT = re.compile(r'^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$')


def _scalar(text):
    v = yaml.safe_load(text)
    if isinstance(v, str) and _FLOAT.match(v):
        return float(v)
    return v


def _fix_floats(node):
    if isinstance(node, dict):
        return {k: _fix_floats(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_fix_floats(v) for v in node]
    if isinstance(node, str) and _FLOAT.match(node):
        return float(node)
    return node


def _get(root, dotted):
    cur = root
    for part in dotted.split('.'):
        if isinstance(cur, list):
            cur = cur[int(part)]
        else:
            cur = cur[part]
    return cur


def _resolve(node, root, depth=0):
    if depth > 32:
        raise ValueError('config: interpolation cycle')
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node)
        if m:                                            # whole value is one reference: keep the referenced type
            return _resolve(_get(root, m.group(1).strip()), root, depth + 1)
        if _INTERP.search(node):
            return _INTERP.sub(lambda mm: str(_resolve(_get(root, mm.group(1).strip()), root, depth + 1)), node)
    return node


def apply_overrides(tree, overrides, strict=True):
    """``a.b.c=value`` (existing key), ``+a.b.c=value`` (new key; every key when not strict), in order."""
    for ov in overrides or ():
        if '=' not in ov:
            raise ValueError(f'config override {ov!r}: expected key=value')
        key, text = ov.split('=', 1)
        add = key.startswith('+') or not strict
        key = key.lstrip('+')
        parts = key.split('.')
        cur = tree
        for part in parts[:-1]:
            if part not in cur or not isinstance(cur[part], dict):
                if not add:
                    raise KeyError(f'config override {ov!r}: no such group {part!r} (use +{key}=... to add)')
                cur[part] = {}
            cur = cur[part]
        if parts[-1] not in cur and not add:
            raise KeyError(f'config override {ov!r}: no such key (use +{key}=... to add)')
        cur[parts[-1]] = _scalar(text)
    return tree


def load_config(path, overrides=(), strict=True):
    """YAML file + overrides -> resolved AttrDict tree."""
    with open(path) as f:
        tree = _fix_floats(yaml.safe_load(f) or {})
    apply_overrides(tree, overrides, strict)
    return AttrDict.wrap(_resolve(tree, tree))


def from_dict(tree, overrides=(), strict=True):
    import copy
    tree = copy.deepcopy(dict(tree))
    apply_overrides(tree, overrides, strict)
    return AttrDict.wrap(_resolve(tree, tree))
