"""Hydra-compatible configuration loading without Hydra (reference: ``@hydra.main(config_path=
'../../configs', config_name='exp/gpv')`` at exp/gpv/train_distr.py:478 and the YAML it names, configs/exp/gpv.yaml).

What the reference's drivers use from Hydra/OmegaConf and what is reproduced here:
  * one YAML file with ``${dotted.path}`` interpolation (configs/exp/gpv.yaml:3-6,15-16,28-30,...): resolved lazily,
    recursively, against the root;
  * ``key=value`` command-line overrides with dotted keys (scripts/train.sh passes ``exp_name=... training.freeze=True``):
    values are parsed as YAML scalars (``True``/``null``/``1e-4``/``[10,15]``), new keys may be added with ``+key=value``;
  * attribute access, ``.items()``, real bools (the model reads ``cfg.roi_head is True`` style flags).
  * the ``defaults:`` list (configs/exp/gpv.yaml:23-25 ``- task: coco_learning_tasks`` / ``- learning_datasets: vqa``): every
    entry ``group: name`` names ``<config root>/<group>/<name>.yaml``, merged into the tree at the package its first line
    declares (``# @package _group_`` -> under ``group``, ``# @package task_configs`` -> under that key, ``_global_`` -> root);
    an override whose key is a defaults group -- ``learning_datasets=all`` as scripts/train.sh passes it (:14-34) -- selects
    the file instead of assigning a value.  The primary file's own keys win over a defaults file's (Hydra 1.0 ordering).
This loads the reference's own ``configs/exp/gpv.yaml``; the drivers' built-in tree is gpv1_amd/default_config.py, whose
``learning_datasets`` group options are the reference's eleven task mixes (configs/learning_datasets/*.yaml).
"""
import os
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


_PACKAGE = re.compile(r'^#\s*@package\s+(\S+)')


def _merge_under(tree, package, sub):
    """merge dict `sub` into `tree` at dotted `package` ('' = root); existing keys of `tree` win"""
    cur = tree
    for part in [p for p in package.split('.') if p]:
        cur = cur.setdefault(part, {})
    for k, v in sub.items():
        if isinstance(v, dict) and isinstance(cur.get(k), dict):
            _merge_under(cur, k, v)
        else:
            cur.setdefault(k, v)


def _group_file(root_dirs, group, name):
    for d in root_dirs:
        p = os.path.join(d, group, f'{name}.yaml')
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f'config: defaults entry {group}: {name} -- no {group}/{name}.yaml under {root_dirs}')


def _compose_defaults(tree, root_dirs, selected):
    """Hydra ``defaults:`` list -> the named group files merged into `tree` (see the module docstring)"""
    for entry in tree.get('defaults') or []:
        if not isinstance(entry, dict):
            continue                                     # (a bare string entry names another primary config: not used by the reference)
        for group, name in entry.items():
            name = selected.get(group, name)
            if name is None:
                continue
            path = _group_file(root_dirs, group, name)
            with open(path) as f:
                text = f.read()
            m = _PACKAGE.match(text.lstrip())
            package = m.group(1) if m else group
            package = {'_group_': group.replace('/', '.'), '_global_': ''}.get(package, package)
            _merge_under(tree, package, _fix_floats(yaml.safe_load(text) or {}))
    return tree


def _split_group_overrides(tree, overrides, groups=None):
    """overrides whose key is a defaults group (``learning_datasets=all``) select a file / a built-in option, the rest assign"""
    names = set(groups or ())
    for entry in tree.get('defaults') or []:
        if isinstance(entry, dict):
            names.update(entry)
    selected, rest = {}, []
    for ov in overrides or ():
        key, _, text = ov.partition('=')
        if key.lstrip('+') in names and '=' in ov:
            selected[key.lstrip('+')] = _scalar(text)
        else:
            rest.append(ov)
    return selected, rest


def load_config(path, overrides=(), strict=True):
    """YAML file (+ its ``defaults:`` groups) + overrides -> resolved AttrDict tree."""
    with open(path) as f:
        tree = _fix_floats(yaml.safe_load(f) or {})
    here = os.path.dirname(os.path.abspath(path))
    root_dirs = [here, os.path.dirname(here)]            # config_name 'exp/gpv' under config_path 'configs': groups sit beside `exp/`
    selected, rest = _split_group_overrides(tree, overrides)
    _compose_defaults(tree, root_dirs, selected)
    apply_overrides(tree, rest, strict)
    return AttrDict.wrap(_resolve(tree, tree))


def from_dict(tree, overrides=(), strict=True, group_options=None):
    """group_options: {group: {name: subtree}} -- the built-in counterpart of the defaults group files
    (default_config.GROUP_OPTIONS): ``learning_datasets=cap`` replaces tree['learning_datasets'] by that option"""
    import copy
    tree = copy.deepcopy(dict(tree))
    group_options = group_options or {}
    selected, rest = _split_group_overrides(tree, overrides, groups=group_options)
    for group, name in selected.items():
        if group not in group_options or name not in group_options[group]:
            raise KeyError(f'config: no option {name!r} for group {group!r} (have {sorted(group_options.get(group, {}))})')
        tree[group] = copy.deepcopy(group_options[group][name])
    apply_overrides(tree, rest, strict)
    return AttrDict.wrap(_resolve(tree, tree))
