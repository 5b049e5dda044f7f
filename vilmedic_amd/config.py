"""YAML config loading with the reference's semantics, without OmegaConf (absent in this image):
``includes`` resolved relative to the config's directory, deep merge, ``a.b.c=value`` command-line overrides and
numeric-string coercion.  ref: bin/utils.py:34-148."""
import copy
import os
import re

import yaml


class Cfg(dict):
    """dict with attribute access, ``.pop``/``.get``/``**`` like an OmegaConf DictConfig sub-tree."""

    def __getattr__(self, k):
        if k.startswith("_"):
            raise AttributeError(k)
        return self.get(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


_num = re.compile(r"^[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)$")


def _coerce(v):
    if isinstance(v, str):
        if _num.match(v):
            return float(v) if any(c in v for c in ".eE") else int(v)
        if v.lower() in ("true", "false"):
            return v.lower() == "true"
        if v.lower() in ("null", "none"):
            return None
    return v


def wrap(x):
    if isinstance(x, dict):
        return Cfg({k: wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [wrap(v) for v in x]
    return _coerce(x)


def merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            merge(a[k], v)
        else:
            a[k] = v
    return a


def load_yaml(path):
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    base = {}
    for inc in raw.pop("includes", []) or []:
        # bin/utils.py:113-115: the path as written (relative to the working directory), else relative to the including file;
        # last resort (a config tree used from another working directory): the include's basename next to the including file
        cands = [inc, os.path.join(os.path.dirname(path), inc), os.path.join(os.path.dirname(path), os.path.basename(inc))]
        inc_path = next((c for c in cands if os.path.exists(c)), cands[1])
        merge(base, load_yaml(inc_path))
    return merge(base, raw)


def apply_dotlist(cfg, overrides):
    for item in overrides:
        key, _, val = item.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = yaml.safe_load(val) if val != "" else None
    return cfg


def get_config(path, overrides=()):
    return wrap(apply_dotlist(load_yaml(path), overrides))


def to_container(x):
    """plain dict / list tree (what is pickled into checkpoints)"""
    if isinstance(x, dict):
        return {k: to_container(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_container(v) for v in x]
    return x


def executor_view(config, name):
    """top-level keys flattened into the executor's own sub-tree (bin/utils.py:140-148)."""
    view = Cfg({k: v for k, v in config.items() if k not in ("trainor", "validator", "ensemblor")})
    view.update(copy.deepcopy(config.get(name) or {}))
    return view
