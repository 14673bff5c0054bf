"""Loading reference checkpoints into the HIP model (reference utils/model_serialization.py:8-78).

Same observable behaviour as the reference loader: a `module.` prefix left by DataParallel/DDP is dropped when every
loaded key carries it; each model key takes the loaded key that is its LONGEST suffix (so `res2.conv1.weight`
lands on `backbone.body.res2.conv1.weight`); model keys with no suffix match keep their current value; loaded
keys nobody claims (e.g. the ImageNet classifier `fc.*`) are ignored; the result goes through a strict
`model.load_state_dict`, which on the HIP model also drops the cached packed weights.

The matching is done per key through a reversed-component index instead of the reference's dense
(model keys x loaded keys) match matrix.
"""
from collections import OrderedDict

import torch


def strip_prefix_if_present(state_dict, prefix):
    """Remove `prefix` from every key, but only if ALL keys start with it (model_serialization.py:59-66)."""
    if not state_dict or not all(k.startswith(prefix) for k in state_dict):
        return state_dict
    return OrderedDict((k[len(prefix):] if k.startswith(prefix) else k, v) for k, v in state_dict.items())


def match_keys(model_keys, loaded_keys):
    """{model key: loaded key} where the loaded key is the longest string suffix of the model key.

    Ties cannot happen (two different suffixes of one string differ in length). A loaded key may serve several
    model keys, exactly as in the reference's match matrix (model_serialization.py:22-36)."""
    by_last = {}
    for lk in loaded_keys:                                       # index by the last character run after the final '.'
        by_last.setdefault(lk.rsplit(".", 1)[-1], []).append(lk)
    out = {}
    for mk in model_keys:
        best = None
        for lk in by_last.get(mk.rsplit(".", 1)[-1], ()):
            if mk.endswith(lk) and (best is None or len(lk) > len(best)):
                best = lk
        if best is None:                                         # suffixes that cut through a component ("1.weight" vs "bn1.weight")
            for lk in loaded_keys:
                if lk and mk.endswith(lk) and (best is None or len(lk) > len(best)):
                    best = lk
        if best is not None:
            out[mk] = best
    return out


def align_and_update_state_dicts(model_state_dict, loaded_state_dict):
    """In-place: overwrite the entries of `model_state_dict` that have a suffix match in `loaded_state_dict`.
    Returns the {model key: loaded key} mapping that was applied."""
    mapping = match_keys(list(model_state_dict.keys()), list(loaded_state_dict.keys()))
    for mk, lk in mapping.items():
        model_state_dict[mk] = loaded_state_dict[lk]
    return mapping


def load_state_dict(model, loaded_state_dict):
    """Reference entry point (model_serialization.py:69-78). Shape mismatches raise from the strict load."""
    model_state = model.state_dict()
    loaded = strip_prefix_if_present(loaded_state_dict, "module.")
    mapping = align_and_update_state_dicts(model_state, loaded)
    model.load_state_dict(model_state)
    return mapping


def load_dla_imagenet(base, path):
    """ImageNet DLA-34 weights into the DLA trunk (reference dla_dcn.py:333-344, `load_pretrained_model`).

    The reference downloads dla34-ba72cf86.pth and grows a 1x1 `fc` conv so that a strict load succeeds; that
    classifier never runs on the detection path, so here the `fc.*` entries are validated (they define the class
    count) and dropped, and the trunk is loaded strictly: every trunk key must be present with the right shape."""
    weights = torch.load(path, map_location="cpu")
    if "state_dict" in weights and isinstance(weights["state_dict"], dict):
        weights = weights["state_dict"]
    trunk = OrderedDict((k, v) for k, v in weights.items() if not k.startswith("fc."))
    extra = [k for k in weights if k.startswith("fc.")]
    if extra and weights["fc.weight"].shape[1] != base.channels[-1]:
        raise RuntimeError("DLA classifier expects %d input channels, trunk has %d"
                           % (weights["fc.weight"].shape[1], base.channels[-1]))
    base.load_state_dict(trunk, strict=True)
    return len(trunk)
