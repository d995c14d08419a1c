"""Checkpoint / config helpers of the reference's tools/utils.py that the inference path touches: `DotDict`, `load_config`,
`traverse_dir` and `load_model` (resume from the highest-numbered `<name>_<step>.pt`, reference tools/utils.py:5-88)."""
import os

import torch
import yaml


class DotDict(dict):
    def __getattr__(*args):
        val = dict.get(*args)
        return DotDict(val) if type(val) is dict else val

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def load_config(path_config):
    with open(path_config, "r") as config:
        return DotDict(yaml.safe_load(config))


def traverse_dir(root_dir, extensions, amount=None, str_include=None, str_exclude=None, is_pure=False, is_sort=False, is_ext=True,
                 second_root_dir=None):
    """Recursive listing of the files under root_dir with one of `extensions` (reference tools/utils.py:5-38): paths relative
    to the top directory when is_pure, extension stripped when not is_ext, optional substring filters and a count limit."""
    top = root_dir if second_root_dir is None else second_root_dir
    found = []
    if not os.path.exists(root_dir):
        return found
    for entry in os.scandir(root_dir):
        if entry.is_file() and any(entry.path.endswith("." + e) for e in extensions):
            if amount is not None and len(found) == amount:
                break
            path = entry.path[len(top) + 1:] if is_pure else entry.path
            if (str_include is not None and str_include not in path) or (str_exclude is not None and str_exclude in path):
                continue
            found.append(path if is_ext else path[: -(len(path.split(".")[-1]) + 1)])
        elif entry.is_dir():
            found += traverse_dir(entry.path, extensions, amount, str_include, str_exclude, is_pure, is_sort, is_ext, root_dir)
    if is_sort:
        found.sort()
    return found


def load_model(expdir, model, optimizer, name="model", postfix="", device="cpu"):
    """Restore `<expdir>/<name>_<step>.pt` with the largest step (reference tools/utils.py:69-88): returns
    (global_step, model, optimizer); nothing found -> (0, model, optimizer) untouched.  Checkpoint = {'global_step', 'model'[,
    'optimizer']}; the model is loaded with strict=False like the reference."""
    prefix = os.path.join(expdir, name + ("_" + postfix if postfix == "" else postfix))
    stems = traverse_dir(expdir, ["pt"], is_ext=False)
    global_step = 0
    if stems:
        steps = [s[len(prefix):] for s in stems]
        best = max(int(s) if s.isdigit() else 0 for s in steps)
        path = prefix + str(best) + ".pt"
        print("restoring model from", path)
        ckpt = torch.load(path, map_location=torch.device(device), weights_only=False)
        global_step = ckpt["global_step"]
        model.load_state_dict(ckpt["model"], strict=False)
        if ckpt.get("optimizer") is not None and optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
    return global_step, model, optimizer
