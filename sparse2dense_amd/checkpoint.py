"""Checkpoint / on-disk formats of the hot path's callers.

  save_checkpoint / load_checkpoint / load_state_dict   /root/reference/det3d/torchie/trainer/checkpoint.py:42-100,146-240
  Trainer.save_checkpoint (epoch file + latest.pth link), Trainer.resume   /root/reference/det3d/torchie/trainer/trainer.py:416-430,558-571
  Waymo frame pickle -> points f32[N,5]                                   /root/reference/det3d/datasets/pipelines/loading.py:61-70,94-97,140-145

The file layout is the reference's: a `.pth` written by `torch.save` holding {"meta": dict, "state_dict": OrderedDict of CPU
tensors, "optimizer": optimizer.state_dict()} (or a bare state_dict).  Because every module of this package keeps the
reference's parameter names, shapes and the spconv [kD,kH,kW,Cin,Cout] weight layout, reference-trained files load directly.
The loader keeps the reference's three key conventions:
  * a leading "module." (DataParallel / DDP) is stripped when the FIRST key has it;
  * `name[4:]`: a model key also accepts the checkpoint key with its first 4 characters removed;
  * "single_det.": a two-stage model's first-stage keys accept the single-stage checkpoint's keys without that prefix.
Unexpected keys and shape mismatches are skipped and reported, never fatal (strict=True raises after loading).
"""
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def load_state_dict(module, state_dict, strict=False, logger=None):
    """Copy `state_dict` into `module` (checkpoint.py:42-100).  Returns dict(unexpected, missing, mismatched)."""
    own = module.state_dict()
    unexpected, mismatched = [], []
    loaded = set()

    def take(dst_name, src):
        src = src.data if isinstance(src, torch.nn.Parameter) else src
        if src.size() != own[dst_name].size():
            return False
        own[dst_name].copy_(src)
        loaded.add(dst_name)
        return True

    with torch.no_grad():
        for name, value in state_dict.items():
            if name not in own:
                unexpected.append(name)
                continue
            if not take(name, value):
                mismatched.append((name, tuple(own[name].size()), tuple(value.size())))
        # the reference's two remaps, applied in its order (they may overwrite an exact-name load, as there)
        for name in own:
            if name[4:] in state_dict:
                take(name, state_dict[name[4:]])
            elif "single_det." in name and name.replace("single_det.", "") in state_dict:
                take(name, state_dict[name.replace("single_det.", "")])
    missing = [k for k in own if k not in state_dict and k not in loaded and "num_batches_tracked" not in k]
    report = dict(unexpected=unexpected, missing=missing, mismatched=mismatched)
    msgs = []
    if unexpected:
        msgs.append("unexpected key in source state_dict: " + ", ".join(unexpected))
    if missing:
        msgs.append("missing keys in source state_dict: " + ", ".join(missing))
    if mismatched:
        msgs.append("these keys have mismatched shape: " + ", ".join(f"{n} (expected {a}, loaded {b})" for n, a, b in mismatched))
    if msgs:
        text = "The model and loaded state dict do not match exactly\n" + "\n".join(msgs)
        if strict:
            raise RuntimeError(text)
        if logger is not None:
            logger.warning(text)
    return report


def load_checkpoint(model, filename, map_location=None, strict=False, logger=None):
    """Load a reference-format checkpoint file into `model` (checkpoint.py:146-197); returns the loaded checkpoint object."""
    if not os.path.isfile(filename):
        raise IOError(f"{filename} is not a checkpoint file")
    checkpoint = torch.load(filename, map_location=map_location, weights_only=False)
    if isinstance(checkpoint, OrderedDict):
        state_dict = checkpoint
    elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        state_dict = checkpoint["state_dict"]
    else:
        raise RuntimeError(f"No state_dict found in checkpoint file {filename}")
    if len(state_dict) and next(iter(state_dict)).startswith("module."):
        state_dict = OrderedDict((k[7:], v) for k, v in state_dict.items())
    load_state_dict(_unwrap(model), state_dict, strict, logger)
    from .dense2d import clear_pack_cache
    clear_pack_cache()   # packed weight images are keyed on the parameter version; a load through .copy_ moves it, belt and braces
    return checkpoint


def weights_to_cpu(state_dict):
    return OrderedDict((k, v.cpu()) for k, v in state_dict.items())


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """{"meta", "state_dict" (CPU tensors), "optimizer"} via torch.save (checkpoint.py:215-240)"""
    if meta is None:
        meta = {}
    elif not isinstance(meta, dict):
        raise TypeError(f"meta must be a dict or None, but got {type(meta)}")
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    checkpoint = {"meta": meta, "state_dict": weights_to_cpu(_unwrap(model).state_dict())}
    if optimizer is not None:
        checkpoint["optimizer"] = optimizer.state_dict()
    torch.save(checkpoint, filename)


def save_epoch_checkpoint(model, out_dir, epoch, iteration, optimizer=None, filename_tmpl="epoch_{}.pth", meta=None):
    """Trainer.save_checkpoint (trainer.py:416-430): epoch file named by the 1-based epoch, `latest.pth` relative symlink."""
    meta = dict(meta or {})
    meta.update(epoch=epoch + 1, iter=iteration)
    name = filename_tmpl.format(epoch + 1)
    path = os.path.join(out_dir, name)
    save_checkpoint(model, path, optimizer=optimizer, meta=meta)
    link = os.path.join(out_dir, "latest.pth")
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(name, link)
    return path


def resume(model, filename, optimizer=None, map_location="cpu", resume_optimizer=True):
    """Trainer.resume (trainer.py:558-571): weights + (epoch, iter) + optimizer state.  Returns (epoch, iter)."""
    checkpoint = load_checkpoint(model, filename, map_location=map_location)
    if optimizer is not None and resume_optimizer and "optimizer" in checkpoint:
        optimizer.load_state_dict(checkpoint["optimizer"])
    return checkpoint["meta"]["epoch"], checkpoint["meta"]["iter"]


def read_waymo_frame(path_or_obj):
    """Waymo frame pickle of the reference's converter -> points f32[N,5] = (x, y, z, tanh(intensity), elongation)
    (loading.py:61-70).  Unlike the reference the pickled arrays are not modified in place."""
    obj = path_or_obj
    if not isinstance(obj, dict):
        with open(path_or_obj, "rb") as f:
            obj = pickle.load(f)
    xyz = np.asarray(obj["lidars"]["points_xyz"])
    feat = np.array(obj["lidars"]["points_feature"], copy=True)
    feat[:, 0] = np.tanh(feat[:, 0])
    return np.concatenate([xyz, feat], axis=-1).astype(np.float32)
