"""Shared helpers for golden fixtures: a deterministic, construction-order-independent parameter
fill (so the reference module in make_golden.py and our module in the tests get bit-identical
weights from their *names and shapes alone*) and compact tensor digests."""
import zlib

import numpy as np
import torch


def fill_params(module, seed=0):
    """Overwrite every parameter/buffer of `module` from a per-name seeded CPU generator."""
    sd = module.state_dict()
    new = {}
    for name in sorted(sd):
        t = sd[name]
        if not torch.is_floating_point(t):
            new[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))
        if name.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif name.endswith("running_mean"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() == 1 and name.endswith(".weight"):  # norm scale
            v = torch.rand(t.shape, generator=g) + 0.5
        elif t.dim() == 1 or name.endswith(".bias"):
            v = torch.randn(t.shape, generator=g) * 0.05
        elif t.dim() == 3 and name.endswith(".weight"):  # LayerNorm([C,H,W]) scale
            v = torch.rand(t.shape, generator=g) + 0.5
        else:
            fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else t.numel()
            v = torch.randn(t.shape, generator=g) * (1.5 / max(fan_in, 1)) ** 0.5
        new[name] = v.to(t.dtype)
    module.load_state_dict(new)
    return module


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def digest(t, stride=None):
    """Compact fingerprint of a tensor: shape, mean, abs-mean, per-channel means, strided sample."""
    t = t.detach().double().cpu()
    out = {"shape": np.asarray(t.shape, np.int64), "mean": np.float64(t.mean()), "absmean": np.float64(t.abs().mean())}
    if t.dim() >= 2:
        dims = [d for d in range(t.dim()) if d != 1]
        out["chmean"] = t.mean(dim=dims).numpy()
    flat = t.reshape(-1)
    if stride is None:
        stride = max(1, flat.numel() // 4096)
    out["sample"] = flat[::stride].numpy().astype(np.float32)
    out["stride"] = np.int64(stride)
    return out


def pack(prefix, d):
    return {f"{prefix}.{k}": v for k, v in d.items()}


def check_digest(t, npz, prefix, rtol, atol):
    d = digest(t, int(npz[f"{prefix}.stride"]))
    assert tuple(d["shape"]) == tuple(npz[f"{prefix}.shape"]), (prefix, d["shape"], npz[f"{prefix}.shape"])
    np.testing.assert_allclose(d["sample"], npz[f"{prefix}.sample"], rtol=rtol, atol=atol, err_msg=prefix)
    np.testing.assert_allclose(d["mean"], npz[f"{prefix}.mean"], rtol=rtol, atol=atol, err_msg=prefix)
    np.testing.assert_allclose(d["absmean"], npz[f"{prefix}.absmean"], rtol=rtol, atol=atol, err_msg=prefix)
    if f"{prefix}.chmean" in npz:
        np.testing.assert_allclose(d["chmean"], npz[f"{prefix}.chmean"], rtol=rtol, atol=atol * 10, err_msg=prefix)


def check_digest_norm(t, npz, prefix, tol):
    """Norm-wise variant for cross-backend runs (MIOpen's fp32 conv algorithms — Winograd/implicit
    GEMM — differ from the CPU's direct sums by ~1e-3 per layer): relative L2 error of the strided
    sample and of the per-channel means."""
    d = digest(t, int(npz[f"{prefix}.stride"]))
    assert tuple(d["shape"]) == tuple(npz[f"{prefix}.shape"]), (prefix, d["shape"], npz[f"{prefix}.shape"])
    a, b = d["sample"].astype(np.float64), npz[f"{prefix}.sample"].astype(np.float64)
    err = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
    assert err <= tol, (prefix, err)
    assert abs(d["absmean"] - npz[f"{prefix}.absmean"]) <= tol * abs(npz[f"{prefix}.absmean"]) + 1e-12, prefix


class _RoundBf16STE(torch.autograd.Function):
    """x -> bf16 -> x.dtype in the forward, the same rounding on the gradient in the backward"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def bf16_emulation_copy(module, device="cpu"):
    """Float64 copy (on `device`: torch's own float64 kernels) of a 2-D neck / head whose leaf layers (convs, transposed convs, batch norms incl. their fused
    ReLU, activations, layer norms) round their OUTPUT - and the gradient flowing back through it - to bf16, and whose
    conv weights are rounded to bf16: the storage roundings of the product's bf16 NHWC mode with exact accumulation in
    between.  3-D (PCR) layers stay unrounded: the product runs them in fp32."""
    import copy
    from torch import nn
    m = copy.deepcopy(module).cpu().double()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, (nn.Conv2d, nn.ConvTranspose2d)):
                mod.weight.copy_(mod.weight.to(torch.bfloat16).double())
    add_bf16_storage_hooks(m)
    return m.to(device)


def add_bf16_storage_hooks(m):
    """in place: the 2-D leaf layers of `m` round their output (and the gradient flowing back through it) to bf16"""
    from torch import nn
    leaf2d = (nn.Conv2d, nn.ConvTranspose2d, nn.BatchNorm2d, nn.GELU, nn.ReLU, nn.LayerNorm)

    def hook(_mod, _inp, out):
        return _RoundBf16STE.apply(out) if torch.is_tensor(out) and out.dim() == 4 else out

    # the last conv of a head branch (1-3 output channels) writes fp32 planar predictions in the product: not rounded
    handles = [mod.register_forward_hook(hook) for mod in m.modules()
               if isinstance(mod, leaf2d) and not (isinstance(mod, nn.Conv2d) and mod.out_channels < 8)]
    return handles


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def tiny_checkpoint_net():
    """the small network of the checkpoint fixtures (tests/golden/make_golden_r03.py): a batch norm BETWEEN non-BN leaves, so that the
    reference's optimizer indices (non-BN leaves first, then BN leaves) differ from model.parameters() order"""
    from torch import nn

    class TinyNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 8, 3)
            self.bn = nn.BatchNorm2d(8)
            self.block = nn.Sequential(nn.Conv2d(8, 8, 1), nn.BatchNorm2d(8), nn.ReLU())
            self.fc = nn.Linear(8, 4)

        def forward(self, x):
            return self.fc(self.block(self.bn(self.conv(x))).mean((2, 3)))
    return TinyNet()


def reference_param_groups(model):
    """restatement of OptimWrapper.create's param groups: split_bn_bias(get_layer_groups(model)) (fastai_optim.py:17-28,
    apis/train.py:159-164) - leaf modules depth-first, non-BN leaves' trainable parameters, then the BN leaves'"""
    from torch import nn

    def flatten(m):
        ch = list(m.children())
        return sum(map(flatten, ch), []) if ch else [m]
    leaves = flatten(model)
    bn = nn.modules.batchnorm._BatchNorm
    l1 = nn.Sequential(*[c for c in leaves if not isinstance(c, bn)])
    l2 = nn.Sequential(*[c for c in leaves if isinstance(c, bn)])
    return [[q for q in l.parameters() if q.requires_grad] for l in (l1, l2)]


# inputs of the round-6 predict fixture (tests/golden/make_golden_r06.py generates the expected outputs with the reference; the GPU test
# feeds the same seeded maps to this package's CenterHead.predict)
PREDICT_FLIP_CIRCLE_CFG = dict(post_center_limit_range=[-80, -80, -10.0, 80, 80, 10.0],
                               nms=dict(nms_pre_max_size=4096, nms_post_max_size=83, nms_iou_threshold=0.7), score_threshold=0.1,
                               pc_range=[-75.2, -75.2], out_size_factor=8, voxel_size=[0.1, 0.1], double_flip=True, circular_nms=True,
                               min_radius=[2.0])


def predict_flip_circle_inputs(h=188, w=188, batch=8):
    """seeded prediction maps [8, C, H, W] (2 samples x 4 flips); the heat map is biased down so that a few hundred cells pass the threshold"""
    out = {}
    for i, (k, c) in enumerate(dict(reg=2, height=1, dim=3, rot=2, hm=3).items()):
        t = seeded((batch, c, h, w), 900 + i)
        out[k] = t * 1.5 - 3.0 if k == "hm" else (t * 0.3 + 0.8 if k == "dim" else (torch.sigmoid(t) if k == "reg" else t))
    return out
