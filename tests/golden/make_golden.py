#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference (read-only, /root/reference) in the
build container.  Run:  python tests/golden/make_golden.py

The reference cannot be imported as shipped (numba/spconv/torchvision/
cv2/... absent, several det3d files missing), so every *package* `__init__` is bypassed with an
empty namespace stub whose `__path__` points at the real directory (leaf modules are the real
files), and the absent third-party modules get inert stubs (`numba.jit` -> identity).  Nothing of
the reference is copied: only inputs and outputs are stored.

Fixtures:
  voxelize_small.npz / voxelize_second8k.npz / voxelize_maxvox.npz / voxelize_pillar.npz
      inputs + outputs of det3d.ops.point_cloud.point_cloud_ops.points_to_voxel (+ reader mean)
  rpn.npz / s2d_rpn.npz / center_head.npz
      digests of forward outputs, input-gradients and a few parameter gradients of the reference
      nn.Modules under the deterministic fill of tests/golden_util.fill_params
  losses.npz
      FastFocalLoss / RegLoss / CenterHead.loss / distillation losses / mask_offset_loss scalars
"""
import importlib
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import digest, fill_params, pack, seeded  # noqa: E402


def _ns(name, path=None, **attrs):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


def install_reference_stubs():
    ident = lambda *a, **k: (a[0] if a and callable(a[0]) and not k else (lambda f: f))
    _ns("numba", jit=ident, njit=ident, cuda=types.SimpleNamespace(jit=ident))
    _ns("numba.cuda", jit=ident)
    _ns("spconv", SparseConv3d=object, SubMConv3d=object, SparseModule=torch.nn.Module,
        SparseSequential=torch.nn.Sequential, SparseConvTensor=object)
    _ns("torchvision"); _ns("torchvision.models", resnet=types.SimpleNamespace())
    _ns("cv2")
    # package namespaces (skip every __init__.py)
    for pkg in ["det3d", "det3d.ops", "det3d.ops.point_cloud", "det3d.models", "det3d.models.readers",
                "det3d.models.necks", "det3d.models.bbox_heads", "det3d.models.losses", "det3d.models.detectors",
                "det3d.models.utils", "det3d.core", "det3d.core.utils", "det3d.core.input", "det3d.utils",
                "det3d.torchie", "det3d.torchie.trainer"]:
        _ns(pkg, os.path.join(REF, *pkg.split(".")))
    # det3d.utils.registry imports `from det3d import torchie` for is_str only
    sys.modules["det3d.torchie"].is_str = lambda s: isinstance(s, str)
    reg = importlib.import_module("det3d.utils.registry")
    sys.modules["det3d.utils"].Registry = reg.Registry
    sys.modules["det3d.utils"].build_from_cfg = reg.build_from_cfg
    # weight-init helpers: the real file is import-clean
    _ns("det3d.torchie.cnn", os.path.join(REF, "det3d/torchie/cnn"))
    wi = importlib.import_module("det3d.torchie.cnn.weight_init")
    for k in ["constant_init", "kaiming_init", "xavier_init", "normal_init", "uniform_init"]:
        setattr(sys.modules["det3d.torchie.cnn"], k, getattr(wi, k))
    sys.modules["det3d.torchie.trainer"].load_checkpoint = lambda *a, **k: None
    misc = importlib.import_module("det3d.models.utils.misc")
    _ns("det3d.utils.dist", dist_common=types.SimpleNamespace(get_world_size=lambda: 1))
    norm = importlib.import_module("det3d.models.utils.norm")
    mu = sys.modules["det3d.models.utils"]
    for k in ["Empty", "GroupNorm", "Sequential", "change_default_args"]:
        setattr(mu, k, getattr(misc, k))
    mu.build_norm_layer = norm.build_norm_layer
    mu.get_paddings_indicator = misc.get_paddings_indicator
    importlib.import_module("det3d.models.registry")
    _ns("det3d.models.builder")
    # center_utils imports circle_nms_jit (numba) and cv2: both stubbed above
    _ns("det3d.core.box_torch_ops")
    sys.modules["det3d.core"].box_torch_ops = sys.modules["det3d.core.box_torch_ops"]
    _ns("det3d.ops.dcn", DeformConv=object)
    _ns("det3d.models.detectors.single_stage", SingleStageDetector=torch.nn.Module)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------------------------
def gen_voxelize():
    pco = importlib.import_module("det3d.ops.point_cloud.point_cloud_ops")
    vg = importlib.import_module("det3d.core.input.voxel_generator")
    ve = importlib.import_module("det3d.models.readers.voxel_encoder")
    from sparse2dense_amd import scene

    def run(name, points, voxel_size, pc_range, max_points, max_voxels):
        gen = vg.VoxelGenerator(voxel_size, pc_range, max_points, max_voxels)
        voxels, coors, num = gen.generate(points, max_voxels)
        reader = ve.VoxelFeatureExtractorV3(num_input_features=points.shape[1])
        mean = reader(torch.from_numpy(voxels), torch.from_numpy(num)).numpy()
        save(name, points=points, voxel_size=np.asarray(voxel_size, np.float32),
             pc_range=np.asarray(pc_range, np.float32), max_points=np.int64(max_points),
             max_voxels=np.int64(max_voxels), grid_size=gen.grid_size, voxels=voxels, coors=coors,
             num_points=num, mean=mean)

    rs = np.random.RandomState(11)
    pts = (rs.rand(4000, 5) * [6, 6, 3, 1, 1] - [3, 3, 1.5, 0, 0]).astype(np.float32)
    # points exactly on cell / range boundaries exercise the fp32 floor((p-lo)/vs) path
    pts[:64, 0] = np.float32(-2.0) + np.arange(64, dtype=np.float32) * np.float32(0.1)
    pts[64:96, 1] = np.float32(2.0)
    pts[96:128, 2] = np.float32(-1.0)
    run("voxelize_small.npz", pts, [0.1, 0.1, 0.15], [-2.0, -2.0, -1.0, 2.0, 2.0, 1.1], 5, 20000)
    # max_voxels cut-off: new voxels dropped, existing ones keep filling (point_cloud_ops.py:46-54)
    run("voxelize_maxvox.npz", pts, [0.2, 0.2, 0.3], [-2.0, -2.0, -1.0, 2.0, 2.0, 1.1], 3, 500)
    s8 = scene.make_scene(8000, seed=7)
    run("voxelize_second8k.npz", s8["points"], scene.WAYMO_VOXEL, scene.WAYMO_RANGE, 5, 150000)
    run("voxelize_pillar.npz", s8["points"], scene.PILLAR_VOXEL, scene.PILLAR_RANGE, 20, 32000)
    # empty result (everything out of range)
    far = pts.copy(); far[:, 0] += 100
    run("voxelize_empty.npz", far[:100], [0.1, 0.1, 0.15], [-2.0, -2.0, -1.0, 2.0, 2.0, 1.1], 5, 100)


def _grads(outputs, inputs, seed):
    loss = 0
    for i, o in enumerate(outputs):
        loss = loss + (o * seeded(o.shape, seed + i)).sum()
    return torch.autograd.grad(loss, inputs, allow_unused=True)


def gen_dense_modules():
    rpn = importlib.import_module("det3d.models.necks.rpn")
    log = logging.getLogger("golden")
    torch.manual_seed(0)
    cfg = dict(layer_nums=[5, 5], ds_layer_strides=[1, 2], ds_num_filters=[128, 256], us_layer_strides=[1, 2],
               us_num_filters=[256, 256], num_input_features=256, logger=log)
    torch.set_num_threads(8)

    # ---- plain RPN (teacher neck), train mode (batch-stat BN) --------------------------------
    net = fill_params(rpn.RPN(**cfg)).train()
    x = seeded((1, 256, 188, 188), 100).abs_().requires_grad_(True)  # post-ReLU-like input
    y = net(x)
    names = ["blocks.0.1.weight", "blocks.1.16.weight", "deblocks.1.0.weight", "blocks.0.2.bias"]
    params = dict(net.named_parameters())
    g = _grads([y], [x] + [params[n] for n in names], 200)
    arrays = dict(state_keys=np.asarray(sorted(net.state_dict().keys())))
    arrays.update(pack("y", digest(y))); arrays.update(pack("gx", digest(g[0])))
    for n, gi in zip(names, g[1:]):
        arrays.update(pack("g:" + n, digest(gi)))
    net.eval()
    arrays.update(pack("y_eval", digest(net(x))))
    save("rpn.npz", **arrays)

    # ---- S2D_RPN (student neck) --------------------------------------------------------------
    net = fill_params(rpn.S2D_RPN(**cfg)).train()
    x = seeded((1, 256, 188, 188), 101).abs_().requires_grad_(True)
    outs = net(x)
    onames = ["x", "gen_offset_2", "gen_mask_2", "gen_offset_4", "gen_mask_4", "F_S_a", "F_S_b"]
    names = ["encoder_1.0.weight", "convnext_block_2.1.weight", "decoder_2.3.weight", "generator_2.3.weight",
             "fusion_sparse.0.weight", "blocks.0.1.weight", "gen_out_2.0.bias"]
    params = dict(net.named_parameters())
    g = _grads(list(outs), [x] + [params[n] for n in names], 300)
    arrays = dict(state_keys=np.asarray(sorted(net.state_dict().keys())),
                  state_shapes=np.asarray([str(tuple(v.shape)) for _, v in sorted(net.state_dict().items())]))
    for n, o in zip(onames, outs):
        arrays.update(pack(n, digest(o)))
    arrays.update(pack("gx", digest(g[0])))
    for n, gi in zip(names, g[1:]):
        arrays.update(pack("g:" + n, digest(gi)))
    net.eval()
    oe = net(x)
    assert oe[1] is None
    arrays.update(pack("x_eval", digest(oe[0]))); arrays.update(pack("F_S_a_eval", digest(oe[5])))
    save("s2d_rpn.npz", **arrays)


def _targets(seed, batch=2):
    from sparse2dense_amd import scene
    ex = {k: [] for k in ["hm", "anno_box", "ind", "mask", "cat"]}
    for b in range(batch):
        s = scene.make_scene(2000, seed=seed + b, n_cars=30, n_walls=3, n_peds=10)
        t = scene.assign_targets(s["gt_boxes"], s["gt_classes"])
        for k in ex:
            ex[k].append(torch.from_numpy(t[k]))
    return {k: [torch.stack(v)] for k, v in ex.items()}


def gen_head_and_losses():
    ch = importlib.import_module("det3d.models.bbox_heads.center_head")
    cl = importlib.import_module("det3d.models.losses.centernet_loss")
    tasks = [dict(num_class=3, class_names=["VEHICLE", "PEDESTRIAN", "CYCLIST"])]
    head = ch.CenterHead(in_channels=512, tasks=tasks, dataset="waymo", weight=2,
                         code_weights=[1.0] * 8,
                         common_heads={"reg": (2, 2), "height": (1, 2), "dim": (3, 2), "rot": (2, 2)})
    fill_params(head).train()
    x = seeded((2, 512, 188, 188), 102, 0.5).requires_grad_(True)
    preds = head(x)
    example = _targets(40)
    arrays = dict(state_keys=np.asarray(sorted(head.state_dict().keys())))
    for k in ["reg", "height", "dim", "rot", "hm"]:
        arrays.update(pack("pred." + k, digest(preds[0][k])))
    losses = head.loss(example, preds)
    loss = losses["loss"][0]
    names = ["shared_conv.0.weight", "tasks.0.hm.3.bias", "tasks.0.reg.0.weight"]
    params = dict(head.named_parameters())
    g = torch.autograd.grad(loss, [x] + [params[n] for n in names])
    arrays["loss"] = loss.detach().numpy()
    arrays["hm_loss"] = losses["hm_loss"][0].numpy()
    arrays["loc_loss"] = losses["loc_loss"][0].detach().numpy()
    arrays["loc_loss_elem"] = losses["loc_loss_elem"][0].numpy()
    arrays["num_positive"] = losses["num_positive"][0].numpy()
    arrays.update(pack("gx", digest(g[0])))
    for n, gi in zip(names, g[1:]):
        arrays.update(pack("g:" + n, digest(gi)))
    for k, v in example.items():
        arrays["ex." + k] = v[0].numpy()
    save("center_head.npz", **arrays)

    # ---- stand-alone loss scalars on seeded maps ---------------------------------------------
    arrays = {}
    out = torch.sigmoid(seeded((2, 3, 188, 188), 500)).clamp(1e-4, 1 - 1e-4)
    ff = cl.FastFocalLoss()(out, example["hm"][0], example["ind"][0], example["mask"][0], example["cat"][0])
    arrays["fastfocal"] = ff.numpy()
    box = seeded((2, 8, 188, 188), 501)
    tb = example["anno_box"][0][..., [0, 1, 2, 3, 4, 5, -2, -1]]
    arrays["regloss"] = cl.RegLoss()(box, example["mask"][0], example["ind"][0], tb).numpy()
    # zero-positive edge case
    zmask = torch.zeros_like(example["mask"][0])
    arrays["fastfocal_nopos"] = cl.FastFocalLoss()(out, example["hm"][0], example["ind"][0], zmask,
                                                   example["cat"][0]).numpy()

    # distillation losses (trainer.py:38-76,783-805): load only the helper functions' module
    import torch.nn.functional as F
    tr_src = open(os.path.join(REF, "det3d/torchie/trainer/trainer.py")).read()
    import ast
    tree = ast.parse(tr_src)
    fn_nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("fastfocalloss", "distill_reg_loss")]
    cu = importlib.import_module("det3d.core.utils.center_utils")
    ns = {"torch": torch, "F": F, "_transpose_and_gather_feat": cu._transpose_and_gather_feat}
    exec(compile(ast.Module(body=fn_nodes, type_ignores=[]), "trainer_helpers", "exec"), ns)
    t_hm = seeded((2, 3, 188, 188), 502)
    arrays["kd_hm"] = ns["fastfocalloss"](out, torch.sigmoid(t_hm), example["ind"][0], example["mask"][0],
                                          example["cat"][0]).numpy()
    t_box = seeded((2, 8, 188, 188), 503)
    arrays["kd_reg"] = ns["distill_reg_loss"](box, t_box, example["mask"][0], example["ind"][0]).numpy()
    # sparse2dense masked MSE x4 (trainer.py:783-789), evaluated literally
    F_D_a = torch.relu(seeded((2, 256, 188, 188), 504)); F_S_a = seeded((2, 256, 188, 188), 505)
    F_D_b = torch.relu(seeded((2, 256, 188, 188), 506)); F_S_b = seeded((2, 256, 188, 188), 507)
    inds = F_D_a > 0
    s2d = F.mse_loss(F_S_a[inds], F_D_a[inds]) * 10
    s2d = s2d + F.mse_loss(F_S_a[~inds], F_D_a[~inds]) * 20
    inds = F_D_b > 0
    s2d = s2d + F.mse_loss(F_S_b[inds], F_D_b[inds]) * 5
    s2d = s2d + F.mse_loss(F_S_b[~inds], F_D_b[~inds]) * 20
    arrays["s2d_mse"] = s2d.numpy()

    # mask_offset_loss (voxelnet.py:171-185) + metric grid (voxelnet.py:230-236)
    vn = importlib.import_module("det3d.models.detectors.voxelnet")
    D, H, W = 10, 47, 47
    gt = torch.zeros(2, 5, D, H, W)
    g = torch.Generator().manual_seed(508)
    occ = torch.rand((2, D, H, W), generator=g) < 0.03
    gt[:, :3] = (seeded((2, 3, D, H, W), 509) * 20) * occ[:, None]
    gt[:, 3:] = torch.rand((2, 2, D, H, W), generator=g) * occ[:, None]
    zs, ys, xs = torch.meshgrid([torch.arange(0, D), torch.arange(0, H), torch.arange(0, W)])
    ys = ys * (150.4 / H) - 75.2 + (150.4 / H) / 2
    xs = xs * (150.4 / W) - 75.2 + (150.4 / H) / 2
    zs = zs * (6 / D) - 2 + (6 / D) / 2
    grid = torch.cat([xs[None], ys[None], zs[None]], 0)[None].repeat(2, 1, 1, 1, 1).to(gt)
    gen_offset = seeded((2, 3, D, H, W), 510)
    gen_mask = seeded((2, 1, D, H, W), 511)
    ml, ol = vn.KD_VoxelNet.mask_offset_loss(None, gen_offset, gen_mask, gt, grid)
    arrays["mask_loss"] = ml.numpy(); arrays["offset_loss"] = ol.numpy()
    arrays["mol_gt"] = gt.numpy(); arrays["mol_shape"] = np.asarray([D, H, W])
    save("losses.npz", **arrays)


def gen_pillars():
    """PointPillars reader + S2D scatter backbone (BASELINE config 5) from the reference modules."""
    pe = importlib.import_module("det3d.models.readers.pillar_encoder")
    from sparse2dense_amd import scene
    g = np.load(os.path.join(HERE, "voxelize_pillar.npz"))
    voxels = torch.from_numpy(g["voxels"]); num = torch.from_numpy(g["num_points"])
    coors = torch.from_numpy(np.concatenate([np.zeros((g["coors"].shape[0], 1), np.int32), g["coors"]], 1))
    pfn = pe.PillarFeatureNet(num_filters=[64, 64], num_input_features=5, with_distance=False,
                              voxel_size=scene.PILLAR_VOXEL, pc_range=scene.PILLAR_RANGE)
    fill_params(pfn).train()
    vin = voxels.clone().requires_grad_(True)
    feats = pfn(vin, num, coors)
    names = ["pfn_layers.0.linear.weight", "pfn_layers.1.linear.weight", "pfn_layers.0.norm.weight"]
    params = dict(pfn.named_parameters())
    gr = _grads([feats], [vin] + [params[n] for n in names], 700)
    arrays = dict(state_keys=np.asarray(sorted(pfn.state_dict().keys())))
    arrays.update(pack("feats", digest(feats))); arrays.update(pack("gvox", digest(gr[0])))
    for n, gi in zip(names, gr[1:]):
        arrays.update(pack("g:" + n, digest(gi)))
    # scatter (plain) on the PFN output
    sc = pe.PointPillarsScatter(num_input_features=64)
    canvas = sc(feats.detach(), coors, 1, np.array([468, 468, 1]))
    arrays.update(pack("canvas", digest(canvas)))
    save("pillar_pfn.npz", **arrays)

    s2d = fill_params(pe.PointPillarsScatter_S2D(num_input_features=64)).train()
    f = feats.detach().clone().requires_grad_(True)
    outs = s2d(f, coors, 1, np.array([468, 468, 1]))
    onames = ["F_S_a", "F_S_b", "gen_offset", "gen_mask"]
    names = ["encoder_1.1.weight", "convnext_block_2.1.weight", "decoder_2.3.weight", "generator.3.weight", "gen_mask.3.bias"]
    params = dict(s2d.named_parameters())
    gr = _grads(list(outs), [f] + [params[n] for n in names], 800)
    arrays = dict(state_keys=np.asarray(sorted(s2d.state_dict().keys())),
                  state_shapes=np.asarray([str(tuple(v.shape)) for _, v in sorted(s2d.state_dict().items())]))
    for n, o in zip(onames, outs):
        arrays.update(pack(n, digest(o)))
    arrays.update(pack("gf", digest(gr[0])))
    for n, gi in zip(names, gr[1:]):
        arrays.update(pack("g:" + n, digest(gi)))
    save("pillar_s2d.npz", **arrays)


if __name__ == "__main__":
    install_reference_stubs()
    which = sys.argv[1:] or ["voxelize", "dense", "head", "pillars"]
    if "voxelize" in which:
        gen_voxelize()
    if "dense" in which:
        gen_dense_modules()
    if "head" in which:
        gen_head_and_losses()
    if "pillars" in which:
        gen_pillars()
