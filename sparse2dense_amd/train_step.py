"""One training iteration of the hot path: what `Trainer.batch_processor_inline` and
`TS_Trainer.batch_processor_inline` (CenterPoint branch) compute
(/root/reference/det3d/torchie/trainer/trainer.py:432-459,726-811) followed by
`OptimizerHook.after_train_iter` (hooks/optimizer.py:15-21: zero_grad, backward, clip 35).

The teacher/student branch in the reference is chosen by `T_model.backbone._get_name()`
(trainer.py:741,775), which breaks under DDP wrapping; here the unwrapped module is inspected.
"""
import torch
import torch.nn.functional as F

from .heads import distill_reg_loss, fast_focal_loss, sparse2dense_loss


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


def parse_losses(losses):
    """sum of the per-task `loss` entries (trainer.py:127-160 `parse_second_losses`, totals only)."""
    return sum(losses["loss"])


def single_stage_loss(model, example):
    """Plain CenterPoint / SECOND-style step: `model(example, return_loss=True)`."""
    losses = model(example, return_loss=True)
    if isinstance(losses, tuple):  # KD_VoxelNet returns (losses, preds)
        losses = losses[0]
    return parse_losses(losses), losses


def distill_loss(T_model, S_model, example):
    """Teacher (eval, no grad) + student forward and every S2D loss term (trainer.py:775-811)."""
    T_model.eval()
    with torch.no_grad():
        T_preds, F_D_a, F_D_b = T_model(example, return_loss=False, return_feature=True, return_recon_feature=True)
    losses, F_S_a, F_S_b, S_preds, mask_loss, offset_loss = S_model(example, return_loss=True, return_feature=True)
    s2d = sparse2dense_loss(F_S_a, F_D_a, F_S_b, F_D_b)
    ind, mask, cat = example["ind"][0], example["mask"][0], example["cat"][0]
    kd_hm = fast_focal_loss(S_preds[0]["hm"], torch.sigmoid(T_preds[0]["hm"]), ind, mask, cat)
    t_box = torch.cat((T_preds[0]["reg"], T_preds[0]["height"], T_preds[0]["dim"], T_preds[0]["rot"]), dim=1)
    head = _unwrap(S_model).bbox_head
    kd_reg = distill_reg_loss(S_preds[0]["anno_box"], t_box, mask, ind)
    cw = getattr(head, "_code_w", None)   # device copy made by CenterHead.loss (no H2D copy per step)
    if cw is None or cw.device != kd_reg.device or cw.dtype != kd_reg.dtype:
        cw = kd_reg.new_tensor(head.code_weights)
    kd_reg = (kd_reg * cw).sum() * head.weight
    losses["loss"][0] = losses["loss"][0] + kd_hm + kd_reg + s2d + (mask_loss + offset_loss)
    losses["sparse2dense_loss"] = [s2d.detach()]
    losses["kd_hm_loss"] = [kd_hm.detach()]
    losses["kd_reg_loss"] = [kd_reg.detach()]
    losses["mask_loss"] = [mask_loss.detach()]
    losses["reconstruction_loss"] = [offset_loss.detach()]
    return parse_losses(losses), losses


def backward_and_clip(loss, params, max_norm=35.0):
    """zero_grad -> backward -> clip_grad_norm_(35) (hooks/optimizer.py:15-21, config :216).  Under data parallelism
    (dp.wrap_ddp, routes "overlap"/"flat") the gradients are averaged over the ranks by the model's GradBuckets: its
    all-reduces are launched from gradient hooks during the backward and awaited here, before the clip.
    max_norm=None: no clip here (solver.OneCycleAdam.clip_and_step folds it into the update)."""
    from . import dp, side
    params = list(params)
    syncs = dp.bucketers_of(params)
    for g in syncs:
        g.prepare()
    for p in params:
        p.grad = None
    loss.backward()
    side.join()   # weight gradients produced on the second stream (side.py; the engine callback has normally joined already)
    for g in syncs:
        g.finish()
    if max_norm is None:
        return None
    return torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], max_norm)


def backward_and_step(loss, params, optimizer, scheduler=None, iteration=None, max_norm=35.0):
    """One OptimizerHook.after_train_iter of the reference (hooks/optimizer.py:15-21) with its optimizer (apis/train.py:168-186,
    fastai OptimWrapper Adam + true weight decay under the OneCycle schedule): schedule step -> zero_grad -> backward (+ overlapped
    gradient all-reduce) -> clip(35) and update in the fused multi-tensor kernels (solver.OneCycleAdam.clip_and_step)."""
    if scheduler is not None:
        scheduler.step(iteration)
    backward_and_clip(loss, params, None)
    norm = optimizer.clip_and_step(max_norm)
    from . import graphed
    graphed.grads_consumed()   # (the next call clears every .grad before its backward: a graphed segment need not park them over its next forward)
    return norm
