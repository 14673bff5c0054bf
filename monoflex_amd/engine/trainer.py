"""One optimisation step and its data-parallel wrapping (reference engine/trainer.py:101-126,
tools/plain_train_net.py:128-137).

Data parallelism: one process per GPU, gradients averaged by torch DDP's bucketed all-reduce over RCCL, overlapped
with the backward kernels.  The reference wraps with find_unused_parameters=True because six parameters (the outer
level3/level4 `project` conv + BN, dead in DLA's Tree.forward) never receive a gradient, which makes DDP walk the
autograd graph every iteration; here those six are excluded statically, so no per-iteration graph traversal happens
and no bucket ever waits for them."""
import torch

DEAD_PARAMETER_SUFFIXES = ("base.level3.project.0.weight", "base.level3.project.1.weight", "base.level3.project.1.bias",
                           "base.level4.project.0.weight", "base.level4.project.1.weight", "base.level4.project.1.bias")


def dead_parameter_names(model):
    """Names of the parameters that are unreachable from the loss, plus the buffers of their BN."""
    names = [n for n, _ in model.named_parameters() if n.endswith(DEAD_PARAMETER_SUFFIXES)]
    stems = {n.rsplit(".", 1)[0] for n in names}
    names += [n for n, _ in model.named_buffers() if n.rsplit(".", 1)[0] in stems]
    return names


def convert_sync_batchnorm(model):
    """Mark the BatchNorm modules the reference's SyncBatchNorm.convert_sync_batchnorm would convert
    (tools/plain_train_net.py:131-132): every nn.BatchNorm1d/2d except the heads' InPlaceABN (upstream's InPlaceABN is
    not a _BatchNorm subclass, so the nine head ABNs keep rank-local statistics)."""
    from ..model.head.detector_predictor import InPlaceABN
    n = 0
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and not isinstance(m, InPlaceABN):
            m.sync_bn = True
            n += 1
    return n


def wrap_data_parallel(model, device_ids=None, bucket_cap_mb=25):
    """DistributedDataParallel(broadcast_buffers=False) with the dead parameters ignored instead of searched for."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    DDP._set_params_and_buffers_to_ignore_for_model(model, dead_parameter_names(model))
    return DDP(model, device_ids=device_ids, broadcast_buffers=False, find_unused_parameters=False,
               bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True)


class PreparedTargets(list):
    """The per-image targets of one batch plus their device-side stacked form, built once per batch (outside any graph
    capture): `.edge` = (edge_indices, edge_lens) for the predictor, `.loss` = (heat maps, stacked fields) for the loss."""


def prepare_targets(model, targets, device, fields=None):
    """Per-image targets -> PreparedTargets. `fields`: the batch-stacked device tensors a DeviceLoader batch carries
    (`batch["fields"]`, produced by the target-encoding kernels): with them nothing is re-stacked per image."""
    from ..model.head.detector_predictor import stack_edge_fields
    m = model.module if hasattr(model, "module") else model
    pt = PreparedTargets(targets)
    if fields is None:
        pt.edge = stack_edge_fields(targets, device)
        pt.loss = m.heads.loss_evaluator.prepare_targets(targets, device)
        return pt
    dev = torch.device(device)
    pt.edge = (fields["edge_indices"].to(device=dev, dtype=torch.int32).contiguous(), fields["edge_len"].to(device=dev, dtype=torch.int32).contiguous())
    names = ("cls_ids", "target_centers", "keypoints", "keypoints_depth_mask", "dimensions", "locations", "rotys", "alphas",
             "orientations", "pad_size", "reg_mask", "reg_weight", "offset_3D", "trunc_mask")
    d = {n: fields[n].to(dev) for n in names}
    d["bboxes"] = fields["2d_bboxes"].to(dev)
    calibs = [t.get_field("calib") for t in targets]
    d["calib"] = calibs
    d["calib_f32"] = torch.tensor([[c.f_u, c.f_v, c.c_u, c.c_v, c.b_x, c.b_y] for c in calibs], dtype=torch.float32).to(dev)
    pt.loss = (fields["hm"].to(dev), d)
    return pt


def train_step(model, optimizer, images, targets, grad_norm_clip=-1.0, scheduler=None):
    """trainer.py:109-126: forward -> summed loss -> zero_grad -> backward (+DDP all-reduce) -> clip -> step."""
    loss_dict, log_loss_dict = model(images, targets)
    losses = sum(loss_dict.values())
    optimizer.zero_grad(set_to_none=True)
    losses.backward()
    if grad_norm_clip > 0:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], grad_norm_clip)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return losses.detach(), loss_dict, log_loss_dict
