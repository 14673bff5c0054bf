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
    not a _BatchNorm subclass, so the nine head ABNs keep rank-local statistics).  Modules that
    `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` already replaced (the reference script's literal call) are
    recognised too: the HIP path never calls a BN module's forward, it reads its parameters / buffers and this flag."""
    n = 0
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):          # InPlaceABN is not one (as upstream)
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
    d["object_rows"] = m.heads.loss_evaluator.pack_objects(d)
    pt.loss = (fields["hm"].to(dev), d)
    return pt


class GraphedTrainStep:
    """One optimisation step replayed from hipGraphs, single GPU or data-parallel.

    The eager step costs tens of ms of host enqueue (hundreds of small launches), and torch DDP's bucket hooks cannot be
    captured reliably, so the data-parallel form is built from three stream-ordered pieces instead:

        graph A   forward -> loss -> backward -> every gradient copied into ONE flat fp32 buffer (a multi-tensor copy)
        RCCL      all-reduce(AVG) of the flat buffer in `comm_chunks` pieces (83.8 MB fp32; 7 xGMI links x 153 GB/s)
        graph B   fused AdamW reading the gradients as views of the flat buffer

    world_size 1 captures the whole step as one graph (no flat copy, no collective) unless `split=True`.  The six dead
    parameters (`dead_parameter_names`) never get a gradient and are left out of the flat buffer.  Static inputs: the
    image batch and the PreparedTargets given at construction; `load_batch` overwrites them in place between replays.
    `use_graphs=False` runs the same three pieces eagerly (CPU / gloo tests of the flat-buffer exchange)."""

    def __init__(self, model, optimizer, images, targets, group=None, comm_chunks=4, warmup=3, split=None, use_graphs=None):
        import torch.distributed as dist
        self.model, self.optimizer, self.images, self.targets = model, optimizer, images, targets
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist_on else 1
        self.group = group
        self.comm_chunks = max(1, int(comm_chunks))
        self.split = (self.world > 1) if split is None else bool(split)
        self.use_graphs = images.is_cuda if use_graphs is None else bool(use_graphs)
        self.graph_a = self.graph_b = self.flat = None
        if self.split:
            dead = set(dead_parameter_names(model))
            self.params = [p for n, p in model.named_parameters() if p.requires_grad and n not in dead]
            self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=images.device)
            self.views, o = [], 0
            for p in self.params:
                self.views.append(self.flat[o:o + p.numel()].view_as(p))
                o += p.numel()
        if not self.use_graphs:
            return
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep: build the optimizer with capturable=True (solver.build_optimizer)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(2, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_a = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):     # (an RCCL watchdog thread may poll events meanwhile)
                self.loss = self._fwd_bwd()
                optimizer.step()
            return
        with torch.cuda.graph(self.graph_a, capture_error_mode="thread_local"):
            self.loss = self._fwd_bwd()
            self._flatten()
        self._point_grads_at_views()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode="thread_local"):
            optimizer.step()

    def _fwd_bwd(self):
        if self.images.is_cuda:
            from .. import autograd as AG
            AG.pack_all_weights()
        loss_dict, _ = self.model(self.images, self.targets)
        losses = sum(loss_dict.values())
        self.optimizer.zero_grad(set_to_none=True)
        losses.backward()
        return losses.detach()

    def _flatten(self):
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        if missing:
            raise RuntimeError("GraphedTrainStep: %d live parameters received no gradient (first index %d)" % (len(missing), missing[0]))
        torch._foreach_copy_(self.views, [p.grad for p in self.params])

    def _point_grads_at_views(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def _exchange(self):
        if not self.dist_on:
            return
        import torch.distributed as dist
        n = self.flat.numel()
        step = (n + self.comm_chunks - 1) // self.comm_chunks
        for i in range(0, n, step):
            dist.all_reduce(self.flat[i:i + step], op=dist.ReduceOp.SUM, group=self.group)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)

    def _eager(self):
        loss = self._fwd_bwd()
        if self.split:
            self._flatten()
            self._exchange()
            self._point_grads_at_views()
        self.optimizer.step()
        return loss

    def load_batch(self, images, targets=None):
        self.images.copy_(images, non_blocking=True)
        if targets is not None:
            for dst, src in zip(_target_tensors(self.targets), _target_tensors(targets)):
                dst.copy_(src, non_blocking=True)

    def __call__(self):
        if not self.use_graphs:
            self.loss = self._eager()
            return self.loss
        self.graph_a.replay()
        if self.graph_b is not None:
            self._exchange()
            self.graph_b.replay()
        return self.loss


def _target_tensors(pt):
    out = list(pt.edge) + [pt.loss[0]]
    out += [v for k, v in sorted(pt.loss[1].items()) if torch.is_tensor(v)]
    return out


def train_step(model, optimizer, images, targets, grad_norm_clip=-1.0, scheduler=None):
    """trainer.py:109-126: forward -> summed loss -> zero_grad -> backward (+DDP all-reduce) -> clip -> step."""
    if images.is_cuda:
        from .. import autograd as AG
        AG.pack_all_weights()                                   # every conv operand of the step from the current parameters, one launch
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


def advance_schedule(iteration, warmup_iters, scheduler, warmup_scheduler):
    """engine/trainer.py:123-126: after the optimizer step of `iteration`, the warm-up schedule (while iteration < warmup_iters)
    or the main schedule is SET to `iteration` -- `step(iteration)`, not `step()`: the main LambdaLR is not advanced during the
    warm-up, so with a plain step() its decay STEPS would fire WARMUP_STEPS iterations late, and a resumed run would restart at 0."""
    from ..solver import step_scheduler
    sched = warmup_scheduler if (warmup_scheduler is not None and iteration < warmup_iters) else scheduler
    if sched is not None:
        step_scheduler(sched, iteration)


def do_train(cfg, distributed, model, data_loader, data_loaders_val, optimizer, scheduler, warmup_scheduler, checkpointer, device,
             arguments):
    """The reference's training loop (engine/trainer.py:62-230) over this build's loaders: forward -> summed loss -> backward
    (+ gradient all-reduce when `model` is DDP-wrapped) -> optional clip -> AdamW -> per-iteration scheduler, periodic
    checkpoints (rank 0) and validation.  TensorBoard logging and best-mAP bookkeeping are not reproduced."""
    import logging
    import time
    from ..utils import comm
    logger = logging.getLogger("monoflex.trainer")
    max_iter, start_iter = cfg.SOLVER.MAX_ITERATION, arguments["iteration"]
    warmup_iters = cfg.SOLVER.WARMUP_STEPS if (cfg.SOLVER.LR_WARMUP and warmup_scheduler is not None) else -1
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    net = model.module if hasattr(model, "module") else model
    model.train()
    t0 = time.time()
    loss_v = None
    for data, iteration in zip(data_loader, range(start_iter, max_iter)):
        images = data["images"].to(device) if hasattr(data["images"], "to") else data["images"]
        targets = [t.to(device) for t in data["targets"]]
        if data.get("fields") is not None:
            targets = prepare_targets(net, targets, device, fields=data["fields"])
        losses, loss_dict, log_loss_dict = train_step(model, optimizer, images, targets, grad_norm_clip=clip)
        advance_schedule(iteration, warmup_iters, scheduler, warmup_scheduler)
        iteration += 1
        arguments["iteration"] = iteration
        if iteration % 10 == 0 or iteration == max_iter:
            loss_v = float(losses)
            logger.info("iter: %d  loss: %.4f  lr: %.8f  %.3f s/iter", iteration, loss_v, optimizer.param_groups[0]["lr"],
                        (time.time() - t0) / (iteration - start_iter))
        if comm.get_rank() == 0 and checkpointer is not None:
            if iteration % cfg.SOLVER.SAVE_CHECKPOINT_INTERVAL == 0:
                checkpointer.save("model_checkpoint", **arguments)
            if iteration == max_iter:
                checkpointer.save("model_final", **arguments)
        if data_loaders_val and cfg.SOLVER.EVAL_INTERVAL > 0 and iteration % cfg.SOLVER.EVAL_INTERVAL == 0:
            from .inference import inference
            import os
            for name, loader in zip(cfg.DATASETS.TEST, data_loaders_val):
                inference(model, loader, dataset_name=name, device=cfg.MODEL.DEVICE, metrics=cfg.TEST.METRIC,
                          output_folder=os.path.join(cfg.OUTPUT_DIR, "inference_{}".format(iteration), name))
            model.train()
            comm.synchronize()
    return loss_v
