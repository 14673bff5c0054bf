"""One optimisation step and its data-parallel wrapping (reference engine/trainer.py:101-126,
tools/plain_train_net.py:128-137).

Data parallelism: one process per GPU, gradients averaged by torch DDP's bucketed all-reduce over RCCL, overlapped
with the backward kernels.  The reference wraps with find_unused_parameters=True because six parameters (the outer
level3/level4 `project` conv + BN, dead in DLA's Tree.forward) never receive a gradient, which makes DDP walk the
autograd graph every iteration; here those six are excluded statically, so no per-iteration graph traversal happens
and no bucket ever waits for them."""
import os

import torch

from ..solver import finish_capture, optimizer_step

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
    capture): `.edge` = (edge_indices, edge_lens) for the predictor, `.loss` = (heat maps, stacked fields) for the loss, `.edge_plan` = the
    edge fusion's index tensors (predictor.edge_plan: functions of the targets alone; None without edge fusion)."""
    edge_plan = None
    arena = None                   # all device tensors above as views of one byte buffer (pack_target_arena), or None
    arena_layout = None


def prepare_targets(model, targets, device, fields=None):
    """Per-image targets -> PreparedTargets. `fields`: the batch-stacked device tensors a DeviceLoader batch carries
    (`batch["fields"]`, produced by the target-encoding kernels): with them nothing is re-stacked per image."""
    from ..model.head.detector_predictor import stack_edge_fields
    m = model.module if hasattr(model, "module") else model
    pt = PreparedTargets(targets)
    if fields is None:
        pt.edge = stack_edge_fields(targets, device)
        pt.loss = m.heads.loss_evaluator.prepare_targets(targets, device)
        pt.edge_plan = _edge_plan(m, pt)
        return pack_target_arena(pt)
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
    pt.edge_plan = _edge_plan(m, pt)
    return pack_target_arena(pt)


def total_loss(loss_dict):
    """The step's scalar loss: the sum of the weighted terms (trainer.py:111).  The fused loss hands it over ready-made (`loss_dict.total`: two
    launches instead of the ten adds of a Python `sum` and their ten backward nodes)."""
    t = getattr(loss_dict, "total", None)
    return t if t is not None else sum(loss_dict.values())


def _edge_plan(m, pt):
    pr = getattr(m.heads, "predictor", None)
    if pr is None or not (getattr(pr, "enable_edge_fusion", False) and getattr(pr, "fused_edge_nodes", False)) or not pt.edge[0].is_cuda:
        return None
    with torch.no_grad():
        return pr.edge_plan(pt.edge[0], pt.edge[1], pt.loss[1].get("object_rows"))


class LossScaler:
    """Dynamic loss scaling for fp16 training (COMPUTE_DTYPE = "fp16"), entirely on the device, so a captured step carries it:

        losses * scale -> backward -> [gradient exchange] -> gradients / scale + non-finite check (one foreach launch)
        -> [clip] -> fused AdamW, which reads `found_inf` and leaves parameters, moments and step counters untouched when it is set
        -> scale *= backoff if found_inf else (scale *= growth every `growth_interval` clean steps)

    fp16 keeps three more mantissa bits than bf16 but only five exponent bits: activation gradients below 6e-5 lose precision and
    below 6e-8 vanish, above 65504 they overflow, so the loss is scaled into range and the scale follows the run (the arithmetic of
    torch.amp.GradScaler, without its host synchronisation).  Data parallel: the check runs on the all-reduced gradients, where an
    inf/nan of any rank has reached every rank, so all ranks take the same decision.  bf16 / fp32 training needs none of this.

    Range of this network (tools/probes/fp16_range_probe.py, random init, B = 8 at 1280 x 384, scale 1): the gradients of the 16-bit
    maps grow from ~1e-7 (median of the dense head-trunk gradients: at fp16's smallest subnormal) to ~1e2 (full-resolution DLA levels)
    on their way down the network, 30 of fp16's 40 binades; scales 2^5 .. 2^9 keep both ends.  The default start (2^8) needs no
    back-off there; a start that is too high costs one skipped step per halving."""

    def __init__(self, device, init_scale=2.0 ** 8, growth_factor=2.0, backoff_factor=0.5, growth_interval=1000):
        self.scale = torch.full((), float(init_scale), dtype=torch.float32, device=device)
        self.growth_tracker = torch.zeros((), dtype=torch.int32, device=device)
        self.found_inf = torch.zeros((), dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)

    @staticmethod
    def for_model(model, device=None):
        """A scaler when the model computes in fp16, else None."""
        net = model.module if hasattr(model, "module") else model
        if getattr(net, "compute_dtype", None) != torch.float16:
            return None
        dev = device if device is not None else next(p.device for p in net.parameters())
        return LossScaler(dev)

    def attach(self, optimizer):
        """The fused AdamW step skips itself while `found_inf` is set (torch.optim's AMP hook)."""
        optimizer.found_inf = self.found_inf
        return self

    def scale_loss(self, losses):
        return losses * self.scale

    def unscale_(self, grads):
        self.found_inf.zero_()
        torch._amp_foreach_non_finite_check_and_unscale_(list(grads), self.found_inf, self.scale.reciprocal())

    def update(self):
        torch._amp_update_scale_(self.scale, self.growth_tracker, self.found_inf, self.growth_factor, self.backoff_factor, self.growth_interval)

    def state_dict(self):
        return {"scale": float(self.scale), "growth_tracker": int(self.growth_tracker)}

    def load_state_dict(self, sd):
        self.scale.fill_(float(sd["scale"]))
        self.growth_tracker.fill_(int(sd.get("growth_tracker", 0)))


class GraphedTrainStep:
    """One optimisation step replayed from hipGraphs, single GPU or data-parallel, with the gradient exchange OVERLAPPED with
    the backward pass (reference: DDP's bucket hooks inside `losses.backward()`, tools/plain_train_net.py:134-137,
    engine/trainer.py:117).

    torch DDP's bucket hooks cannot be captured, and the eager step costs tens of ms of host enqueue (hundreds of small
    launches), so the data-parallel form is built from stream-ordered pieces.  The model cuts its forward pass at segment
    boundaries (`model.set_backward_cuts`, KeypointDetector: heads + IDAUp | DLAUp | level5/4 | level3..stem), which splits
    `losses.backward()` into K pieces; the parameters are laid out in ONE flat fp32 gradient buffer in the order their pieces
    finish, so every piece owns a contiguous slice:

        graph 0      pack weights -> forward (with cuts) -> loss -> backward piece 0 -> its gradients into slice 0
        graph k      backward piece k -> slice k                                        (k = 1 .. K-1)
        after graph k is enqueued: the comm stream waits for it and all-reduces slice k over RCCL (83.8 MB fp32 in total over
                     7 xGMI links x 153 GB/s) WHILE graph k+1 runs on the compute stream
        graph B      waits for the K collectives, [gradient clipping,] fused AdamW reading the gradients as views of the buffer

    Only the last (smallest: level3..stem, 5 MB) slice's collective is exposed.  world_size 1 captures the whole step as one
    graph (no cuts, no flat copy, no collective) unless `split=True`.  A model without the cut protocol is one piece.  The six
    dead parameters (`dead_parameter_names`) never get a gradient and are left out of the buffer.  Static inputs: the image batch
    and the PreparedTargets given at construction; `load_batch` overwrites them in place between replays.  The learning rate is a
    device scalar (solver.build_optimizer(capturable=True)), so schedulers keep working between replays.  `use_graphs=False` runs
    the same pieces eagerly (CPU / gloo tests of the segmented exchange).

    SyncBN (tools/plain_train_net.py:131-132, runs/monoflex.yaml:59 USE_SYNC_BN True) stays on this fast path: a synchronised BN layer
    is statistics kernel -> all-reduce of [sum, sum of squares, rows] -> finalize + apply (and reduce -> all-reduce of [sum g, sum g.xhat]
    -> apply on the way back), and those 2 x 57 small collectives are CAPTURED with the kernels around them -- RCCL collectives replay from
    a hipGraph -- so they cost their latency on the device and nothing on the host.  Each is a true data dependency of the next layer
    (layer l+1 reads the output normalised with layer l's global statistics), so they cannot be batched; they run on a communicator of
    their own (`autograd.set_sync_bn_group`), because the gradient slices are all-reduced on the default communicator in host order
    between graph launches."""

    def __init__(self, model, optimizer, images, targets, group=None, comm_chunks=None, warmup=3, split=None, use_graphs=None,
                 grad_norm_clip=-1.0, scaler="auto", graph_sync_bn=None):
        import torch.distributed as dist
        self.model, self.optimizer, self.images, self.targets = model, optimizer, images, targets
        self.net = model.module if hasattr(model, "module") else model
        self.scaler = LossScaler.for_model(self.net, images.device) if scaler == "auto" else scaler      # fp16 activations: dynamic loss scaling
        if self.scaler is not None:
            self.scaler.attach(optimizer)
        self.dist_on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist_on else 1
        self.group = group
        self.split = (self.world > 1) if split is None else bool(split)
        self.use_graphs = images.is_cuda if use_graphs is None else bool(use_graphs)
        self.grad_norm_clip = float(grad_norm_clip)
        self.graphs, self.graph_b, self.flat = [], None, None
        self.graph_a = None                                            # (kept: the first captured graph)
        self.overlap = False
        self.nseg = 1
        self.sync_bn = self.world > 1 and any(getattr(mod, "sync_bn", False) or isinstance(mod, torch.nn.SyncBatchNorm) for mod in self.net.modules())
        self.bn_group = None
        # SyncBN on more than one rank: the CAPTURED form (statistics all-reduces inside the hipGraphs on a second communicator while the
        # host all-reduces gradient slices on the default one) has only ever run on a one-rank RCCL group inside this build's sessions (one GPU
        # per box).  Until a multi-GPU run has passed it is OPT-IN -- graph_sync_bn=True or MFX_GRAPH_SYNC_BN=1 -- and needs the nccl
        # backend (a gloo collective cannot be captured); otherwise a SyncBN model takes the same pieces eagerly (use_graphs False: every
        # collective on the default communicator, in stream order).  bench.py's `train` legs opt in (they run behind a watchdog, after the
        # `train_local_bn` leg that has no collective inside its graphs).
        if graph_sync_bn is None:
            graph_sync_bn = os.environ.get("MFX_GRAPH_SYNC_BN", "0") == "1"
        self.graph_sync_bn = False
        if self.sync_bn and self.use_graphs:
            if graph_sync_bn and dist.get_backend(group) == "nccl":
                from .. import autograd as AG
                self.bn_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None)    # (collective: every rank constructs the step)
                AG.set_sync_bn_group(self.bn_group)
                self.graph_sync_bn = True
            else:
                self.use_graphs = False
        if self.split:
            dead = set(dead_parameter_names(self.net))
            named = [(n, p) for n, p in self.net.named_parameters() if p.requires_grad and n not in dead]
            seg_of = getattr(self.net, "backward_segment_of", None)
            if seg_of is not None and hasattr(self.net, "set_backward_cuts"):
                self.nseg = int(self.net.BACKWARD_SEGMENTS)
                named.sort(key=lambda np_: seg_of(np_[0]))             # stable: named_parameters() order inside a segment
                segs = [seg_of(n) for n, _ in named]
            else:
                segs = [0] * len(named)
            self.params = [p for _, p in named]
            self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=images.device)
            self.views, o = [], 0
            self.seg_params, self.seg_views = [[] for _ in range(self.nseg)], [[] for _ in range(self.nseg)]
            sizes = [0] * self.nseg
            for (n, p), k in zip(named, segs):                         # (sorted by segment: every segment is one contiguous slice)
                v = self.flat[o:o + p.numel()].view_as(p)
                self.views.append(v)
                self.seg_params[k].append(p)
                self.seg_views[k].append(v)
                sizes[k] += p.numel()
                o += p.numel()
            self.seg_bounds, b = [], 0
            for k in range(self.nseg):
                self.seg_bounds.append((b, b + sizes[k]))
                b += sizes[k]
            self.overlap = self.nseg > 1
            self.comm = torch.cuda.Stream(device=images.device) if images.is_cuda else None
            self._works = []
        if not self.use_graphs:
            return
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("GraphedTrainStep: build the optimizer with capturable=True (solver.build_optimizer)")
            if not torch.is_tensor(g["lr"]):
                raise ValueError("GraphedTrainStep: the learning rate must be a device scalar (solver.build_optimizer(capturable=True)); "
                                 "a Python float would be baked into the captured optimizer step")
        # The capture needs warm-up runs (lazy allocations, the optimizer's state tensors, RCCL's first-call setup), and those are
        # real optimisation steps on batch 0.  They must not count: the state they touch (parameters, BN buffers incl. running
        # statistics and num_batches_tracked, AdamW moments and step counters, the loss scale) is snapshotted here and written back
        # IN PLACE after the capture, so the captured graphs keep their addresses and the first replay is step 1 of the run (or step
        # k+1 of a resumed one) -- a run with the captured step follows the trajectory of the reference's loop (engine/trainer.py:103-126).
        snap = self._snapshot_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(2, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            import gc
            gc.collect()                                       # (models of earlier steps that only a reference cycle kept alive: gone before the tables are built)
            self._pack_scope = None
            if images.is_cuda:
                from .. import autograd as AG
                self._pack_scope = AG.pack_scope(self.net.parameters())      # this model's operands only, tables owned by this step (AG.pack_scope)
                with self._pack_scope:
                    self._capture()
            else:
                self._capture()
            finish_capture(self.optimizer)                    # (the one-launch AdamW's pointer tables: solver.MultiTensorAdamW)
        finally:
            self._restore_state(snap)
            torch.cuda.synchronize()

    def _snapshot_state(self):
        with torch.no_grad():
            return {"params": [p.detach().clone() for p in self.net.parameters()],
                    "buffers": [b.detach().clone() for b in self.net.buffers()],
                    "opt": {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                            for p, st in self.optimizer.state.items()},
                    "scaler": None if self.scaler is None else (self.scaler.scale.clone(), self.scaler.growth_tracker.clone())}

    def _restore_state(self, snap):
        with torch.no_grad():
            for p, v in zip(self.net.parameters(), snap["params"]):
                p.copy_(v)
            for b, v in zip(self.net.buffers(), snap["buffers"]):
                b.copy_(v)
            for p, st in self.optimizer.state.items():
                old = snap["opt"].get(id(p))
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if old is not None and torch.is_tensor(old.get(k)):
                            v.copy_(old[k])
                        else:
                            v.zero_()                       # created by the warm-up: AdamW's initial state (zero moments, step 0)
            if self.scaler is not None:
                self.scaler.scale.copy_(snap["scaler"][0])
                self.scaler.growth_tracker.copy_(snap["scaler"][1])
                self.scaler.found_inf.zero_()

    def _capture(self):
        g0 = torch.cuda.CUDAGraph()
        self.graph_a = g0
        if not self.split:
            with torch.cuda.graph(g0, capture_error_mode="thread_local"):     # (an RCCL watchdog thread may poll events meanwhile)
                self.loss = self._fwd_bwd()
                self._update()
            self.graphs = [g0]
            return
        thunks = None
        for k in range(self.nseg):
            gk = g0 if k == 0 else torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, pool=None if k == 0 else g0.pool(), capture_error_mode="thread_local"):
                if k == 0:
                    self.loss, thunks = self._forward_cut()
                self._arena_piece = k
                with self._side_wgrads():
                    thunks[k]()
                self._arena_piece = 0
                self._flatten(k)
            self.graphs.append(gk)
        self._point_grads_at_views()
        self.graph_b = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_b, pool=g0.pool(), capture_error_mode="thread_local"):
            self._update()

    # ---- pieces ---------------------------------------------------------------------------------------------------
    def _pack(self):
        if self.images.is_cuda:
            from .. import autograd as AG
            AG.pack_all_weights()

    def _side_wgrads(self):
        """Context: weight gradients of the backward passes inside run on the library's side stream (autograd.on_wgrad_stream), joined when each pass ends."""
        from .. import autograd as AG

        class _Ctx:
            def __enter__(self_):
                self_.prev = AG.WGRAD_SIDE[0]
                AG.WGRAD_SIDE[0] = bool(self.images.is_cuda)
                # ... and the bias-gradient sums of the pass come from one arena zeroed here.  One arena state per STEP: the pieces of a split
                # backward continue in it (`begin` only in front of the first)
                self_.arena = None
                if self.images.is_cuda and getattr(self, "_arena_piece", 0) == 0:
                    self_.arena = AG.sum_arena(self.images.device)
                    self_.arena.__enter__()
                elif self.images.is_cuda:
                    self_.was_on, AG.SUM_ARENA.on = AG.SUM_ARENA.on, True

            def __exit__(self_, *a):
                AG.WGRAD_SIDE[0] = self_.prev
                if self_.arena is not None:
                    self_.arena.__exit__(*a)
                elif self.images.is_cuda:
                    AG.SUM_ARENA.on = self_.was_on
        return _Ctx()

    def _fwd_bwd(self):
        """Whole step in one autograd graph (single-GPU form)."""
        self._pack()
        loss_dict, _ = self.model(self.images, self.targets)
        losses = total_loss(loss_dict)
        self.optimizer.zero_grad(set_to_none=True)
        with self._side_wgrads():
            self._scaled(losses).backward()
        return losses.detach()

    def _scaled(self, losses):
        return losses if self.scaler is None else self.scaler.scale_loss(losses)

    def _update(self):
        """[unscale + non-finite check,] [clip,] optimizer step [, loss-scale update] on the step's final gradients."""
        if self.scaler is not None:
            self.scaler.unscale_([self.flat] if self.flat is not None else [p.grad for p in self.net.parameters() if p.grad is not None])
        self._clip()
        optimizer_step(self.optimizer)
        if self.scaler is not None:
            self.scaler.update()

    def _forward_cut(self):
        """Forward pass with the model's gradient cuts -> (detached total loss, the K backward pieces in execution order)."""
        self._pack()
        self.optimizer.zero_grad(set_to_none=True)
        if self.nseg == 1:
            loss_dict, _ = self.model(self.images, self.targets)
            losses = total_loss(loss_dict)
            scaled = self._scaled(losses)
            return losses.detach(), [lambda: scaled.backward()]
        cuts = {}

        def cut(name, tensors):
            leaves = [t.detach().requires_grad_(t.requires_grad) for t in tensors]
            cuts[name] = (list(tensors), list(leaves))         # (copies: IDAUp overwrites the entries of the list it is handed)
            return leaves
        self.net.set_backward_cuts(cut)
        try:
            loss_dict, _ = self.model(self.images, self.targets)
        finally:
            self.net.set_backward_cuts(None)
        losses = total_loss(loss_dict)
        return losses.detach(), self.net.backward_thunks(self._scaled(losses), cuts)

    def _flatten(self, k):
        params, views = self.seg_params[k], self.seg_views[k]
        missing = [i for i, p in enumerate(params) if p.grad is None]
        if missing:
            raise RuntimeError("GraphedTrainStep: %d live parameters of backward piece %d received no gradient (first index %d)"
                               % (len(missing), k, missing[0]))
        if params:
            torch._foreach_copy_(views, [p.grad for p in params])

    def _point_grads_at_views(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def _clip(self):
        if self.grad_norm_clip > 0:                                       # (device-side: total norm, clamp and scale without a host sync)
            torch.nn.utils.clip_grad_norm_([p for p in self.net.parameters() if p.grad is not None], self.grad_norm_clip, foreach=True)

    def _exchange(self, k):
        """All-reduce (mean) of slice k.  GPU: enqueued on the comm stream behind everything the compute stream holds so far, so it
        runs while the next backward piece computes; `_finish_exchange` makes the compute stream wait for all of them."""
        if not self.dist_on:
            return
        import torch.distributed as dist
        b, e = self.seg_bounds[k]
        if e <= b:
            return
        sl = self.flat[b:e]
        if self.comm is None:                                             # CPU / gloo
            dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                sl.mul_(1.0 / self.world)
            return
        self.comm.wait_stream(torch.cuda.current_stream())
        avg = dist.get_backend(self.group) == "nccl"              # RCCL averages in the collective; other backends (gloo on device
        with torch.cuda.stream(self.comm):                         # tensors: tests) sum, and the buffer is scaled once at the end
            self._works.append(dist.all_reduce(sl, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._scale_after = not avg

    def _finish_exchange(self):
        if self.dist_on and self.comm is not None:
            for w in self._works:
                w.wait()                                                  # the compute stream waits for the collective; the host does not
            self._works = []
            torch.cuda.current_stream().wait_stream(self.comm)
            if getattr(self, "_scale_after", False) and self.world > 1:
                self.flat.mul_(1.0 / self.world)

    def _eager(self):
        if not self.split:
            loss = self._fwd_bwd()
            self._update()
            return loss
        loss, thunks = self._forward_cut()
        for k, t in enumerate(thunks):
            self._arena_piece = k
            with self._side_wgrads():
                t()
            self._arena_piece = 0
            self._flatten(k)
            self._exchange(k)
        self._finish_exchange()
        self._point_grads_at_views()
        self._update()
        return loss

    def eager_on(self, images, targets):
        """The same step run EAGERLY on a batch of another shape (a partial last batch of a data-parallel run) with this object's flat
        gradient buffer, views, scaler and communicator: exactly the collectives of a replay, so ranks that replay and ranks that fall back
        still meet.  (the fallback used to build a fresh GraphedTrainStep -- an 84 MB buffer and a re-attached scaler -- per batch.)"""
        if not self.split:
            raise RuntimeError("eager_on: only the segmented (data-parallel) step has an eager form that shares its buffers")
        keep = (self.images, self.targets)
        self.images, self.targets = images, targets
        try:
            self.loss = self._eager()
        finally:
            self.images, self.targets = keep
        return self.loss

    def close(self, destroy=True):
        """Detach and destroy the SyncBN communicator this step installed (autograd.set_sync_bn_group).  The global is process-wide: it is cleared
        only while it still IS this object's group -- a newer step (re-capture, a second GraphedTrainStep in a bench / test) that installed its own
        must not lose it when the older object is collected."""
        group, self.bn_group = getattr(self, "bn_group", None), None
        if group is None:
            return
        from .. import autograd as AG
        if AG._SYNC_BN_GROUP[0] is group:
            AG.set_sync_bn_group(None)
        if not destroy:                                                   # (garbage collection runs at different times on different ranks: only an explicit close() tears the communicator down)
            return
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                dist.destroy_process_group(group)
        except Exception:                                                 # noqa: BLE001  (backend already torn down)
            pass

    def __del__(self):                                                    # identity-guarded (see close): collecting an old step never touches a newer one's group
        try:
            self.close(destroy=False)
        except Exception:                                                 # noqa: BLE001  (interpreter shutdown)
            pass

    def load_batch(self, images, targets=None):
        self.images.copy_(images, non_blocking=True)
        if targets is not None:
            da, sa = getattr(self.targets, "arena", None), getattr(targets, "arena", None)
            if da is not None and sa is not None and da.device == sa.device and self.targets.arena_layout == targets.arena_layout:
                da.copy_(sa, non_blocking=True)                    # every target tensor of the batch in one copy (pack_target_arena)
                return
            dsts, srcs = _target_tensors(self.targets), _target_tensors(targets)
            if dsts and dsts[0].is_cuda and all(s.is_cuda and s.dtype == d_.dtype and s.shape == d_.shape for s, d_ in zip(srcs, dsts)):
                # one multi-tensor copy per dtype instead of ~20 device-to-device memcpy launches (8 us of host gap each in front of every replay)
                torch._foreach_copy_(dsts, srcs)
            else:
                for dst, src in zip(dsts, srcs):
                    dst.copy_(src, non_blocking=True)

    def __call__(self):
        if not self.use_graphs:
            self.loss = self._eager()
            return self.loss
        if self.graph_b is None:
            self.graphs[0].replay()
            return self.loss
        for k, g in enumerate(self.graphs):
            g.replay()
            self._exchange(k)
        self._finish_exchange()
        self.graph_b.replay()
        return self.loss


def _clone_targets(pt):
    """A PreparedTargets whose device tensors are private copies (the captured graphs keep reading them), packed in one arena."""
    out = PreparedTargets(list(pt))
    out.edge = tuple(t.clone() for t in pt.edge)
    out.loss = (pt.loss[0].clone(), {k: (v.clone() if torch.is_tensor(v) else v) for k, v in pt.loss[1].items()})
    if getattr(pt, "edge_plan", None) is not None:
        out.edge_plan = {k: v.clone() for k, v in pt.edge_plan.items()}
    return pack_target_arena(out)


def _target_tensors(pt):
    out = list(pt.edge) + [pt.loss[0]]
    out += [v for k, v in sorted(pt.loss[1].items()) if torch.is_tensor(v)]
    if getattr(pt, "edge_plan", None) is not None:
        out += [v for k, v in sorted(pt.edge_plan.items())]
    return out


def _set_target_tensors(pt, ts):
    """Inverse of `_target_tensors`: put `ts` (same order) back into the PreparedTargets."""
    ts = list(ts)
    n = len(pt.edge)
    pt.edge = tuple(ts[:n])
    heat, d = ts[n], dict(pt.loss[1])
    i = n + 1
    for k in sorted(d):
        if torch.is_tensor(d[k]):
            d[k] = ts[i]
            i += 1
    pt.loss = (heat, d)
    if getattr(pt, "edge_plan", None) is not None:
        pt.edge_plan = dict(zip(sorted(pt.edge_plan), ts[i:]))


TARGET_ARENA = [os.environ.get("MFX_TARGET_ARENA", "1") != "0"]      # 0: every target tensor its own allocation (A/B)


def pack_target_arena(pt):
    """Move every device tensor of a PreparedTargets into ONE byte buffer (`pt.arena`; the tensors become views at 16-byte aligned offsets), so that
    handing a batch to a captured step (GraphedTrainStep.load_batch) is one device-to-device copy instead of one per tensor -- 27 copies of a
    few hundred bytes, ~9 us each with their gaps, in front of every replay (r06: 250 us of the 17.9 ms step).  `pt.arena_layout` identifies the layout."""
    ts = [t.contiguous() for t in _target_tensors(pt)]
    if not TARGET_ARENA[0] or not ts or any(t.device != ts[0].device for t in ts):
        return pt
    offs, o = [], 0
    for t in ts:
        o = (o + 15) // 16 * 16
        offs.append(o)
        o += t.numel() * t.element_size()
    arena = torch.zeros(max(o, 16), dtype=torch.uint8, device=ts[0].device)
    views = []
    for t, off in zip(ts, offs):
        v = arena[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
        v.copy_(t)
        views.append(v)
    _set_target_tensors(pt, views)
    pt.arena = arena
    pt.arena_layout = tuple((off, str(t.dtype), tuple(t.shape)) for t, off in zip(ts, offs))
    return pt


def train_step(model, optimizer, images, targets, grad_norm_clip=-1.0, scheduler=None, scaler=None):
    """trainer.py:109-126: forward -> summed loss -> zero_grad -> backward (+DDP all-reduce) -> clip -> step.
    `scaler`: a LossScaler (attached to `optimizer`) when the model computes in fp16."""
    if images.is_cuda:
        from .. import autograd as AG
        AG.pack_all_weights()                                   # every conv operand of the step from the current parameters, one launch
    loss_dict, log_loss_dict = model(images, targets)
    losses = total_loss(loss_dict)
    optimizer.zero_grad(set_to_none=True)
    (losses if scaler is None else scaler.scale_loss(losses)).backward()
    if scaler is not None:
        scaler.unscale_([p.grad for p in model.parameters() if p.grad is not None])
    if grad_norm_clip > 0:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], grad_norm_clip)
    optimizer_step(optimizer)
    if scaler is not None:
        scaler.update()
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
    # The fast step: when the loader hands over device-encoded batches of one static shape (DeviceLoader: `fields`), the optimizer
    # was built capturable (solver.build_optimizer does that on a GPU), the whole step is replayed from hipGraphs (GraphedTrainStep:
    # ~3.5x less wall time than the eager launches, gradient exchange overlapped with backward, SyncBN's statistics collectives
    # captured); every later batch is copied into the captured buffers.  Anything else (DDP-wrapped model, CPU tensors, ragged
    # shapes) takes the eager step, the reference's literal loop.
    graphed, graphed_shapes = None, None
    scaler = LossScaler.for_model(net, device) if torch.device(device).type == "cuda" else None     # fp16 activations: dynamic loss scaling
    if scaler is not None:
        scaler.attach(optimizer)
        if arguments.get("loss_scaler"):                             # resumed run: continue from the checkpointed scale
            scaler.load_state_dict(arguments["loss_scaler"])
    want_graph = bool(cfg.SOLVER.get("GRAPHED_STEP", True)) and not hasattr(model, "module") \
        and all(g.get("capturable", False) and torch.is_tensor(g["lr"]) for g in optimizer.param_groups)
    if want_graph and hasattr(net, "heads") and hasattr(net.heads, "loss_evaluator"):
        net.heads.loss_evaluator.log_as_float = False                   # (logged values stay device scalars: no host sync inside the step)
    for data, iteration in zip(data_loader, range(start_iter, max_iter)):
        images = data["images"].to(device) if hasattr(data["images"], "to") else data["images"]
        targets = [t.to(device) for t in data["targets"]]
        if data.get("fields") is not None:
            targets = prepare_targets(net, targets, device, fields=data["fields"])
        img_t = getattr(images, "tensors", images)               # (an ImageList from the loader, or a plain batch tensor)
        use_graph = want_graph and isinstance(targets, PreparedTargets) and torch.is_tensor(img_t) and img_t.is_cuda
        if use_graph:
            shapes = (tuple(img_t.shape),) + tuple(tuple(t.shape) for t in _target_tensors(targets))
            if graphed is None:
                graphed = GraphedTrainStep(model, optimizer, img_t.clone(), _clone_targets(targets), grad_norm_clip=clip, scaler=scaler)
                graphed_shapes = shapes
                logger.info("training step: %s (%d graph(s), overlap %s, SyncBN %s)", "captured as hipGraphs" if graphed.use_graphs else "eager pieces",
                            len(graphed.graphs) + (graphed.graph_b is not None), graphed.overlap,
                            "off" if not graphed.sync_bn else ("captured collectives" if graphed.graph_sync_bn else "eager collectives (MFX_GRAPH_SYNC_BN=1 captures them)"))
            if shapes == graphed_shapes:
                graphed.load_batch(img_t, targets)
                losses = graphed()
            elif graphed.split:
                # a batch of another shape (a partial last batch) in a data-parallel run: the model is not DDP-wrapped on this path, so
                # the reference's eager step would skip the gradient exchange and the ranks would drift apart silently.  The same
                # segmented step runs eagerly instead -- cut backward, flat gradient buffer, one all-reduce per piece -- which issues
                # exactly the collectives the captured step issues, so ranks that replay and ranks that fall back still meet
                # (the loaders of this build hand every rank the same batch shape at the same iteration: data/samplers.py).
                losses = graphed.eager_on(img_t, targets)
            else:                                               # single process: the reference's eager step
                losses, _, _ = train_step(model, optimizer, images, targets, grad_norm_clip=clip, scaler=scaler)
        else:
            losses, loss_dict, log_loss_dict = train_step(model, optimizer, images, targets, grad_norm_clip=clip, scaler=scaler)
        advance_schedule(iteration, warmup_iters, scheduler, warmup_scheduler)
        iteration += 1
        arguments["iteration"] = iteration
        if iteration % 10 == 0 or iteration == max_iter:
            loss_v = float(losses)
            from .. import lib as L
            if torch.cuda.is_available() and not L.bn_onepass_ok():
                raise RuntimeError("a one-launch BatchNorm timed out at its grid barrier (another process is computing on this device?): "
                                   "results since the last check are invalid -- rerun with MFX_OPTIONS=bn_onepass=0")
            logger.info("iter: %d  loss: %.4f  lr: %.8f  %.3f s/iter", iteration, loss_v, optimizer.param_groups[0]["lr"],
                        (time.time() - t0) / (iteration - start_iter))
        if comm.get_rank() == 0 and checkpointer is not None:
            if scaler is not None and (iteration % cfg.SOLVER.SAVE_CHECKPOINT_INTERVAL == 0 or iteration == max_iter):
                arguments["loss_scaler"] = scaler.state_dict()             # (travels with the checkpoint's `arguments`)
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
    if graphed is not None:
        graphed.close()                                              # (the SyncBN communicator it may have installed process-wide)
    return loss_v
