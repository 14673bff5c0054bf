"""N>1 path on CPU: two processes over gloo exercise the sharding and the max-over-ranks timing reduction
bench.py uses (the data path itself has no collective: inference ranks are independent replicas)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoflex_amd import parallel, synthetic as S
    r, w, lr = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    first, count = parallel.image_shard(rank, world, global_batch=5)
    imgs = S.synthetic_images(count, 8, 16, seed=parallel.shard_seed(1000, rank, 3))
    parallel.barrier()
    rate, elapsed, total = parallel.aggregate_throughput(0.5 + rank, images_local=count)     # rank 1 is the slow one
    seen = parallel.ranks_seen()                                                               # the all-gather bench.py's line carries
    out.put((rank, first, count, float(imgs.sum()), rate, elapsed, total, seen))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, f0, c0, s0, rate0, el0, tot0, seen0), (_, f1, c1, s1, rate1, el1, tot1, seen1) = res
    assert seen0 == seen1 == [0, 1]                          # both ranks answered the all-gather
    assert (f0, c0, f1, c1) == (0, 3, 3, 2)                  # contiguous, disjoint, covers the 5 images
    assert s0 != s1                                          # disjoint synthetic streams
    assert el0 == el1 == 1.5 and tot0 == tot1 == 5           # max over ranks, sum over ranks
    assert abs(rate0 - 5 / 1.5) < 1e-9 and rate0 == rate1


def test_bench_line_order_puts_leg_scalars_and_ranks_before_the_long_blobs():
    """every leg's headline number is a top-level scalar, `ranks_seen` / `rccl_ranks` are in the line, and the per-family blob comes
    last (a truncated capture of the line keeps the numbers)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    res = {"metric": "m", "value": 1.0, "unit": "images/s", "n_gpus": 2, "steps": 3, "warmup": 1, "ms_per_step": 2.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "config": {"workload": "w"}, "roofline": {"frac": 0.5},
           "roofline_families": {"heads": {"x": 1}}, "cpu_baseline": {"value": 0.5}, "b32": {"value": 10.0}, "fp16x2_parity": {"value": 4.0},
           "fp32_parity": {"error": "boom"}, "train": {"value": 7.0, "ms_per_step": 18.0}}
    out = bench.order_line(res, [0, 1], 2)
    keys = list(out)
    assert keys[:13] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
    assert out["b32_images_per_s"] == 10.0 and out["fp16x2_images_per_s"] == 4.0 and out["train_ms_per_step"] == 18.0 and out["train_images_per_s"] == 7.0
    assert "fp32_images_per_s" not in out                    # a failed leg has no number; its error object stays
    assert out["fp32_parity"] == {"error": "boom"} and out["rccl_ranks"] == 2 and out["ranks_seen"] == [0, 1]
    assert keys[-1] == "roofline_families" and keys.index("b32_images_per_s") < keys.index("b32")
    assert set(res) <= set(out)


def test_single_process_is_passthrough():
    from monoflex_amd import parallel
    assert parallel.ranks_seen() == [0]
    assert parallel.image_shard(0, 1, 8) == (0, 8)
    rate, el, tot = parallel.aggregate_throughput(2.0, 16)
    assert (rate, el, tot) == (8.0, 2.0, 16)


class _Toy(torch.nn.Module):
    """Stand-in with the reference's dead-parameter names (backbone.base.level{3,4}.project.*) next to live ones."""

    def __init__(self):
        super().__init__()
        self.backbone = torch.nn.Module()
        self.backbone.base = torch.nn.Module()
        for lvl in ("level3", "level4"):
            holder = torch.nn.Module()
            holder.project = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1, bias=False), torch.nn.BatchNorm2d(4))
            setattr(self.backbone.base, lvl, holder)
        self.live = torch.nn.Linear(6, 3)

    def forward(self, x):
        return self.live(x).pow(2).sum()


def _dp_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoflex_amd import parallel
    from monoflex_amd.engine.trainer import dead_parameter_names, wrap_data_parallel
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    m = _Toy()
    assert len(dead_parameter_names(m)) == 6 + 6                     # 6 parameters + their BN buffers
    net = wrap_data_parallel(m)
    xs = torch.arange(12, dtype=torch.float32).view(2, 6) / 10       # rank r trains on row r
    grads = []
    for it in range(2):          # with the dead parameters merely unused (not ignored) DDP raises on the 2nd iteration
        m.zero_grad()
        net(xs[rank:rank + 1]).backward()
        grads.append(m.live.weight.grad.clone())
    out.put((rank, grads[1].tolist(), [p.grad is None for n, p in m.named_parameters() if "project" in n]))
    dist.destroy_process_group()


def test_training_data_parallel_ignores_dead_parameters_and_averages():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    torch.manual_seed(0)
    m = _Toy()
    xs = torch.arange(12, dtype=torch.float32).view(2, 6) / 10
    (m(xs[0:1]) + m(xs[1:2])).backward()
    want = m.live.weight.grad / 2                                    # DDP averages over ranks
    for rank, g, dead_none in res:
        assert torch.allclose(torch.tensor(g), want, atol=1e-6) and all(dead_none)


def _sampler_worker(rank, world, port, out):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from monoflex_amd.data.samplers import InferenceSampler, IterationBatchSampler, TrainingSampler, shared_random_seed
    torch.manual_seed(100 + rank)                                  # ranks have different local RNG states ...
    seed = shared_random_seed()                                    # ... and still agree on the seed
    tr = TrainingSampler(10)                                       # seed=None: agreed through the broadcast
    batches = list(IterationBatchSampler(TrainingSampler(10, seed=7), batch_size=3, num_iterations=4))
    out.put((rank, seed, list(__import__("itertools").islice(iter(tr), 15)), batches, list(InferenceSampler(7)), tr._seed))
    dist.destroy_process_group()


def test_samplers_shard_one_shared_stream_over_two_ranks():
    """distributed_sampler.py:43-54,193-196: rank r sees entries r, r+2, ... of one endless seeded permutation stream; the
    inference sampler cuts contiguous shards."""
    import itertools
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_sampler_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    (_, s0, t0, b0, i0, ts0), (_, s1, t1, b1, i1, ts1) = res
    assert s0 == s1 and ts0 == ts1
    merged = [v for pair in zip(t0, t1) for v in pair]             # interleaving the two ranks restores the stream
    g = torch.Generator(); g.manual_seed(ts0)
    stream = torch.randperm(10, generator=g).tolist() + torch.randperm(10, generator=g).tolist() + torch.randperm(10, generator=g).tolist()
    assert merged == stream[:30] and sorted(merged[:10]) == list(range(10))
    assert len(b0) == 4 and all(len(b) == 3 for b in b0) and not set(map(tuple, b0)) & set(map(tuple, b1))
    g.manual_seed(7)
    s7 = torch.randperm(10, generator=g).tolist() + torch.randperm(10, generator=g).tolist() + torch.randperm(10, generator=g).tolist()
    assert [v for b in b0 for v in b] == s7[0:24:2] and [v for b in b1 for v in b] == s7[1:24:2]
    assert i0 == [0, 1, 2, 3] and i1 == [4, 5, 6]


def test_samplers_single_process():
    from monoflex_amd.data.samplers import InferenceSampler, TrainingSampler
    import itertools
    assert list(InferenceSampler(5)) == [0, 1, 2, 3, 4] and len(InferenceSampler(5)) == 5
    assert list(itertools.islice(iter(TrainingSampler(4, shuffle=False, seed=0)), 9)) == [0, 1, 2, 3, 0, 1, 2, 3, 0]


class _ToyDetector(_Toy):
    """The training surface of KeypointDetector: model(images, targets) -> (loss_dict, log_loss_dict)."""

    def forward(self, images, targets=None):
        return {"a_loss": self.live(images).pow(2).sum(), "b_loss": self.live(images).sum() * 0.5}, {}


def _flat_step_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoflex_amd import parallel
    from monoflex_amd.engine.trainer import GraphedTrainStep
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    m = _ToyDetector()
    opt = torch.optim.SGD([p for n, p in m.named_parameters() if "project" not in n], lr=0.1)
    xs = torch.arange(12, dtype=torch.float32).view(2, 6) / 10
    step = GraphedTrainStep(m, opt, xs[rank:rank + 1].clone(), None, use_graphs=False)
    assert step.split and step.flat.numel() == 3 * 6 + 3                 # live.weight + live.bias only: dead ones left out
    w0 = m.live.weight.detach().clone()
    step()
    g1 = step.flat.clone()
    step.load_batch(xs[1 - rank:2 - rank])                               # swap the shards: the averaged gradient of a linear
    step()                                                               # model's batch does not depend on who holds which row
    out.put((rank, g1.tolist(), m.live.weight.detach().tolist(), w0.tolist(),
             [p.grad is None for n, p in m.named_parameters() if "project" in n]))
    dist.destroy_process_group()


def test_flat_gradient_exchange_of_the_graphed_step_over_two_ranks():
    """engine.trainer.GraphedTrainStep, split form, run eagerly on CPU over gloo: every live gradient lands in ONE flat fp32
    buffer, the chunked all-reduce averages it over the ranks, the optimizer reads it through views; the dead parameters are
    neither in the buffer nor touched."""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_flat_step_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    torch.manual_seed(0)
    m = _ToyDetector()
    xs = torch.arange(12, dtype=torch.float32).view(2, 6) / 10
    (sum(m(xs[0:1])[0].values()) + sum(m(xs[1:2])[0].values())).backward()
    want = torch.cat((m.live.weight.grad.flatten(), m.live.bias.grad.flatten())) / 2
    for rank, g1, w_after, w0, dead_none in res:
        assert torch.allclose(torch.tensor(g1), want, atol=1e-6) and all(dead_none)
    assert torch.allclose(torch.tensor(res[0][2]), torch.tensor(res[1][2]), atol=1e-7)                # ranks stay in lock step
    assert not torch.allclose(torch.tensor(res[0][2]), torch.tensor(res[0][3]))                     # and the weights moved


class _ToyStaged(torch.nn.Module):
    """Three stages with the cut protocol of KeypointDetector (set_backward_cuts / backward_segment_of / backward_thunks): the
    head's gradients are ready first (segment 0), the stem's last (segment 2); `dead` never receives a gradient."""
    BACKWARD_SEGMENTS = 3

    def __init__(self):
        super().__init__()
        self.stem, self.mid, self.head = torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 2)
        self._cut = None

    def backward_segment_of(self, name):
        return {"head": 0, "mid": 1, "stem": 2}[name.split(".")[0]]

    def set_backward_cuts(self, cut):
        self._cut = cut

    @staticmethod
    def backward_thunks(losses, cuts):
        def piece(name):
            def run():
                o, l = cuts[name]
                torch.autograd.backward(o, [x.grad for x in l])
            return run
        return [lambda: losses.backward(), piece("mid_out"), piece("stem_out")]

    def forward(self, images, targets=None):
        a = torch.tanh(self.stem(images))
        if self._cut is not None:
            a = self._cut("stem_out", [a])[0]
        b = torch.tanh(self.mid(a))
        if self._cut is not None:
            b = self._cut("mid_out", [b])[0]
        y = self.head(b)
        return {"a_loss": y.pow(2).sum(), "b_loss": y.sum() * 0.5}, {}


def _staged_step_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoflex_amd import parallel
    from monoflex_amd.engine.trainer import GraphedTrainStep
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    m = _ToyStaged()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    xs = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    step = GraphedTrainStep(m, opt, xs[2 * rank:2 * rank + 2].clone(), None, use_graphs=False)
    calls = []
    exchange = step._exchange

    def spy(k):
        # at the moment slice k is exchanged, the LATER pieces have not run: their parameters hold no gradient yet
        calls.append((k, [all(p.grad is None for p in step.seg_params[q]) for q in range(k + 1, step.nseg)]))
        exchange(k)
    step._exchange = spy
    step()
    names = [n for n, p in m.named_parameters()]
    order = [next(n for n, q in m.named_parameters() if q is p) for p in step.params]
    out.put((rank, step.flat.tolist(), step.seg_bounds, calls, order, step.overlap, {n: p.detach().tolist() for n, p in m.named_parameters()}, names))
    dist.destroy_process_group()


def test_segmented_gradient_exchange_over_two_ranks():
    """engine.trainer.GraphedTrainStep with a model that cuts its backward pass into pieces (the overlap form of the data-parallel
    step), run eagerly over gloo, world_size 2: the flat buffer is laid out in piece order (head | mid | stem) with contiguous,
    disjoint slices that cover it; slice k is exchanged right after piece k and BEFORE piece k+1 has produced anything; after the
    step the buffer holds the mean over the ranks of the gradients an uncut single-process backward computes; ranks stay in lock step."""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_staged_step_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    torch.manual_seed(0)
    m = _ToyStaged()
    xs = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    (sum(m(xs[0:2])[0].values()) + sum(m(xs[2:4])[0].values())).backward()             # no cuts: one autograd graph
    grads = {n: p.grad / 2 for n, p in m.named_parameters()}
    for rank, flat, bounds, calls, order, overlap, params, names in res:
        assert overlap and order == ["head.weight", "head.bias", "mid.weight", "mid.bias", "stem.weight", "stem.bias"]
        assert bounds == [(0, 10), (10, 34), (34, 69)] and len(flat) == 69
        assert [k for k, _ in calls] == [0, 1, 2] and all(all(later) for _, later in calls)
        want = torch.cat([grads[n].flatten() for n in order])
        assert torch.allclose(torch.tensor(flat), want, atol=1e-6)
    assert all(torch.allclose(torch.tensor(res[0][6][n]), torch.tensor(res[1][6][n]), atol=1e-7) for n in res[0][7])


def _scaled_step_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoflex_amd import parallel
    from monoflex_amd.engine.trainer import GraphedTrainStep, LossScaler
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    m = _ToyStaged()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-2, fused=True)
    scaler = LossScaler(torch.device("cpu"), init_scale=8.0, growth_interval=2)
    xs = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    step = GraphedTrainStep(m, opt, xs[2 * rank:2 * rank + 2].clone(), None, use_graphs=False, scaler=scaler)
    trace = []

    def snap():
        return {n: p.detach().clone() for n, p in m.named_parameters()}
    for it in range(4):
        if it == 1 and rank == 1:
            step.images[0, 0] = float("inf")                     # ONE rank's shard overflows on the second step
        if it == 2 and rank == 1:
            step.images.copy_(xs[2:4])
        before = snap()
        loss = float(step())
        after = snap()
        trace.append((it, loss, float(scaler.scale), float(scaler.found_inf), int(scaler.growth_tracker),
                      all(torch.equal(before[n], after[n]) for n in before), {n: v.tolist() for n, v in after.items()},
                      sorted(int(st["step"]) for st in opt.state.values())))
    out.put((rank, trace))
    dist.destroy_process_group()


def test_loss_scaler_takes_the_same_decision_on_every_rank():
    """fp16's dynamic loss scaling in the data-parallel step (engine.trainer.LossScaler inside GraphedTrainStep, eager over gloo,
    world_size 2): the gradients are checked AFTER the exchange, so an overflow in ONE rank's shard reaches both ranks -- both skip
    that step (parameters, AdamW counters untouched; scale halved), both apply the next ones, the scale grows back after
    `growth_interval` clean steps, and the ranks' parameters stay identical throughout."""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_scaled_step_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, t0), (_, t1) = res
    for a, b in zip(t0, t1):
        assert a[2:6] == b[2:6] and a[7] == b[7], (a[:6], b[:6])       # scale, found_inf, tracker, skipped?, AdamW counters: same on both ranks
        assert all(torch.equal(torch.tensor(a[6][n]), torch.tensor(b[6][n])) for n in a[6])         # replicas in lock step, bitwise
    # step 0 applied at 8; step 1 skipped (found_inf, scale 8 -> 4, parameters and counters untouched); steps 2, 3 applied, and the
    # second clean step in a row doubles the scale again
    assert [(t[2], t[3], t[4], t[5]) for t in t0] == [(8.0, 0.0, 1, False), (4.0, 1.0, 0, True), (4.0, 0.0, 1, False), (8.0, 0.0, 0, False)]
    assert [set(t[7]) for t in t0] == [{1}, {1}, {2}, {3}]


def test_bench_respawns_itself_under_torchrun_for_n_ranks(monkeypatch):
    """`python bench.py --gpus N` outside a torchrun environment starts N ranks on 127.0.0.1 (reference engine/launch.py:23-89)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--mode", "train"])
    args = bench.parse()
    assert bench.respawn_under_torchrun(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--mode", "train"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _gram_sync_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from monoflex_amd.gram_heads import _AllReduceSum
    torch.manual_seed(7)
    x = torch.randn(4, 6, 5, 3, dtype=torch.float64)                     # the GLOBAL batch; rank r owns images 2r, 2r+1
    w = torch.randn(8, 3, 3, 3, dtype=torch.float64)
    y_loc = torch.nn.functional.conv2d(x[2 * rank:2 * rank + 2].permute(0, 3, 1, 2), w, None, 1, 1)
    sums_loc = torch.cat((y_loc.sum((0, 2, 3)), (y_loc * y_loc).sum((0, 2, 3)))).requires_grad_()
    sums = _AllReduceSum.apply(sums_loc, dist.group.WORLD)
    coef = torch.arange(1, 17, dtype=torch.float64) * (rank + 1)            # every rank's loss weighs the global sums its own way
    (sums * coef).sum().backward()
    out.put((rank, sums.detach().numpy().tolist(), sums_loc.grad.numpy().tolist()))
    dist.destroy_process_group()


def test_gram_heads_statistics_all_reduce_is_differentiable_over_two_ranks():
    """monoflex_amd/gram_heads._AllReduceSum (the SyncBN form of the Gram-matrix heads' [sum y, sum y^2]): forward = the global-batch sums on every
    rank, backward = the SUM of the ranks' upstream gradients (each rank's loss depends on every rank's local sums)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gram_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    torch.manual_seed(7)
    x = torch.randn(4, 6, 5, 3, dtype=torch.float64)
    w = torch.randn(8, 3, 3, 3, dtype=torch.float64)
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, None, 1, 1)
    want = torch.cat((y.sum((0, 2, 3)), (y * y).sum((0, 2, 3))))
    gwant = torch.arange(1, 17, dtype=torch.float64) * 3.0                  # coef of rank 0 + coef of rank 1
    for _, sums, grad in res:
        assert torch.allclose(torch.tensor(sums, dtype=torch.float64), want, atol=1e-9)
        assert torch.allclose(torch.tensor(grad, dtype=torch.float64), gwant, atol=1e-12)


def _syncbn_gate_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from monoflex_amd import autograd as AG, parallel
    from monoflex_amd.engine.trainer import GraphedTrainStep
    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    m = _ToyStaged()
    m.bn = torch.nn.BatchNorm1d(4)                                       # (never run: the step only looks for the flag)
    m.bn.sync_bn = True
    for p in m.bn.parameters():
        p.requires_grad_(False)
    opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=0.1)
    xs = torch.arange(36, dtype=torch.float32).view(6, 6) / 10
    # a SyncBN model on two ranks with graphs REQUESTED and the captured collectives opted in, on a backend that cannot capture them:
    # the step must take its pieces eagerly and must not install a process-wide SyncBN communicator
    step = GraphedTrainStep(m, opt, xs[2 * rank:2 * rank + 2].clone(), None, use_graphs=True, graph_sync_bn=True)
    gated = (step.sync_bn, step.use_graphs, step.graph_sync_bn, step.bn_group is None, AG._SYNC_BN_GROUP[0] is None)
    step()
    flat_a = step.flat.clone()
    w_a = {n: p.detach().clone() for n, p in m.named_parameters()}
    # a batch of ANOTHER shape through the same object's buffers (the partial-last-batch fallback of do_train)
    flat_id = step.flat.data_ptr()
    step.eager_on(xs[4 + rank:5 + rank].clone(), None)
    same_buffer = step.flat.data_ptr() == flat_id and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(step.params, step.views))
    step.close()
    out.put((rank, gated, flat_a.tolist(), step.flat.tolist(), same_buffer, {n: p.detach().tolist() for n, p in m.named_parameters() if p.requires_grad},
             {n: v.tolist() for n, v in w_a.items() if m.get_parameter(n).requires_grad}))
    dist.destroy_process_group()


def test_sync_bn_step_without_opt_in_or_without_rccl_runs_its_pieces_eagerly_and_eager_on_shares_the_buffers():
    """ADVICE r4 (engine/trainer.py): (a) the captured SyncBN collectives are opt-in AND need the nccl backend -- a gloo group falls back to
    the eager pieces even when asked, installs no second communicator; (b) `eager_on` runs an odd-shaped batch through the SAME flat
    gradient buffer (no fresh step object per batch) and both ranks still meet: equal averaged gradients, equal parameters afterwards."""
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    ps = [ctx.Process(target=_syncbn_gate_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, gated, flat_a, flat_b, same_buffer, w_end, w_mid in res:
        assert gated == (True, False, False, True, True), gated
        assert same_buffer
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] and res[0][2] != res[0][3]           # averaged gradients agree on both steps
    assert res[0][5] == res[1][5] and res[0][6] == res[1][6] and res[0][5] != res[0][6]           # parameters in lock step, and they moved
    # the second step's gradient is the mean over the two ranks' single rows
    torch.manual_seed(0)
    m = _ToyStaged()
    m.load_state_dict({k: torch.tensor(v) for k, v in res[0][6].items()}, strict=False)
    xs = torch.arange(36, dtype=torch.float32).view(6, 6) / 10
    (sum(m(xs[4:5])[0].values()) + sum(m(xs[5:6])[0].values())).backward()
    by_name = dict(m.named_parameters())
    want = torch.cat([by_name[n].grad.flatten() for n in ("head.weight", "head.bias", "mid.weight", "mid.bias", "stem.weight", "stem.bias")]) / 2
    assert torch.allclose(torch.tensor(res[0][3]), want, atol=1e-6)
