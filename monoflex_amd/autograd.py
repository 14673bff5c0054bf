"""Differentiable NHWC operators for the training path (reference: torch autograd over nn.Conv2d, BatchNorm2d,
MaxPool2d, ConvTranspose2d and the `_DCNv2` Function, engine/trainer.py:109-117).

torch.autograd only records the graph and routes gradients; every forward and backward below runs in
libmonoflex_hip.so.  Activations are NHWC (B,H,W,C) CUDA tensors; parameters keep the reference's shapes
(conv OIHW, BN vectors, deconv (C,1,k,k)), so optimizers and checkpoints see the usual tensors.

Gradient recipes (DESIGN.md section 4.6 has the kernels):
  conv      dx = conv(dy [zero-inserted when stride 2], W flipped + in/out swapped)   (the forward MFMA kernels)
            dW = mfx_conv_wgrad_oihw (LDS-patch / transposed-read / slab kernels), stem: mfx_stem_wgrad_bf16; db = mfx_colsum
            operands packed once per step for every parameter: pack_all_weights() -> mfx_pack_conv_weights_batched
  BN+act    mfx_bn_train_fwd / mfx_bn_train_bwd (two launches each; statistics optionally from the producing conv's epilogue:
            conv2d_bn_stats); with SyncBN: mfx_bn_stats -> all-reduce -> mfx_bn_finalize -> mfx_bn_act_fwd and the split backward
  DCNv2     mfx_dcn_nhwc / mfx_dcn_backward_v2 (tile-owned grad_input; offsets, mask, weight, bias gradients)
  heads     dense: conv + BN + conv1x1; regression branches in training: SparseRegHeadsFn (object centres only)
  loss      FocalLossFn (heat map), ObjectLossFn (the nine per-object terms, forward-mode gradient rows)
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as L
from . import ops
from .ops import _dt, _ptr, _stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _device_guarded(cls):
    """Class decorator: forward/backward of an autograd Function run with their tensors' device current (they launch on
    `torch.cuda.current_stream()` and keep per-device scratch), so a model on cuda:N works from any current device."""
    for name in ("forward", "backward"):
        setattr(cls, name, staticmethod(ops.on_tensor_device(getattr(cls, name))))
    return cls


# ---- weight gradients on a side stream (r06) ----------------------------------------------------------------------------------------------
# The backward pass is a CHAIN of data-gradient kernels (conv dgrad -> BN backward -> conv dgrad ...), most of them one-round launches that leave the chip
# half empty; the weight-gradient kernels (3.7 ms of the 18.3 ms step: MFMA GEMMs over the pixels + their partial-sum reduces) hang off that chain as leaves --
# nothing reads a dW before the optimizer.  With WGRAD_SIDE[0] set (engine.trainer.GraphedTrainStep sets it around its backward passes) AND MFX_WGRAD_STREAM=1 in
# the environment they are enqueued on a second stream that waits for the event of their inputs and is joined ONCE, by a callback the autograd
# engine runs when the backward pass ends (so p.grad is complete before anything after backward() looks at it).  Inside a hipGraph capture the fork / join
# become graph edges.  Tensors read or written across the fork carry `record_stream` marks, so the caching allocator does not hand their memory to the other
# stream early (inside a capture: not before the capture ends).  Never on for plain `loss.backward()` calls: a DDP reducer hook would read a
# gradient the moment autograd accumulates it, before the join.
# MEASURED (r06 call 24, B = 8, one hipGraph, same box, alternating): 18.49 / 19.04 ms with the side stream, 18.62 / 18.62 ms without -- no gain: at B = 8 the
# backward chain's kernels are throughput-bound, not idle-bound (the same picture as two sub-batch streams in inference: a wash at B = 8, +6 % at B = 32).
# The GPU suite is green with it on (551 passed).  OFF by default.
WGRAD_SIDE = [False]
_WGRAD_STREAM_ON = __import__("os").environ.get("MFX_WGRAD_STREAM", "0") == "1"
_side_streams = {}
_join_pending = [False]


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _join_side_streams():
    _join_pending[0] = False
    for key, side in _side_streams.items():
        torch.cuda.current_stream(torch.device(*key)).wait_stream(side)


def on_wgrad_stream(fn, inputs):
    """Run `fn()` (weight-gradient launches; returns a tensor or a tuple of tensors / None) on the side stream when enabled; `inputs` are the tensors it reads."""
    if not (WGRAD_SIDE[0] and _WGRAD_STREAM_ON) or not inputs or not inputs[0].is_cuda:
        return fn()
    dev = inputs[0].device
    cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = fn()
    for t in inputs:
        t.record_stream(side)
    for t in (out if isinstance(out, tuple) else (out,)):
        if t is not None:
            t.record_stream(cur)                              # produced on the side stream, consumed (optimizer, flat-buffer copy) on the main one after the join
    if not _join_pending[0]:
        _join_pending[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
    return out


def _wgrad(x, dy, kh, kw, stride, pad, Ho, Wo, Ck=None, x_pixstride=None, in_hw=None):
    """fp32 (Cout, kh*kw, Ck) weight gradient."""
    B, H, W, Cx = x.shape
    if in_hw is not None:
        H, W = in_hw
    Ck = Ck or Cx
    Cout = dy.shape[-1]
    dw = torch.empty((Cout, kh * kw, Ck), dtype=torch.float32, device=x.device)
    L.check(L.load().mfx_conv_wgrad_nhwc(_ptr(x), _ptr(dy), _ptr(dw), B, H, W, x_pixstride or Cx, Ck, kh, kw, stride, pad, pad,
                                         Ho, Wo, Cout, Cout, _dt(x.dtype), _stream()), "mfx_conv_wgrad_nhwc")
    return dw


class _SumArena:
    """Zeroed fp32 scratch the bias-gradient column sums of one backward pass are carved from: ONE fill at the top of the pass instead of one
    per biased conv (23 launches of a training step).  Opt-in (`sum_arena`: engine/trainer.GraphedTrainStep, which owns the step's gradients):
    the vectors handed out stay valid until the next `begin()` on the device -- the same lifetime as the gradients of a captured step."""
    ELEMS = 1 << 15

    def __init__(self):
        self.on = False
        self.bufs = {}              # device -> [buffer, floats used]

    def begin(self, device):
        b = self.bufs.get(device)
        if b is None:
            if torch.cuda.is_current_stream_capturing():
                return
            b = self.bufs[device] = [torch.zeros(self.ELEMS, dtype=torch.float32, device=device), 0]
        b[0].zero_()
        b[1] = 0

    def take(self, device, n):
        b = self.bufs.get(device)
        if not self.on or not SUM_ARENA_ON[0] or b is None or b[1] + n > self.ELEMS:
            return None
        o = b[1]
        b[1] += (n + 3) // 4 * 4
        return b[0][o:o + n]


SUM_ARENA = _SumArena()
SUM_ARENA_ON = [os.environ.get("MFX_SUM_ARENA", "1") != "0"]       # 0: a zero fill per bias-gradient sum, as before (A/B)


class sum_arena:
    """Context: the backward passes inside take their bias-gradient sums from SUM_ARENA (zeroed here, once)."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        self.prev = SUM_ARENA.on
        if self.device.type == "cuda":
            SUM_ARENA.on = True
            SUM_ARENA.begin(self.device)
        return self

    def __exit__(self, *a):
        SUM_ARENA.on = self.prev


def _colsum(t):
    C = t.shape[-1]
    M = t.numel() // C
    out = SUM_ARENA.take(t.device, C) if SUM_ARENA.on else None
    if out is not None:
        L.check(L.load().mfx_colsum_add(_ptr(t), _ptr(out), M, C, C, _dt(t.dtype), _stream()), "mfx_colsum_add")
        return out
    out = torch.empty(C, dtype=torch.float32, device=t.device)
    L.check(L.load().mfx_colsum(_ptr(t), _ptr(out), M, C, C, _dt(t.dtype), _stream()), "mfx_colsum")
    return out


def _pad_channels(n, dtype):
    """Output-channel padding: a power of two (>= one 16-byte chunk) so the padded map is a valid conv input."""
    e = 4 if dtype == torch.float32 else 8
    if n >= 64:
        return (n + 63) // 64 * 64
    p = e
    while p < n:
        p *= 2
    return p


class _PackRegistry:
    """Persistent packed copies of the conv parameters (forward operand, data-gradient operand) with the parameter version they
    were packed from.  A parameter changes once per optimisation step, so `pack_all_weights()` at the top of a step re-packs
    every registered operand in ONE launch (mfx_pack_conv_weights_batched); `_pack_weight` then only hands out the buffers.
    Without that call (or for an operand seen for the first time) the operand is packed on demand, one launch, as before."""

    def __init__(self):
        self.entries = {}              # key -> dict(ref, weight version, buffers, descriptor fields)
        self.tables = {}               # (device, dtype) -> (keys, descs tensor, prefix tensor, total)
        self.dirty = set()
        self.scope_ids = None
        self.scope = None              # pack_scope: {(device, dtype) -> table} restricted to one model's parameters, used INSTEAD of self.tables
        self.capture_unpacked = set()  # (device, dtype) whose batched packing could not be recorded in the running capture
        self.arenas = {}               # device -> fragment arena of the 64-channel DCN operands (_arena_slot)

    def lookup(self, weight, dtype, mode, rows, ck, stride, pad_h, pad_w, register=True):
        import weakref
        key = (id(weight), dtype, mode, rows, ck, stride, pad_h, pad_w)
        e = self.entries.get(key) if register else None
        if e is not None and (e["ref"]() is not weight or e["ptr"] != weight.data_ptr()):
            e = None                                           # the id was recycled, or the parameter moved (.to(), load)
        if e is None:
            Cout, Cin, kh, kw = weight.shape
            E = 4 if dtype == torch.float32 else 8
            if (ck & (ck - 1) and kh * kw > 1) or ck < E or ck % E:
                raise ValueError("conv operand: channels per tap must be a power of two >= %d (any multiple of %d for 1x1), got %d" % (E, E, ck))
            kr = 8 * E if kh * kw * ck >= 8 * E else 4 * E      # (a 64-byte K stays 64 bytes: ops.pack_conv)
            K_pad = (kh * kw * ck + kr - 1) // kr * kr
            cp = ops.cout_pad(rows)
            packed = torch.empty((cp, K_pad), dtype=dtype, device=weight.device)
            frag = f16 = None
            if kh == 3 and kw == 3 and stride in (1, 2) and pad_h == 1 and pad_w == 1:
                if register and dtype == torch.bfloat16 and ck == 64 and cp in (32, 64) and stride == 1 and mode == 0:
                    # a 64-channel DCN layer's operands (_with_f16_fragments): the fragments of all such layers share one arena, so their
                    # IEEE-fp16 copies are ONE cast per step (pack_all) instead of one per operand
                    frag, f16 = self._arena_slot(weight.device, cp, K_pad)
                if frag is None:
                    frag = torch.empty_like(packed)
            e = dict(ref=None, ptr=weight.data_ptr(), version=-1, packed=packed, frag=frag, cp=cp, K_pad=K_pad,
                     shape=(Cout, Cin, kh, kw), mode=mode, ck=ck, fp32=weight.dtype == torch.float32, f16=f16, f16_version=-2)
            if register:
                tk = (weight.device, dtype)
                # the entry (and its packed buffers) goes away with the parameter
                e["ref"] = weakref.ref(weight, lambda _r, k=key, t=tk, reg=self: reg._drop(k, t))
                self.entries[key] = e
                self.dirty.add(tk)
            else:
                e["ref"] = weakref.ref(weight)
        return e

    ARENA_ELEMS = 48 * 64 * 576            # bf16 fragments of up to 48 64-channel 3x3 operands (5.3 MB + its fp16 twin) per device

    def _arena_slot(self, dev, cp, K_pad):
        """(bf16 fragment buffer, its fp16 twin) carved from the device's fragment arena, or (None, None) when it is full / a capture is running."""
        if torch.cuda.is_current_stream_capturing():
            return None, None
        a = self.arenas.get(dev)
        if a is None:
            a = self.arenas[dev] = dict(b16=torch.zeros(self.ARENA_ELEMS, dtype=torch.bfloat16, device=dev),
                                        f16=torch.zeros(self.ARENA_ELEMS, dtype=torch.float16, device=dev), used=0)
        n = cp * K_pad
        if a["used"] + n > self.ARENA_ELEMS:
            return None, None
        o = a["used"]
        a["used"] += n                                          # (slots are not recycled: a model's operands are registered once)
        return a["b16"][o:o + n].view(cp, K_pad), a["f16"][o:o + n].view(cp, K_pad)

    def _drop(self, key, table_key):
        if self.entries.pop(key, None) is not None:
            self.dirty.add(table_key)

    def pack_one(self, e, weight, dtype):
        Cout, Cin, kh, kw = e["shape"]
        w32 = weight.detach()
        w32 = _c(w32 if w32.dtype == torch.float32 else w32.float())
        L.check(L.load().mfx_pack_conv_weight(_ptr(w32), Cout, Cin, kh, kw, e["mode"], _ptr(e["packed"]), _ptr(e["frag"]), e["cp"], e["K_pad"],
                                              e["ck"], _dt(dtype), _stream()), "mfx_pack_conv_weight")
        e["version"] = weight._version

    def _rebuild(self, dev, dtype, only=None):
        """The (keys, descriptor table, chunk prefix, chunk count) of the registered operands on (dev, dtype) -- all of them (stored in self.tables), or
        those of the parameters whose id() is in `only` (returned, not stored: pack_scope)."""
        import numpy as np
        keys, descs, prefix, total = [], [], [0], 0
        chunk = L.load().mfx_pack_chunk_elems()
        for key, e in list(self.entries.items()):
            w = e["ref"]()
            if w is None or w.data_ptr() != e["ptr"]:
                del self.entries[key]
                continue
            if key[1] != dtype or w.device != dev or not e["fp32"] or not w.is_contiguous() or (only is not None and id(w) not in only):
                continue
            d = L.PackDesc()
            d.w, d.packed, d.frag = w.data_ptr(), e["packed"].data_ptr(), (e["frag"].data_ptr() if e["frag"] is not None else None)
            d.Cout, d.Cin, d.kh, d.kw = e["shape"]
            d.mode, d.rows_pad, d.K_pad, d.ck = e["mode"], e["cp"], e["K_pad"], e["ck"]
            keys.append(key); descs.append(bytes(d))
            total += (e["cp"] * e["K_pad"] + chunk - 1) // chunk
            prefix.append(total)
        if not keys:
            if only is None:
                self.tables.pop((dev, dtype), None)
            return None
        dt = torch.from_numpy(np.frombuffer(b"".join(descs), dtype=np.uint8).copy()).to(dev)
        pt = torch.tensor(prefix, dtype=torch.int64).to(dev)
        if only is None:
            self.tables[(dev, dtype)] = (keys, dt, pt, total)
        return keys, dt, pt, total

    def pack_all(self):
        capturing = torch.cuda.is_current_stream_capturing()
        self.capture_unpacked = set()
        scoped = self.scope is not None
        if not scoped:
            for tk in list(self.dirty):
                if not capturing:                              # the tables are uploaded from the host: not inside a capture
                    self._rebuild(*tk)
                    self.dirty.discard(tk)
                else:
                    self.capture_unpacked.add(tk)              # operands registered since the last rebuild are not in the table
        for (dev, dtype), (keys, dt, pt, total) in list((self.scope if scoped else self.tables).items()):
            live = [self.entries.get(k) for k in keys]
            if any(e is None or e["ref"]() is None or e["ref"]().data_ptr() != e["ptr"] for e in live):
                if scoped:
                    raise RuntimeError("pack_scope: a parameter of the scoped model went away or moved")
                if capturing:
                    # the table names dead parameters and cannot be rebuilt inside a capture: nothing is recorded for (dev, dtype) here, and
                    # `_pack_weight` packs every operand it hands out during this capture on demand (one launch each) -- slower replays, never
                    # stale operands.  (engine/trainer.GraphedTrainStep captures inside a `pack_scope`, which cannot get here.)
                    self.capture_unpacked.add((dev, dtype))
                    continue
                self._rebuild(dev, dtype)
                if (dev, dtype) not in self.tables:
                    continue
                keys, dt, pt, total = self.tables[(dev, dtype)]
                live = [self.entries[k] for k in keys]
            with torch.cuda.device(dev):
                L.check(L.load().mfx_pack_conv_weights_batched(_ptr(dt), _ptr(pt), len(keys), total, _dt(dtype), _stream()),
                        "mfx_pack_conv_weights_batched")
            for e in live:
                e["version"] = e["ref"]()._version
            a = self.arenas.get(dev)
            if a is not None and dtype == torch.bfloat16 and a["used"]:
                a["f16"][:a["used"]].copy_(a["b16"][:a["used"]])          # bf16 -> fp16 of every arena fragment (exactness: _with_f16_fragments)
                for e in live:
                    if e.get("f16") is not None:
                        e["f16_version"] = e["version"]


_PACKS = _PackRegistry()


class _PadRegistry:
    """Persistent zero-padded fp32 copies of bias parameters whose conv pads its output channels (the 27 -> 32 channel DCN offset
    conv, the 3-channel class head): `F.pad` was a fill plus a copy launch per layer and step.  The tail is zeroed once; the head
    is refreshed from the parameter -- all entries in one `_foreach_copy_` at the top of a step (`pack_all_weights`), or on demand
    when the parameter's version moved."""

    def __init__(self):
        self.entries = {}

    def get(self, vec, n):
        import weakref
        key = (id(vec), n)
        e = self.entries.get(key)
        if e is not None and (e["ref"]() is not vec or e["ptr"] != vec.data_ptr()):
            e = None
        if e is None:
            e = dict(ref=weakref.ref(vec, lambda _r, k=key, reg=self: reg.entries.pop(k, None)), ptr=vec.data_ptr(), version=-1,
                     buf=torch.zeros(n, dtype=torch.float32, device=vec.device), m=vec.numel())
            self.entries[key] = e
        if e["version"] != vec._version:
            e["buf"][:e["m"]].copy_(vec.detach())
            e["version"] = vec._version
        return e["buf"]

    def refresh_all(self, only=None):
        live = []
        for key, e in list(self.entries.items()):
            v = e["ref"]()
            if v is None or v.data_ptr() != e["ptr"]:
                del self.entries[key]
            elif only is None or id(v) in only:
                live.append((e, v))
        if live:                                               # unconditional, like the batched packing: a captured step replays it
            torch._foreach_copy_([e["buf"][:e["m"]] for e, _ in live], [v.detach() for _, v in live])
            for e, v in live:
                e["version"] = v._version


_PADS = _PadRegistry()


def _padded_bias(bias, n):
    """fp32 [n] shift vector of a conv whose output channels are padded to n: the bias followed by zeros."""
    if bias.numel() == n:
        return _c(bias.detach().float())
    if isinstance(bias, torch.nn.Parameter) and bias.dtype == torch.float32 and bias.is_contiguous():
        return _PADS.get(bias, n)
    return torch.nn.functional.pad(bias.detach().float(), (0, n - bias.numel()))


def pack_all_weights():
    """Re-pack every conv operand the training path has used so far, one launch per (device, dtype), and refresh the padded bias
    vectors; call at the top of a step."""
    _PACKS.pack_all()
    _PADS.refresh_all(_PACKS.scope_ids)


class pack_scope:
    """Context for CAPTURING a step of one model: inside it `pack_all_weights()` touches only the operands / padded biases of `params`, through
    descriptor tables built here (before the capture: they are uploaded from the host) and kept alive by this object -- keep it as long as the graph.
    Why: the registry's own tables name the operands of EVERY model that is alive at the time and are replaced whenever that set changes.  A graph
    captured against them would (a) read a freed table after the next rebuild and (b) keep re-packing the operands of other models after those
    have been freed -- writes into memory that belongs to somebody else by then.  r06: seen as order-dependent mismatches of the graph-vs-eager tests
    once an unrelated reference cycle delayed the collection of earlier tests' models (tools/probes/syncbn_flaky_dbg.py)."""

    def __init__(self, params):
        ps = list(params)
        self.ids = {id(p) for p in ps}
        self.keep = ps
        self.tables = {}
        for key in {(e["ref"]().device, k[1]) for k, e in list(_PACKS.entries.items()) if e["ref"]() is not None and id(e["ref"]()) in self.ids}:
            t = _PACKS._rebuild(key[0], key[1], only=self.ids)
            if t is not None:
                self.tables[key] = t

    def __enter__(self):
        self.prev = (_PACKS.scope, _PACKS.scope_ids)
        _PACKS.scope, _PACKS.scope_ids = self.tables, self.ids
        return self

    def __exit__(self, *a):
        _PACKS.scope, _PACKS.scope_ids = self.prev


def _pack_weight(weight, dtype, mode, rows, ck, stride, pad_h, pad_w, shift=None):
    """The conv operand of an fp32 OIHW parameter (mode 0 forward, 1 data gradient), packed on the device from the CURRENT
    parameter values: handed out from the step's batched packing when the parameter has not changed since, else packed now."""
    Cout, Cin, kh, kw = weight.shape
    if isinstance(weight, torch.nn.Parameter):
        e = _PACKS.lookup(weight, dtype, mode, rows, ck, stride, pad_h, pad_w)
        if e["version"] != weight._version or ((weight.device, dtype) in _PACKS.capture_unpacked and torch.cuda.is_current_stream_capturing()):
            _PACKS.pack_one(e, weight, dtype)
    else:                                                      # a temporary (stacked head weights, a view): nothing to remember
        e = _PACKS.lookup(weight, dtype, mode, rows, ck, stride, pad_h, pad_w, register=False)
        _PACKS.pack_one(e, weight, dtype)
    p = ops.PackedConv(e["packed"], None, shift, kh, kw, stride, pad_h, pad_w, 1, ck, rows, e["cp"], e["K_pad"], L.ACT_NONE, e["frag"])
    p.transient = True                                         # rebuilt every step: no derived operand is worth packing behind it (ops.dcn_ps_applies)
    p.entry = e
    return p


def _with_f16_fragments(p, x):
    """The DCN LDS-patch kernel (dcn_patch.hip; 64 -> 64 layers on large maps, bf16) multiplies with an IEEE fp16 copy of the
    fragment-major weights.  bf16 -> fp16 is element-wise and exact for 2^-14 <= |w| <= 65504 (bf16's 8 mantissa bits fit fp16's 11; smaller
    weights round to fp16 subnormals / zero -- an absolute error below 3e-8 --, larger ones do not occur in a network whose activations
    are finite in fp16), and both fragment layouts hold 8 elements per 16 bytes, so the copy is one small cast of the bf16 fragments the step's batched packing already produced (36.9 k elements per layer): the
    training forward then runs the third-generation kernel on the five full-resolution layers (102 -> 75 us each) instead of the
    first-generation gather."""
    if p.w_frag is not None and p.Ck == 64 and p.Cout_pad in (32, 64) and x.shape[0] * x.shape[1] * x.shape[2] >= 65536:      # (32: the module's offset/mask conv, run inside that kernel)
        if p.w.dtype == torch.bfloat16:
            e = getattr(p, "entry", None)
            if e is not None and e.get("f16") is not None and e["f16_version"] == e["version"] and e["frag"] is p.w_frag:
                p.w_frag_f16 = e["f16"]                        # this step's batched cast (_PackRegistry.pack_all)
            else:
                p.w_frag_f16 = p.w_frag.to(torch.float16)
        elif p.w.dtype == torch.float16:
            p.w_frag_f16 = p.w_frag                            # fp16 mode: the fragments already are IEEE fp16
    return p


@_device_guarded
class Conv2dFn(Function):
    """y = conv2d(x, weight) (+ bias), k in {1,3}, stride in {1,2}, pad = k//2.  Output channels are padded up to a
    multiple of the 16-byte chunk (extra channels are exactly zero); callers slice."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, out_dtype=None, act=L.ACT_NONE, stats=None, alias=False):
        """`act` = ACT_DCN_OFFMASK fuses the sigmoid of the DCN mask channels into the epilogue; the CONSUMER (DCNFn with
        post_sigmoid=True) then hands back the gradient of the pre-activation, which is what backward() below expects.
        `stats` = the scratch of the train-mode BN that follows: the conv's epilogue adds the output's statistics to it where
        the kernel supports that (ops.conv2d.last_stats_done).
        `alias` = True: returns (y, x') with x' the input again, as an output of THIS node -- for an input that has a second consumer next to
        the conv (the identity residual of a BasicBlock, dla_dcn.py:84-98).  The gradient of x' then arrives here instead of at an autograd
        add, and the data-gradient conv takes it as its epilogue residual: one element-wise pass over the map less per block."""
        x_in = x
        x = _c(x)
        Cout, Cin, kh, kw = weight.shape
        # (padded to the chunk of the COMPUTE type also where the map is written in fp32: the backward pass then casts dy and has nothing to pad)
        cpad = _pad_channels(Cout, x.dtype)
        shift = None
        if bias is not None:
            cp = ops.cout_pad(cpad)
            shift = _padded_bias(bias, cp)
        p = _pack_weight(weight, x.dtype, 0, cpad, Cin, stride, pad, pad, shift)
        p.act = act
        if stats is not None and (cpad != Cout or ops.cout_pad(cpad) != Cout or act != L.ACT_NONE):
            stats = None                                       # padded output channels / an activation: leave the statistics to the BN
        ops.conv2d.last_stats_done = False
        y = ops.conv2d(x, p, out_dtype=out_dtype, stats=stats)     # bf16 mode: fp32 out for DCN offsets and the head maps
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, bias is not None, Cout)
        if alias:
            return y, x_in.view(x_in.shape)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, g_alias=None):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias, Cout = ctx.cfg
        res = None
        if g_alias is not None and ctx.needs_input_grad[0]:
            res = _c(g_alias if g_alias.dtype == x.dtype else g_alias.to(x.dtype))
        if dy is None:                                          # (only the alias was used downstream)
            return res, None, None, None, None, None, None, None, None
        dx, dw, db = _conv_backward(x, weight, dy, stride, pad, has_bias, Cout, ctx.needs_input_grad[:3], res=res)
        return dx, dw, db, None, None, None, None, None, None


def _conv_backward(x, weight, dy, stride, pad, has_bias, Cout, needs, res=None):
    """Gradients of y = conv2d(x, weight) (+ bias) given dy: (dx, dW, db).  `res` (same shape / dtype as dx) is added to dx in the
    data-gradient conv's epilogue -- the other gradient path of x, so autograd has nothing left to add."""
    dy = _c(dy)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    emin = 4 if x.dtype == torch.float32 else 8
    if dy.shape[-1] < emin:                                # fp32-out head conv in bf16 mode: 4 channels < one bf16 chunk
        dy = torch.nn.functional.pad(dy, (0, emin - dy.shape[-1]))
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cp = dy.shape
    kh, kw = weight.shape[2], weight.shape[3]
    dx = dw = db = None
    if needs[0]:
        # the kernel rotated by 180 degrees with in/out swapped, as the operand of a stride-1 'full' correlation
        cin_pad = _pad_channels(Cin, x.dtype)
        pt = _pack_weight(weight, x.dtype, 1, cin_pad, Cp, 1, kh - 1 - pad, kw - 1 - pad)
        g = dy
        if stride == 2:
            g = torch.empty((B, H, W, Cp), dtype=dy.dtype, device=dy.device)
            L.check(L.load().mfx_zero_insert2_nhwc(_ptr(dy), _ptr(g), B, Ho, Wo, Cp, H, W, _dt(dy.dtype), _stream()),
                    "mfx_zero_insert2_nhwc")
        dx = ops.conv2d(g, pt, res=res if (res is not None and cin_pad == Cin) else None)
        if cin_pad != Cin:
            dx = dx[..., :Cin]
            if res is not None:
                dx = dx + res
    need_w, need_b = bool(needs[1]), bool(has_bias and needs[2])
    if need_w or need_b:
        def leaves():                                           # the parameter gradients: leaves of the backward graph (see on_wgrad_stream)
            dw_ = db_ = None
            if need_w:
                dwf = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                ws = ops._splitk_workspace(x.device)          # per-slab partial tiles (bf16 path), shared per stream
                L.check(L.load().mfx_conv_wgrad_oihw(_ptr(x), _ptr(dy), _ptr(dwf), B, H, W, Cin, Cin, kh, kw, stride, pad, pad, Ho, Wo, Cp, Cp,
                                                     Cout, Cin, _dt(x.dtype), _ptr(ws), ws.numel() * 4, _stream()), "mfx_conv_wgrad_oihw")
                dw_ = dwf if weight.dtype == torch.float32 else dwf.to(weight.dtype)
            if need_b:
                db_ = _colsum(dy)[:Cout]
            return dw_, db_
        dw, db = on_wgrad_stream(leaves, [dy, x])
    return dx, dw, db


@_device_guarded
class HeadConvGatherFn(Function):
    """A head's 1x1 conv of its trunk activation f AND the gather of f's rows at the edge-sequence pixels, as ONE node
    (reference detector_predictor.py:125-147: the class / 3d_offset trunks feed both their head conv and the edge fusion).
    As two nodes the gather's gradient is a dense zero map of f's size (126 MB at B=8) with ~6.7k rows scattered into it, which
    autograd then adds to the conv's data gradient in another full pass; here the rows are added into the data gradient in place.
    Returns (y with padded channels, rows [len(rowmap), C])."""

    @staticmethod
    def forward(ctx, f, weight, bias, rowmap):
        f = _c(f)
        Cout, Cin = weight.shape[0], weight.shape[1]
        cpad = _pad_channels(Cout, f.dtype)
        shift = _padded_bias(bias, ops.cout_pad(cpad)) if bias is not None else None
        y = ops.conv2d(f, _pack_weight(weight, f.dtype, 0, cpad, Cin, 1, 0, 0, shift), out_dtype=torch.float32)
        e = f.view(-1, Cin).index_select(0, rowmap)
        ctx.save_for_backward(f, weight, rowmap)
        ctx.cfg = (bias is not None, Cout)
        return y, e

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, de):
        f, weight, rowmap = ctx.saved_tensors
        has_bias, Cout = ctx.cfg
        needs = (True,) + tuple(ctx.needs_input_grad[1:3])       # the rows' gradient lands in dx
        dx, dw, db = _conv_backward(f, weight, dy, 1, 0, has_bias, Cout, needs)
        if not dx.is_contiguous():
            dx = dx.contiguous()
        dx.view(-1, dx.shape[-1]).index_add_(0, rowmap, _c(de).to(dx.dtype))       # replicate-padded ends / padding rows repeat: atomics
        return dx, dw, db, None


@_device_guarded
class EdgeScatterAddFn(Function):
    """out = base with o[b, l, :] added to channels [lo, lo + co) of the border pixel edge_xy[b, l], l < edge_len[b] (reference
    detector_predictor.py:139-147; the first edge_len points of an image are unique).  `rows` = flat pixel row of every (b, l),
    `valid` = (l < edge_len[b]) as fp32 [B, L, 1]: both depend on the targets only."""

    @staticmethod
    def forward(ctx, base, o, edge_xy, edge_len, lo, rows, valid):
        out = base.clone() if base.is_contiguous() else base.contiguous()
        o = _c(o.float())
        ops.edge_scatter_add(out, lo, o.shape[-1], o, edge_xy, edge_len)
        ctx.save_for_backward(rows, valid)
        ctx.cfg = (lo, tuple(o.shape))
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        rows, valid = ctx.saved_tensors
        lo, (B, Lmax, co) = ctx.cfg
        g = _c(g)
        do = g.view(-1, g.shape[-1]).index_select(0, rows)[:, lo:lo + co].reshape(B, Lmax, co) * valid
        return g, do, None, None, None, None, None


@_device_guarded
class CatConv1x1Fn(Function):
    """Root (dla_dcn.py:203-220): y = conv1x1(cat(xs, channel axis), weight) without materialising the concat.
    Backward per source i: dx_i = dy @ W[:, seg_i], dW[:, seg_i] = dy^T x_i."""

    @staticmethod
    def forward(ctx, weight, *xs):
        xs = [_c(t) for t in xs]
        chans = tuple(t.shape[3] for t in xs)
        Cout, Ctot = weight.shape[0], weight.shape[1]
        pk = _pack_weight(weight, xs[0].dtype, 0, Cout, Ctot, 1, 0, 0)          # one launch: [Cout_pad][Ctot], K-contiguous
        Cseg = min(chans)
        assert sum(chans) == Ctot and all(c % Cseg == 0 for c in chans) and Cseg & (Cseg - 1) == 0 and pk.K_pad == Ctot
        p = ops.PackedCat(pk.w, None, None, Cseg, Cout, pk.Cout_pad, Ctot, L.ACT_NONE)
        ctx.save_for_backward(weight, *xs)
        return ops.cat_conv1x1(xs, p)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        weight, *xs = ctx.saved_tensors
        dy = _c(dy)
        Cout, Ctot = weight.shape[0], weight.shape[1]
        need_dx = any(ctx.needs_input_grad[1 + i] for i in range(len(xs)))
        # data-gradient operand of the whole Root in one launch: WT[c][o] = W[o][c]; source i uses rows [off, off + C_i)
        wt = _pack_weight(weight, dy.dtype, 1, Ctot, Cout, 1, 0, 0) if need_dx else None
        dxs, off = [], 0
        for i, x in enumerate(xs):
            C = x.shape[3]
            if ctx.needs_input_grad[1 + i]:
                pt = ops.PackedConv(wt.w[off:off + C], None, None, 1, 1, 1, 0, 0, 1, Cout, C, C, wt.K_pad, L.ACT_NONE, None)
                dxs.append(ops.conv2d(dy, pt))
            else:
                dxs.append(None)
            off += C
        dw = None
        if ctx.needs_input_grad[0]:
            def leaves():                                       # the Root's weight gradient, source by source: a leaf of the backward graph (on_wgrad_stream)
                parts = [_wgrad(x, dy, 1, 1, 1, 0, x.shape[1], x.shape[2]).view(Cout, x.shape[3]) for x in xs]
                return torch.cat(parts, 1).view_as(weight).to(weight.dtype)
            dw = on_wgrad_stream(leaves, [dy, *xs])
        return (dw, *dxs)


_STEM_WGRAD_GENERIC = [False]    # True: the generic dilated-tap weight-gradient kernel also for the 16-channel bf16 stem (test switch)


_STEM_FWD_GENERIC = [False]      # True: the stem's forward goes through the generic implicit-GEMM kernel (test switch)


@_device_guarded
class StemConvFn(Function):
    """7x7 / stride 1 / pad 3 convolution of the NCHW fp32 image batch (dla_dcn.py:268-272); no data gradient."""

    @staticmethod
    def forward(ctx, images, weight, dtype):
        B, _, H, W = images.shape
        xp = ops.pack_image(images, dtype)
        one = torch.ones(weight.shape[0], device=images.device)
        p = ops.pack_stem(weight, dtype, one, torch.zeros_like(one), act=L.ACT_NONE)
        if dtype in (torch.bfloat16, torch.float16) and weight.shape[0] == 16 and not _STEM_FWD_GENERIC[0]:
            y = ops.stem_conv(images, p)          # the inference stem kernel, raw output (60 vs 170 us at B=8); xp is only kept for the backward
        else:
            y = ops.conv2d(xp, p, out_hw=(H, W))
        ctx.save_for_backward(xp, weight)
        ctx.hw = (H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xp, weight = ctx.saved_tensors
        H, W = ctx.hw
        Cout = weight.shape[0]
        dy = _c(dy)
        if xp.dtype in (torch.bfloat16, torch.float16):
            # 16-bit activations: the matrix-core kernel needs 16-byte K chunks -> 8-element super-taps (two 4-channel pixels), kw = 4, dilation 2
            # (the layout the forward stem uses); dW[o][th*4 + j][u*4 + c] is the gradient of w[o][c][th][2j + u]
            B, Hp, Wp, _ = xp.shape
            dws = torch.empty((dy.shape[-1], 28, 8), dtype=torch.float32, device=xp.device)
            if dy.shape[-1] == 16 and not _STEM_WGRAD_GENERIC[0]:
                ws = ops._splitk_workspace(xp.device)             # Toeplitz-row kernel: image and dy read once
                L.check(L.load().mfx_stem_wgrad_16(_ptr(xp), _ptr(dy), _ptr(dws), B, H, W, Hp, Wp, _dt(xp.dtype), _ptr(ws), ws.numel() * 4, _stream()),
                        "mfx_stem_wgrad_16")
            else:
                L.check(L.load().mfx_conv_wgrad_nhwc_dil(_ptr(xp), _ptr(dy), _ptr(dws), B, Hp, Wp, 4, 8, 7, 4, 1, 0, 0, 2, H, W, dy.shape[-1],
                                                         dy.shape[-1], _dt(xp.dtype), _stream()), "mfx_conv_wgrad_nhwc_dil")
            dw = dws[:Cout].view(Cout, 7, 4, 2, 4).reshape(Cout, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2).contiguous()
        else:
            dwf = _wgrad(xp, dy, 7, 7, 1, 0, H, W, Ck=4, x_pixstride=4)      # padded image: pad 0 in padded coordinates
            dw = dwf[:Cout].view(Cout, 7, 7, 4)[..., :3].permute(0, 3, 1, 2).contiguous()
        return None, dw, None


_SYNC_BN_GROUP = [None]        # process group of the SyncBN statistics collectives (None: the default group); see set_sync_bn_group
_SYNC_BN_FORCE = [False]       # tests: issue the collectives on a one-rank group too (captured-collective coverage on a single GPU)


def set_sync_bn_group(group):
    """Route the SyncBN statistics all-reduces through `group` (None: the default group).  engine.trainer.GraphedTrainStep gives them a
    communicator of their own: they are CAPTURED inside the step's hipGraphs and replayed by the GPU, while the gradient slices are
    all-reduced on the default communicator from the host between graphs -- two issue orders that must not share one communicator."""
    _SYNC_BN_GROUP[0] = group


def _sync_group(sync):
    """Process group for synchronised BN, or None (single process / local statistics)."""
    import torch.distributed as dist
    if sync and dist.is_available() and dist.is_initialized():
        g = _SYNC_BN_GROUP[0] if _SYNC_BN_GROUP[0] is not None else dist.group.WORLD
        if dist.get_world_size(g) > 1 or _SYNC_BN_FORCE[0]:
            return g
    return None


_BN_SEPARATE = [False]          # True: the five-launch form (stats, finalize, apply; reduce, apply) also in the single-process case
_BN_READ_OUTPUT = [False]       # True: the backward always reads the forward output for the activation derivative (test switch)
_CONV_STATS_OFF = [__import__('os').environ.get('MFX_CONV_STATS', '1') == '0']       # True: convs never accumulate the following BN's statistics (test switch; env MFX_CONV_STATS=0)
_CONV_STATS_MAX_COUT = [128]
RESIDUAL_ALIAS = [__import__('os').environ.get('MFX_RESIDUAL_ALIAS', '1') != '0']      # BasicBlock identity residual through Conv2dFn's alias output (env MFX_RESIDUAL_ALIAS=0: autograd adds the two gradients)
_BN_SCRATCH = {}


def _bn_scratch(gamma):
    """Persistent zero scratch of one BN layer (keyed by its weight's storage): the kernels leave it zero after every call."""
    key = (gamma.data_ptr(), gamma.device)
    s = _BN_SCRATCH.get(key)
    if s is None:
        s = torch.zeros(L.load().mfx_bn_scratch_bytes() // 4, dtype=torch.float32, device=gamma.device)
        _BN_SCRATCH[key] = s
    return s


@_device_guarded
class BNActFn(Function):
    """Train-mode BatchNorm (batch statistics, biased variance) + activation (+ residual before the activation).
    With `sync` and an initialised multi-rank process group the statistics are those of the global batch
    (SyncBatchNorm, tools/plain_train_net.py:131-132): one all-reduce of [sum, sumsq, rows] forward and one of
    [sum g, sum g*xhat] backward, 2C(+1) floats each over RCCL."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, res, act, momentum, eps, sync, nbt=None, stats_done=False):
        x = _c(x)
        C = x.shape[-1]
        M = x.numel() // C
        lib_ = L.load()
        group = _sync_group(sync)
        g32 = gamma.detach() if gamma.dtype == torch.float32 else gamma.detach().float()
        b32 = beta.detach() if beta.dtype == torch.float32 else beta.detach().float()
        if group is None and C <= 512 and M > 0 and not _BN_SEPARATE[0]:
            # single process: statistics + apply in two launches on the layer's persistent (self-clearing) scratch
            scratch = _bn_scratch(gamma)
            mr = torch.empty(2 * C, dtype=torch.float32, device=x.device)
            y = torch.empty_like(x)
            res_c = _c(res) if res is not None else None
            rm = running_mean if running_mean is not None and running_mean.dtype == torch.float32 else None
            if nbt is not None and (nbt.dtype != torch.int64 or rm is None):
                nbt.add_(1)
                nbt = None
            L.check(lib_.mfx_bn_train_fwd(_ptr(x), _ptr(res_c), _ptr(y), _ptr(_c(g32)), _ptr(_c(b32)), _ptr(rm),
                                          _ptr(running_var) if rm is not None else None, _ptr(nbt), ctypes.c_float(momentum), ctypes.c_float(eps),
                                          M, C, act, _dt(x.dtype), _ptr(scratch), _ptr(mr), mr.data_ptr() + 4 * C, int(bool(stats_done)), _stream()),
                    "mfx_bn_train_fwd")
            ctx.save_for_backward(x, y, mr[:C], mr[C:], _c(g32))
            ctx.cfg = (act, res is not None, None, M)
            ctx.scratch = scratch
            ctx.beta32 = _c(b32)
            return y
        if stats_done:
            raise RuntimeError("bn_act: the conv's epilogue filled the statistics scratch, but this BN takes the separate-kernel path")
        if nbt is not None:
            nbt.add_(1)
        ctx.scratch = None
        st = torch.empty(2 * C + 1, dtype=torch.float32, device=x.device)
        L.check(lib_.mfx_bn_stats(_ptr(x), _ptr(st), st.data_ptr() + 4 * C, M, C, _dt(x.dtype), _stream()), "mfx_bn_stats")
        Mt = M
        if group is not None:
            import torch.distributed as dist
            st[2 * C:].fill_(float(M))                         # (a fill kernel: an indexed scalar assignment is a host copy, illegal while capturing)
            dist.all_reduce(st, group=group)
            Mt = M * dist.get_world_size(group)                # equal per-rank batches (weak scaling), no host sync
        out = torch.empty(4 * C, dtype=torch.float32, device=x.device)          # mean | rstd | scale | shift
        mean, rstd, scale, shift = out[:C], out[C:2 * C], out[2 * C:3 * C], out[3 * C:]
        L.check(lib_.mfx_bn_finalize(_ptr(st), st.data_ptr() + 4 * C, _ptr(_c(g32)), _ptr(_c(b32)), _ptr(running_mean), _ptr(running_var),
                                     ctypes.c_float(momentum), ctypes.c_float(eps), Mt, _ptr(mean), out.data_ptr() + 4 * C,
                                     out.data_ptr() + 8 * C, out.data_ptr() + 12 * C, C, _stream()), "mfx_bn_finalize")
        y = torch.empty_like(x)
        res_c = _c(res) if res is not None else None
        L.check(lib_.mfx_bn_act_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(res_c), _ptr(y), M, C, act, _dt(x.dtype), _stream()),
                "mfx_bn_act_fwd")
        ctx.save_for_backward(x, y, mean, rstd, _c(g32))
        ctx.cfg = (act, res is not None, group, Mt)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, da):
        x, y, mean, rstd, g32 = ctx.saved_tensors
        act, has_res, group, Mt = ctx.cfg
        da = _c(da)
        C = x.shape[-1]
        M = x.numel() // C
        if ctx.scratch is not None:
            dx = torch.empty_like(x)
            dres = torch.empty_like(x) if has_res else None
            dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
            dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
            # without a residual the activation's derivative is recomputed from x (the forward output is not read)
            a_in = y if (has_res or act == L.ACT_NONE or _BN_READ_OUTPUT[0]) else None
            L.check(L.load().mfx_bn_train_bwd(_ptr(x), _ptr(a_in), _ptr(da), _ptr(mean), _ptr(rstd), _ptr(g32), _ptr(ctx.beta32), _ptr(dx),
                                              _ptr(dres), _ptr(dgamma), _ptr(dbeta), M, C, act, _dt(x.dtype), _ptr(ctx.scratch), _stream()),
                    "mfx_bn_train_bwd")
            return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None
        sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        sg, sgx = sums[:C], sums[C:]
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        lib_ = L.load()
        L.check(lib_.mfx_bn_bwd_reduce(_ptr(x), _ptr(y), _ptr(da), _ptr(mean), _ptr(rstd), _ptr(sg), sums.data_ptr() + 4 * C,
                                       M, C, act, _dt(x.dtype), _stream()), "mfx_bn_bwd_reduce")
        if group is not None:
            import torch.distributed as dist
            local = sums.clone()                                # parameter gradients stay rank-local: DDP averages them
            dist.all_reduce(sums, group=group)
        else:
            local = sums
        L.check(lib_.mfx_bn_bwd_apply(_ptr(x), _ptr(y), _ptr(da), _ptr(mean), _ptr(rstd), _ptr(g32), _ptr(sg), sums.data_ptr() + 4 * C,
                                      _ptr(dx), _ptr(dres), M, Mt, C, act, _dt(x.dtype), _stream()), "mfx_bn_bwd_apply")
        return dx, local[C:].clone(), local[:C].clone(), None, None, dres, None, None, None, None, None, None


@_device_guarded
class MaxPool2x2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.maxpool2x2(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        L.check(L.load().mfx_maxpool2x2_bwd_nhwc(_ptr(x), _ptr(_c(dy)), _ptr(dx), B, H, W, C, _dt(x.dtype), _stream()),
                "mfx_maxpool2x2_bwd_nhwc")
        return dx


@_device_guarded
class UpsampleAddFn(Function):
    """y = depthwise ConvTranspose2d(x; k=2f, s=f, p=f/2) + skip   (dla_dcn.py:409-411, 419-425)."""

    @staticmethod
    def forward(ctx, x, weight, skip, f):
        x = _c(x)
        wt = ops.pack_upsample(weight)
        ctx.save_for_backward(x, wt)
        ctx.f = f
        ctx.wshape = weight.shape
        return ops.upsample_add(x, wt, f, skip=_c(skip))

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        f = ctx.f
        dy = _c(dy)
        B, H, W, C = x.shape
        dx = torch.empty_like(x)
        dw = torch.empty(ctx.wshape, dtype=torch.float32, device=x.device)                  # written in the parameter's layout (C, 1, 2f, 2f)
        nws = L.load().mfx_upsample_bwd_workspace_bytes(B, H, C, f)
        ws = torch.empty(nws // 4, dtype=torch.float32, device=x.device)
        L.check(L.load().mfx_upsample_bwd_nhwc_oihw(_ptr(x), _ptr(wt), _ptr(dy), _ptr(dx), _ptr(dw), B, H, W, C, f, _dt(x.dtype), _ptr(ws), nws,
                                                    _stream()), "mfx_upsample_bwd_nhwc_oihw")
        return dx, dw, dy, None


@_device_guarded
class DCNFn(Function):
    """Modulated deformable conv on NHWC fp32: y = DCNv2(x; offsets, sigmoid(mask logits), weight) + bias.
    `offmask_raw` is the (B,H,W,32) output of the 27-channel offset/mask conv (channels 0..17 offsets,
    18..26 mask logits, reference dcn_v2.py:118-122)."""

    @staticmethod
    def forward(ctx, x, offmask_raw, weight, bias, stride, pad, dil, post_sigmoid=False):
        x, raw = _c(x), _c(offmask_raw).float()
        if post_sigmoid:                                       # the offset/mask conv already applied the sigmoid (fused epilogue)
            om = raw
        else:
            om = raw.clone()
            om[..., 18:27] = torch.sigmoid(raw[..., 18:27])
        Cout, Cin = weight.shape[0], weight.shape[1]
        shift = None
        if bias is not None:
            cp = ops.cout_pad(Cout)
            shift = _padded_bias(bias, cp)
        p = _pack_weight(weight, x.dtype, 0, Cout, Cin, stride, pad, pad, shift)
        p.dil_w = dil
        if stride == 1 and pad == 1 and dil == 1:
            _with_f16_fragments(p, x)
        y = ops.dcn(x, om, p)
        ctx.save_for_backward(x, om, weight)
        ctx.cfg = (stride, pad, dil)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, om, weight = ctx.saved_tensors
        stride, pad, dil = ctx.cfg
        dy = _c(dy)
        B, H, W, C = x.shape
        Cout, _, kh, kw = weight.shape
        lib_ = L.load()
        pow2 = lambda v: v >= 64 and (v & (v - 1)) == 0
        if (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1) and pow2(C) and pow2(Cout) and not _DCN_BWD_V1[0]:
            # tile-owned backward (dcn_bwd_tile.hip): dx in the activation dtype, gradient of the RAW offset/mask conv output
            dt = _dt(x.dtype)
            ws = ops._workspace(lib_.mfx_dcn_backward_v2_workspace_bytes(B, C, H, W, Cout, dt), x.device)
            dx = torch.empty_like(x)
            draw = torch.empty_like(om)
            dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
            db = torch.empty(Cout, dtype=torch.float32, device=x.device)
            wc = _c(weight.detach() if weight.dtype == torch.float32 else weight.detach().float())
            L.check(lib_.mfx_dcn_backward_v2(_ptr(x), _ptr(om), _ptr(wc), _ptr(dy if dy.dtype == x.dtype else dy.to(x.dtype)), _ptr(dx), _ptr(draw),
                                             _ptr(dw), _ptr(db), B, C, H, W, Cout, dt, _ptr(ws), ws.numel(), _stream()), "mfx_dcn_backward_v2")
            return dx, draw, dw if weight.dtype == torch.float32 else dw.to(weight.dtype), db, None, None, None, None
        nbytes = lib_.mfx_dcn_backward_nhwc_workspace_bytes(B, C, H, W, Cout, kh, kw, stride, pad, dil)
        ws = ops._workspace(nbytes, x.device)
        xdtype = x.dtype
        bf16_path = xdtype == torch.bfloat16 and C >= 64
        if xdtype != torch.float32 and not bf16_path:
            x, dy = x.float(), dy.float()
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        dom = torch.empty_like(om)
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        wc = weight.detach().float().contiguous()
        entry = lib_.mfx_dcn_backward_nhwc_bf16 if bf16_path else lib_.mfx_dcn_backward_nhwc
        L.check(entry(_ptr(x), _ptr(om), _ptr(wc), _ptr(dy if dy.dtype == x.dtype else dy.to(x.dtype)), _ptr(dx), _ptr(dom), _ptr(dw), _ptr(db),
                                           B, C, H, W, Cout, kh, kw, stride, pad, dil, _ptr(ws), ws.numel(), _stream()),
                "mfx_dcn_backward_nhwc")
        m = om[..., 18:27]
        dom[..., 18:27] = dom[..., 18:27] * m * (1 - m)               # through the sigmoid
        dom[..., 27:] = 0
        return dx.to(xdtype), dom, dw.to(weight.dtype), db, None, None, None, None


_RAW16 = [os.environ.get("MFX_DCN_RAW16", "1") != "0"]     # the DCN backward's offset / mask gradient rows in the activation type (0: fp32 + a cast, A/B)
_DCN_BWD_V1 = [False]          # tests: force the first-generation (global-atomics) backward for comparison


@_device_guarded
class DCNModuleFn(Function):
    """The DCN module of the training path as ONE node (reference dcn_v2.py:118-128: conv_offset_mask -> sigmoid -> dcn_v2_conv):
    x feeds both the offset/mask conv and the deformable sampling, so as two nodes autograd adds their two input gradients in a
    separate pass per layer; here the offset conv's data-gradient conv takes the DCN's input gradient as its epilogue residual.
    3x3 / stride 1 / pad 1 / dilation 1, channels a power of two >= 64 (the tile-owned backward)."""

    @staticmethod
    def forward(ctx, x, w_off, b_off, weight, bias):
        x = _c(x)
        n_off, Cin = w_off.shape[0], w_off.shape[1]
        cpad = _pad_channels(n_off, torch.float32)
        cp = ops.cout_pad(cpad)
        sh = _padded_bias(b_off, cp)
        po = _pack_weight(w_off, x.dtype, 0, cpad, Cin, 1, 1, 1, sh)
        po.act = L.ACT_DCN_OFFMASK
        Cout = weight.shape[0]
        cpm = ops.cout_pad(Cout)
        shift = None
        if bias is not None:
            shift = _padded_bias(bias, cpm)
        pm = _with_f16_fragments(_pack_weight(weight, x.dtype, 0, Cout, weight.shape[1], 1, 1, 1, shift), x)
        # offsets | sigmoid(mask logits), (B,H,W,32) fp32: from the offset conv, or written by the DCN kernel that ran it (ops.dcn_module)
        y, om = ops.dcn_module(x, _with_f16_fragments(po, x) if n_off == 27 else po, pm, need_offmask=True)
        ctx.save_for_backward(x, om, w_off, weight)
        ctx.n_off = n_off
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, om, w_off, weight = ctx.saved_tensors
        dy = _c(dy)
        B, H, W, C = x.shape
        Cout = weight.shape[0]
        lib_ = L.load()
        dt = _dt(x.dtype)
        ws = ops._workspace(lib_.mfx_dcn_backward_v2_workspace_bytes(B, C, H, W, Cout, dt), x.device)
        dx = torch.empty_like(x)
        # the raw offset / mask gradient in the activation type: what the offset conv's backward consumes (16-bit layers: no fp32 map, no cast)
        draw = torch.empty(om.shape, dtype=x.dtype if _RAW16[0] else torch.float32, device=x.device)
        dw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
        db = torch.empty(Cout, dtype=torch.float32, device=x.device)
        wc = _c(weight.detach() if weight.dtype == torch.float32 else weight.detach().float())
        L.check(lib_.mfx_dcn_backward_v2_rt(_ptr(x), _ptr(om), _ptr(wc), _ptr(dy if dy.dtype == x.dtype else dy.to(x.dtype)), _ptr(dx), _ptr(draw), int(_RAW16[0]),
                                            _ptr(dw), _ptr(db), B, C, H, W, Cout, dt, _ptr(ws), ws.numel(), _stream()), "mfx_dcn_backward_v2_rt")
        dx, dw_off, db_off = _conv_backward(x, w_off, draw, 1, 1, True, ctx.n_off, (True, True, True), res=dx)
        return dx, dw_off, db_off, dw if weight.dtype == torch.float32 else dw.to(weight.dtype), db


@_device_guarded
class FanOutConvFn(Function):
    """Several bias-free stride-1 convolutions of ONE input (the nine head trunks on the backbone feature map) as one node: the
    data-gradient convs are chained through their epilogue residual, so the input gradient comes out summed instead of as one map
    per consumer for autograd to add."""

    @staticmethod
    def forward(ctx, x, pad, *weights):
        x = _c(x)
        ys = []
        for w in weights:
            Cout, Cin = w.shape[0], w.shape[1]
            cpad = _pad_channels(Cout, x.dtype)
            if cpad != Cout:
                raise ValueError("FanOutConvFn: output channels must fill whole 16-byte chunks")
            ys.append(ops.conv2d(x, _pack_weight(w, x.dtype, 0, cpad, Cin, 1, pad, pad)))
        ctx.save_for_backward(x, *weights)
        ctx.pad = pad
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        x, weights = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dx, dws = None, []
        for i, (w, dy) in enumerate(zip(weights, dys)):
            if dy is None:
                dws.append(None)
                continue
            need_x = ctx.needs_input_grad[0]
            d, dw, _ = _conv_backward(x, w, dy, 1, ctx.pad, False, w.shape[0], (need_x, ctx.needs_input_grad[2 + i], False), res=dx)
            dx = d if need_x else None
            dws.append(dw)
        return (dx, None, *dws)


def fanout_conv(x, weights, pad):
    """[conv2d(x, w, stride 1, pad) for w in weights] with ONE summed input gradient (FanOutConvFn)."""
    return list(FanOutConvFn.apply(x, pad, *weights))


def dcn_module(x, w_off, b_off, weight, bias, stride, pad, dil):
    """conv_offset_mask + DCNv2 of one module, differentiable; one fused node where the tile-owned backward applies."""
    Cout, C, kh, kw = weight.shape
    pow2 = lambda v: v >= 64 and (v & (v - 1)) == 0
    if (kh, kw, stride, pad, dil) == (3, 3, 1, 1, 1) and pow2(C) and pow2(Cout) and b_off is not None and not _DCN_BWD_V1[0] and not _DCN_TWO_NODES[0]:
        return DCNModuleFn.apply(x, w_off, b_off, weight, bias)
    om = Conv2dFn.apply(x, w_off, b_off, stride, pad, torch.float32, L.ACT_DCN_OFFMASK)
    return DCNFn.apply(x, om, weight, bias, stride, pad, dil, True)


_DCN_TWO_NODES = [False]       # tests: offset conv and DCN as two autograd nodes (their input gradients added by autograd)


@_device_guarded
class FocalLossFn(Function):
    """Penalty-reduced focal loss of the class heat map in one kernel (csrc/loss_kernels.hip): fp32 NHWC logits (B,H,W,C) and
    the NCHW Gaussian target map -> (loss_sum, num_pos); sigmoid + clamp(1e-4, 1-1e-4) are part of the kernel, and
    d(loss_sum)/d(logit) is written in the same pass, so backward is one scale."""

    @staticmethod
    def forward(ctx, logits, heat, alpha, beta):
        z, t = _c(logits).float(), _c(heat).float()
        B, H, W, C = z.shape
        sums = torch.empty(2, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        L.check(L.load().mfx_focal_loss(_ptr(z), _ptr(t), B, H, W, C, ctypes.c_float(alpha), ctypes.c_float(beta), _ptr(sums), _ptr(dz),
                                        _stream()), "mfx_focal_loss")
        ctx.save_for_backward(dz)
        loss_sum, num_pos = sums[0], sums[1]
        ctx.mark_non_differentiable(num_pos)                  # (the very tensor that is returned, not another view of it)
        return loss_sum, num_pos

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, g_np):
        (dz,) = ctx.saved_tensors
        return dz * g_loss, None, None, None


@_device_guarded
class ObjectLossFn(Function):
    """The nine per-object regression terms (detector_loss.py:116-482) in one launch (csrc/object_loss_math.h): fp32 NHWC map
    with the 50 regression channels at [ch_off, ch_off+50) of each pixel + the packed target rows -> (terms[10], logged[14]);
    the gradient row of every term is produced by the same launch (forward-mode tangents, one lane per channel), backward is a
    scatter-add of sum_t gout[t] * G[n][t][:] into a zero map."""

    @staticmethod
    def forward(ctx, reg_nhwc, rows, cfg, ch_off):
        reg = _c(reg_nhwc)
        if reg.dtype != torch.float32 or rows.dtype != torch.float32:
            raise TypeError("object loss: fp32 regression map and fp32 target rows")
        rows = _c(rows)
        if reg.dim() == 2:                                     # gathered table (N, ld): the kernels' B = 0 form
            if reg.shape[0] != rows.shape[0]:
                raise ValueError("object loss: a gathered table needs one row per object row")
            B, H, W, ld = 0, 1, reg.shape[0], reg.shape[1]
        else:
            B, H, W, ld = reg.shape
        N = rows.shape[0]
        vals = torch.empty(L.OBJ_VALUES, dtype=torch.float32, device=reg.device)
        G = torch.empty(N, L.OBJ_TERMS, 64, dtype=torch.float32, device=reg.device)
        L.check(L.load().mfx_object_loss(_ptr(reg), B, H, W, ld, ch_off, _ptr(rows), N, ctypes.byref(cfg), _ptr(vals), _ptr(G), _stream()),
                "mfx_object_loss")
        ctx.save_for_backward(G, rows)
        ctx.geom = (B, H, W, ld, ch_off)
        terms, logged = vals[:L.OBJ_TERMS], vals[L.OBJ_TERMS:]
        ctx.mark_non_differentiable(logged)
        return terms, logged

    @staticmethod
    @once_differentiable
    def backward(ctx, g_terms, g_logged):
        G, rows = ctx.saved_tensors
        B, H, W, ld, ch_off = ctx.geom
        dreg = torch.zeros((W, ld) if B == 0 else (B, H, W, ld), dtype=torch.float32, device=G.device)
        g = _c(g_terms.float())
        L.check(L.load().mfx_object_loss_backward(_ptr(G), _ptr(g), _ptr(rows), rows.shape[0], B, H, W, _ptr(dreg), ld, ch_off, _stream()),
                "mfx_object_loss_backward")
        return dreg, None, None, None


@_device_guarded
class SparseRegHeadsFn(Function):
    """Regression branches of the training step at the object centres only (csrc/head_sparse.hip): per branch the dense trunk
    conv output y (B,H,W,256), its ABN holder and the stacked 1x1 heads (w2 (k,256[,1,1]), b2 (k)) -> out fp32 [N][ld_out], row n
    = object row n of `rows`, branch i at columns [offs[i], offs[i]+k_i).  The ABN's batch statistics are taken over the dense
    map (running statistics and num_batches_tracked move as in bn_act); backward is one dense pass per branch."""

    @staticmethod
    def forward(ctx, rows, abns, offs, ld_out, stats_done, *ts):
        nb = len(abns)
        ys, gammas, betas, w2s, b2s = (ts[i * nb:(i + 1) * nb] for i in range(5))
        ys = [_c(y) for y in ys]
        y0 = ys[0]
        B, H, W, C = y0.shape
        lib_ = L.load()
        d = L.HeadSparseDesc()
        d.nbranch, d.N, d.B, d.H, d.W, d.C, d.dtype, d.ld_out = nb, rows.shape[0], B, H, W, C, _dt(y0.dtype), ld_out
        rows = _c(rows)
        d.rows = rows.data_ptr()
        keep = []
        stats = torch.empty(nb, 2, C, dtype=torch.float32, device=y0.device)              # mean | rstd per branch
        for i, abn in enumerate(abns):
            g32, b32 = _c(gammas[i].detach().float()), _c(betas[i].detach().float())
            w2 = _c(w2s[i].detach().float().reshape(w2s[i].shape[0], C))
            b2 = _c(b2s[i].detach().float()) if b2s[i] is not None else None
            mom = abn.momentum if abn.momentum is not None else 0.1
            nbt = abn.num_batches_tracked if (abn.track_running_stats and abn.num_batches_tracked is not None
                                              and abn.num_batches_tracked.dtype == torch.int64) else None
            L.check(lib_.mfx_bn_train_stats(_ptr(ys[i]), _ptr(g32), _ptr(b32), _ptr(abn.running_mean), _ptr(abn.running_var), _ptr(nbt),
                                            ctypes.c_float(mom), ctypes.c_float(abn.eps), B * H * W, C, _dt(y0.dtype), _ptr(_bn_scratch(gammas[i])),
                                            _ptr(stats[i, 0]), _ptr(stats[i, 1]), int(bool(stats_done[i])), _stream()), "mfx_bn_train_stats")
            d.y[i], d.mean[i], d.rstd[i] = ys[i].data_ptr(), stats[i, 0].data_ptr(), stats[i, 1].data_ptr()
            d.gamma[i], d.beta[i], d.w2[i], d.b2[i] = g32.data_ptr(), b32.data_ptr(), w2.data_ptr(), (b2.data_ptr() if b2 is not None else None)
            d.k[i], d.out_off[i] = w2.shape[0], offs[i]
            keep += [g32, b32, w2, b2]
        out = torch.zeros(rows.shape[0], ld_out, dtype=torch.float32, device=y0.device)
        d.out = out.data_ptr()
        L.check(lib_.mfx_head_sparse_fwd(ctypes.byref(d), _stream()), "mfx_head_sparse_fwd")
        ctx.save_for_backward(rows, stats, *ys, *[t for t in keep if t is not None])
        ctx.meta = (nb, offs, ld_out, [w.shape for w in w2s], [b is not None for b in b2s], [w.shape[0] for w in w2s])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        nb, offs, ld_out, wshapes, has_b, ks = ctx.meta
        saved = list(ctx.saved_tensors)
        rows, stats = saved[0], saved[1]
        ys = saved[2:2 + nb]
        rest = saved[2 + nb:]
        B, H, W, C = ys[0].shape
        dev = ys[0].device
        d = L.HeadSparseDesc()
        d.nbranch, d.N, d.B, d.H, d.W, d.C, d.dtype, d.ld_out = nb, rows.shape[0], B, H, W, C, _dt(ys[0].dtype), ld_out
        d.rows = rows.data_ptr()
        dout = _c(dout.float())
        d.dout = dout.data_ptr()
        sizes = [2 * C + k * C + k for k in ks]
        arena = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        g = torch.empty(nb, rows.shape[0], C, dtype=torch.float32, device=dev)
        d.g, d.arena, d.arena_bytes = g.data_ptr(), arena.data_ptr(), arena.numel() * 4
        dxs, sums, dw2s, db2s, o, r = [], [], [], [], 0, 0
        for i in range(nb):
            g32, b32, w2 = rest[r], rest[r + 1], rest[r + 2]
            r += 3
            b2 = None
            if has_b[i]:
                b2 = rest[r]; r += 1
            k = ks[i]
            sm, dw, db = arena[o:o + 2 * C], arena[o + 2 * C:o + 2 * C + k * C], arena[o + 2 * C + k * C:o + sizes[i]]
            o += sizes[i]
            dx = torch.empty_like(ys[i])
            d.y[i], d.mean[i], d.rstd[i] = ys[i].data_ptr(), stats[i, 0].data_ptr(), stats[i, 1].data_ptr()
            d.gamma[i], d.beta[i], d.w2[i], d.b2[i] = g32.data_ptr(), b32.data_ptr(), w2.data_ptr(), (b2.data_ptr() if b2 is not None else None)
            d.k[i], d.out_off[i] = k, offs[i]
            d.sums[i], d.dw2[i], d.db2[i], d.dx[i] = sm.data_ptr(), dw.data_ptr(), db.data_ptr(), dx.data_ptr()
            dxs.append(dx); sums.append(sm); dw2s.append(dw.view(wshapes[i])); db2s.append(db if has_b[i] else None)
        L.check(L.load().mfx_head_sparse_bwd(ctypes.byref(d), _stream()), "mfx_head_sparse_bwd")
        dgammas = [sm[C:] for sm in sums]
        dbetas = [sm[:C] for sm in sums]
        return (None, None, None, None, None, *dxs, *dgammas, *dbetas, *dw2s, *db2s)


def conv2d(x, weight, bias=None, stride=1, pad=0, out_dtype=None):
    """Differentiable NHWC conv; returns exactly weight.shape[0] channels."""
    y = Conv2dFn.apply(x, weight, bias, stride, pad, out_dtype)
    return y if y.shape[-1] == weight.shape[0] else y[..., :weight.shape[0]]


def bn_fuses_statistics(bn, sync=None):
    """True when train-mode `bn` takes the two-launch path whose statistics a producing conv may accumulate itself."""
    if _BN_SEPARATE[0] or _CONV_STATS_OFF[0] or bn.weight.numel() > 512:
        return False
    if sync is None:
        sync = bool(getattr(bn, 'sync_bn', False)) or isinstance(bn, torch.nn.SyncBatchNorm)
    return _sync_group(sync) is None


def conv2d_bn_stats(x, weight, bias, stride, pad, bn, alias=False):
    """conv2d whose epilogue also accumulates the batch statistics of its output for the train-mode BN `bn` that follows.
    Returns (y, stats_done): pass stats_done on to bn_act / SparseRegHeadsFn.  `alias`: (y, stats_done, x') -- Conv2dFn's second output."""
    if alias:
        fused = bn_fuses_statistics(bn) and weight.shape[0] <= _CONV_STATS_MAX_COUT[0] and x.shape[0] * x.shape[1] * x.shape[2] != 0
        y, xr = Conv2dFn.apply(x, weight, bias, stride, pad, None, L.ACT_NONE, _bn_scratch(bn.weight) if fused else None, True)
        done = bool(fused and ops.conv2d.last_stats_done)
        return (y if y.shape[-1] == weight.shape[0] else y[..., :weight.shape[0]]), done, xr
    # wide outputs pay more for the epilogue's atomics (one per column and wave, each covering only the wave's 128 pixels) than
    # the separate pass costs: 64->256 @ 96x320 went 80 -> 150 us against a 33 us statistics pass; up to 128 channels it is +1..2 us
    if not bn_fuses_statistics(bn) or weight.shape[0] > _CONV_STATS_MAX_COUT[0] or x.shape[0] * x.shape[1] * x.shape[2] == 0:
        return conv2d(x, weight, bias, stride, pad), False
    y = Conv2dFn.apply(x, weight, bias, stride, pad, None, L.ACT_NONE, _bn_scratch(bn.weight))
    done = ops.conv2d.last_stats_done
    return (y if y.shape[-1] == weight.shape[0] else y[..., :weight.shape[0]]), done


def bn_act(x, bn, act, res=None, sync=None, stats_done=False):
    """Train-mode BN module `bn` (+act, +res) on an NHWC tensor; updates bn.running_* like nn.BatchNorm2d.
    sync=None follows the module's `sync_bn` attribute (set by engine.trainer.convert_sync_batchnorm)."""
    mom = bn.momentum if bn.momentum is not None else 0.1
    nbt = bn.num_batches_tracked if (bn.track_running_stats and bn.num_batches_tracked is not None) else None
    if sync is None:                    # torch's own converter (the reference script's literal call) leaves SyncBatchNorm holders
        sync = bool(getattr(bn, 'sync_bn', False)) or isinstance(bn, torch.nn.SyncBatchNorm)
    return BNActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, res, act, mom, bn.eps, sync, nbt, stats_done)
