"""torch.autograd.Function wrappers over the C ABI (include/mogan_hip.h).

These are the ONLY callers of libmogan_hip.so.  PyTorch supplies device memory, streams and the
autograd tape; all arithmetic on tensors happens inside the HIP kernels.  Every function raises
(MoganHipError) if the library is missing or an input is not on the GPU -- there is no fallback.
"""
import contextlib
import os

import torch

from . import lib
from .lib import call, ptr, stream_ptr, workspace

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_GLU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4, 5

# Test hook: when set to a list, every (Leaky)ReLU launch appends (activation code, its OUTPUT tensor).  A gradient check against a CPU
# reference can then hand the reference the same sign decisions: with millions of pre-activations per layer a few land
# within fp32 noise of the kink, and one flipped decision moves the weight gradients below it by ~1e-3 (tests/
# test_fullwidth_parity_gpu.py).  None (the default) costs one global lookup per launch.
ACT_TRACE = None


# Deferred running statistics (include/mogan_hip.h: mogan_bn_running_update).  While this is a list, training-mode BatchNorm
# launches leave running_mean / running_var alone and append what the update needs; bn_apply_deferred() performs the updates
# later, in list order -- the arithmetic of a call runs early, its place in the reference's call order is kept
# (miscc/losses.py: the "wrong pair" head of a discriminator update).
BN_DEFER = None


def bn_apply_deferred(pending):
    for mean, invstd, n, rm, rv, eps, momentum in pending:
        if rm is not None or rv is not None:
            call("mogan_bn_running_update", ptr(mean), ptr(invstd), ptr(rm), ptr(rv), mean.numel(), int(n), float(eps),
                 float(momentum), stream_ptr())
    del pending[:]


TRACE_GROUPS = 1            # set by a paired pass for the launches that do not know about groups (no BatchNorm inside)


def _trace(act, y, groups=1):
    """one ACT_TRACE entry per reference call: a tensor that stands for `groups` calls (groups batches one behind the other)
    is recorded as its per-call slices; inside a paired pass (TRACE_GROUPS > 1) the entries carry the call's index as a third
    element (discriminator_loss puts them back into the reference's call order: every layer of D(real), then of D(fake))"""
    if groups <= 1:
        ACT_TRACE.append((act, y))
        return
    n = y.shape[0] // groups
    for g in range(groups):
        ACT_TRACE.append((act, y[g * n:(g + 1) * n], g) if TRACE_GROUPS > 1 else (act, y[g * n:(g + 1) * n]))


# When a parameter already owns a dense .grad (the trainer's flat gradient buckets), the weight-gradient
# kernels accumulate straight into it (C += ...) and the Function returns None for that input: same result as
# autograd's `grad += new`, without the temporary and the extra read-modify-write pass (~350 launches/step).
DIRECT_GRAD = True


# Data-parallel overlap hook (attngan/trainer.ChunkedReducer): address ranges of flat gradient buckets whose owner wants to
# know when a parameter's .grad has just received a contribution (the kernel is queued on `stream`).  Empty unless N > 1.
GRAD_HOOKS = []


def _grad_hit(g, stream=None):
    if GRAD_HOOKS:
        p = g.data_ptr()
        for lo, hi, cb in GRAD_HOOKS:
            if lo <= p < hi:
                cb(p, stream)
                break


def _grad_buf(param):
    if not DIRECT_GRAD or param is None:
        return None
    g = getattr(param, "grad", None)
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.shape != param.shape:
        return None
    return g


# Weight gradients have no consumer until the optimizer step, so (in DIRECT_GRAD mode) they are launched on a
# side stream paired with the stream the backward runs on: the dgrad chain continues on the main stream and the
# wgrad launches fill the CUs its tails leave idle.  Opt-in per backward call through `with wgrad_overlap():`
# (the engines wrap every .backward() in it); leaving the context makes the current stream wait for the side
# stream, so the all-reduce / Adam / anybody reading .grad afterwards is ordered behind the weight gradients.
WGRAD_SIDE_STREAM = False
CAPTURE_WGRAD_OK = set()     # raw handles of streams that may fork a wgrad stream while being captured (depth-1 forks only)
_WGRAD_ENV = os.environ.get("MOGAN_WGRAD_STREAM", "1") != "0"
_wgrad_streams = {}


class _WgradFrame:
    """What ONE `with wgrad_overlap():` (one backward pass) owns: `used` = side streams that received its launches, `keep` =
    operands of its in-flight side-stream launches (released after the join so the caching allocator cannot hand their memory to
    a main-stream kernel that runs concurrently), `parked` = keys of the contributions it parked in _wgrad_pending.  A join
    waits for and clears only its own frame -- a nested or interleaved context on another stream keeps its waits and operands."""
    __slots__ = ("used", "keep", "parked")

    def __init__(self):
        self.used, self.keep, self.parked = [], [], []


_wgrad_frames = [_WgradFrame()]      # [0] = launches outside any context (joined by a bare join_wgrad())


# Merged weight gradients: a D update back-propagates through the same discriminator twice (real, fake).  For the deep
# layers (few output pixels, tens of MB of weights) a weight-gradient launch is bound by the read-modify-write of dW, not by
# its K = B*OH*OW; the first contribution is therefore parked and launched together with the second as ONE pass over the
# concatenated batch (dW = dY_1 X_1^t + dY_2 X_2^t -- the same sum), leftovers at the join.
MERGE_WGRAD = os.environ.get("MOGAN_MERGE_WGRAD", "1") != "0"
_wgrad_pending = {}
# under hipGraph capture too (the branch graphs of trainer.TrainEngine replay the merged launches): the parked contribution and
# the concatenated operands live in the graph's memory pool
MERGE_WGRAD_CAPTURED = True
_wgrad_ctx_depth = 0         # parking needs somebody to flush: only inside `with wgrad_overlap():`
_MERGE_K = 2048
_MERGE_W = 1 << 21


def _wgrad_launch(dy, x, w_shape, geom, g):
    stride, ph, pw, up = geom
    if WGRAD_SIDE_STREAM:
        cur, side = _wgrad_stream()
        side.wait_stream(cur)                     # dy (and the zeroed / partly accumulated grad) are ready
        with torch.cuda.stream(side):
            conv2d_wgrad(dy, x, w_shape, stride, ph, pw, up, out=g, accumulate=True)
        fr = _wgrad_frames[-1]
        if side not in fr.used:
            fr.used.append(side)
        fr.keep.append((dy, x))                   # freed only after the join (see join_wgrad)
        _grad_hit(g, side)
    else:
        conv2d_wgrad(dy, x, w_shape, stride, ph, pw, up, out=g, accumulate=True)
        _grad_hit(g)


def _wgrad_accumulate(dy, x, w, geom, g):
    """dW += wgrad(dy, x) into the parameter's .grad buffer g, possibly deferred / merged (see MERGE_WGRAD)."""
    K = dy.shape[0] * dy.shape[2] * dy.shape[3]
    if not (MERGE_WGRAD and _wgrad_ctx_depth > 0 and K <= _MERGE_K and w.numel() >= _MERGE_W) \
            or (torch.cuda.is_current_stream_capturing() and not MERGE_WGRAD_CAPTURED):
        _wgrad_launch(dy, x, w.shape, geom, g)
        return
    cur = torch.cuda.current_stream()
    key = (g.data_ptr(), cur.cuda_stream)
    prev = _wgrad_pending.pop(key, None)
    if prev is None:
        # parked on THIS stream: autograd may be running the node on another stream than the one the context was entered on (the
        # tape of a replayed generator forward executes on its capture stream), so the flush goes by the context's own list of
        # keys and launches each leftover on the stream it was parked on
        _wgrad_pending[key] = (dy, x, tuple(w.shape), geom, g, cur)
        _wgrad_frames[-1].parked.append(key)
        return
    pdy, px, _, pgeom = prev[:4]
    if pgeom == geom and pdy.shape[1:] == dy.shape[1:] and px.shape[1:] == x.shape[1:]:
        _wgrad_launch(_cat_batch(pdy, dy), _cat_batch(px, x), w.shape, geom, g)
    else:
        _wgrad_launch(pdy, px, w.shape, pgeom, g)
        _wgrad_launch(dy, x, w.shape, geom, g)


def _cat_batch(a, b):
    """torch.cat([a, b]) along the batch axis of two dense tensors as one mogan_concat_fwd launch (the images of a tensor are the
    "channels" of a one-row concat)."""
    a, b = _c(a), _c(b)
    if a.shape[1:] != b.shape[1:]:
        raise lib.MoganHipError("_cat_batch: %r and %r differ behind the batch axis" % (tuple(a.shape), tuple(b.shape)))
    per = a[0].numel()
    out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=torch.float32, device=a.device)
    arrs = _cat_arrays([(a.shape[0], 1, 0, 0, 0), (b.shape[0], 1, 0, 0, 0)], [a.data_ptr(), b.data_ptr()])
    call("mogan_concat_fwd", *arrs, 2, ptr(out), 1, per, stream_ptr())
    return out


class SplitBatchFn(torch.autograd.Function):
    """x (N, ...) -> (x[:B], x[B:]) as views; the two gradients come back as ONE dense tensor through one concat launch
    (autograd's own slice backward is a zero fill + a copy + an add per half)."""

    @staticmethod
    def forward(ctx, x, B):
        ctx.B, ctx.shape = B, tuple(x.shape)
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, da, db):
        B, shape = ctx.B, ctx.shape
        if da is None:
            da = torch.zeros((B,) + shape[1:], dtype=torch.float32, device=db.device)
        if db is None:
            db = torch.zeros((shape[0] - B,) + shape[1:], dtype=torch.float32, device=da.device)
        return _cat_batch(da, db), None


def split_batch(x, B):
    return SplitBatchFn.apply(_c(x), int(B))


def _wgrad_flush():
    """launch the single contributions the innermost context parked and nobody merged -- each on the stream it was parked on; the
    current stream then waits for every such stream that is not itself (the optimizer reads .grad behind the context)"""
    cur = torch.cuda.current_stream()
    fr = _wgrad_frames[-1]
    keys, fr.parked = fr.parked, []
    if len(_wgrad_frames) == 1:                      # a bare flush outside any context: everything that is still parked
        keys = list(_wgrad_pending)
    capturing = torch.cuda.is_current_stream_capturing()
    for key in keys:
        item = _wgrad_pending.pop(key, None)
        if item is None:
            continue                                  # merged with a second contribution meanwhile
        dy, x, w_shape, geom, g, st = item
        if st.cuda_stream == cur.cuda_stream:
            _wgrad_launch(dy, x, w_shape, geom, g)
        else:
            with torch.cuda.stream(st):
                _wgrad_launch(dy, x, w_shape, geom, g)
            if not capturing:
                cur.wait_stream(st)


@contextlib.contextmanager
def wgrad_overlap():
    global WGRAD_SIDE_STREAM, _wgrad_ctx_depth
    # not under hipGraph capture: hipStreamEndCapture segfaults (ROCm 7.2) on the nested fork pattern
    # capture stream -> branch stream -> wgrad stream; captured steps keep the weight gradients in line
    ok = _WGRAD_ENV
    if ok and torch.cuda.is_current_stream_capturing():
        ok = torch.cuda.current_stream().cuda_stream in CAPTURE_WGRAD_OK
    prev, WGRAD_SIDE_STREAM = WGRAD_SIDE_STREAM, ok
    _wgrad_ctx_depth += 1
    _wgrad_frames.append(_WgradFrame())
    try:
        yield
    finally:
        try:
            _wgrad_flush()                # parked single contributions (still under this context's stream policy)
        finally:
            _wgrad_ctx_depth -= 1
            WGRAD_SIDE_STREAM = prev
            join_wgrad()
            _wgrad_frames.pop()


def _wgrad_stream():
    cur = torch.cuda.current_stream()
    side = _wgrad_streams.get(cur.cuda_stream)
    if side is None:
        side = _wgrad_streams[cur.cuda_stream] = torch.cuda.Stream()
    return cur, side


def precreate_wgrad_stream(stream):
    """Create the weight-gradient side stream of `stream` now.  HIP multiplexes streams onto a few hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default) in creation order, and two streams that share a queue serialize -- so the engines
    create their streams in one fixed order (branch streams, their wgrad streams, communication streams last) instead
    of leaving it to the first backward pass."""
    if stream.cuda_stream not in _wgrad_streams:
        _wgrad_streams[stream.cuda_stream] = torch.cuda.Stream()
    return _wgrad_streams[stream.cuda_stream]


def join_wgrad():
    """The current stream waits for the weight-gradient launches of the backward pass that just ended: for its own paired side
    stream AND for every side stream that received launches since the last join.  The second part matters when autograd ran the
    backward on ANOTHER stream than the caller's -- the nodes of a tape recorded on a capture stream (the generator's replayed
    forward, trainer.TrainEngine g_fwd_only) execute on that stream, so their weight gradients went to ITS side stream; joining
    only the caller's pair would let the optimizer read gradients that are still being written."""
    fr = _wgrad_frames[-1]
    if not _wgrad_streams:
        del fr.keep[:]
        return
    cur = torch.cuda.current_stream()
    side = _wgrad_streams.get(cur.cuda_stream)
    if side is not None:
        cur.wait_stream(side)
    capturing = torch.cuda.is_current_stream_capturing()
    for s_ in fr.used:
        if s_ is not side and not capturing:
            cur.wait_stream(s_)
    del fr.used[:]
    del fr.keep[:]


def _c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------- convolution
def conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, up):
    H, W = Hs << up, Ws << up
    return (H + 2 * ph - KH) // stride + 1, (W + 2 * pw - KW) // stride + 1


# nearest-x2 upsample + conv3x3(p1) runs as the transposed 4x4-s2 convolution (include/mogan_hip.h: mogan_upconv3x3_*):
# 2.25x fewer FLOPs.  MOGAN_UPCONV4=0 keeps the fused-upsample 3x3 kernels (the kernel tests compare both).
UPCONV4 = True


def _is_upconv(w_shape, stride, ph, pw, up):
    return UPCONV4 and bool(up) and stride == 1 and ph == 1 and pw == 1 and w_shape[2] == 3 and w_shape[3] == 3


# ------------------------------------------------------------------------------- packed weights (csrc/mogan_pgemm.hip)
# The weight-heavy convolutions (deep discriminator layers) read their filters from a PACKED copy: the three bf16 pieces of
# every weight in matrix-instruction order, one copy per direction (forward / data gradient).  The copy belongs to the
# parameter's owner: trainer.FlatAdam attaches a WeightPacks object to every 4-D parameter of its bucket and re-packs the
# copies in use right after each optimizer step (one pack per weight version, used by the real, the fake and the generator
# pass); anything else that writes weights calls FlatAdam.touch() / invalidate_all_packs().  A parameter without the
# attribute (plain modules, the kernel tests' default) takes the unpacked kernels.
PK_ENABLED = True
PK_STATS = {"fwd": 0, "dgrad": 0, "wgrad": 0, "packs": 0}      # launches through the packed path (tests, diagnostics)
PK_WGRAD = True          # the deep layers' weight gradients on the packed kernels too
_pk_wgrad_elig = {}
_PK_GLOBAL = [0]


def invalidate_all_packs():
    """Every packed weight copy of the process is stale (weights were written behind the owners' backs: load_state_dict,
    load_params, a test poking .data); they are rebuilt at their next use."""
    _PK_GLOBAL[0] += 1


class WeightPacks:
    """Packed copies of ONE convolution weight: slot[dgrad] = [buffer, version, global epoch, geometry, event, stream,
    packed inside a hipGraph capture]."""

    def __init__(self, w, version_cell=None):
        self.w = w
        self.cell = version_cell if version_cell is not None else [0]
        self.slots = {}
        self.elig = {}
        # prepared Winograd filter images (round 6; include/mogan_hip.h "Prepared filter images"): wino[dgrad] = [buffer, version,
        # global epoch, event, stream, prepared inside a capture]; wbytes[(dgrad, geometry)] = image size, 0 = not a Winograd layer
        self.wino = {}
        self.wbytes = {}
        # upsample + conv3x3 as the transposed 4x4 s2 convolution (mogan_upconv3x3_*): the virtual filters K = T w T^t of this weight
        # version, [buffer (Cin, Cout, 4, 4), version, global epoch, event, stream, built inside a capture], and a child WeightPacks
        # over K (same version cell) that holds the filter images of the kernels running the virtual convolution
        self.k4 = None
        self.k4pk = None

    def _pack(self, dgrad, slot):
        Cout, Cin, KH, KW = self.w.shape
        st = stream_ptr()
        stride, ph, pw = slot[3]
        call("mogan_pk_weight_pack", ptr(self.w), slot[0].data_ptr(), Cout, Cin, KH, KW, stride, ph, pw, dgrad, st)
        slot[1], slot[2] = self.cell[0], _PK_GLOBAL[0]
        ev = torch.cuda.Event()
        ev.record()
        slot[4], slot[5], slot[6] = ev, st, bool(lib._capturing())
        PK_STATS["packs"] += 1

    def repack(self):
        """re-pack every copy in use now, on the current stream (the owner just changed the weight): both directions from one
        read of the master where both are in use (mogan_pk_weight_pack_both)"""
        s0, s1 = self.slots.get(0), self.slots.get(1)
        Cout, Cin, KH, KW = self.w.shape
        if s0 is not None and s1 is not None and s0[3] == s1[3] and Cin % 32 == 0 and Cout % 32 == 0:
            stride, ph, pw = s0[3]
            st = stream_ptr()
            call("mogan_pk_weight_pack_both", ptr(self.w), s0[0].data_ptr(), s1[0].data_ptr(), Cout, Cin, KH, KW, stride, ph, pw, st)
            ev = torch.cuda.Event()
            ev.record()
            cap = bool(lib._capturing())
            for slot in (s0, s1):
                slot[1], slot[2], slot[4], slot[5], slot[6] = self.cell[0], _PK_GLOBAL[0], ev, st, cap
            PK_STATS["packs"] += 1
            return
        for dgrad, slot in self.slots.items():
            self._pack(dgrad, slot)

    # -- Winograd images -------------------------------------------------------------------------------------------------
    def wino_stale(self):
        """(dgrad, slot) of every image in use that is older than the weight"""
        return [(d, sl) for d, sl in self.wino.items() if sl[1] != self.cell[0] or sl[2] != _PK_GLOBAL[0]]

    def _wino_mark(self, sl, ev, st, cap):
        sl[1], sl[2], sl[3], sl[4], sl[5] = self.cell[0], _PK_GLOBAL[0], ev, st, cap

    def wino_pointer(self, dgrad, B, Hs, Ws, stride, ph, pw, up):
        """device pointer of this weight's prepared filter image for the direction, current and ordered before a use on the
        current stream -- or None: the convolution takes a kernel without one (or the library is a native-fp32 build).  The kind
        of image follows the filter size: the Winograd kernels' for 3x3 s1, dconv2_fwd_kernel's for 4x4 s2 (mogan_conv_prep_bytes)"""
        key = (dgrad, B, Hs, Ws, stride, ph, pw, up)
        nb = self.wbytes.get(key)
        if nb is None:
            Cout, Cin, KH, KW = self.w.shape
            nb = self.wbytes[key] = int(lib.load().mogan_conv_prep_bytes(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up, dgrad))
        if not nb:
            return None
        sl = self.wino.get(dgrad)
        if sl is None or sl[0].numel() < nb:
            sl = self.wino[dgrad] = [torch.empty(nb, dtype=torch.uint8, device=self.w.device), -1, -1, None, None, False]
        if sl[1] != self.cell[0] or sl[2] != _PK_GLOBAL[0]:
            wino_prep([(self, dgrad, sl)])
        elif sl[4] != stream_ptr() and (sl[5] or not lib._capturing()):
            torch.cuda.current_stream().wait_event(sl[3])
        return sl[0].data_ptr()

    # -- K of the up-convolution --------------------------------------------------------------------------------------------
    def k4_stale(self):
        return self.k4 is not None and (self.k4[1] != self.cell[0] or self.k4[2] != _PK_GLOBAL[0])

    def k4_build(self):
        """K = T w T^t of the current weight on the current stream (one launch)"""
        Cout, Cin = int(self.w.shape[0]), int(self.w.shape[1])
        st = stream_ptr()
        call("mogan_upconv3x3_k4", ptr(self.w), self.k4[0].data_ptr(), Cout, Cin, st)
        ev = torch.cuda.Event()
        ev.record()
        self.k4[1], self.k4[2], self.k4[3], self.k4[4], self.k4[5] = self.cell[0], _PK_GLOBAL[0], ev, st, bool(lib._capturing())
        PK_STATS["k4_builds"] = PK_STATS.get("k4_builds", 0) + 1

    def upconv_pointers(self, dgrad, B, Hs, Ws):
        """(K pointer, filter-image pointer or None) for the up-convolution of an (B, Cin, Hs, Ws) input with this 3x3 weight:
        direction 0 (forward) runs the DATA GRADIENT of the virtual 4x4 s2 convolution C4 (in = Cout, out = Cin) on 2Hs x 2Ws,
        direction 1 its forward.  Both current and ordered before a use on the current stream."""
        Cout, Cin = int(self.w.shape[0]), int(self.w.shape[1])
        if self.k4 is None:
            k = torch.empty((Cin, Cout, 4, 4), dtype=torch.float32, device=self.w.device)
            self.k4 = [k, -1, -1, None, None, False]
            self.k4pk = WeightPacks(k, self.cell)
        if self.k4_stale():
            self.k4_build()
        elif self.k4[4] != stream_ptr() and (self.k4[5] or not lib._capturing()):
            torch.cuda.current_stream().wait_event(self.k4[3])
        img = self.k4pk.wino_pointer(0 if dgrad else 1, B, 2 * Hs, 2 * Ws, 2, 1, 1, 0) if D2_PREP else None
        return self.k4[0].data_ptr(), img

    def _fresh(self, key, slot):
        """make the copy in `slot` current for a use on the current stream"""
        if slot[1] != self.cell[0] or slot[2] != _PK_GLOBAL[0]:
            self._pack(key, slot)
        elif slot[5] != stream_ptr() and (slot[6] or not lib._capturing()):
            # packed on another stream: order behind that pack (a capturing stream must not wait for an event recorded
            # outside its capture -- and need not: the device is synchronised before a capture begins)
            torch.cuda.current_stream().wait_event(slot[4])
        return slot[0].data_ptr()

    def pointer(self, dgrad, B, Hs, Ws, stride, ph, pw):
        """device pointer of the packed copy for this call's geometry, or None: take the unpacked kernels"""
        key = (dgrad, B, Hs, Ws, stride, ph, pw)
        e = self.elig.get(key)
        Cout, Cin, KH, KW = self.w.shape
        if e is None:
            e = self.elig[key] = bool(lib.load().mogan_pk_conv_eligible(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, dgrad))
        if not e:
            return None
        slot = self.slots.get(dgrad)
        if slot is None:
            nbytes = int(lib.load().mogan_pk_weight_bytes(Cout, Cin, KH, KW, stride, dgrad))
            buf = torch.empty(nbytes, dtype=torch.uint8, device=self.w.device)
            slot = self.slots[dgrad] = [buf, -1, -1, (stride, ph, pw), None, None, False]
        elif slot[3] != (stride, ph, pw):
            return None                                   # one weight, two convolution geometries: not a case of the step
        return self._fresh(dgrad, slot)


def wino_prep(items):
    """(WeightPacks, dgrad, slot) triples -> their prepared filter images rebuilt on the current stream in one launch per kind
    (mogan_conv_prep_group: the Winograd images of the 3x3 weights, dconv2's images of the 4x4 s2 weights)"""
    import ctypes
    n = len(items)
    if not n:
        return
    VP, CI = ctypes.c_void_p * n, ctypes.c_int * n
    ws = VP(*[it[0].w.data_ptr() for it in items])
    ps = VP(*[it[2][0].data_ptr() for it in items])
    co = CI(*[int(it[0].w.shape[0]) for it in items])
    ci = CI(*[int(it[0].w.shape[1]) for it in items])
    kh = CI(*[int(it[0].w.shape[2]) for it in items])
    dg = CI(*[int(it[1]) for it in items])
    st = stream_ptr()
    call("mogan_conv_prep_group", n, ctypes.cast(ws, ctypes.c_void_p), ctypes.cast(ps, ctypes.c_void_p), ctypes.cast(co, ctypes.c_void_p),
         ctypes.cast(ci, ctypes.c_void_p), ctypes.cast(kh, ctypes.c_void_p), ctypes.cast(dg, ctypes.c_void_p), st)
    ev = torch.cuda.Event()
    ev.record()
    cap = bool(lib._capturing())
    for pk, _, sl in items:
        pk._wino_mark(sl, ev, st, cap)
    PK_STATS["wino_preps"] = PK_STATS.get("wino_preps", 0) + 1


def k4_build_group(pks):
    """K = T w T^t of every pack in `pks` on the current stream in one launch (mogan_upconv3x3_k4_group)"""
    import ctypes
    n = len(pks)
    if not n:
        return
    VP, CI = ctypes.c_void_p * n, ctypes.c_int * n
    ws = VP(*[pk.w.data_ptr() for pk in pks])
    ks = VP(*[pk.k4[0].data_ptr() for pk in pks])
    co = CI(*[int(pk.w.shape[0]) for pk in pks])
    ci = CI(*[int(pk.w.shape[1]) for pk in pks])
    st = stream_ptr()
    call("mogan_upconv3x3_k4_group", n, ctypes.cast(ws, ctypes.c_void_p), ctypes.cast(ks, ctypes.c_void_p), ctypes.cast(co, ctypes.c_void_p),
         ctypes.cast(ci, ctypes.c_void_p), st)
    ev = torch.cuda.Event()
    ev.record()
    cap = bool(lib._capturing())
    for pk in pks:
        pk.k4[1], pk.k4[2], pk.k4[3], pk.k4[4], pk.k4[5] = pk.cell[0], _PK_GLOBAL[0], ev, st, cap
    PK_STATS["k4_builds"] = PK_STATS.get("k4_builds", 0) + 1


def repack_all(packs):
    """Every derived weight image of a bucket brought up to date on the current stream (the owner changed the weights): the
    packed panels pack by pack, the Winograd filter images of all of them in one launch."""
    items, k4s = [], []
    for pk in packs:
        pk.repack()
        items += [(pk, d, sl) for d, sl in pk.wino.items()]
        if pk.k4 is not None:                  # the virtual filters of an up-convolution, then the images built from them
            k4s.append(pk)
            items += [(pk.k4pk, d, sl) for d, sl in pk.k4pk.wino.items()]
    k4_build_group(k4s)
    wino_prep(items)


def _wino_prep_ptr(w, dgrad, B, Hs, Ws, stride, ph, pw, up):
    pk = getattr(w, "_mogan_pk", None)
    if pk is None or not WINO_PREP:
        return None
    return pk.wino_pointer(dgrad, B, Hs, Ws, stride, ph, pw, up)


WINO_PREP = True      # (module attribute: False = every convolution prepares its filters per call, as without an owner)


D2_PREP = True        # (module attribute: False = the 4x4 s2 convolutions prepare dconv2's filter image per call)
UPCONV_OWNED = True   # (module attribute: False = the up-convolutions build K = T w T^t per call, mogan_upconv3x3_fwd / _dgrad)


def _prep_kind(KH, KW, stride, up):
    """filter sizes that may have a prepared image: 3x3 s1 (Winograd) and 4x4 s2 (dconv2_fwd_kernel)"""
    return (KH == 3 and KW == 3 and stride == 1) or (D2_PREP and KH == 4 and KW == 4 and stride == 2 and not up)


def attach_packs(w, version_cell=None):
    pk = getattr(w, "_mogan_pk", None)
    if pk is None:
        pk = w._mogan_pk = WeightPacks(w, version_cell)
    elif version_cell is not None and pk.cell is not version_cell:
        # a new owner (a second FlatAdam over the same network): its version counter rules from now on
        pk.cell = version_cell
        for slot in list(pk.slots.values()) + list(pk.wino.values()):
            slot[1] = -1
        if pk.k4 is not None:
            pk.k4[1] = -1
            pk.k4pk.cell = version_cell
            for slot in pk.k4pk.wino.values():
                slot[1] = -1
    return pk


def _packed(w, dgrad, B, Hs, Ws, stride, ph, pw, up):
    if up or not PK_ENABLED:
        return None
    pk = getattr(w, "_mogan_pk", None)
    if pk is None:
        return None
    return pk.pointer(dgrad, B, Hs, Ws, stride, ph, pw)


def conv2d_forward(x, w, stride, ph, pw, up):
    B, Cin, Hs, Ws = x.shape
    Cout, _, KH, KW = w.shape
    wp = _packed(w, 0, B, Hs, Ws, stride, ph, pw, up)
    if wp is not None:
        OH, OW = conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, 0)
        y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
        wsp, wsn = workspace(x.device)
        call("mogan_conv2d_fwd_pk", ptr(x), wp, ptr(y), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, wsp, wsn, stream_ptr())
        PK_STATS["fwd"] += 1
        return y
    if _is_upconv(w.shape, stride, ph, pw, up):
        y = torch.empty((B, Cout, 2 * Hs, 2 * Ws), dtype=torch.float32, device=x.device)
        wsp, wsn = workspace(x.device)
        pk = getattr(w, "_mogan_pk", None)
        if pk is not None and WINO_PREP and UPCONV_OWNED:     # the owner's K = T w T^t (and filter image) of this weight version
            k4, img = pk.upconv_pointers(0, B, Hs, Ws)
            call("mogan_conv2d_dgrad_wp", ptr(x), k4, img, ptr(y), B, Cout, 2 * Hs, 2 * Ws, Cin, 4, 4, 2, 1, 1, 0, wsp, wsn, stream_ptr())
            return y
        call("mogan_upconv3x3_fwd", ptr(x), ptr(w), ptr(y), B, Cin, Hs, Ws, Cout, wsp, wsn, stream_ptr())
        return y
    OH, OW = conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, up)
    y = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
    wsp, wsn = workspace(x.device)
    wprep = _wino_prep_ptr(w, 0, B, Hs, Ws, stride, ph, pw, up) if _prep_kind(KH, KW, stride, up) else None
    if wprep is not None:        # the owner's prepared Winograd filter image of this weight version (no transform launch here)
        call("mogan_conv2d_fwd_wp", ptr(x), ptr(w), wprep, ptr(y), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up,
             wsp, wsn, stream_ptr())
        return y
    call("mogan_conv2d_fwd", ptr(x), ptr(w), ptr(y), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up,
         wsp, wsn, stream_ptr())
    return y


def conv2d_dgrad(dy, w, x_shape, stride, ph, pw, up):
    B, Cin, Hs, Ws = x_shape
    Cout, _, KH, KW = w.shape
    wsp, wsn = workspace(dy.device)
    wp = _packed(w, 1, B, Hs, Ws, stride, ph, pw, up)
    if wp is not None:
        dx = torch.empty((B, Cin, Hs, Ws), dtype=torch.float32, device=dy.device)
        call("mogan_conv2d_dgrad_pk", ptr(dy), wp, ptr(dx), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, wsp, wsn, stream_ptr())
        PK_STATS["dgrad"] += 1
        return dx
    if _is_upconv(w.shape, stride, ph, pw, up):          # the gradient comes out at the source resolution
        dx = torch.empty((B, Cin, Hs, Ws), dtype=torch.float32, device=dy.device)
        pk = getattr(w, "_mogan_pk", None)
        if pk is not None and WINO_PREP and UPCONV_OWNED:
            k4, img = pk.upconv_pointers(1, B, Hs, Ws)
            call("mogan_conv2d_fwd_wp", ptr(dy), k4, img, ptr(dx), B, Cout, 2 * Hs, 2 * Ws, Cin, 4, 4, 2, 1, 1, 0, wsp, wsn, stream_ptr())
            return dx
        call("mogan_upconv3x3_dgrad", ptr(dy), ptr(w), ptr(dx), B, Cin, Hs, Ws, Cout, wsp, wsn, stream_ptr())
        return dx
    du = torch.empty((B, Cin, Hs << up, Ws << up), dtype=torch.float32, device=dy.device)
    wprep = _wino_prep_ptr(w, 1, B, Hs, Ws, stride, ph, pw, up) if _prep_kind(KH, KW, stride, up) else None
    if wprep is not None:
        call("mogan_conv2d_dgrad_wp", ptr(dy), ptr(w), wprep, ptr(du), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up,
             wsp, wsn, stream_ptr())
    else:
        call("mogan_conv2d_dgrad", ptr(dy), ptr(w), ptr(du), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up,
             wsp, wsn, stream_ptr())
    if not up:
        return du
    dx = torch.empty((B, Cin, Hs, Ws), dtype=torch.float32, device=dy.device)
    call("mogan_down2_sum", ptr(du), ptr(dx), B * Cin, Hs, Ws, stream_ptr())
    return dx


def conv2d_wgrad(dy, x, w_shape, stride, ph, pw, up, out=None, accumulate=False):
    B, Cin, Hs, Ws = x.shape
    Cout, _, KH, KW = w_shape
    wsp, wsn = workspace(dy.device)
    dw = out if out is not None else torch.empty(w_shape, dtype=torch.float32, device=dy.device)
    if PK_ENABLED and PK_WGRAD and not up:
        key = (B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, wsn)
        e = _pk_wgrad_elig.get(key)
        if e is None:
            e = _pk_wgrad_elig[key] = bool(lib.load().mogan_pk_wgrad_eligible(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, wsn))
        if e:
            call("mogan_conv2d_wgrad_pk", ptr(dy), ptr(x), ptr(dw), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw,
                 1 if accumulate else 0, wsp, wsn, stream_ptr())
            PK_STATS["wgrad"] = PK_STATS.get("wgrad", 0) + 1
            return dw
    if _is_upconv(w_shape, stride, ph, pw, up):
        call("mogan_upconv3x3_wgrad", ptr(dy), ptr(x), ptr(dw), B, Cin, Hs, Ws, Cout, 1 if accumulate else 0, wsp, wsn,
             stream_ptr())
        return dw
    call("mogan_conv2d_wgrad", ptr(dy), ptr(x), ptr(dw), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, up,
         1 if accumulate else 0, wsp, wsn, stream_ptr())
    return dw


class Conv2dFn(torch.autograd.Function):
    """y = conv2d(upsample2x?(x), w) (+ bias).  model.py:35-55,587,598-609,626,664-677."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, ph, pw, up):
        x, w = _c(x), _c(w)
        y = conv2d_forward(x, w, stride, ph, pw, up)
        if bias is not None:
            call("mogan_bias_add", ptr(y), ptr(_c(bias)), y.shape[0], y.shape[1], y.shape[2] * y.shape[3],
                 stream_ptr())
        ctx.save_for_backward(x, w)
        ctx.bias_ref = bias
        ctx.geom = (stride, ph, pw, up, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, ph, pw, up, has_bias = ctx.geom
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = conv2d_dgrad(dy, w, x.shape, stride, ph, pw, up)
        if ctx.needs_input_grad[1]:
            g = _grad_buf(w)
            if g is not None:
                _wgrad_accumulate(dy, x, w, (stride, ph, pw, up), g)
            else:
                dw = conv2d_wgrad(dy, x, w.shape, stride, ph, pw, up)
        if has_bias and ctx.needs_input_grad[2]:
            g = _grad_buf(ctx.bias_ref)
            db = g if g is not None else torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
            call("mogan_bias_grad", ptr(dy), ptr(db), dy.shape[0], dy.shape[1], dy.shape[2] * dy.shape[3],
                 1 if g is not None else 0, stream_ptr())
            if g is not None:
                db = None
                _grad_hit(g)
        return dx, dw, db, None, None, None, None


class ConvAffineReluFn(torch.autograd.Function):
    """z = relu(scale[c] * conv2d(x, w) + shift[c]) in one launch (the affine + ReLU ride in the conv epilogue or in its
    split-K reduction): BasicConv2d of the frozen eval-mode Inception trunk (model.py:258-299).  Saves x, w and the
    OUTPUT z; backward: g = dz * scale * (z > 0), then the ordinary data / weight gradients of the conv."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, stride, ph, pw):
        x, w = _c(x), _c(w)
        B, Cin, Hs, Ws = x.shape
        Cout, _, KH, KW = w.shape
        OH, OW = conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, 0)
        z = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
        wsp, wsn = workspace(x.device)
        call("mogan_conv2d_affine_fwd", ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(z), B, Cin, Hs, Ws, Cout, KH, KW,
             stride, ph, pw, 1, wsp, wsn, stream_ptr())
        ctx.save_for_backward(x, w, z, scale)
        ctx.geom = (stride, ph, pw)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, z, scale = ctx.saved_tensors
        stride, ph, pw = ctx.geom
        dz = _c(dz)
        g = torch.empty_like(z)
        call("mogan_affine_relu_bwd_out", ptr(z), ptr(dz), ptr(scale), ptr(g), z.shape[0], z.shape[1],
             z.shape[2] * z.shape[3], stream_ptr())
        dx = conv2d_dgrad(g, w, x.shape, stride, ph, pw, 0) if ctx.needs_input_grad[0] else None
        dw = conv2d_wgrad(g, x, w.shape, stride, ph, pw, 0) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None, None, None


class ConvLReLUFn(torch.autograd.Function):
    """z = LeakyReLU(conv2d(x, w)) with the activation in the convolution's epilogue (mogan_conv2d_lrelu_fwd): the first layer
    of every discriminator (model.py:597-598, 660-661).  Saves x, w and the OUTPUT z; backward: g = dz * (z > 0 ? 1 : slope)
    (z has the pre-activation's sign), then the ordinary data / weight gradients."""

    @staticmethod
    def forward(ctx, x, w, stride, ph, pw, slope):
        x, w = _c(x), _c(w)
        B, Cin, Hs, Ws = x.shape
        Cout, _, KH, KW = w.shape
        OH, OW = conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, 0)
        z = torch.empty((B, Cout, OH, OW), dtype=torch.float32, device=x.device)
        wsp, wsn = workspace(x.device)
        rc = lib.load().mogan_conv2d_lrelu_fwd(ptr(x), ptr(w), ptr(z), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, slope,
                                               wsp, wsn, stream_ptr())
        if rc == 1:                                        # not a geometry for the fused epilogue
            z = conv2d_forward(x, w, stride, ph, pw, 0)
            call("mogan_act_fwd", ptr(z), ptr(z), B, Cout, OH * OW, ACT_LRELU, slope, stream_ptr())
        elif rc != 0:
            raise lib.MoganHipError("mogan_conv2d_lrelu_fwd failed: %s" % lib._ERRORS.get(rc, rc))
        if ACT_TRACE is not None:
            _trace(ACT_LRELU, z, TRACE_GROUPS)
        ctx.save_for_backward(x, w, z)
        ctx.cfg = (stride, ph, pw, slope)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, z = ctx.saved_tensors
        stride, ph, pw, slope = ctx.cfg
        g = torch.empty_like(z)
        call("mogan_act_bwd", ptr(z), ptr(_c(dz)), ptr(g), z.shape[0], z.shape[1], z.shape[2] * z.shape[3], ACT_LRELU, slope,
             stream_ptr())
        dx = conv2d_dgrad(g, w, x.shape, stride, ph, pw, 0) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            gb = _grad_buf(w)
            if gb is not None:
                _wgrad_accumulate(g, x, w, (stride, ph, pw, 0), gb)
            else:
                dw = conv2d_wgrad(g, x, w.shape, stride, ph, pw, 0)
        return dx, dw, None, None, None, None


class LogitsHeadFn(torch.autograd.Function):
    """p = sigmoid(conv(x, w) + bias).view(-1) for a convolution whose filter covers the whole map (D_GET_LOGITS.outlogits,
    model.py:626-627, 640-641): one launch forward, one backward (mogan_logits_head_*)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x, w = _c(x), _c(w)
        B, Cout, K = x.shape[0], w.shape[0], w[0].numel()
        p = torch.empty((B, Cout, 1, 1), dtype=torch.float32, device=x.device)
        call("mogan_logits_head_fwd", ptr(x), ptr(w), ptr(_c(bias)) if bias is not None else None, ptr(p), B, K, Cout,
             stream_ptr())
        ctx.save_for_backward(x, w, p)
        ctx.bias_ref = bias
        return p

    @staticmethod
    def backward(ctx, dp):
        x, w, p = ctx.saved_tensors
        bias = ctx.bias_ref
        B, Cout, K = x.shape[0], w.shape[0], w[0].numel()
        dp = _c(dp)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1]
        want_b = bias is not None and ctx.needs_input_grad[2]
        gw = _grad_buf(w) if want_w else None
        gb = _grad_buf(bias) if want_b else None
        # one accumulate flag for both: direct only when every wanted gradient has its buffer
        direct = (want_w or want_b) and (not want_w or gw is not None) and (not want_b or gb is not None)
        dw = gw if direct else (torch.empty_like(w) if want_w else None)
        db = (gb if direct else torch.empty_like(bias)) if want_b else None
        call("mogan_logits_head_bwd", ptr(dp), ptr(p), ptr(x), ptr(w), ptr(dx), ptr(dw), ptr(db), B, K, Cout,
             1 if direct else 0, stream_ptr())
        if direct:
            if gw is not None:
                _grad_hit(gw)
            if gb is not None:
                _grad_hit(gb)
            dw = db = None
        return dx, dw, db


def logits_head(x, w, bias):
    return LogitsHeadFn.apply(x, w, bias)


def logits_head_ok(x, conv):
    """conv + Sigmoid as the fused head: the filter covers the whole (unpadded) map, <= 4 logits, K a multiple of 4"""
    w = conv.weight
    ph, pw = conv.padding if isinstance(conv.padding, tuple) else (conv.padding, conv.padding)
    if not (x.dim() == 4 and ph == 0 and pw == 0 and w.shape[2] == x.shape[2] and w.shape[3] == x.shape[3] and w.shape[0] <= 4
            and w[0].numel() % 4 == 0 and x.shape[1] == w.shape[1]):
        return False
    # the kernels read x and w as 16-byte quads: a dense view whose storage offset is not a multiple of four floats (a batch
    # slice of an odd-sized tensor) takes the convolution + sigmoid path instead of failing with MOGAN_ERR_SHAPE
    # (a non-contiguous operand is copied to a fresh, aligned tensor by the Function)
    return (not x.is_contiguous() or x.data_ptr() % 16 == 0) and (not w.is_contiguous() or w.data_ptr() % 16 == 0)


def conv2d_lrelu(x, w, stride, padding, slope=0.2):
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    return ConvLReLUFn.apply(x, w, int(stride), int(ph), int(pw), float(slope))


def conv2d_affine_relu(x, w, scale, shift, stride=1, padding=0):
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    return ConvAffineReluFn.apply(x, w, scale, shift, int(stride), int(ph), int(pw))


def conv2d(x, w, bias=None, stride=1, padding=0, up=False):
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    return Conv2dFn.apply(x, w, bias, int(stride), int(ph), int(pw), 1 if up else 0)


# ------------------------------------------------------------------------------- deep block: conv + BN + activation
DEEP_STATS = {"fwd": 0, "bwd": 0, "panel_hits": 0}
DEEP_ENABLED = True       # 0: packed GEMMs, but BatchNorm / activation as separate launches
_deep_elig = {}


def pk_debug_force(take_all, cfg=-1, split=0):
    """test hook (mogan_pk_debug_force) + the host-side eligibility caches it invalidates"""
    lib.load().mogan_pk_debug_force(int(take_all), int(cfg), int(split))
    _deep_elig.clear()
    _pk_wgrad_elig.clear()


def deep_block_eligible(x, w, stride, ph, pw, act, groups=1):
    """conv -> BatchNorm(train) -> act as the fused deep block (csrc/mogan_pgemm.hip): the weight has packed copies, both
    directions of the convolution take the packed path and the output map is small enough for the one-block-per-8-channels
    tail kernels.  groups: BatchNorm calls the batch stands for (see deep_conv_bn_act)."""
    if not (PK_ENABLED and DEEP_ENABLED) or getattr(w, "_mogan_pk", None) is None or x.dim() != 4:
        return False
    B, Cin, Hs, Ws = x.shape
    Cout, _, KH, KW = w.shape
    key = (B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, act, groups)
    e = _deep_elig.get(key)
    if e is None:
        e = _deep_elig[key] = bool(lib.load().mogan_deep_block_eligible(B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, act,
                                                                        groups))
    if e:
        # one weight used with two convolution geometries has packed copies for the first one only (WeightPacks.pointer
        # returns None for the other): such a call takes the unfused path
        pk = w._mogan_pk
        for d in (0, 1):
            slot = pk.slots.get(d)
            if slot is not None and slot[3] != (stride, ph, pw):
                return False
    return e


class DeepConvBNActFn(torch.autograd.Function):
    """z = act(BatchNorm2d_train(conv2d(x, w))) for the deep discriminator layers (model.py:575-613: downBlock,
    Block3x3_leakRelu; 616-642: jointConv) in two launches forward (packed-weight GEMM, tail) and two + the weight gradient
    backward.  The output carries the pixel panel of z for the next deep block (attribute _mogan_panel).  groups = 2: the batch
    is [real; fake] of a discriminator update (miscc/losses.py:136-174) -- one convolution, BatchNorm statistics per half."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, act, slope, eps, momentum, stride, ph, pw, groups):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        B, Cin, Hs, Ws = x.shape
        Cout, _, KH, KW = w.shape
        OH, OW = conv_out_hw(Hs, Ws, KH, KW, stride, ph, pw, 0)
        dev = x.device
        pk = w._mogan_pk
        wp = pk.pointer(0, B, Hs, Ws, stride, ph, pw)
        if wp is None:          # (deep_block_eligible checks both directions; a NULL panel must never reach the kernel)
            raise lib.MoganHipError("deep block: no packed forward copy of this weight for geometry %r" % ((stride, ph, pw),))
        f32 = dict(dtype=torch.float32, device=dev)
        y = torch.empty((B, Cout, OH, OW), **f32)
        z = torch.empty((B, Cout, OH, OW), **f32)
        stats = torch.empty((2, groups, Cout), **f32)
        zpanel = torch.empty(int(lib.load().mogan_pk_panel_bytes(B, Cout, OH * OW)), dtype=torch.uint8, device=dev)
        xp = getattr(x, "_mogan_panel", None)
        if xp is not None and (xp[1] != x.data_ptr() or xp[2] != x._version):
            xp = None                                    # not the panel of these values
        if xp is not None:
            DEEP_STATS["panel_hits"] += 1
        wsp, wsn = workspace(dev)
        if BN_DEFER is not None:
            if groups != 1:
                raise lib.MoganHipError("deferred running statistics: one BatchNorm call per launch only")
            BN_DEFER.append((stats[0, 0], stats[1, 0], B * OH * OW, running_mean, running_var, eps, momentum))
            running_mean = running_var = None
        call("mogan_deep_conv_bn_act_fwd", ptr(x), ptr(xp[0]) if xp is not None else None, wp, ptr(gamma), ptr(beta),
             ptr(running_mean), ptr(running_var), ptr(y), ptr(stats), ptr(z), ptr(zpanel), B, Cin, Hs, Ws, Cout, KH, KW,
             stride, ph, pw, eps, momentum, act, slope, groups, wsp, wsn, stream_ptr())
        DEEP_STATS["fwd"] += 1
        if ACT_TRACE is not None and act in (ACT_RELU, ACT_LRELU):
            _trace(act, z, groups)
        ctx.save_for_backward(x, w, y, stats, gamma, beta)
        ctx.cfg = (act, slope, stride, ph, pw, groups)
        z._mogan_panel = (zpanel, z.data_ptr(), z._version)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, y, stats, gamma, beta = ctx.saved_tensors
        act, slope, stride, ph, pw, groups = ctx.cfg
        dz = _c(dz)
        B, Cin, Hs, Ws = x.shape
        Cout, _, KH, KW = w.shape
        dev = x.device
        dy = torch.empty_like(y)
        want_dx = ctx.needs_input_grad[0]
        dx = torch.empty(x.shape, dtype=torch.float32, device=dev) if want_dx else None
        wpd = w._mogan_pk.pointer(1, B, Hs, Ws, stride, ph, pw) if want_dx else None
        if want_dx and wpd is None:
            raise lib.MoganHipError("deep block: no packed data-gradient copy of this weight for geometry %r" % ((stride, ph, pw),))
        gg, gb = _grad_buf(gamma), _grad_buf(beta)
        want_gb = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        direct = gg is not None and gb is not None and ctx.needs_input_grad[2] and ctx.needs_input_grad[3]
        dg = db = None
        if direct:
            pg, pb = gg, gb
        elif want_gb:
            dgb = torch.empty((2, Cout), dtype=torch.float32, device=dev)
            pg, pb = dgb[0], dgb[1]
        else:
            pg = pb = None
        wsp, wsn = workspace(dev)
        call("mogan_deep_conv_bn_act_bwd", ptr(dz), ptr(y), ptr(stats), ptr(gamma), ptr(beta), wpd, ptr(dy), ptr(pg), ptr(pb),
             1 if direct else 0, ptr(dx), B, Cin, Hs, Ws, Cout, KH, KW, stride, ph, pw, act, slope, groups, wsp, wsn,
             stream_ptr())
        DEEP_STATS["bwd"] += 1
        if direct:
            _grad_hit(gg)
            _grad_hit(gb)
        elif want_gb:
            dg, db = pg, pb
        dw = None
        if ctx.needs_input_grad[1]:
            g = _grad_buf(w)
            if g is not None:
                _wgrad_accumulate(dy, x, w, (stride, ph, pw, 0), g)
            else:
                dw = conv2d_wgrad(dy, x, w.shape, stride, ph, pw, 0)
        return (dx, dw, dg, db) + (None,) * 10


def deep_conv_bn_act(x, w, gamma, beta, running_mean, running_var, act, slope, eps, momentum, stride, ph, pw, groups=1):
    return DeepConvBNActFn.apply(x, w, gamma, beta, running_mean, running_var, int(act), float(slope), float(eps),
                                 float(momentum), int(stride), int(ph), int(pw), int(groups))


# ------------------------------------------------------------------------------- strided bmm / linear
def bmm_raw(a, b, out, accumulate=False):
    """out[z] (+)= a[z] @ b[z] for 3-D strided views (no copies)."""
    Z_, M, K = a.shape
    N = b.shape[2]
    wsp, wsn = workspace(a.device)
    call("mogan_bmm", ptr(a), ptr(b), ptr(out), Z_, M, N, K,
         a.stride(0), a.stride(1), a.stride(2), b.stride(0), b.stride(1), b.stride(2),
         out.stride(0), out.stride(1), out.stride(2), 1 if accumulate else 0, wsp, wsn, stream_ptr())
    return out


class BmmFn(torch.autograd.Function):
    """(Z,M,K) x (Z,K,N) -> (Z,M,N); inputs may be arbitrary strided views (torch.bmm call sites of
    GlobalAttention.py:46,66)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.float(), b.float()
        out = torch.empty((a.shape[0], a.shape[1], b.shape[2]), dtype=torch.float32, device=a.device)
        bmm_raw(a, b, out)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        dout = _c(dout)
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty(a.shape, dtype=torch.float32, device=a.device)
            bmm_raw(dout, b.transpose(1, 2), da)
        if ctx.needs_input_grad[1]:
            db = torch.empty(b.shape, dtype=torch.float32, device=a.device)
            bmm_raw(a.transpose(1, 2), dout, db)
        return da, db


def bmm(a, b):
    return BmmFn.apply(a, b)


class LinearFn(torch.autograd.Function):
    """y = x @ w.T (+ bias); nn.Linear at model.py:324,365,371."""

    @staticmethod
    def forward(ctx, x, w, bias):
        x, w = _c(x), _c(w)
        y = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
        bmm_raw(x.unsqueeze(0), w.t().unsqueeze(0), y.unsqueeze(0))
        if bias is not None:
            call("mogan_bias_add", ptr(y), ptr(_c(bias)), y.shape[0], y.shape[1], 1, stream_ptr())
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            bmm_raw(dy.unsqueeze(0), w.unsqueeze(0), dx.unsqueeze(0))
        if ctx.needs_input_grad[1]:
            g = _grad_buf(w)
            if g is not None:
                bmm_raw(dy.t().unsqueeze(0), x.unsqueeze(0), g.unsqueeze(0), accumulate=True)
                _grad_hit(g)
            else:
                dw = torch.empty(w.shape, dtype=torch.float32, device=x.device)
                bmm_raw(dy.t().unsqueeze(0), x.unsqueeze(0), dw.unsqueeze(0))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g = _grad_buf(ctx.bias_ref)
            db = g if g is not None else torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
            call("mogan_bias_grad", ptr(dy), ptr(db), dy.shape[0], dy.shape[1], 1, 1 if g is not None else 0,
                 stream_ptr())
            if g is not None:
                db = None
                _grad_hit(g)
        return dx, dw, db


def linear(x, w, bias=None):
    return LinearFn.apply(x, w, bias)


# ------------------------------------------------------------------------------- batch norm (+act)
def _bchw(x):
    if x.dim() == 2:
        return x.shape[0], x.shape[1], 1
    return x.shape[0], x.shape[1], x[0, 0].numel()


class BNActFn(torch.autograd.Function):
    """Training-mode BatchNorm1d/2d fused with GLU / LeakyReLU / ReLU and an optional residual add
    (model.py:52-54,72-80,96-101,366-373,577-611).  Updates the running statistics in place like
    nn.BatchNorm does (momentum 0.1, unbiased variance)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, act, slope, eps, momentum):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        B, C, HW = _bchw(x)
        dev = x.device
        stats = torch.empty((2, C), dtype=torch.float32, device=dev)
        need = lib.bn_ws_bytes(B, C, HW)
        wsp, wsn = workspace(dev)
        if need > wsn:
            raise lib.MoganHipError("workspace too small for bn (%d > %d)" % (need, wsn))
        # one call: small maps (<= 4096 values per channel) take ONE launch (a block per channel reduces, finalises and applies),
        # larger ones two (partial sums; apply with the reduction of the partial sums folded in: per wave, an xor-butterfly, no
        # barrier -- round 2's version of that fold, every block re-reducing behind a barrier, was 3 % slower in the step and
        # dropped; this one is +0.3 %), planes whose size is not a multiple of 4 three.
        Cy = C // 2 if act == ACT_GLU else C
        y = torch.empty((B, Cy) + tuple(x.shape[2:]), dtype=torch.float32, device=dev)
        res = _c(residual) if residual is not None else None
        if BN_DEFER is not None:
            BN_DEFER.append((stats[0], stats[1], B * HW, running_mean, running_var, eps, momentum))
            running_mean = running_var = None
        call("mogan_bn_act_fwd_fused", ptr(x), ptr(gamma), ptr(beta), ptr(res), ptr(running_mean), ptr(running_var),
             ptr(stats[0]), ptr(stats[1]), ptr(y), B, C, HW, act, slope, eps, momentum, wsp, wsn, stream_ptr())
        if ACT_TRACE is not None and act in (ACT_RELU, ACT_LRELU):
            ACT_TRACE.append((act, y))
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (act, slope, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        act, slope, has_res = ctx.cfg
        dy = _c(dy)
        B, C, HW = _bchw(x)
        dx = torch.empty_like(x)
        gg, gb = _grad_buf(gamma), _grad_buf(beta)
        direct = gg is not None and gb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        wsp, wsn = workspace(x.device)
        if direct:
            dg, db = gg, gb
        else:
            dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
            dg, db = dgb[0], dgb[1]
        call("mogan_bn_act_bwd", ptr(x), ptr(dy), ptr(stats[0]), ptr(stats[1]), ptr(gamma), ptr(beta), ptr(dx),
             ptr(dg), ptr(db), B, C, HW, act, slope, 1 if direct else 0, wsp, wsn, stream_ptr())
        if direct:
            _grad_hit(gg)
            _grad_hit(gb)
            dg = db = None
        return dx, dg, db, (dy if has_res else None), None, None, None, None, None, None


class BNActGroupedFn(torch.autograd.Function):
    """`groups` training-mode BatchNorm(+activation) calls on the groups of B images of one (groups*B, C, ...) tensor: own batch
    statistics per group, running statistics updated group after group (the per-object BatchNorm calls of the object pathways,
    model.py:395-407, 662-672; SURVEY F11 -- and, round 5, the [real; fake] batch of a discriminator update,
    miscc/losses.py:136-174).  One launch each way where a group has <= 4096 values per channel; larger maps: the two-launch
    kernels of BNActFn once per group on the group's slice of x / y (the groups are contiguous: no copies, no concatenation)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, act, slope, eps, momentum, groups):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        N, C, HW = _bchw(x)
        B = N // groups
        stats = torch.empty((2, groups, C), dtype=torch.float32, device=x.device)
        Cy = C // 2 if act == ACT_GLU else C
        y = torch.empty((N, Cy) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        one = bn_groups_ok(x, groups)
        if BN_DEFER is not None:
            raise lib.MoganHipError("deferred running statistics: one BatchNorm call per launch only")
        if one:
            call("mogan_bn_act_grouped_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(stats[0]),
                 ptr(stats[1]), ptr(y), groups, B, C, HW, act, slope, eps, momentum, stream_ptr())
        else:
            wsp, wsn = workspace(x.device)
            if lib.bn_ws_bytes(B, C, HW) > wsn:
                raise lib.MoganHipError("workspace too small for bn")
            for g in range(groups):
                call("mogan_bn_act_fwd_fused", ptr(x[g * B:(g + 1) * B]), ptr(gamma), ptr(beta), None, ptr(running_mean),
                     ptr(running_var), ptr(stats[0, g]), ptr(stats[1, g]), ptr(y[g * B:(g + 1) * B]), B, C, HW, act, slope, eps,
                     momentum, wsp, wsn, stream_ptr())
        if ACT_TRACE is not None and act in (ACT_RELU, ACT_LRELU):
            _trace(act, y, groups)                       # (one entry per reference call, in call order)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (act, slope, groups, one)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        act, slope, groups, one = ctx.cfg
        dy = _c(dy)
        N, C, HW = _bchw(x)
        B = N // groups
        dx = torch.empty_like(x)
        gg, gb = _grad_buf(gamma), _grad_buf(beta)
        direct = gg is not None and gb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        if direct:
            dg, db = gg, gb
        elif one:
            dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)
            dg, db = dgb[0], dgb[1]
        else:
            dgb = torch.zeros((2, C), dtype=torch.float32, device=x.device)      # (the per-group calls accumulate)
            dg, db = dgb[0], dgb[1]
        if one:
            call("mogan_bn_act_grouped_bwd", ptr(x), ptr(dy), ptr(stats[0]), ptr(stats[1]), ptr(gamma), ptr(beta), ptr(dx),
                 ptr(dg), ptr(db), groups, B, C, HW, act, slope, 1 if direct else 0, stream_ptr())
        else:
            wsp, wsn = workspace(x.device)
            for g in range(groups):
                call("mogan_bn_act_bwd", ptr(x[g * B:(g + 1) * B]), ptr(dy[g * B:(g + 1) * B]), ptr(stats[0, g]), ptr(stats[1, g]),
                     ptr(gamma), ptr(beta), ptr(dx[g * B:(g + 1) * B]), ptr(dg), ptr(db), B, C, HW, act, slope, 1, wsp, wsn,
                     stream_ptr())
        if direct:
            _grad_hit(gg)
            _grad_hit(gb)
            dg = db = None
        return dx, dg, db, None, None, None, None, None, None, None


def bn_groups_ok(x, groups):
    """can `groups` BatchNorm calls on this (groups*B, C, ...) tensor go out as ONE launch (B*HW <= 4096 values per channel)?
    (bn_act(groups=...) itself takes any size: larger maps run the two-launch kernels per group, in place)"""
    N, C, HW = _bchw(x)
    return groups > 1 and N % groups == 0 and bool(lib.load().mogan_bn_act_grouped_eligible(groups, N // groups, C, HW))


def bn_act(x, gamma, beta, running_mean, running_var, act=ACT_NONE, slope=0.2, residual=None, eps=1e-5,
           momentum=0.1, groups=1):
    if groups > 1:
        assert residual is None
        return BNActGroupedFn.apply(x, gamma, beta, running_mean, running_var, act, float(slope), float(eps), float(momentum),
                                    int(groups))
    return BNActFn.apply(x, gamma, beta, residual, running_mean, running_var, act, float(slope), float(eps),
                         float(momentum))


class AffineActFn(torch.autograd.Function):
    """Eval-mode BN (running statistics folded into scale/shift) + activation; only dx is produced
    (frozen Inception trunk of CNN_ENCODER, model.py:218-219,227-242)."""

    @staticmethod
    def forward(ctx, x, scale, shift, act, slope):
        x = _c(x)
        B, C, HW = _bchw(x)
        y = torch.empty_like(x)
        call("mogan_affine_act_fwd", ptr(x), ptr(scale), ptr(shift), ptr(y), B, C, HW, act, slope, stream_ptr())
        ctx.save_for_backward(x, scale, shift)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift = ctx.saved_tensors
        act, slope = ctx.cfg
        B, C, HW = _bchw(x)
        dx = torch.empty_like(x)
        call("mogan_affine_act_bwd", ptr(x), ptr(_c(dy)), ptr(scale), ptr(shift), ptr(dx), B, C, HW, act, slope,
             stream_ptr())
        return dx, None, None, None, None


def affine_act(x, scale, shift, act=ACT_RELU, slope=0.0):
    return AffineActFn.apply(x, scale, shift, act, float(slope))


# ------------------------------------------------------------------------------- activations
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = _c(x)
        B, C, HW = _bchw(x)
        Cy = C // 2 if act == ACT_GLU else C
        y = torch.empty((B, Cy) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        call("mogan_act_fwd", ptr(x), ptr(y), B, C, HW, act, slope, stream_ptr())
        if ACT_TRACE is not None and act in (ACT_RELU, ACT_LRELU):
            ACT_TRACE.append((act, y))
        ctx.save_for_backward(x)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        act, slope = ctx.cfg
        B, C, HW = _bchw(x)
        dx = torch.empty_like(x)
        call("mogan_act_bwd", ptr(x), ptr(_c(dy)), ptr(dx), B, C, HW, act, slope, stream_ptr())
        return dx, None, None


def act(x, kind, slope=0.2):
    return ActFn.apply(x, kind, float(slope))


def glu(x):
    return ActFn.apply(x, ACT_GLU, 0.0)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        call("mogan_add", ptr(a), ptr(b), ptr(y), a.numel(), stream_ptr())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


class GroupSumFn(torch.autograd.Function):
    """(G*B, ...) -> (B, ...): x_0 + x_1 + ... over the G groups of B samples, in that order (model.py:113,406,671)."""

    @staticmethod
    def forward(ctx, x, G):
        x = _c(x)
        y = torch.empty((x.shape[0] // G,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        call("mogan_group_sum", ptr(x), ptr(y), y.numel(), G, stream_ptr())
        ctx.G = G
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty((dy.shape[0] * ctx.G,) + tuple(dy.shape[1:]), dtype=torch.float32, device=dy.device)
        call("mogan_group_bcast", ptr(dy), ptr(dx), dy.numel(), ctx.G, stream_ptr())
        return dx, None


def group_sum(x, G):
    return GroupSumFn.apply(x, int(G))


# ------------------------------------------------------------------------------- softmax
class SoftmaxFn(torch.autograd.Function):
    """softmax(scale * x) over `dim`, optionally restricted to the first lens[...] entries of each
    column (DAMSM: captions of different lengths).  GlobalAttention.py:50,58."""

    @staticmethod
    def forward(ctx, x, dim, scale, lens):
        x = _c(x)
        dim = dim % x.dim()
        outer = 1
        for s in x.shape[:dim]:
            outer *= s
        inner = 1
        for s in x.shape[dim + 1:]:
            inner *= s
        L = x.shape[dim]
        y = torch.empty_like(x)
        call("mogan_softmax_fwd", ptr(x), ptr(y), ptr(lens), outer, L, inner, scale, stream_ptr())
        ctx.save_for_backward(y)
        ctx.lens = lens
        ctx.cfg = (outer, L, inner, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        outer, L, inner, scale = ctx.cfg
        dx = torch.empty_like(y)
        call("mogan_softmax_bwd", ptr(y), ptr(_c(dy)), ptr(dx), ptr(ctx.lens), outer, L, inner, scale, stream_ptr())
        return dx, None, None, None


def softmax(x, dim, scale=1.0, lens=None):
    return SoftmaxFn.apply(x, dim, float(scale), lens)


# ------------------------------------------------------------------------------- spatial transformer
class STNFn(torch.autograd.Function):
    """model.py:17-21 (affine_grid + grid_sample, bilinear, zeros)."""

    @staticmethod
    def forward(ctx, x, theta, Hout, Wout, align_corners):
        x, theta = _c(x), _c(theta)
        B, C, Hin, Win = x.shape
        y = torch.empty((B, C, Hout, Wout), dtype=torch.float32, device=x.device)
        call("mogan_stn_fwd", ptr(x), ptr(theta), ptr(y), B, C, Hin, Win, Hout, Wout, align_corners, stream_ptr())
        ctx.save_for_backward(theta)
        ctx.cfg = (B, C, Hin, Win, Hout, Wout, align_corners)
        return y

    @staticmethod
    def backward(ctx, dy):
        (theta,) = ctx.saved_tensors
        B, C, Hin, Win, Hout, Wout, ac = ctx.cfg
        dx = torch.empty((B, C, Hin, Win), dtype=torch.float32, device=dy.device)
        call("mogan_stn_bwd", ptr(_c(dy)), ptr(theta), ptr(dx), B, C, Hin, Win, Hout, Wout, ac, stream_ptr())
        return dx, None, None, None, None


def stn(x, theta, size, align_corners=False):
    return STNFn.apply(x, theta, int(size[2]), int(size[3]), 1 if align_corners else 0)


class STNSharedFn(torch.autograd.Function):
    """stn() whose source is shared between the objects or constant over the plane, without the materialised copies
    (include/mogan_hip.h: mogan_stn_*_ex): x (xB, C, Hin, Win) read by sample b as image b % xB, or -- plane -- x (xB, C), the
    label vector the reference repeats over Hin x Win first (model.py:109-111, 663-665).  theta (B', G, 2, 3) in the loader's
    order when theta_G = G (samples are object-major), else (N, 2, 3)."""

    @staticmethod
    def forward(ctx, x, theta, N, Hin, Win, Hout, Wout, align_corners, plane, theta_G):
        x, theta = _c(x), _c(theta)
        xB, C = x.shape[0], x.shape[1]
        y = torch.empty((N, C, Hout, Wout), dtype=torch.float32, device=x.device)
        call("mogan_stn_fwd_ex", ptr(x), ptr(theta), ptr(y), N, C, Hin, Win, Hout, Wout, align_corners, xB, plane, theta_G,
             stream_ptr())
        ctx.save_for_backward(theta)
        ctx.cfg = (N, C, Hin, Win, Hout, Wout, align_corners, xB, plane, theta_G, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        (theta,) = ctx.saved_tensors
        N, C, Hin, Win, Hout, Wout, ac, xB, plane, theta_G, xshape = ctx.cfg
        dx = torch.empty(xshape, dtype=torch.float32, device=dy.device)
        call("mogan_stn_bwd_ex", ptr(_c(dy)), ptr(theta), ptr(dx), N, C, Hin, Win, Hout, Wout, ac, xB, plane, theta_G,
             stream_ptr())
        return (dx,) + (None,) * 9


def stn_shared(x, theta, N, in_hw, out_hw, align_corners=False, plane=False, theta_G=0):
    return STNSharedFn.apply(x, theta, int(N), int(in_hw[0]), int(in_hw[1]), int(out_hw[0]), int(out_hw[1]),
                             1 if align_corners else 0, 1 if plane else 0, int(theta_G))


# ------------------------------------------------------------------------------- channel concat with broadcast sources
class CatFn(torch.autograd.Function):
    """torch.cat(parts, 1) where a part may be a code repeated over the plane, one tensor repeated for every object, or the
    per-object slices of a (B, G, C) tensor in the object-major batch -- one launch each way (mogan_concat_fwd / _bwd; model.py:
    400-401, 418, 457, 633-634, 666, 703).  meta[i] = (C, rows, sb, sg, bcast) of part i."""

    @staticmethod
    def forward(ctx, meta, N, spatial, *parts):
        import ctypes
        parts = [_c(t) for t in parts]
        n = len(parts)
        HW = 1
        for d in spatial:
            HW *= d
        Ctot = sum(m[0] for m in meta)
        dst = torch.empty((N, Ctot) + tuple(spatial), dtype=torch.float32, device=parts[0].device)
        arrs = _cat_arrays(meta, [t.data_ptr() for t in parts])
        call("mogan_concat_fwd", *arrs, n, ptr(dst), N, HW, stream_ptr())
        ctx.cfg = (meta, N, HW, [tuple(t.shape) for t in parts])
        return dst

    @staticmethod
    def backward(ctx, ddst):
        meta, N, HW, shapes = ctx.cfg
        ddst = _c(ddst)
        grads = [torch.empty(shp, dtype=torch.float32, device=ddst.device) if ctx.needs_input_grad[3 + i] else None
                 for i, shp in enumerate(shapes)]
        if any(g is not None for g in grads):
            arrs = _cat_arrays(meta, [g.data_ptr() if g is not None else None for g in grads])
            call("mogan_concat_bwd", ptr(ddst), *arrs, len(shapes), N, HW, stream_ptr())
        return (None, None, None) + tuple(grads)


def _cat_arrays(meta, pointers):
    import ctypes
    n = len(meta)
    pp = (ctypes.c_void_p * n)(*pointers)
    cc = (ctypes.c_int * n)(*[m[0] for m in meta])
    rr = (ctypes.c_int * n)(*[m[1] for m in meta])
    sb = (ctypes.c_longlong * n)(*[m[2] for m in meta])
    sg = (ctypes.c_longlong * n)(*[m[3] for m in meta])
    bc = (ctypes.c_int * n)(*[m[4] for m in meta])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    return cast(pp), cast(cc), cast(rr), cast(sb), cast(sg), cast(bc)


def cat_channels(parts, N, spatial=()):
    """parts: list of (tensor, mode); mode = "full" (N, C, *spatial) | "plane" (N, C) | ("rep", G) (N/G, C, *spatial) repeated for
    G objects | ("rep_plane", G) (N/G, C) | ("obj", G) (B, G, C, *spatial), batch n = g B + b | ("obj_plane", G) (B, G, C).
    Returns the (N, sum C, *spatial) concatenation."""
    HW = 1
    for d in spatial:
        HW *= int(d)
    meta, ts = [], []
    for t, mode in parts:
        kind, G = (mode, 1) if isinstance(mode, str) else mode
        if kind == "full":
            C = t.shape[1]; m = (C, N, C * HW, 0, 0)
        elif kind == "plane":
            C = t.shape[1]; m = (C, N, C, 0, 1)
        elif kind == "rep":
            C = t.shape[1]; m = (C, N // G, C * HW, 0, 0)
        elif kind == "rep_plane":
            C = t.shape[1]; m = (C, N // G, C, 0, 1)
        elif kind == "obj":
            C = t.shape[2]; m = (C, N // G, G * C * HW, C * HW, 0)
        elif kind == "obj_plane":
            C = t.shape[2]; m = (C, N // G, G * C, C, 1)
        else:
            raise ValueError(mode)
        want = {"full": N, "plane": N, "rep": N // G, "rep_plane": N // G, "obj": N // G, "obj_plane": N // G}[kind]
        lead = 3 if kind.startswith("obj") else 2                 # (batch[, object], channel) in front of the plane
        tail = () if kind.endswith("plane") else tuple(int(d) for d in spatial)
        # the kernels index every part with the destination's plane size: a part whose trailing dims differ would be read out
        # of bounds where torch.cat raises a shape error
        if (t.shape[0] != want or (kind.startswith("obj") and t.shape[1] != G) or t.dim() != lead + len(tail)
                or tuple(t.shape[lead:]) != tail):
            raise lib.MoganHipError("cat_channels: part of shape %r does not fit mode %r at N = %d, plane %r"
                                    % (tuple(t.shape), mode, N, tuple(spatial)))
        meta.append(m)
        ts.append(t)
    if len(ts) > 4:
        raise lib.MoganHipError("cat_channels: at most 4 parts")
    return CatFn.apply(tuple(meta), int(N), tuple(int(d) for d in spatial), *ts)


def bbox_to_theta(bbox):
    bbox = _c(bbox).view(-1, 4)
    n = bbox.shape[0]
    th = torch.empty((n, 2, 3), dtype=torch.float32, device=bbox.device)
    thi = torch.empty((n, 2, 3), dtype=torch.float32, device=bbox.device)
    call("mogan_bbox_to_theta", ptr(bbox), ptr(th), ptr(thi), n, stream_ptr())
    return th, thi


# ------------------------------------------------------------------------------- word attention
class AttnFn(torch.autograd.Function):
    """GlobalAttention.py:96-121 after conv_context. h (B,idf,Q), src (B,idf,T), mask (B,T) uint8."""

    @staticmethod
    def forward(ctx, h, src, mask, mask_mode):
        h, src = _c(h), _c(src)
        B, idf, Q = h.shape
        T = src.shape[2]
        wc = torch.empty((B, idf, Q), dtype=torch.float32, device=h.device)
        attn = torch.empty((B, T, Q), dtype=torch.float32, device=h.device)
        call("mogan_attn_fwd", ptr(h), ptr(src), ptr(mask), ptr(wc), ptr(attn), B, idf, Q, T, mask_mode,
             stream_ptr())
        ctx.save_for_backward(h, src, attn)
        return wc, attn

    @staticmethod
    def backward(ctx, dwc, dattn):
        h, src, attn = ctx.saved_tensors
        B, idf, Q = h.shape
        T = src.shape[2]
        dwc = _c(dwc)
        dattn = _c(dattn) if dattn is not None else None
        dh = torch.empty_like(h)
        dscore = torch.empty_like(attn)
        call("mogan_attn_bwd", ptr(src), ptr(attn), ptr(dwc), ptr(dattn), ptr(dh), ptr(dscore), B, idf, Q, T,
             stream_ptr())
        dsrc = None
        if ctx.needs_input_grad[1]:
            dsrc = torch.empty_like(src)
            bmm_raw(h, dscore.transpose(1, 2), dsrc)                    # (B,idf,Q) x (B,Q,T)
            bmm_raw(dwc, attn.transpose(1, 2), dsrc, accumulate=True)
        return dh, dsrc, None, None


def attention(h, src, mask=None, mask_mode=0):
    if mask is not None:
        mask = mask.to(torch.uint8).contiguous()
    return AttnFn.apply(h, src, mask, mask_mode)


# ------------------------------------------------------------------------------- losses
class BCEFn(torch.autograd.Function):
    """nn.BCELoss()(p, const target) (miscc/losses.py:158-168,198-201)."""

    @staticmethod
    def forward(ctx, p, target):
        p = _c(p).view(-1)
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        call("mogan_bce_fwd", ptr(p), target, 1.0, ptr(loss), p.numel(), 0, stream_ptr())
        ctx.save_for_backward(p)
        ctx.target = target
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        dp = torch.empty_like(p)
        call("mogan_bce_bwd", ptr(p), ctx.target, 1.0, ptr(_c(g).view(1)), ptr(dp), p.numel(), stream_ptr())
        return dp, None


def bce(p, target):
    return BCEFn.apply(p, float(target))


class BCELogitsFn(torch.autograd.Function):
    """nn.BCEWithLogitsLoss()(x, const target) (code/coco/stackgan/miscc/utils.py:73,114)."""

    @staticmethod
    def forward(ctx, x, target):
        x = _c(x).view(-1)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        call("mogan_bce_logits_fwd", ptr(x), target, 1.0, ptr(loss), x.numel(), 0, stream_ptr())
        ctx.save_for_backward(x)
        ctx.target = target
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("mogan_bce_logits_bwd", ptr(x), ctx.target, 1.0, ptr(_c(g).view(1)), ptr(dx), x.numel(), stream_ptr())
        return dx, None


def bce_with_logits(x, target):
    return BCELogitsFn.apply(x, float(target))


class KLFn(torch.autograd.Function):
    """KL_loss (miscc/losses.py:230-234)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _c(mu), _c(logvar)
        loss = torch.empty(1, dtype=torch.float32, device=mu.device)
        call("mogan_kl_fwd", ptr(mu), ptr(logvar), ptr(loss), mu.numel(), stream_ptr())
        ctx.save_for_backward(mu, logvar)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        call("mogan_kl_bwd", ptr(mu), ptr(logvar), ptr(_c(g).view(1)), ptr(dmu), ptr(dlv), mu.numel(), stream_ptr())
        return dmu, dlv


def kl_loss(mu, logvar):
    return KLFn.apply(mu, logvar)


class ReparamFn(torch.autograd.Function):
    """CA_NET.reparametrize (model.py:333-340)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _c(mu), _c(logvar), _c(eps)
        c = torch.empty_like(mu)
        call("mogan_reparam_fwd", ptr(mu), ptr(logvar), ptr(eps), ptr(c), mu.numel(), stream_ptr())
        ctx.save_for_backward(logvar, eps)
        return c

    @staticmethod
    def backward(ctx, dc):
        logvar, eps = ctx.saved_tensors
        dmu, dlv = torch.empty_like(logvar), torch.empty_like(logvar)
        call("mogan_reparam_bwd", ptr(logvar), ptr(eps), ptr(_c(dc)), ptr(dmu), ptr(dlv), logvar.numel(), stream_ptr())
        return dmu, dlv, None


def reparam(mu, logvar, eps):
    return ReparamFn.apply(mu, logvar, eps)


# ------------------------------------------------------------------------------- DAMSM matching losses
def _scalar(device):
    return torch.empty((), dtype=torch.float32, device=device)


def _g_ptr(g):
    """Pointer of a 0-dim upstream gradient (None -> NULL: that output did not take part in the loss)."""
    return ptr(_c(g)) if g is not None else None


class _PairCE:
    """Shared tail of the word and the sentence loss: the two cross-entropies over a (B,B) similarity matrix."""

    @staticmethod
    def forward(ctx, sim, labels, mask):
        R = sim.shape[0]
        dev = sim.device
        prow, pcol = torch.empty_like(sim), torch.empty_like(sim)
        nll = torch.empty(2 * R, dtype=torch.float32, device=dev)
        out2 = torch.empty(2, dtype=torch.float32, device=dev)
        call("mogan_damsm_ce_fwd", ptr(sim), ptr(labels), ptr(mask), R, sim.shape[1], ptr(prow), ptr(pcol), ptr(nll),
             ptr(out2), stream_ptr())
        ctx.ce = (prow, pcol, labels)
        return out2[0], out2[1]

    @staticmethod
    def backward(ctx, g0, g1):
        prow, pcol, labels = ctx.ce
        dsim = torch.empty_like(prow)
        call("mogan_damsm_ce_bwd", ptr(prow), ptr(pcol), ptr(labels), _g_ptr(g0), _g_ptr(g1), prow.shape[0], prow.shape[1],
             ptr(dsim), stream_ptr())
        return dsim


class DamsmWordsFn(torch.autograd.Function):
    """words_loss for all (image, caption) pairs (miscc/losses.py:62-132, GlobalAttention.py:31-69) -> (loss0, loss1,
    attention a2 (B,Bc,T,S)).  Gradient: region features only."""

    @staticmethod
    def forward(ctx, feat, words, lens, labels, mask, g1, g2, g3):
        feat, words = _c(feat), _c(words)
        B, C = feat.shape[0], feat.shape[1]
        S = feat[0, 0].numel()
        Bc, T = words.shape[0], words.shape[2]
        dev = feat.device
        f32 = dict(dtype=torch.float32, device=dev)
        sim = torch.empty((B, Bc), **f32)
        a1, a2 = torch.empty((B, Bc, S, T), **f32), torch.empty((B, Bc, T, S), **f32)
        wc, wt = torch.empty((B, C, Bc, T), **f32), torch.empty((C, Bc, T), **f32)
        call("mogan_damsm_words_fwd", ptr(feat), ptr(words), ptr(lens), B, Bc, C, S, T, g1, g2, g3, ptr(sim), ptr(a1),
             ptr(a2), ptr(wc), ptr(wt), stream_ptr())
        l0, l1 = _PairCE.forward(ctx, sim, labels, mask)
        ctx.save_for_backward(feat, words, a1, a2, wc, wt)
        ctx.cfg = (lens, g1, g2, g3)
        ctx.mark_non_differentiable(a2)
        return l0, l1, a2

    @staticmethod
    def backward(ctx, gl0, gl1, _ga2):
        feat, words, a1, a2, wc, wt = ctx.saved_tensors
        lens, g1, g2, g3 = ctx.cfg
        B, C = feat.shape[0], feat.shape[1]
        S = feat[0, 0].numel()
        Bc, T = words.shape[0], words.shape[2]
        dsim = _PairCE.backward(ctx, gl0, gl1)
        dwc = torch.empty_like(wc)
        dst = torch.empty_like(a2)
        call("mogan_damsm_words_bwd", ptr(feat), ptr(words), ptr(lens), ptr(a1), ptr(a2), ptr(wc), ptr(dsim), B, Bc, C, S, T,
             g1, g2, g3, ptr(dwc), ptr(dst), stream_ptr())
        dfeat = torch.empty_like(feat)
        out = dfeat.view(B, C, S)
        bmm_raw(dwc.view(B, C, Bc * T), a2.view(B, Bc * T, S), out)
        bmm_raw(wt.view(1, C, Bc * T).expand(B, C, Bc * T), dst.view(B, Bc * T, S), out, accumulate=True)
        return dfeat, None, None, None, None, None, None, None


def damsm_words(feat, words, lens, labels, mask, gamma1, gamma2, gamma3):
    return DamsmWordsFn.apply(feat, words, lens, labels, mask, float(gamma1), float(gamma2), float(gamma3))


class DamsmSentFn(torch.autograd.Function):
    """sent_loss (miscc/losses.py:20-59) -> (loss0, loss1).  Gradient: cnn code only."""

    @staticmethod
    def forward(ctx, cnn, rnn, labels, mask, g3, eps):
        cnn, rnn = _c(cnn), _c(rnn)
        B, C, Bc = cnn.shape[0], cnn.shape[1], rnn.shape[0]
        sim = torch.empty((B, Bc), dtype=torch.float32, device=cnn.device)
        call("mogan_damsm_sent_fwd", ptr(cnn), ptr(rnn), B, Bc, C, g3, eps, ptr(sim), stream_ptr())
        l0, l1 = _PairCE.forward(ctx, sim, labels, mask)
        ctx.save_for_backward(cnn, rnn)
        ctx.cfg = (g3, eps)
        return l0, l1

    @staticmethod
    def backward(ctx, gl0, gl1):
        cnn, rnn = ctx.saved_tensors
        g3, eps = ctx.cfg
        dsim = _PairCE.backward(ctx, gl0, gl1)
        dcnn = torch.empty_like(cnn)
        call("mogan_damsm_sent_bwd", ptr(cnn), ptr(rnn), ptr(dsim), cnn.shape[0], rnn.shape[0], cnn.shape[1], g3, eps,
             ptr(dcnn), stream_ptr())
        return dcnn, None, None, None, None, None


def damsm_sent(cnn, rnn, labels, mask, gamma3, eps=1e-8):
    return DamsmSentFn.apply(cnn, rnn, labels, mask, float(gamma3), float(eps))


class ScalarSumFn(torch.autograd.Function):
    """sum_k w_k * x_k over up to 8 zero-dim device scalars in ONE launch (forward) / one launch (backward): the loss sums
    of miscc/losses.py:169-174,203,221 and trainer.py:330 otherwise cost a chain of 0-dim aten kernels each way."""

    @staticmethod
    def forward(ctx, weights, *xs):
        import ctypes
        n = len(xs)
        xs = [_c(x) for x in xs]
        out = _scalar(xs[0].device)
        pin = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
        pw = (ctypes.c_float * n)(*weights)
        call("mogan_scalar_sum", ctypes.cast(pin, ctypes.c_void_p), ctypes.cast(pw, ctypes.c_void_p), n, ptr(out),
             stream_ptr())
        ctx.weights = weights
        return out

    @staticmethod
    def backward(ctx, g):
        import ctypes
        idx = [k for k in range(len(ctx.weights)) if ctx.needs_input_grad[1 + k]]
        if not idx:
            return (None,) * (1 + len(ctx.weights))
        g = _c(g)
        outs = [_scalar(g.device) for _ in idx]
        n = len(idx)
        pout = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
        pw = (ctypes.c_float * n)(*[ctx.weights[k] for k in idx])
        call("mogan_scalar_scale", ptr(g), ctypes.cast(pw, ctypes.c_void_p), n, ctypes.cast(pout, ctypes.c_void_p),
             stream_ptr())
        grads = [None] * len(ctx.weights)
        for k, o in zip(idx, outs):
            grads[k] = o
        return (None,) + tuple(grads)


def scalar_sum(xs, weights=None):
    """sum_k weights[k] * xs[k] of zero-dim fp32 device tensors (len <= 8; longer lists are summed in chunks)."""
    xs = list(xs)
    weights = [1.0] * len(xs) if weights is None else [float(w) for w in weights]
    while len(xs) > 8:
        head = ScalarSumFn.apply(tuple(weights[:8]), *xs[:8])
        xs, weights = [head] + xs[8:], [1.0] + weights[8:]
    return ScalarSumFn.apply(tuple(weights), *xs)


# ------------------------------------------------------------------------------- pooling / resize
class PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind, k, s, pad, OH, OW):
        x = _c(x)
        B, C, H, W = x.shape
        if kind == "max":
            oh, ow = (H - k) // s + 1, (W - k) // s + 1
            y = torch.empty((B, C, oh, ow), dtype=torch.float32, device=x.device)
            idx = torch.empty((B, C, oh, ow), dtype=torch.uint8, device=x.device)
            call("mogan_maxpool_fwd", ptr(x), ptr(y), ptr(idx), B * C, H, W, k, s, stream_ptr())
            ctx.save_for_backward(idx)
        elif kind == "avg":
            oh, ow = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
            y = torch.empty((B, C, oh, ow), dtype=torch.float32, device=x.device)
            call("mogan_avgpool_fwd", ptr(x), ptr(y), B * C, H, W, k, s, pad, stream_ptr())
        else:
            y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
            call("mogan_bilinear_fwd", ptr(x), ptr(y), B * C, H, W, OH, OW, stream_ptr())
        ctx.cfg = (kind, k, s, pad, OH, OW, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        kind, k, s, pad, OH, OW, shp = ctx.cfg
        B, C, H, W = shp
        dy = _c(dy)
        dx = torch.empty(shp, dtype=torch.float32, device=dy.device)
        if kind == "max":
            (idx,) = ctx.saved_tensors
            call("mogan_maxpool_bwd", ptr(idx), ptr(dy), ptr(dx), B * C, H, W, k, s, stream_ptr())
        elif kind == "avg":
            call("mogan_avgpool_bwd", ptr(dy), ptr(dx), B * C, H, W, k, s, pad, stream_ptr())
        else:
            call("mogan_bilinear_bwd", ptr(dy), ptr(dx), B * C, H, W, OH, OW, stream_ptr())
        return dx, None, None, None, None, None, None


def max_pool2d(x, k, s):
    return PoolFn.apply(x, "max", k, s, 0, 0, 0)


def avg_pool2d(x, k, s=None, pad=0):
    return PoolFn.apply(x, "avg", k, s or k, pad, 0, 0)


def bilinear_resize(x, OH, OW):
    return PoolFn.apply(x, "bilinear", 0, 0, 0, OH, OW)


# ------------------------------------------------------------------------------- optimizer
def adam_step(p, g, m, v, ema, lr, beta1, beta2, eps, step=0, dev_state=None, eps_mode=0, grad_scale=1.0,
              ema_decay=0.999):
    """In-place fused Adam (+EMA) over flat fp32 buckets (trainer.py:137-148,341-342)."""
    call("mogan_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), p.numel(), lr, beta1, beta2, eps, step,
         ptr(dev_state), eps_mode, grad_scale, ema_decay, stream_ptr())


# ------------------------------------------------------------------------------- text encoder (csrc/mogan_lstm.hip)
def lstm_encoder_forward(captions, lens, emb_weight, rnn, h0=None, c0=None):
    """Embedding + one-layer bidirectional LSTM over packed captions, eval mode, no gradient, as ONE launch
    (mogan_lstm_encoder_fwd); `rnn` is the nn.LSTM whose parameters are used.  Returns (words (B, 2H, Tmax), sent (B, 2H)) or
    None where the kernel does not cover the module (then the caller keeps the stock path)."""
    import ctypes
    H = rnn.hidden_size
    if (not isinstance(rnn, torch.nn.LSTM) or rnn.num_layers != 1 or not rnn.bidirectional or not rnn.batch_first or H != 128
            or not rnn.bias or getattr(rnn, "proj_size", 0)):
        return None
    B, T = captions.shape
    V, E = emb_weight.shape
    lens = [int(v) for v in lens]
    Tmax = max(lens) if lens else 0
    if B > 64 or Tmax < 1 or Tmax > 32 or Tmax > T or E > 320 or E % 4 or min(lens) < 0 or captions.dtype != torch.int64:
        return None
    ws = [rnn.weight_ih_l0, rnn.weight_ih_l0_reverse, rnn.weight_hh_l0, rnn.weight_hh_l0_reverse,
          rnn.bias_ih_l0, rnn.bias_ih_l0_reverse, rnn.bias_hh_l0, rnn.bias_hh_l0_reverse]
    if any(w.dtype != torch.float32 or not w.is_contiguous() or w.data_ptr() % 16 for w in ws) or not emb_weight.is_contiguous():
        return None
    dev = captions.device
    cap = captions if captions.is_contiguous() else captions.contiguous()
    words = torch.empty((B, 2 * H, Tmax), dtype=torch.float32, device=dev)
    sent = torch.empty((B, 2 * H), dtype=torch.float32, device=dev)
    P2 = ctypes.c_void_p * 2
    arr = [P2(ws[2 * i].data_ptr(), ws[2 * i + 1].data_ptr()) for i in range(4)]
    lens_c = (ctypes.c_int * B)(*lens)
    h0p = ptr(_c(h0)) if h0 is not None else None
    c0p = ptr(_c(c0)) if c0 is not None else None
    call("mogan_lstm_encoder_fwd", cap.data_ptr(), ctypes.cast(lens_c, ctypes.c_void_p), emb_weight.data_ptr(),
         ctypes.cast(arr[0], ctypes.c_void_p), ctypes.cast(arr[1], ctypes.c_void_p), ctypes.cast(arr[2], ctypes.c_void_p),
         ctypes.cast(arr[3], ctypes.c_void_p), h0p, c0p, words.data_ptr(), sent.data_ptr(), B, T, Tmax, V, E, H, stream_ptr())
    PK_STATS["lstm_fused"] = PK_STATS.get("lstm_fused", 0) + 1
    return words, sent
