"""torch.library custom ops over libganconv.so: conv forward / data-gradient / weight-gradient.

Three ops, closed under differentiation (every backward formula is written with the same three ops), so the
WGAN-GP gradient penalty -- torch.autograd.grad(..., create_graph=True) followed by .backward(), reference
BasicGANs/loss_functions.py:21-28 -- runs through the same kernels:

    y  = conv_fprop(x, w)        dy/dx -> conv_dgrad(gy, w)     dy/dw -> conv_wgrad(x, gy)
    gx = conv_dgrad(gy, w)       d/dgy -> conv_fprop(u, w)      d/dw  -> conv_wgrad(u, gy)
    gw = conv_wgrad(x, gy)       d/dx  -> conv_dgrad(gy, v)     d/dgy -> conv_fprop(x, v)

All tensors are logically NCHW (what the reference's modules exchange) and physically channels-last; weights
are logically [Cout, Cin, kh, kw] with any strides.  These ops replace aten::convolution /
aten::convolution_backward / _convolution_double_backward at the reference call sites listed in
include/ganconv.h.  There is no fallback: on a CPU tensor, a missing library or an unsupported geometry they
raise RuntimeError.
"""
import ctypes
import os
import threading
import weakref
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from ._lib import ConvDesc, Epilogue

_PRECISION = {"bf16": _lib.GC_PREC_BF16, "fp32x3": _lib.GC_PREC_F32X3}
_default_precision = _lib.GC_PREC_BF16


def set_precision(name):
    """'bf16': operands rounded to bf16, fp32 accumulate (fast).  'fp32x3': three-term bf16 split of fp32
    operands (about 1e-5 relative error, three times the matrix work)."""
    global _default_precision
    _default_precision = _PRECISION[name]


def get_precision():
    return _default_precision


# ---- packed-weight cache -------------------------------------------------------------------------------------
# A module that owns a Parameter registers a key; the cache entry remembers which storage/version it was packed
# from and is refreshed when the parameter has been modified in place (optimizer step, weights_init, load).
# The caches below are shared by the thread that runs the forward pass and the autograd engine's worker threads
# (backward formulas run there): every compound look-up / insert holds _lock.
_lock = threading.RLock()
_wcache = {}          # module key -> {(pass, precision, image bytes): _Image}
_divisors = {}        # module key -> (weight data_ptr, tensor holding the divisors, index into it, epoch): repack_divided
_div_epoch = [0]
_next_key = [1]


def new_weight_key():
    with _lock:
        k = _next_key[0]
        _next_key[0] += 1
        return k


def drop_weight_key(key):
    with _lock:
        _wcache.pop(key, None)
        _divisors.pop(key, None)


def _dtype_code(t):
    if t.dtype == torch.float32:
        return _lib.GC_F32
    if t.dtype == torch.bfloat16:
        return _lib.GC_BF16
    raise RuntimeError(f"ganconv: unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- backward overlap: weight gradient on a second stream --------------------------------------------------------
# The data gradient and the weight gradient of a convolution are independent given the output gradient.  With the
# overlap switched on, the weight-gradient kernels (incl. the split reduction) are launched on a side stream that
# forks from the current stream after the operands are ready and is joined before the autograd formula returns;
# mid-size layers leave half the CUs idle during their load and store phases, which the other kernel then fills.
# All memory is allocated in the current stream's context and kept alive until the join, so the caching allocator
# never sees the side stream.  Works inside HIP-graph capture (the fork/join becomes two graph branches).
_overlap = False
_side_streams = {}
_capture_origin = [None]     # the stream a HIP-graph capture was started on (see capturing_on)


class capturing_on:
    """Context a graph recorder enters around its capture: names the stream the capture starts on.  A weight-gradient
    side stream is forked only from that stream while a capture is running: a fork from a stream that is itself forked
    inside the capture made hipStreamEndCapture abort (ROCm 7.2; tools/probe/nested_fork_capture.hip)."""

    def __init__(self, stream=None):
        self.stream = stream

    def __enter__(self):
        st = self.stream if self.stream is not None else torch.cuda.current_stream()
        self.was, _capture_origin[0] = _capture_origin[0], st.cuda_stream
        return self

    def __exit__(self, *exc):
        _capture_origin[0] = self.was


def set_split_reduction(on):
    """Allows (default) or forbids split reductions: small convolutions whose partial sums are added with fp32 atomics,
    which makes their last bit vary from run to run.  Forbid them to get bit-wise reproducible convolutions (bias
    gradients stay atomically summed).  Returns the previous setting; recorded HIP graphs keep the setting they were
    recorded with."""
    return bool(_lib.lib().gc_set_split_reduction(1 if on else 0))


def set_backward_overlap(on):
    global _overlap
    _overlap = bool(on)


def get_backward_overlap():
    return _overlap


class suspended_backward_overlap:
    """Context: no weight-gradient side stream inside.  For callers that already run independent chains on streams of
    their own: a side stream forked from a forked stream inside a HIP-graph capture crashed hipStreamEndCapture
    (ROCm 7.2), and the chains fill the GPU anyway."""

    def __enter__(self):
        global _overlap
        self.was, _overlap = _overlap, False

    def __exit__(self, *exc):
        global _overlap
        _overlap = self.was


class _SideLaunch:
    """Context of one weight-gradient launch on the side stream (entered by the autograd formulas)."""
    active = None

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        with _lock:
            side = _side_streams.get(device)
            if side is None:
                side = _side_streams[device] = torch.cuda.Stream(device)
        self.side = side
        self.forked = False
        self.keep = []        # workspaces and operand copies in use on the side stream, released at the join
        # under a capture, fork only from the stream the capture started on
        self.allowed = not (torch.cuda.is_current_stream_capturing() and _capture_origin[0] != self.main.cuda_stream)

    def __enter__(self):
        _SideLaunch.active = self
        return self

    def __exit__(self, *exc):
        _SideLaunch.active = None

    def fork(self, *keep):
        """Called by the op right before its launches: everything enqueued so far (operand copies) is waited for."""
        if not self.allowed:
            return ctypes.c_void_p(self.main.cuda_stream)
        self.side.wait_stream(self.main)
        self.forked = True
        self.keep.extend(keep)
        return ctypes.c_void_p(self.side.cuda_stream)

    def join(self):
        if self.forked:
            self.main.wait_stream(self.side)
        self.keep.clear()


class weight_gradient_sink:
    """Context: the convolution backward formulas hand the weight gradients of the given parameters to this object
    instead of returning them to autograd (which receives None for them: call torch.autograd.grad with allow_unused=True,
    or read the buffers instead of .grad).  Each weight-gradient kernel writes into (first contribution) or adds to (later
    ones, e.g. the two a double backward produces per weight) the parameter's buffer, so autograd's own accumulation
    kernels disappear; with `fork=True` all of them run on one stream beside the calling one, in launch order, and the
    calling stream waits for them only when the context closes.  The data-gradient chain of a backward pass, the part
    with the dependencies, then never queues behind a weight-gradient kernel.

    Buffers are written, never read, before their first contribution: they need no zeroing.  `touched` tells afterwards
    which buffers received one.  Not for passes that record a graph for further differentiation (create_graph=True):
    those take the ordinary path.  Process-wide like the other switches: one sink at a time."""
    active = None

    def __init__(self, params, buffers, fork=False, defer=False, flush_streams=(), expected=0, early_streams=()):
        self.flush_streams = list(flush_streams)     # deferred mode: the weights' launches are dealt out over these streams and the calling one
        # deferred mode, optional: a weight whose `expected`-th contribution has arrived is launched right away on one of
        # `early_streams` (which waits for the streams that produced the contributions), beside the rest of the passes
        self.expected, self.early_streams = int(expected), list(early_streams)
        self.events, self.early_used, self.n_early = {}, [], 0
        self.buffers = {p.data_ptr(): (p, b) for p, b in zip(params, buffers)}
        self.touched = set()
        self.fork = fork
        self.defer = defer       # collect the contributions and launch them at flush(): one reduction per weight
        self.pending = {}        # id of the parameter's storage -> [(x, gy, geometry)]
        self.keep = []
        self.forked = False

    def __enter__(self):
        if weight_gradient_sink.active is not None:
            raise RuntimeError("ganconv: a weight_gradient_sink is already active")
        dev = next(iter(self.buffers.values()))[1].device if self.buffers else None
        self.main = torch.cuda.current_stream(dev)
        self.side = None
        if self.fork and dev is not None:
            # under a capture, fork only from the stream the capture started on
            if not (torch.cuda.is_current_stream_capturing() and _capture_origin[0] != self.main.cuda_stream):
                with _lock:
                    self.side = _sink_streams.get(dev)
                    if self.side is None:
                        self.side = _sink_streams[dev] = torch.cuda.Stream(dev)
        if self.early_streams and dev is not None:
            self.fork0 = torch.cuda.Event()
            self.fork0.record(self.main)
        weight_gradient_sink.active = self
        return self

    def __exit__(self, *exc):
        weight_gradient_sink.active = None
        if exc[0] is None:
            self.flush()
        if self.forked:
            self.main.wait_stream(self.side)
        for st in self.early_used:
            self.main.wait_stream(st)
        self.keep.clear()
        self.pending.clear()
        self.events.clear()
        return False

    def contribution(self, key, item):
        """Deferred mode: one (x, gy, geometry) pair for the weight `key`, produced on the current stream."""
        items = self.pending.setdefault(key, [])
        items.append(item)
        if not (self.early_streams and self.expected):
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events.setdefault(key, []).append(ev)
        if len(items) == self.expected:
            lane = self.early_streams[self.n_early % len(self.early_streams)]
            self.n_early += 1
            if lane not in self.early_used:
                lane.wait_event(self.fork0)       # the lane joins a running capture from the stream the capture started on
                self.early_used.append(lane)
            for e in self.events.pop(key):
                lane.wait_event(e)
            self.keep.extend(items)               # the operands stay allocated until the calling stream has joined the lane
            del self.pending[key]
            with torch.cuda.stream(lane):
                self._flush_weight(key, list(items))

    def flush(self):
        """Deferred mode: launches what was collected, on the current stream -- per weight one kernel that reduces over
        the pixels of all its (x, gy) pairs (up to four per launch) and one split reduction.  The caller makes the
        current stream wait for every stream a contribution was produced on before calling this.  With `flush_streams`
        the weights take turns over those streams and the current one (the layers' reductions are independent: a thin
        layer's memory-bound kernel then runs beside a wide layer's matrix-bound one); the current stream waits for all
        of them before this returns."""
        if not self.pending:
            return
        main = torch.cuda.current_stream()
        lanes = [main] + [s for s in self.flush_streams if s != main]
        if len(lanes) > 1:
            fork = torch.cuda.Event()
            fork.record(main)
            waiting = set()
            for i, key in enumerate(self.pending):
                st = lanes[i % len(lanes)]
                if st is not main and st not in waiting:
                    st.wait_event(fork)
                    waiting.add(st)
                with torch.cuda.stream(st):
                    self._flush_weight(key, self.pending[key])
            for st in lanes[1:]:
                if st in waiting:
                    main.wait_stream(st)
        else:
            for key, items in self.pending.items():
                self._flush_weight(key, items)
        self.pending.clear()

    def _flush_weight(self, key, items):
        w, buf = self.buffers[key]
        while items:
            group, items = items[:4], items[4:]
            geo = group[0][2]
            if len(group) > 1 and all(it[2] == geo for it in group):
                _wgrad_launch_multi([(it[0], it[1]) for it in group], buf, *geo, accumulate=key in self.touched)
                self.touched.add(key)
            else:
                for x, gy, g in group:
                    kh, kw, stride, padding, precision, alpha = g
                    _wgrad_launch(x, gy, buf, kh, kw, stride, padding, precision, alpha, key in self.touched, lambda *keep: _stream())
                    self.touched.add(key)

    def buffer_for(self, w):
        """(buffer, accumulate flag) when w is one of the sink's parameters and nothing differentiates this pass."""
        if torch.is_grad_enabled():
            return None
        ent = self.buffers.get(w.data_ptr())
        if ent is None or ent[0].shape != w.shape:
            return None
        return ent[1], (w.data_ptr() in self.touched)

    def stream_for(self, w, *keep):
        """The stream of the next launch for w; everything enqueued on the calling stream so far is waited for."""
        self.touched.add(w.data_ptr())
        if self.side is None:
            return _stream()
        # the formulas run on the autograd engine's thread, on the stream their forward pass ran on
        self.side.wait_stream(torch.cuda.current_stream())
        self.forked = True
        self.keep.extend(keep)          # operands and workspaces stay allocated until the join
        return ctypes.c_void_p(self.side.cuda_stream)


_sink_streams = {}


def _check_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ganconv: tensors must live on the GPU (this library has no CPU path)")


def _nhwc(t):
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, dtype_code, precision, y_dtype_code=None):
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Ho, Wo, Cout
    d.kh, d.kw, d.stride_h, d.stride_w, d.pad_h, d.pad_w = kh, kw, stride, stride, padding, padding
    d.x_dtype = dtype_code
    d.y_dtype = dtype_code if y_dtype_code is None else y_dtype_code
    d.precision = precision if (d.x_dtype == _lib.GC_F32 and d.y_dtype == _lib.GC_F32) else _lib.GC_PREC_BF16
    return d


# ---- bf16 operand copies ------------------------------------------------------------------------------------------
# In the bf16 arithmetic mode the kernels round their fp32 operands to bf16 anyway.  Wide layers run faster when that
# rounding is done once per tensor (16-byte LDS-DMA loads of bf16 straight into LDS, half the bytes) instead of in
# every kernel that reads it: an activation is read by the forward conv and by the weight-gradient conv, a gradient
# by the data-gradient and the weight-gradient conv.  The cache entry keeps the source tensor alive, so its address
# cannot be re-used by another tensor while the entry exists; a version bump (in-place write) misses.
_cast_cache = {}
_CAST_CACHE_MAX = 128


def clear_cast_cache():
    """Drops the bf16 operand copies and the statistics rows a convolution left for a BatchNorm that never asked for them
    (an evaluation-mode norm behind a training-mode convolution, a dtype change in between): both keep activations alive."""
    with _lock:
        _cast_cache.clear()
        _row_cache.clear()


def clear_weight_cache():
    """Forget every packed weight image.  Needed before recording a HIP graph: an image packed before the capture
    would be a cache hit inside it, the pack kernel would not be part of the graph, and every replay would read
    the weights as they were when the graph was recorded."""
    with _lock:
        _wcache.clear()


def _cast_insert(key, t, tb):
    with _lock:
        if len(_cast_cache) >= _CAST_CACHE_MAX:
            _cast_cache.pop(next(iter(_cast_cache)), None)
        _cast_cache[key] = (t, tb)


def _to_bf16(t):
    key = (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()))
    with _lock:
        ent = _cast_cache.get(key)
    if ent is not None:
        return ent[1]
    tb = t.to(torch.bfloat16, memory_format=torch.channels_last)
    _cast_insert(key, t, tb)
    return tb


def _bf16_twin(t):
    """For a kernel that can write a bf16 copy of its fp32 output `t` on the side: the buffer for that copy,
    registered as t's cast, or None when no convolution would use it (fp32x3 mode, thin or unaligned channels)."""
    C = t.shape[1]
    if _default_precision != _lib.GC_PREC_BF16 or t.dtype != torch.float32 or C % 8 != 0 or C <= 32 or t.numel() * 2 >= (1 << 31) - (1 << 21):
        return None
    tb = torch.empty_like(t, dtype=torch.bfloat16, memory_format=torch.channels_last)
    _cast_insert((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride())), t, tb)
    return tb


# ---- batch statistics from the convolution in front of a BatchNorm -------------------------------------------------
# A convolution called inside channel_statistics_request() asks its kernel for rows of per-channel (sum, sum of squares)
# of the values it stores (gc_epilogue.stat_rows).  When the kernel chosen for the problem provides them, the rows are
# remembered under the output tensor, and the training-mode BatchNorm that reads this tensor next takes its batch
# statistics from them (gc_batchnorm_act_fwd_rows) instead of from a sweep of its own over the tensor.
_want_rows = [False]
_row_cache = {}       # (data_ptr, shape) -> (output tensor, rows, number of rows): consumed once
_ROW_CACHE_MAX = 8


class channel_statistics_request:
    def __enter__(self):
        self.was, _want_rows[0] = _want_rows[0], True

    def __exit__(self, *exc):
        _want_rows[0] = self.was


def _rows_for(ep, out, C):
    """Adds a request for statistics rows to the epilogue descriptor (created when there was none) if one is pending and
    the output qualifies; returns (descriptor, (rows, counter) or None)."""
    P = out.numel() // max(C, 1)
    if not _want_rows[0] or C % 4 or P <= 1024 or (ep is not None and ep.grad_mask):
        return ep, None
    if ep is None:
        ep = Epilogue()
    cap = 4 * ((P + 127) // 128 + 16)
    rows = torch.empty(cap * C * 2, dtype=torch.float32, device=out.device)
    written = ctypes.c_int32(0)
    ep.stat_rows, ep.stat_rows_capacity, ep.stat_rows_written = rows.data_ptr(), cap, ctypes.pointer(written)
    return ep, (rows, written)


def _offer_scratch(ep, out, reduction):
    """Scratch memory for a split reduction (gc_epilogue.workspace) when the output is small and the reduction long;
    returns (descriptor, tensor to keep alive)."""
    n = out.numel()
    if n == 0 or n > (1 << 20) or reduction < 2048:
        return ep, None
    if ep is None:
        ep = Epilogue()
    ws = torch.empty(min(n * 4 * 32, 32 << 20), dtype=torch.uint8, device=out.device)
    ep.workspace, ep.workspace_bytes = ws.data_ptr(), ws.numel()
    return ep, ws


def _remember_rows(out, got):
    if got is not None and got[1].value > 0:
        with _lock:
            if len(_row_cache) >= _ROW_CACHE_MAX:
                _row_cache.pop(next(iter(_row_cache)), None)
            _row_cache[(out.data_ptr(), tuple(out.shape))] = (out, got[0], got[1].value)


def _optr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_OUT_SAME, _OUT_F32, _OUT_BF16 = -1, 0, 1


def _out_dtype(t, code):
    return t.dtype if code == _OUT_SAME else (torch.bfloat16 if code == _OUT_BF16 else torch.float32)


_SPLIT_MAX_OUT = int(os.environ.get("GC_SPLIT_MAX_OUT", 65536))


def _small_output_in_f32(numel, dtype, act):
    """Small outputs with long reductions run as split reductions, which add fp32 partial tiles into the output: a bf16
    result of that size is computed in fp32 and rounded afterwards (a few KB)."""
    return dtype == torch.bfloat16 and act == 0 and numel <= _SPLIT_MAX_OUT


def _code_of(dtype):
    return _OUT_BF16 if dtype == torch.bfloat16 else _OUT_F32


def _dma_ok(t, ic, oc, precision):
    """fp32 activation in bf16 mode whose conv qualifies for the LDS-DMA kernel (bf16 in, fp32 out)."""
    return (precision == _lib.GC_PREC_BF16 and t.dtype == torch.float32 and ic % 8 == 0 and ic > 32 and oc > 16
            and t.numel() * 2 < (1 << 31) - (1 << 21))   # the kernel addresses its input with 32-bit buffer offsets


class _Image:
    """A cached packed weight image and what it was made from / on."""
    __slots__ = ("ptr", "version", "shape", "buf", "desc", "wref", "stream", "event", "alpha", "recorded", "done", "div_epoch")

    def mark_packed(self, w, event=None):
        """Called right after a pack launch on the current stream (event: one recorded behind a launch that packed several)."""
        self.ptr, self.version, self.shape = w.data_ptr(), w._version, tuple(w.shape)
        st = torch.cuda.current_stream(w.device)
        self.stream = st.cuda_stream
        if event is None:
            event = torch.cuda.Event()
            event.record(st)
        self.event = event
        self.recorded = torch.cuda.is_current_stream_capturing()      # the event is a node of a graph being recorded
        self.done = False

    def ready(self, w):
        """Makes the current stream wait for the pack if that ran on another stream (two chains of one training phase on
        two streams share the images).  A pack that ran eagerly and is known to have finished (weight_images_settled) needs
        no wait (inside a recording the wait would become an event node of its own per use)."""
        st = torch.cuda.current_stream(w.device)
        if st.cuda_stream != self.stream and not self.done:
            st.wait_event(self.event)
        return self.buf


def _packed(desc, w, pass_, wkey, alpha=1.0):
    """bf16 image of alpha * w for `pass_`; cached per registered key and refreshed when the weight's version moved.
    The image layout depends on the layer geometry as well as on the weight (a window that covers its whole input packs
    its data-gradient image differently), so an entry is also keyed by the image size the library reports for `desc`:
    one module used at two input sizes keeps one image per layout."""
    L = _lib.lib()
    nbytes = L.gc_packed_weight_bytes(ctypes.byref(desc), pass_)
    if nbytes == 0:
        _lib.check(-1, "gc_packed_weight_bytes")
    ik = (pass_, int(desc.precision), int(nbytes))
    alpha = float(alpha)
    with _lock:
        # a key registered by repack_divided: the image of this weight holds (alpha * w) / divisor, current as of `epoch`
        dv = _divisors.get(wkey) if wkey else None
        if dv is not None and (dv[0] != w.data_ptr() or w.dtype != torch.float32):
            dv = None
        epoch = dv[3] if dv is not None else 0

        def pack(w32, dst):
            if dv is None:
                _lib.check(L.gc_pack_weight(ctypes.byref(desc), pass_, ctypes.c_void_p(w32.data_ptr()), _lib.strides4(w32), alpha,
                                            ctypes.c_void_p(dst.data_ptr()), _stream()), "gc_pack_weight")
            else:
                _pack_divided([(desc, pass_, w32, alpha, dv[1].data_ptr() + 4 * dv[2], dst)])

        if wkey:
            ent = _wcache.get(wkey, {}).get(ik)
            if ent is not None and ent.ptr == w.data_ptr() and ent.shape == tuple(w.shape):
                if ent.version == w._version and ent.alpha == alpha and ent.div_epoch == epoch:
                    return ent.ready(w)
                if w.dtype == torch.float32:
                    # same storage, new values: refresh the image in place (keeps repack() able to find it)
                    pack(w, ent.buf)
                    ent.alpha, ent.div_epoch = alpha, epoch
                    ent.mark_packed(w)
                    return ent.buf
        w32 = w if w.dtype == torch.float32 else w.float()
        buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        pack(w32, buf)
        if wkey:
            ent = _Image()
            ent.buf, ent.desc, ent.wref, ent.alpha, ent.div_epoch = buf, ConvDesc.from_buffer_copy(desc), weakref.ref(w), alpha, epoch
            ent.mark_packed(w)
            _wcache.setdefault(wkey, {})[ik] = ent
        return buf


def _pack_divided(jobs):
    """One gc_pack_weights_div launch (per 24 images) for jobs of (descriptor, pass, fp32 weight, alpha, device address of the
    divisor, destination)."""
    n = len(jobs)
    descs = (ConvDesc * n)(*[j[0] for j in jobs])
    passes = (ctypes.c_int32 * n)(*[j[1] for j in jobs])
    ws = (ctypes.c_void_p * n)(*[j[2].data_ptr() for j in jobs])
    strides = (ctypes.c_int64 * (4 * n))(*[s_ for j in jobs for s_ in j[2].stride()])
    alphas = (ctypes.c_float * n)(*[j[3] for j in jobs])
    divs = (ctypes.c_void_p * n)(*[j[4] for j in jobs])
    packed = (ctypes.c_void_p * n)(*[j[5].data_ptr() for j in jobs])
    _lib.check(_lib.lib().gc_pack_weights_div(n, descs, passes, ws, strides, alphas, divs, packed, _stream()), "gc_pack_weights_div")


def repack_divided(wkeys, weights, divisors, index=None, forward_only=False):
    """Registers divisors[index[i]] (index None: divisors[i]; an element of one fp32 device tensor: the sigma of a spectral norm)
    as the divisor of the packed images of weights[i] (module key wkeys[i]) and refreshes, in ONE launch, every image those keys have cached: the
    convolutions that follow are called with the WEIGHT ITSELF (a view of it, see divisor_epoch) and find images of
    weight / divisor without the tensor weight / divisor and without a pack launch of their own.  An image that does not
    exist yet is packed with the divisor at its first use, and so is a data-gradient image left out here (forward_only: a
    pass that no backward pass will follow).  Returns the epoch of this registration: the images are those of
    THIS call's divisors until the next call for the same keys."""
    with _lock:
        _div_epoch[0] += 1
        epoch = _div_epoch[0]
        jobs, ents = [], []
        if divisors.dtype != torch.float32 or not divisors.is_contiguous() or not divisors.is_cuda:
            raise RuntimeError("ganconv: repack_divided takes the divisors as one contiguous fp32 device tensor")
        for j, (key, w) in enumerate(zip(wkeys, weights)):
            i = j if index is None else int(index[j])
            if not key:
                continue
            _divisors[key] = (w.data_ptr(), divisors, i, epoch)
            for ik, ent in _wcache.get(key, {}).items():
                if w.dtype != torch.float32 or w.data_ptr() != ent.ptr or tuple(w.shape) != ent.shape:
                    continue
                if forward_only and ik[0] != _lib.GC_PASS_FPROP:
                    continue          # (stays at its old epoch: refreshed at its next use, should there be one)
                jobs.append((ent.desc, ik[0], w, ent.alpha, divisors.data_ptr() + 4 * i, ent.buf))
                ents.append((ent, w))
        if jobs:
            _pack_divided(jobs)
            st = torch.cuda.current_stream(jobs[0][2].device)
            ev = torch.cuda.Event()
            ev.record(st)
            for ent, w in ents:
                ent.div_epoch = epoch
                ent.mark_packed(w, ev)
        return epoch


def divisor_epoch(wkey):
    """Epoch of the divisor the images of `wkey` hold now (0: none registered)."""
    dv = _divisors.get(wkey)
    return dv[3] if dv is not None else 0


def weight_images_settled():
    """Call after a device synchronisation: every cached image that was packed eagerly has finished, so later uses on
    other streams skip the wait for its pack (GraphedStep does this in front of a recording that keeps the images)."""
    with _lock:
        for images in _wcache.values():
            for ent in images.values():
                if not ent.recorded:
                    ent.done = True


def repack(wkeys):
    """Refreshes, in ONE launch, every cached weight image of the given module keys whose parameter has changed since
    it was packed (call right after an optimizer step: the convolutions that follow then find current images instead
    of re-packing one weight at a time).  Images are rewritten in place."""
    with _lock:
        jobs = []
        for key in wkeys:
            if key in _divisors:          # (refreshed per forward pass by repack_divided)
                continue
            for ik, ent in _wcache.get(key, {}).items():
                w = ent.wref()
                if (w is None or w.dtype != torch.float32 or w.data_ptr() != ent.ptr or tuple(w.shape) != ent.shape
                        or w._version == ent.version):
                    continue
                jobs.append((ik, ent, w))
        if not jobs:
            return 0
        n = len(jobs)
        descs = (ConvDesc * n)(*[e.desc for _, e, _ in jobs])
        passes = (ctypes.c_int32 * n)(*[ik[0] for ik, _, _ in jobs])
        ws = (ctypes.c_void_p * n)(*[w.data_ptr() for _, _, w in jobs])
        strides = (ctypes.c_int64 * (4 * n))(*[s_ for _, _, w in jobs for s_ in w.stride()])
        alphas = (ctypes.c_float * n)(*[e.alpha for _, e, _ in jobs])
        packed = (ctypes.c_void_p * n)(*[e.buf.data_ptr() for _, e, _ in jobs])
        _lib.check(_lib.lib().gc_pack_weights(n, descs, passes, ws, strides, alphas, packed, _stream()), "gc_pack_weights")
        for _, e, w in jobs:
            e.mark_packed(w)
        return n


def _epilogue(bias, act=0, slope=0.0, out=None):
    """Epilogue descriptor (or None when there is nothing to fuse) and the tensors it points to.  `out` (the fp32
    output) gets a bf16 side copy when a later convolution can use one."""
    twin = _bf16_twin(out) if out is not None else None
    if bias is None and act == 0 and twin is None:
        return None, None
    b = None
    ep = Epilogue()
    ep.act, ep.slope = act, slope
    if bias is not None:
        b = bias if bias.dtype == torch.float32 and bias.is_contiguous() else bias.float().contiguous()
        ep.bias = b.data_ptr()
    if twin is not None:
        ep.out_bf16 = twin.data_ptr()
    return ep, (b, twin)


def _act_backward(g, y, act, slope):
    """Gradient through a fused epilogue activation, from its OUTPUT y (differentiable torch ops, any order)."""
    if act == _lib.GC_ACT_NONE:
        return g
    if act == _lib.GC_ACT_LRELU:
        return torch.ops.aten.leaky_relu_backward(g, y, slope, True)
    if act == _lib.GC_ACT_RELU:
        return torch.ops.aten.threshold_backward(g, y, 0.0)
    if act == _lib.GC_ACT_TANH:
        return g * (1 - y * y)
    return g * y * (1 - y)   # sigmoid


# ---- op implementations (run below autograd) -------------------------------------------------------------------
@torch.library.custom_op("ganconv::conv_fprop", mutates_args=())
def conv_fprop(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int, padding: int, precision: int, wkey: int,
               act: int = 0, slope: float = 0.0, out_dtype: int = -1, alpha: float = 1.0, defer_act_grad: bool = False,
               x_act_slope: float = -1.0) -> Tensor:
    """act(conv2d(x, alpha * w) + bias); act is a gc_act code fused into the kernel's epilogue.  The last two arguments
    change nothing in the forward pass; they move the derivative of a ReLU / LeakyReLU epilogue out of a sweep of its own
    and into the data-gradient kernel of the convolution that consumes the activated tensor.  defer_act_grad: the backward
    formula treats the incoming gradient as the gradient with respect to the PRE-activation value; valid when the output's
    only consumer is a convolution called with x_act_slope >= 0.  x_act_slope >= 0: x is the output of such an epilogue
    (slope 0: ReLU), and the gradient returned for x is already multiplied by the activation's derivative.  out_dtype: -1 the
    input's dtype, 0 float32, 1 bfloat16 (activations may be stored in either; accumulation is fp32).  alpha: a per-layer
    scalar folded into the packed weight image (EqualizedLR's sqrt(2 / fan_in), StyleGAN/model.py:33-37): the weight the
    op sees stays the stored parameter, so its cached image is reused until the parameter itself changes."""
    _check_cuda(x, w, bias)
    N, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = w.shape
    if Cin_w != Cin:
        raise RuntimeError(f"ganconv: weight expects {Cin_w} input channels, input has {Cin}")
    if H + 2 * padding < kh or W + 2 * padding < kw:
        raise RuntimeError(f"ganconv: kernel ({kh}x{kw}) larger than padded input ({H + 2 * padding}x{W + 2 * padding})")
    Ho, Wo = (H + 2 * padding - kh) // stride + 1, (W + 2 * padding - kw) // stride + 1
    x = _nhwc(x)
    want = _out_dtype(x, out_dtype)
    # A small output with an activation (a critic's score layer with its sigmoid: one pixel tile, 128 reduction steps): the
    # kernel computes the plain fp32 result, so that its reduction can be split over workgroups, and the activation is
    # applied to the few KB afterwards.
    act_after = act if (act != 0 and N * Cout * Ho * Wo <= _SPLIT_MAX_OUT) else 0
    if act_after:
        act = 0
    narrow = _small_output_in_f32(N * Cout * Ho * Wo, want, act)
    y = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32 if narrow else want, device=x.device, memory_format=torch.channels_last)
    if _dma_ok(x, Cin, Cout, precision):
        x = _to_bf16(x)
    d = _desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, _dtype_code(x), precision, _dtype_code(y))
    wp = _packed(d, w, _lib.GC_PASS_FPROP, wkey, alpha)
    ep, keep = _epilogue(bias, act, slope, y if (act != 0 and y.dtype == torch.float32) else None)
    ep, rows = (ep, None) if (narrow or act_after) else _rows_for(ep, y, Cout)
    ep, scratch = _offer_scratch(ep, y, Cin * kh * kw)
    _lib.check(_lib.lib().gc_conv2d_fprop(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                          ctypes.c_void_p(y.data_ptr()), ctypes.byref(ep) if ep is not None else None, _stream()),
               "gc_conv2d_fprop")
    _remember_rows(y, rows)
    if act_after:
        y = (torch.relu_(y) if act_after == _lib.GC_ACT_RELU else torch.nn.functional.leaky_relu_(y, slope) if act_after == _lib.GC_ACT_LRELU
             else torch.tanh_(y) if act_after == _lib.GC_ACT_TANH else torch.sigmoid_(y))
    return y.to(want) if narrow else y


@conv_fprop.register_fake
def _(x, w, bias, stride, padding, precision, wkey, act=0, slope=0.0, out_dtype=-1, alpha=1.0, defer_act_grad=False, x_act_slope=-1.0):
    N, _, H, W = x.shape
    Cout, _, kh, kw = w.shape
    return x.new_empty((N, Cout, (H + 2 * padding - kh) // stride + 1, (W + 2 * padding - kw) // stride + 1),
                       dtype=_out_dtype(x, out_dtype))


# A caller may offer the memory a convolution's result should be written to (the generator's images straight into the second
# half of the critic's real || generated batch: no concatenation launch).  The next forward-of-a-transposed-convolution /
# data-gradient launch whose result has exactly the offered tensor's shape, dtype and NHWC layout takes it.
_offered_outputs = []


class output_into:
    """with output_into(buf): the launch inside whose result matches `buf` (shape, dtype, dense NHWC) writes into it and
    returns it instead of a fresh tensor.  An offer nobody took is withdrawn at exit."""

    def __init__(self, buf):
        self.buf = buf

    def __enter__(self):
        _offered_outputs.append(self.buf)
        return self

    def __exit__(self, *exc):
        for i, b in enumerate(_offered_outputs):
            if b is self.buf:
                del _offered_outputs[i]
                break
        return False


def _result_tensor(shape, dtype, device):
    for i, b in enumerate(_offered_outputs):
        if tuple(b.shape) == tuple(shape) and b.dtype == dtype and b.device == device and b.is_contiguous(memory_format=torch.channels_last):
            return _offered_outputs.pop(i)
    return torch.empty(shape, dtype=dtype, device=device, memory_format=torch.channels_last)


def _dgrad_launch(gy, w, bias, H, W, stride, padding, precision, wkey, act, slope, out_dtype, alpha, mask=None, mask_slope=0.0):
    _check_cuda(gy, w, bias, mask)
    N, Cout, Ho, Wo = gy.shape
    Cout_w, Cin, kh, kw = w.shape
    if Cout_w != Cout:
        raise RuntimeError(f"ganconv: weight has {Cout_w} output channels, gradient has {Cout}")
    if (H + 2 * padding - kh) // stride + 1 != Ho or (W + 2 * padding - kw) // stride + 1 != Wo:
        raise RuntimeError(f"ganconv: input size ({H}x{W}) is inconsistent with the gradient's ({Ho}x{Wo})")
    gy = _nhwc(gy)
    want = _out_dtype(gy, out_dtype)
    narrow = mask is None and _small_output_in_f32(N * Cin * H * W, want, act)
    if narrow:
        gx = torch.empty((N, Cin, H, W), dtype=torch.float32, device=gy.device, memory_format=torch.channels_last)
    else:
        gx = _result_tensor((N, Cin, H, W), want, gy.device)
    if _dma_ok(gy, Cout, Cin, precision):
        gy = _to_bf16(gy)
    d = _desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, _dtype_code(gx), precision, _dtype_code(gy))
    wp = _packed(d, w, _lib.GC_PASS_DGRAD, wkey, alpha)
    if mask is not None:
        if mask.shape != gx.shape:
            raise RuntimeError(f"ganconv: the activated tensor {tuple(mask.shape)} does not have the gradient's shape {tuple(gx.shape)}")
        mask = _nhwc(mask)
        if mask.dtype != gx.dtype:
            mask = mask.to(gx.dtype)
        ep, keep = Epilogue(), (mask,)
        ep.act, ep.slope, ep.grad_mask = _lib.GC_ACT_NONE, mask_slope, mask.data_ptr()
        if bias is not None:
            raise RuntimeError("ganconv: a masked data gradient takes no bias")
    else:
        ep, keep = _epilogue(bias, act, slope, gx if (act != 0 and gx.dtype == torch.float32) else None)
    ep, rows = (ep, None) if narrow else _rows_for(ep, gx, Cin)
    ep, scratch = _offer_scratch(ep, gx, Cout * kh * kw // max(stride * stride, 1))
    _lib.check(_lib.lib().gc_conv2d_dgrad(ctypes.byref(d), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                          ctypes.c_void_p(gx.data_ptr()), ctypes.byref(ep) if ep is not None else None, _stream()),
               "gc_conv2d_dgrad")
    _remember_rows(gx, rows)
    return gx.to(want) if narrow else gx


@torch.library.custom_op("ganconv::conv_dgrad", mutates_args=())
def conv_dgrad(gy: Tensor, w: Tensor, bias: Optional[Tensor], H: int, W: int, stride: int, padding: int, precision: int,
               wkey: int, act: int = 0, slope: float = 0.0, out_dtype: int = -1, alpha: float = 1.0) -> Tensor:
    """gx [N, Cin, H, W] of the conv x -> y described by (alpha * w [Cout,Cin,kh,kw], stride, padding), given gy.
    `bias` ([Cin]) is added to gx: the bias of a transposed convolution whose forward pass this is.
    out_dtype as for conv_fprop."""
    return _dgrad_launch(gy, w, bias, H, W, stride, padding, precision, wkey, act, slope, out_dtype, alpha)


@torch.library.custom_op("ganconv::conv_dgrad_masked", mutates_args=())
def conv_dgrad_masked(gy: Tensor, w: Tensor, x: Tensor, x_act_slope: float, stride: int, padding: int, precision: int, wkey: int,
                      out_dtype: int = -1, alpha: float = 1.0) -> Tensor:
    """conv_dgrad(gy, w) * (x > 0 ? 1 : x_act_slope): the gradient of the conv x -> y with respect to the value in front
    of the ReLU / LeakyReLU epilogue that produced x, the activation's derivative applied in the data-gradient kernel's
    epilogue (the kernel reads x at the position it stores)."""
    return _dgrad_launch(gy, w, None, x.shape[2], x.shape[3], stride, padding, precision, wkey, 0, 0.0, out_dtype, alpha, x, x_act_slope)


@conv_dgrad_masked.register_fake
def _(gy, w, x, x_act_slope, stride, padding, precision, wkey, out_dtype=-1, alpha=1.0):
    return gy.new_empty(x.shape, dtype=_out_dtype(gy, out_dtype))


@conv_dgrad.register_fake
def _(gy, w, bias, H, W, stride, padding, precision, wkey, act=0, slope=0.0, out_dtype=-1, alpha=1.0):
    return gy.new_empty((gy.shape[0], w.shape[1], H, W), dtype=_out_dtype(gy, out_dtype))


@torch.library.custom_op("ganconv::conv_fprop_masked", mutates_args=())
def conv_fprop_masked(x: Tensor, w: Tensor, mask: Tensor, mask_slope: float, stride: int, padding: int, precision: int, wkey: int,
                      out_dtype: int = -1, alpha: float = 1.0) -> Tensor:
    """conv2d(x, alpha * w) * (mask > 0 ? 1 : mask_slope): a forward convolution whose result is multiplied, in the kernel's
    output stage, by the derivative of the ReLU / LeakyReLU whose output is `mask` (gc_epilogue.grad_mask).  The second-order
    pass of a convolution whose activation's derivative is deferred to its consumer: the gradient with respect to the masked
    data gradient, handed on already masked (see the hand-over described at fuse_second_order_mask)."""
    _check_cuda(x, w, mask)
    N, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = w.shape
    if Cin_w != Cin:
        raise RuntimeError(f"ganconv: weight expects {Cin_w} input channels, input has {Cin}")
    Ho, Wo = (H + 2 * padding - kh) // stride + 1, (W + 2 * padding - kw) // stride + 1
    x = _nhwc(x)
    want = _out_dtype(x, out_dtype)
    y = torch.empty((N, Cout, Ho, Wo), dtype=want, device=x.device, memory_format=torch.channels_last)
    if tuple(mask.shape) != tuple(y.shape):
        raise RuntimeError(f"ganconv: the activated tensor {tuple(mask.shape)} does not have the result's shape {tuple(y.shape)}")
    mask = _nhwc(mask)
    if mask.dtype != y.dtype:
        mask = mask.to(y.dtype)
    if _dma_ok(x, Cin, Cout, precision):
        x = _to_bf16(x)
    d = _desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, _dtype_code(x), precision, _dtype_code(y))
    wp = _packed(d, w, _lib.GC_PASS_FPROP, wkey, alpha)
    ep = Epilogue()
    ep.act, ep.slope, ep.grad_mask = _lib.GC_ACT_NONE, mask_slope, mask.data_ptr()
    _lib.check(_lib.lib().gc_conv2d_fprop(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                          ctypes.c_void_p(y.data_ptr()), ctypes.byref(ep), _stream()), "gc_conv2d_fprop")
    return y


@conv_fprop_masked.register_fake
def _(x, w, mask, mask_slope, stride, padding, precision, wkey, out_dtype=-1, alpha=1.0):
    return x.new_empty(mask.shape, dtype=_out_dtype(x, out_dtype))


_IMAGE_3X3 = os.environ.get("GC_NO_IMAGE_3X3") is None


def _image_layer(x, gy, geom, precision):
    """fp32 image, bf16 gradient, in the geometry of the image layer's weight-gradient kernel (conv_wgrad_image.hip),
    which rounds the image to bf16 in registers: no bf16 copy of the image is needed."""
    if geom is None or x.dtype != torch.float32 or gy.dtype != torch.bfloat16 or precision != _lib.GC_PREC_BF16:
        return False
    kh, kw, stride, padding = geom
    (_, Cin, H, W), (_, Cout, Ho, Wo) = x.shape, gy.shape
    if Wo % 32 or Ho % 4 or Cout % 16 or Cout > 64:
        return False
    if stride == 1 and ((kh == 3 and kw == 3 and padding == 1) or (kh == 1 and kw == 1 and padding == 0)):
        # (BigGAN's first discriminator block: 3 x 3 on the image; StyleGAN's from-RGB layers: 1 x 1)
        return Cin <= 4 and H == Ho and W == Wo and _IMAGE_3X3
    return Cin <= 8 and kh == 4 and kw == 4 and stride == 2 and padding == 1 and H == 2 * Ho and W == 2 * Wo


def _wgrad_operands(x, gy, precision, geom=None):
    x, gy = _nhwc(x), _nhwc(gy)
    if _image_layer(x, gy, geom, precision):
        return x, gy
    if x.dtype != gy.dtype:      # mixed storage (e.g. an fp32 image feeding bf16 activations): the operands are bf16 anyway
        x = x if x.dtype == torch.bfloat16 else _to_bf16(x)
        gy = gy if gy.dtype == torch.bfloat16 else _to_bf16(gy)
    # bf16 copies only where the LDS-DMA weight-gradient kernel can use them; the other kernels convert fp32 operands
    # in registers (narrow layers are bound by reading these tensors once, a copy would only add traffic)
    Cin, Cout = x.shape[1], gy.shape[1]
    if (precision == _lib.GC_PREC_BF16 and x.dtype == torch.float32 and Cin % 8 == 0 and Cout % 8 == 0 and Cin > 32
            and max(x.numel(), gy.numel()) * 2 < (1 << 31) - (1 << 21)):
        x, gy = _to_bf16(x), _to_bf16(gy)
    return x, gy


def _wgrad_launch(x, gy, gw, kh, kw, stride, padding, precision, alpha, accumulate, stream_of, want_bias=False):
    """gw (+)= alpha * weight gradient.  stream_of(*tensors to keep alive) names the launch stream once the operands exist.
    want_bias: returns gy summed over (n, h, w) -- by the weight-gradient kernel itself where it holds gy in its operand
    storage type already (no extra rounding) and its plan sums it on the way, otherwise by channel_sum."""
    N, Cin, H, W = x.shape
    N2, Cout, Ho, Wo = gy.shape
    if N2 != N or (H + 2 * padding - kh) // stride + 1 != Ho or (W + 2 * padding - kw) // stride + 1 != Wo:
        raise RuntimeError("ganconv: x and gy shapes are inconsistent with the convolution geometry")
    gy_given = gy
    x, gy = _wgrad_operands(x, gy, precision, (kh, kw, stride, padding))
    d = _desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, _dtype_code(x), precision, _dtype_code(gy))
    L = _lib.lib()
    ws_bytes = L.gc_wgrad_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    gb = None
    if want_bias and gy_given.dtype == gy.dtype and L.gc_wgrad_sums_bias(ctypes.byref(d), _ptr(x), _ptr(gy)):
        gb = torch.empty(Cout, dtype=torch.float32, device=x.device)
    stream = stream_of(ws, x, gy, gw, gb)
    _lib.check(L.gc_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(gy.data_ptr()),
                                 ctypes.c_void_p(gw.data_ptr()), _lib.strides4(gw), float(alpha), 1 if accumulate else 0,
                                 None if gb is None else _ptr(gb), ctypes.c_void_p(ws.data_ptr()), ws.numel(), stream), "gc_conv2d_wgrad")
    if want_bias and gb is None:
        gb = channel_sum(gy_given)
    return gb


@torch.library.custom_op("ganconv::conv_wgrad", mutates_args=())
def conv_wgrad(x: Tensor, gy: Tensor, kh: int, kw: int, stride: int, padding: int, precision: int, alpha: float = 1.0) -> Tensor:
    """alpha * gw, gw [Cout, Cin, kh, kw] (fp32) being the weight gradient of the conv x -> y given x and gy (alpha: the
    scalar that multiplied the weight in the forward pass, see conv_fprop)."""
    _check_cuda(x, gy)
    gw = torch.empty((gy.shape[1], x.shape[1], kh, kw), dtype=torch.float32, device=x.device)
    side = _SideLaunch.active
    _wgrad_launch(x, gy, gw, kh, kw, stride, padding, precision, alpha, False, side.fork if side is not None else (lambda *keep: _stream()))
    return gw


@torch.library.custom_op("ganconv::conv_wgrad_bias", mutates_args=())
def conv_wgrad_bias(x: Tensor, gy: Tensor, kh: int, kw: int, stride: int, padding: int, precision: int,
                    alpha: float = 1.0) -> tuple[Tensor, Tensor]:
    """(conv_wgrad(x, gy, ...), gy.sum((0, 2, 3)) in float32): weight and bias gradient of a convolution from one launch
    where the weight-gradient kernel sums gy next to its products (channel_sum otherwise)."""
    _check_cuda(x, gy)
    gw = torch.empty((gy.shape[1], x.shape[1], kh, kw), dtype=torch.float32, device=x.device)
    side = _SideLaunch.active
    gb = _wgrad_launch(x, gy, gw, kh, kw, stride, padding, precision, alpha, False,
                       side.fork if side is not None else (lambda *keep: _stream()), want_bias=True)
    return gw, gb


@conv_wgrad_bias.register_fake
def _(x, gy, kh, kw, stride, padding, precision, alpha=1.0):
    return (x.new_empty((gy.shape[1], x.shape[1], kh, kw), dtype=torch.float32), x.new_empty((gy.shape[1],), dtype=torch.float32))


def _wgrad_launch_multi(pairs, gw, kh, kw, stride, padding, precision, alpha, accumulate):
    """gw (+)= alpha * the weight gradient summed over several (x, gy) pairs of one layer, one launch + one reduction."""
    x0, gy0 = pairs[0]
    _, Cin, H, W = x0.shape
    _, Cout, Ho, Wo = gy0.shape
    for x, gy in pairs:
        if x.shape[1:] != x0.shape[1:] or gy.shape[1:] != gy0.shape[1:] or x.shape[0] != gy.shape[0] or x.dtype != x0.dtype or gy.dtype != gy0.dtype:
            raise RuntimeError("ganconv: the pairs of a multi-segment weight gradient must share geometry and storage type")
    d = _desc(x0.shape[0], H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, _dtype_code(x0), precision, _dtype_code(gy0))
    L = _lib.lib()
    n = len(pairs)
    Ns = (ctypes.c_int32 * n)(*[p[0].shape[0] for p in pairs])
    xs = (ctypes.c_void_p * n)(*[p[0].data_ptr() for p in pairs])
    gys = (ctypes.c_void_p * n)(*[p[1].data_ptr() for p in pairs])
    ws_bytes = L.gc_wgrad_multi_workspace_bytes(ctypes.byref(d), n, Ns)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x0.device)
    _lib.check(L.gc_conv2d_wgrad_multi(ctypes.byref(d), n, xs, gys, Ns, ctypes.c_void_p(gw.data_ptr()), _lib.strides4(gw), float(alpha),
                                       1 if accumulate else 0, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _stream()),
               "gc_conv2d_wgrad_multi")


def _wgrad_to_sink(w, x, gy, stride, padding, precision, alpha):
    """Hands the weight gradient of w to the active sink; False when there is none for it (the caller computes it)."""
    sink = weight_gradient_sink.active
    if sink is None or not x.is_cuda:
        return False
    ent = sink.buffer_for(w)
    if ent is None:
        return False
    buf, accumulate = ent
    if buf.dtype != torch.float32 or not buf.is_contiguous():
        raise RuntimeError("ganconv: weight_gradient_sink buffers must be contiguous float32 tensors")
    if sink.defer:      # operands in their kernel form now (on the stream that produced them), the launch at flush()
        xo, gyo = _wgrad_operands(x, gy, precision, (w.shape[2], w.shape[3], stride, padding))
        sink.contribution(w.data_ptr(), (xo, gyo, (w.shape[2], w.shape[3], stride, padding, precision, alpha)))
        return True
    _wgrad_launch(x, gy, buf, w.shape[2], w.shape[3], stride, padding, precision, alpha, accumulate, lambda *keep: sink.stream_for(w, *keep))
    return True


@conv_wgrad.register_fake
def _(x, gy, kh, kw, stride, padding, precision, alpha=1.0):
    return x.new_empty((gy.shape[1], x.shape[1], kh, kw), dtype=torch.float32)


@torch.library.custom_op("ganconv::channel_sum", mutates_args=())
def channel_sum(x: Tensor) -> Tensor:
    """x.sum((0, 2, 3)) in float32: the bias gradient of a convolution (fixed summation order when C % 4 == 0)."""
    _check_cuda(x)
    x = _nhwc(x)
    N, C, H, W = x.shape
    P = N * H * W
    L = _lib.lib()
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    nws = L.gc_channel_sum_workspace_bytes(P, C)
    ws = torch.empty(nws, dtype=torch.uint8, device=x.device) if nws > 0 else None
    _lib.check(L.gc_channel_sum(_ptr(x), _ptr(out), _optr(ws), nws, P, C, _dtype_code(x), _stream()), "gc_channel_sum")
    return out


@channel_sum.register_fake
def _(x):
    return x.new_empty(x.shape[1], dtype=torch.float32)


@torch.library.custom_op("ganconv::act_backward_sum", mutates_args=())
def act_backward_sum(gy: Tensor, y: Tensor, act: int, slope: float) -> tuple[Tensor, Tensor]:
    """(gy * act'(y), its sum over (N, H, W)): the gradient through an epilogue activation and the bias gradient of its
    convolution in one sweep.  No autograd formula: the backward formulas use it only when nothing differentiates them."""
    _check_cuda(gy, y)
    gy, y = _nhwc(gy), _nhwc(y)
    N, C, H, W = gy.shape
    P = N * H * W
    L = _lib.lib()
    out = torch.empty_like(gy, memory_format=torch.channels_last)
    colsum = torch.empty(C, dtype=torch.float32, device=gy.device)
    nws = L.gc_channel_sum_workspace_bytes(P, C)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=gy.device)
    _lib.check(L.gc_act_backward_channel_sum(_ptr(gy), _ptr(y), _ptr(out), _ptr(colsum), _ptr(ws), ws.numel(), P, C, _dtype_code(gy),
                                             act, slope, _stream()), "gc_act_backward_channel_sum")
    return out, colsum


@act_backward_sum.register_fake
def _(gy, y, act, slope):
    return torch.empty_like(gy), gy.new_empty(gy.shape[1], dtype=torch.float32)


def _act_sum_fusable(g, y):
    return (g.is_cuda and g.dtype == y.dtype and g.dtype in (torch.float32, torch.bfloat16) and g.shape == y.shape and g.shape[1] % 4 == 0
            and g.numel() > 0 and not (torch.is_grad_enabled() and (g.requires_grad or y.requires_grad)))


def _channel_sum_setup(ctx, inputs, output):
    ctx.shape, ctx.dtype = inputs[0].shape, inputs[0].dtype


def _channel_sum_backward(ctx, g):
    # linear in x: the gradient is a broadcast (reached when a penalty differentiates through a bias gradient)
    return g[None, :, None, None].expand(ctx.shape).to(ctx.dtype)


torch.library.register_autograd("ganconv::channel_sum", _channel_sum_backward, setup_context=_channel_sum_setup)


# ---- autograd: each formula uses only the three ops above -------------------------------------------------------
def _edges(*ts):
    """The autograd node each tensor's gradient is handed to (None for a leaf or a tensor that takes no gradient)."""
    return tuple(t.grad_fn if (t is not None and t.requires_grad) else None for t in ts)


_inputs_only = False
_skip_leaf_inputs = False


class parameter_gradients_only:
    """While active, the convolution backward formulas compute no gradient with respect to an activation operand that
    is a leaf of the autograd graph (an input image marked requires_grad).  For torch.autograd.grad(loss, parameters)
    behind a gradient penalty: the mixed images were differentiated by in the first pass and are leaves that require
    grad, so the second pass would run the first layer's data-gradient kernel for a gradient nobody asked for (the engine
    cannot be asked about leaves).  Process-wide like input_gradients_only."""

    def __enter__(self):
        global _skip_leaf_inputs
        self.prev, _skip_leaf_inputs = _skip_leaf_inputs, True
        return self

    def __exit__(self, *exc):
        global _skip_leaf_inputs
        _skip_leaf_inputs = self.prev
        return False


class input_gradients_only:
    """While active, the convolution backward formulas compute gradients with respect to activations only (weights and
    biases get None).  For a backward pass that is restricted to a network's input, such as the first pass of a gradient
    penalty: torch.autograd.grad(scores, images, create_graph=True) runs every node on the way, and needs_input_grad,
    fixed when the forward pass ran, still asks each of them for its weight gradient, a fifth of a critic iteration's
    kernel time that nothing reads.  Process-wide, not per thread: the formulas run on the autograd engine's own thread
    while the caller waits inside torch.autograd.grad."""

    def __enter__(self):
        global _inputs_only
        self.prev, _inputs_only = _inputs_only, True
        return self

    def __exit__(self, *exc):
        global _inputs_only
        _inputs_only = self.prev
        return False


def _wanted(ctx, i):
    """Whether the running backward pass uses the gradient of input i (0: the activation operand, 1: the weight, 2: the
    bias).  Beyond needs_input_grad: nothing for parameters inside input_gradients_only, and for an interior tensor the
    engine's own answer (torch.autograd.grad(..., inputs=[...]) executes only the nodes that lead to those inputs; the
    engine does not answer for leaves)."""
    if not ctx.needs_input_grad[i] or (_inputs_only and i > 0):
        return False
    node = ctx.edges[i]
    if node is None:
        return not (_skip_leaf_inputs and i == 0)
    try:
        return torch._C._will_engine_execute_node(node)
    except RuntimeError:          # no information
        return True


# Second-order pass through a deferred activation derivative (the gradient penalty's double backward through the critic's
# first LeakyReLU, BasicGANs/models.py:78).  First order: layer 2 returns G = m * dgrad(gy2, w2) (conv_dgrad_masked, m the
# derivative of layer 1's activation) and layer 1 computes dgrad(G, w1).  Second order: layer 1's node returns
# t = fprop(u, w1) as the gradient with respect to G, and conv_dgrad_masked's node needs m * t.  Instead of a sweep there,
# layer 1's node produces m * t in its kernel's output stage (conv_fprop_masked).  The two nodes agree on it as follows:
# conv_dgrad_masked registers its result G (by address, with a weak reference) together with (m, slope, token); the
# convolution whose backward formula receives exactly that tensor as its incoming gradient passes (m, slope, token) to its
# conv_dgrad node; in the second-order pass that node multiplies by m and notes the tensor it returned under the token; the
# conv_dgrad_masked node skips its own multiplication only when the tensor it receives is that one, and raises if a
# premasked tensor reached it changed (a second reader of G would make autograd add its gradient to the premasked one).
fuse_second_order_mask = os.environ.get("GC_NO_SECOND_ORDER_MASK") is None      # (the variable: an A/B switch for measurements)
_premask_offers = {}      # address of G -> (weak reference to G, m, slope, token)
_premask_done = {}        # token -> weak reference to the premasked tensor the conv_dgrad node returned
_premask_pending = [None]
_premask_token = [0]


def _premask_offer(out, mask, slope):
    _premask_token[0] += 1
    while len(_premask_offers) >= 16:
        _premask_offers.pop(next(iter(_premask_offers)))
    while len(_premask_done) >= 16:
        _premask_done.pop(next(iter(_premask_done)))
    _premask_offers[out.data_ptr()] = (weakref.ref(out), mask, float(slope), _premask_token[0])
    return _premask_token[0]


def _premask_for(gy):
    """(mask, slope, token) when `gy` is the registered result of a conv_dgrad_masked call, else None."""
    ent = _premask_offers.pop(gy.data_ptr(), None) if (fuse_second_order_mask and _premask_offers) else None
    if ent is None:
        return None
    g = ent[0]()
    if g is None or g.data_ptr() != gy.data_ptr() or g.shape != gy.shape or g.dtype != gy.dtype or g._version != gy._version:
        return None
    return ent[1:]


def _input_weight_grads(x, w, gy, stride, padding, precision, wkey, alpha, need_x, need_w, x_act_slope=-1.0, need_b=False):
    """(gx, gw, gb) of y = conv(x, alpha * w) + b from gy: the data-gradient and weight-gradient launches (the weight gradient
    goes to an active weight_gradient_sink instead when one accepts it; with backward overlap on, it runs on the side
    stream).  The bias gradient comes out of the weight-gradient launch where that kernel sums gy on the way."""
    gx = gw = gb = None

    def dgrad():
        if x_act_slope >= 0.0:
            return conv_dgrad_masked(gy, w, x, x_act_slope, stride, padding, precision, wkey, _code_of(x.dtype), alpha)
        _premask_pending[0] = _premask_for(gy) if gy.requires_grad else None      # (picked up by the conv_dgrad node's setup)
        try:
            return conv_dgrad(gy, w, None, x.shape[2], x.shape[3], stride, padding, precision, wkey, 0, 0.0, _code_of(x.dtype), alpha)
        finally:
            _premask_pending[0] = None

    def wgrad():
        if need_b:
            return conv_wgrad_bias(x, gy, w.shape[2], w.shape[3], stride, padding, precision, alpha)
        return conv_wgrad(x, gy, w.shape[2], w.shape[3], stride, padding, precision, alpha), None

    if need_w and _wgrad_to_sink(w, x, gy, stride, padding, precision, alpha):
        need_w = False
    if _overlap and need_x and need_w and gy.is_cuda:
        with _SideLaunch(gy.device) as side:
            gw, gb = wgrad()
        gx = dgrad()
        side.join()
        gw = gw.to(w.dtype)
    else:
        if need_x:
            gx = dgrad()
        if need_w:
            gw, gb = wgrad()
            gw = gw.to(w.dtype)
    if need_b and gb is None:
        gb = channel_sum(gy)
    return gx, gw, gb


def _fprop_setup(ctx, inputs, output):
    x, w, bias, stride, padding, precision, wkey, act, slope, _, alpha, defer_act_grad, x_act_slope = inputs
    ctx.edges = _edges(x, w, bias)
    if defer_act_grad:
        if act not in (_lib.GC_ACT_RELU, _lib.GC_ACT_LRELU):
            raise RuntimeError("ganconv: defer_act_grad is for ReLU / LeakyReLU epilogues")
        act = 0                      # the consumer's data-gradient kernel applies the derivative
    if act != 0:
        ctx.save_for_backward(x, w, output)
    else:
        ctx.save_for_backward(x, w)
    ctx.cfg = (stride, padding, precision, wkey, bias is not None, act, slope, alpha)
    ctx.x_act_slope = x_act_slope


def _fprop_backward(ctx, gy):
    stride, padding, precision, wkey, has_bias, act, slope, alpha = ctx.cfg
    need_x, need_w, need_b = _wanted(ctx, 0), _wanted(ctx, 1), has_bias and _wanted(ctx, 2)
    gb = None
    if act != 0:
        x, w, y = ctx.saved_tensors
        if need_b and _act_sum_fusable(gy, y):
            gy, gb = act_backward_sum(gy, y, act, slope)
        else:
            gy = _act_backward(gy, y, act, slope)
    else:
        x, w = ctx.saved_tensors
    gx, gw, gb2 = _input_weight_grads(x, w, gy, stride, padding, precision, wkey, alpha, need_x, need_w, ctx.x_act_slope,
                                      need_b and gb is None)
    return gx, gw, (gb if gb is not None else gb2), None, None, None, None, None, None, None, None, None, None


torch.library.register_autograd("ganconv::conv_fprop", _fprop_backward, setup_context=_fprop_setup)


def _dgrad_setup(ctx, inputs, output):
    gy, w, bias, H, W, stride, padding, precision, wkey, act, slope, _, alpha = inputs
    ctx.edges = _edges(gy, w, bias)
    if act != 0:
        ctx.save_for_backward(gy, w, output)
    else:
        ctx.save_for_backward(gy, w)
    ctx.cfg = (stride, padding, precision, wkey, bias is not None, act, slope, alpha)
    ctx.premask, _premask_pending[0] = (_premask_pending[0] if act == 0 else None), None


def _dgrad_backward(ctx, u):
    stride, padding, precision, wkey, has_bias, act, slope, alpha = ctx.cfg
    need_gy, need_w, need_b = _wanted(ctx, 0), _wanted(ctx, 1), has_bias and _wanted(ctx, 2)
    g_b = None
    if act != 0:
        gy, w, out = ctx.saved_tensors
        if need_b and _act_sum_fusable(u, out):
            u, g_b = act_backward_sum(u, out, act, slope)
        else:
            u = _act_backward(u, out, act, slope)
    else:
        gy, w = ctx.saved_tensors
    g_gy = g_w = None

    def fprop_of_u():
        premask = getattr(ctx, "premask", None)
        if premask is None:
            return conv_fprop(u, w, None, stride, padding, precision, wkey, 0, 0.0, _code_of(gy.dtype), alpha)
        mask, mask_slope, token = premask       # gy is a masked data gradient: its gradient leaves already masked
        t = conv_fprop_masked(u, w, mask, mask_slope, stride, padding, precision, wkey, _code_of(gy.dtype), alpha)
        _premask_done[token] = weakref.ref(t)
        return t

    if need_w and _wgrad_to_sink(w, u, gy, stride, padding, precision, alpha):
        need_w = False
    if _overlap and need_gy and need_w and u.is_cuda:
        with _SideLaunch(u.device) as side:
            g_w = conv_wgrad(u, gy, w.shape[2], w.shape[3], stride, padding, precision, alpha)
        g_gy = fprop_of_u()
        side.join()
        g_w = g_w.to(w.dtype)
    else:
        if need_gy:
            g_gy = fprop_of_u()
        if need_w:
            g_w = conv_wgrad(u, gy, w.shape[2], w.shape[3], stride, padding, precision, alpha).to(w.dtype)
    if need_b and g_b is None:
        g_b = channel_sum(u)
    return g_gy, g_w, g_b, None, None, None, None, None, None, None, None, None, None


torch.library.register_autograd("ganconv::conv_dgrad", _dgrad_backward, setup_context=_dgrad_setup)


def _dgrad_masked_setup(ctx, inputs, output):
    gy, w, x, x_act_slope, stride, padding, precision, wkey, _, alpha = inputs
    ctx.edges = _edges(gy, w)
    ctx.save_for_backward(gy, w, x)
    ctx.cfg = (x_act_slope, stride, padding, precision, wkey, alpha)
    ctx.token = _premask_offer(output, x, x_act_slope) if fuse_second_order_mask else 0


def _dgrad_masked_backward(ctx, u):
    """The op is linear in gy and in w for a fixed mask, and the mask's derivative with respect to x is zero almost
    everywhere: u passes through the same mask, then through the adjoints of the data gradient."""
    x_act_slope, stride, padding, precision, wkey, alpha = ctx.cfg
    gy, w, x = ctx.saved_tensors
    need_gy, need_w = _wanted(ctx, 0), _wanted(ctx, 1)
    done = _premask_done.pop(ctx.token, None) if getattr(ctx, "token", 0) else None
    if done is not None:          # the node that produced u multiplied it by the activation's derivative already
        t = done()
        if t is None or t.data_ptr() != u.data_ptr() or t.shape != u.shape:
            raise RuntimeError("ganconv: a gradient that left its kernel already multiplied by the deferred activation derivative "
                               "reached conv_dgrad_masked changed (a second reader of the masked data gradient?); set "
                               "gan_amd.ops.fuse_second_order_mask = False for this graph")
    else:
        u = _act_backward(u, x, _lib.GC_ACT_LRELU if x_act_slope != 0.0 else _lib.GC_ACT_RELU, x_act_slope)
    g_gy = g_w = None
    if need_w and _wgrad_to_sink(w, u, gy, stride, padding, precision, alpha):
        need_w = False
    if need_gy:
        g_gy = conv_fprop(u, w, None, stride, padding, precision, wkey, 0, 0.0, _code_of(gy.dtype), alpha)
    if need_w:
        g_w = conv_wgrad(u, gy, w.shape[2], w.shape[3], stride, padding, precision, alpha).to(w.dtype)
    return g_gy, g_w, None, None, None, None, None, None, None, None


torch.library.register_autograd("ganconv::conv_dgrad_masked", _dgrad_masked_backward, setup_context=_dgrad_masked_setup)


def _fprop_masked_setup(ctx, inputs, output):
    x, w, mask, mask_slope, stride, padding, precision, wkey, _, alpha = inputs
    ctx.edges = _edges(x, w)
    ctx.save_for_backward(x, w, mask)
    ctx.cfg = (mask_slope, stride, padding, precision, wkey, alpha)


def _fprop_masked_backward(ctx, g):
    """Linear in x and in w for a fixed mask: g passes through the same mask, then through the convolution's adjoints."""
    mask_slope, stride, padding, precision, wkey, alpha = ctx.cfg
    x, w, mask = ctx.saved_tensors
    g = _act_backward(g, mask, _lib.GC_ACT_LRELU if mask_slope != 0.0 else _lib.GC_ACT_RELU, mask_slope)
    gx, gw, _ = _input_weight_grads(x, w, g, stride, padding, precision, wkey, alpha, _wanted(ctx, 0), _wanted(ctx, 1))
    return gx, gw, None, None, None, None, None, None, None, None


torch.library.register_autograd("ganconv::conv_fprop_masked", _fprop_masked_backward, setup_context=_fprop_masked_setup)


def _wgrad_setup(ctx, inputs, output):
    x, gy, kh, kw, stride, padding, precision, alpha = inputs
    ctx.edges = _edges(x, gy)
    ctx.save_for_backward(x, gy)
    ctx.cfg = (stride, padding, precision, alpha)


def _wgrad_backward(ctx, v):
    x, gy = ctx.saved_tensors
    stride, padding, precision, alpha = ctx.cfg
    g_x = g_gy = None
    if _wanted(ctx, 0):
        g_x = conv_dgrad(gy, v, None, x.shape[2], x.shape[3], stride, padding, precision, 0, 0, 0.0, _code_of(x.dtype), alpha)
    if _wanted(ctx, 1):
        g_gy = conv_fprop(x, v, None, stride, padding, precision, 0, 0, 0.0, _code_of(gy.dtype), alpha)
    return g_x, g_gy, None, None, None, None, None, None


torch.library.register_autograd("ganconv::conv_wgrad", _wgrad_backward, setup_context=_wgrad_setup)


def _wgrad_bias_backward(ctx, v, vb):
    """Both outputs are linear in gy (and the weight gradient in x): gy receives conv(x, v) + vb per channel."""
    x, gy = ctx.saved_tensors
    stride, padding, precision, alpha = ctx.cfg
    g_x = g_gy = None
    if v is not None and _wanted(ctx, 0):
        g_x = conv_dgrad(gy, v, None, x.shape[2], x.shape[3], stride, padding, precision, 0, 0, 0.0, _code_of(x.dtype), alpha)
    if _wanted(ctx, 1):
        if v is not None:
            g_gy = conv_fprop(x, v, vb, stride, padding, precision, 0, 0, 0.0, _code_of(gy.dtype), alpha)
        elif vb is not None:
            g_gy = vb.to(gy.dtype).view(1, -1, 1, 1).expand_as(gy)
    return g_x, g_gy, None, None, None, None, None, None


torch.library.register_autograd("ganconv::conv_wgrad_bias", _wgrad_bias_backward, setup_context=_wgrad_setup)


def _blur_launch(x, mask, slope, mask_input=False):
    _check_cuda(x, mask)
    N, C, H, W = x.shape
    x = _nhwc(x)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    if mask is None:
        _lib.check(_lib.lib().gc_dwblur3x3(_ptr(x), _ptr(y), N, H, W, C, _dtype_code(x), _stream()), "gc_dwblur3x3")
    else:
        if mask.shape != x.shape:
            raise RuntimeError("ganconv: the activated tensor does not have the gradient's shape")
        mask = _nhwc(mask if mask.dtype == x.dtype else mask.to(x.dtype))
        _lib.check(_lib.lib().gc_dwblur3x3_masked(_ptr(x), _ptr(y), _ptr(mask), slope, 1 if mask_input else 0, N, H, W, C, _dtype_code(x),
                                                  _stream()),
                   "gc_dwblur3x3_masked")
    return y


@torch.library.custom_op("ganconv::blur3x3", mutates_args=())
def blur3x3(x: Tensor, x_act_slope: float = -1.0) -> Tensor:
    """Depthwise [1 2 1]x[1 2 1]/16 blur with zero padding 1 (reference StyleGAN/model.py:82-150).  x_act_slope >= 0 (the
    forward result is unchanged): x is the output of a ReLU (0) / LeakyReLU epilogue called with defer_act_grad, and the
    gradient returned for x is already multiplied by that activation's derivative (one sweep: blur3x3_masked)."""
    return _blur_launch(x, None, 0.0)


@blur3x3.register_fake
def _(x, x_act_slope=-1.0):
    return torch.empty_like(x)


@torch.library.custom_op("ganconv::blur3x3_masked", mutates_args=())
def blur3x3_masked(g: Tensor, x: Tensor, x_act_slope: float, mask_input: bool = False) -> Tensor:
    """blur3x3(g) * (x > 0 ? 1 : x_act_slope), or with mask_input blur3x3(g * (x > 0 ? 1 : x_act_slope)): each other's adjoint."""
    return _blur_launch(g, x, x_act_slope, mask_input)


@blur3x3_masked.register_fake
def _(g, x, x_act_slope, mask_input=False):
    return torch.empty_like(g)


def _blur_setup(ctx, inputs, output):
    ctx.slope = inputs[1]
    if ctx.slope >= 0.0:
        ctx.save_for_backward(inputs[0])


def _blur_backward(ctx, g):
    # the kernel is symmetric, so the blur is self-adjoint: its backward (and double backward) is the blur itself
    if ctx.slope >= 0.0:
        return blur3x3_masked(g, ctx.saved_tensors[0], ctx.slope), None
    return blur3x3(g), None


def _blur_masked_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1])
    ctx.slope, ctx.mask_input = inputs[2], inputs[3]


def _blur_masked_backward(ctx, u):
    """Linear in g for a fixed mask: the adjoint applies the mask on the other side of the blur."""
    (x,) = ctx.saved_tensors
    return blur3x3_masked(u, x, ctx.slope, not ctx.mask_input), None, None, None


torch.library.register_autograd("ganconv::blur3x3", _blur_backward, setup_context=_blur_setup)
torch.library.register_autograd("ganconv::blur3x3_masked", _blur_masked_backward, setup_context=_blur_masked_setup)


# ---- InstanceNorm (+ LeakyReLU), with explicit first- and second-order backward kernels --------------------------
def _in_args(x, *same):
    """(N, H*W, C, dtype code) of an NHWC fp32 / bf16 tensor; the other tensors must share its dtype."""
    for t in same:
        if t.dtype != x.dtype:
            raise RuntimeError(f"ganconv: norm kernels take tensors of one dtype (got {x.dtype} and {t.dtype})")
    N, C, H, W = x.shape
    return N, H * W, C, _dtype_code(x)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


@torch.library.custom_op("ganconv::instnorm_act", mutates_args=())
def instnorm_act(x: Tensor, eps: float, act: int, slope: float) -> Tensor:
    """act((x - mean) * rsqrt(var + eps)) per (sample, channel); act = GC_ACT_NONE or GC_ACT_LRELU."""
    _check_cuda(x)
    x = _nhwc(x)
    N, HW, C, dt = _in_args(x)
    z = torch.empty_like(x, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_instnorm_act_fwd(_ptr(x), _ptr(z), _optr(_bf16_twin(z)), N, HW, C, dt, eps, act, slope, _stream()),
               "gc_instnorm_act_fwd")
    return z


@instnorm_act.register_fake
def _(x, eps, act, slope):
    return torch.empty_like(x)


@torch.library.custom_op("ganconv::instnorm_act_bwd", mutates_args=())
def instnorm_act_bwd(x: Tensor, gz: Tensor, eps: float, act: int, slope: float) -> Tensor:
    _check_cuda(x, gz)
    x, gz = _nhwc(x), _nhwc(gz)
    N, HW, C, dt = _in_args(x, gz)
    gx = torch.empty_like(x, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_instnorm_act_bwd(_ptr(x), _ptr(gz), _ptr(gx), _optr(_bf16_twin(gx)), N, HW, C, dt, eps, act, slope, _stream()),
               "gc_instnorm_act_bwd")
    return gx


@instnorm_act_bwd.register_fake
def _(x, gz, eps, act, slope):
    return torch.empty_like(x)


@torch.library.custom_op("ganconv::instnorm_act_bwd_bwd", mutates_args=())
def instnorm_act_bwd_bwd(x: Tensor, gz: Tensor, u: Tensor, eps: float, act: int, slope: float) -> tuple[Tensor, Tensor]:
    _check_cuda(x, gz, u)
    x, gz, u = _nhwc(x), _nhwc(gz), _nhwc(u)
    N, HW, C, dt = _in_args(x, gz, u)
    g_x = torch.empty_like(x, memory_format=torch.channels_last)
    g_gz = torch.empty_like(x, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_instnorm_act_bwd_bwd(_ptr(x), _ptr(gz), _ptr(u), _ptr(g_x), _ptr(g_gz), _optr(_bf16_twin(g_x)),
                                                  _optr(_bf16_twin(g_gz)), N, HW, C, dt, eps, act, slope, _stream()),
               "gc_instnorm_act_bwd_bwd")
    return g_x, g_gz


@instnorm_act_bwd_bwd.register_fake
def _(x, gz, u, eps, act, slope):
    return torch.empty_like(x), torch.empty_like(x)


def _in_setup(ctx, inputs, output):
    x, eps, act, slope = inputs
    ctx.save_for_backward(x)
    ctx.cfg = (eps, act, slope)


def _in_backward(ctx, gz):
    (x,) = ctx.saved_tensors
    return instnorm_act_bwd(x, gz, *ctx.cfg), None, None, None


torch.library.register_autograd("ganconv::instnorm_act", _in_backward, setup_context=_in_setup)


def _in_bwd_setup(ctx, inputs, output):
    x, gz, eps, act, slope = inputs
    ctx.save_for_backward(x, gz)
    ctx.cfg = (eps, act, slope)


def _in_bwd_backward(ctx, u):
    x, gz = ctx.saved_tensors
    g_x, g_gz = instnorm_act_bwd_bwd(x, gz, u, *ctx.cfg)
    return g_x, g_gz, None, None, None


torch.library.register_autograd("ganconv::instnorm_act_bwd", _in_bwd_backward, setup_context=_in_bwd_setup)


# ---- BatchNorm (+ ReLU / LeakyReLU), first-order backward --------------------------------------------------------
def _cond_affine(weight, bias, N, C):
    if bias is None or tuple(weight.shape) != (N, C) or tuple(bias.shape) != (N, C):
        raise RuntimeError(f"ganconv: conditional batchnorm takes gain and bias of shape [N, C] = [{N}, {C}]")
    return weight.float().contiguous(), bias.float().contiguous()


def _bn_forward_launch(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, training, eps, act, slope):
    """One fused forward: (y, mean, rstd, unbiased variance).  Training mode with running buffers given: the finalize
    kernel also folds the batch statistics into them (momentum) and counts the batch."""
    _check_cuda(x, weight, bias, running_mean, running_var)
    x = _nhwc(x)
    N, C, H, W = x.shape
    P = N * H * W
    dt = _dtype_code(x)
    y = torch.empty_like(x, memory_format=torch.channels_last)
    mean, rstd, var = (torch.empty(C, dtype=torch.float32, device=x.device) for _ in range(3))
    L = _lib.lib()
    if weight is not None and weight.dim() == 2:     # conditional: gain and bias per sample and channel
        weight, bias = _cond_affine(weight, bias, N, C)
        ws = torch.empty(max(L.gc_cond_batchnorm_workspace_bytes(N, H * W, C), 16), dtype=torch.uint8, device=x.device)
        with _lock:
            ent = _row_cache.pop((x.data_ptr(), tuple(x.shape)), None)
        if ent is not None and training and ent[0].dtype == x.dtype and ent[0]._version == x._version:
            _lib.check(L.gc_cond_batchnorm_act_fwd_rows(_ptr(x), _ptr(y), _ptr(weight), _ptr(bias), _optr(running_mean), _optr(running_var),
                                                        _optr(num_batches_tracked), _ptr(mean), _ptr(rstd), _ptr(var), _ptr(ent[1]), ent[2],
                                                        _ptr(ws), ws.numel(), N, H * W, C, dt, eps, momentum, act, slope, _stream()),
                       "gc_cond_batchnorm_act_fwd_rows")
            return y, mean, rstd, var
        _lib.check(L.gc_cond_batchnorm_act_fwd(_ptr(x), _ptr(y), _ptr(weight), _ptr(bias), _optr(running_mean), _optr(running_var),
                                               _optr(num_batches_tracked), _ptr(mean), _ptr(rstd), _ptr(var), _ptr(ws), ws.numel(), N,
                                               H * W, C, dt, eps, momentum, act, slope, 1 if training else 0, _stream()),
                   "gc_cond_batchnorm_act_fwd")
        return y, mean, rstd, var
    ws = torch.empty(max(L.gc_batchnorm_workspace_bytes(P, C), 16), dtype=torch.uint8, device=x.device)
    with _lock:
        ent = _row_cache.pop((x.data_ptr(), tuple(x.shape)), None)
    if ent is not None and training and ent[0].dtype == x.dtype and ent[0]._version == x._version:
        _lib.check(L.gc_batchnorm_act_fwd_rows(_ptr(x), _ptr(y), _optr(_bf16_twin(y)), _optr(weight), _optr(bias), _optr(running_mean),
                                               _optr(running_var), _optr(num_batches_tracked), _ptr(mean), _ptr(rstd), _ptr(var),
                                               _ptr(ent[1]), ent[2], _ptr(ws), ws.numel(), P, C, dt, eps, momentum, act, slope, _stream()),
                   "gc_batchnorm_act_fwd_rows")
        return y, mean, rstd, var
    _lib.check(L.gc_batchnorm_act_fwd(_ptr(x), _ptr(y), _optr(_bf16_twin(y)), _optr(weight), _optr(bias), _optr(running_mean),
                                      _optr(running_var), _optr(num_batches_tracked), _ptr(mean), _ptr(rstd), _ptr(var), _ptr(ws),
                                      ws.numel(), P, C, dt, eps, momentum, act, slope, 1 if training else 0, _stream()),
               "gc_batchnorm_act_fwd")
    return y, mean, rstd, var


@torch.library.custom_op("ganconv::batchnorm_act", mutates_args=())
def batchnorm_act(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], running_mean: Optional[Tensor],
                  running_var: Optional[Tensor], training: bool, eps: float, act: int, slope: float) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    """act(batch_norm(x)) with torch.nn.BatchNorm2d semantics.  training: batch statistics; otherwise the running
    ones (read only).  Returns (y, mean, rstd, unbiased variance); the running statistics are never written here (a
    caller folds the last two into them with batchnorm_update_running_, or uses batchnorm_act_train, which does it
    inside the same launches)."""
    rm, rv = (None, None) if training else (running_mean, running_var)
    return _bn_forward_launch(x, weight, bias, rm, rv, None, 0.0, training, eps, act, slope)


@batchnorm_act.register_fake
def _(x, weight, bias, running_mean, running_var, training, eps, act, slope):
    C = x.shape[1]
    f = dict(dtype=torch.float32)
    return torch.empty_like(x), x.new_empty(C, **f), x.new_empty(C, **f), x.new_empty(C, **f)


@torch.library.custom_op("ganconv::batchnorm_update_running_", mutates_args=("running_mean", "running_var", "num_batches_tracked"))
def batchnorm_update_running_(running_mean: Tensor, running_var: Tensor, num_batches_tracked: Optional[Tensor], mean: Tensor,
                              var_unbiased: Tensor, momentum: float) -> None:
    _check_cuda(running_mean, running_var, num_batches_tracked, mean, var_unbiased)
    _lib.check(_lib.lib().gc_batchnorm_update_running(_ptr(running_mean), _ptr(running_var), _optr(num_batches_tracked), _ptr(mean),
                                                      _ptr(var_unbiased), running_mean.numel(), momentum, _stream()),
               "gc_batchnorm_update_running")


@torch.library.custom_op("ganconv::batchnorm_act_bwd", mutates_args=())
def batchnorm_act_bwd(x: Tensor, gy: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], mean: Tensor, rstd: Tensor,
                      act: int, slope: float) -> tuple[Tensor, Tensor, Tensor]:
    """(gx, gweight, gbias) of a training-mode batchnorm_act."""
    _check_cuda(x, gy)
    x, gy = _nhwc(x), _nhwc(gy)
    if gy.dtype != x.dtype:
        raise RuntimeError(f"ganconv: norm kernels take tensors of one dtype (got {x.dtype} and {gy.dtype})")
    N, C, H, W = x.shape
    P = N * H * W
    dt = _dtype_code(x)
    gx = torch.empty_like(x, memory_format=torch.channels_last)
    L = _lib.lib()
    if weight is not None and weight.dim() == 2:
        weight, bias = _cond_affine(weight, bias, N, C)
        gw, gb = torch.empty_like(weight), torch.empty_like(bias)
        ws = torch.empty(max(L.gc_cond_batchnorm_workspace_bytes(N, H * W, C), 16), dtype=torch.uint8, device=x.device)
        _lib.check(L.gc_cond_batchnorm_act_bwd(_ptr(x), _ptr(gy), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(gw),
                                               _ptr(gb), _ptr(ws), ws.numel(), N, H * W, C, dt, act, slope, _stream()),
                   "gc_cond_batchnorm_act_bwd")
        return gx, gw, gb
    gw = torch.empty(C, dtype=torch.float32, device=x.device)
    gb = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(L.gc_batchnorm_workspace_bytes(P, C), 16), dtype=torch.uint8, device=x.device)
    _lib.check(L.gc_batchnorm_act_bwd(_ptr(x), _ptr(gy), _optr(weight), _optr(bias), _ptr(mean), _ptr(rstd), _ptr(gx),
                                      _optr(_bf16_twin(gx)), _ptr(gw), _ptr(gb), _ptr(ws), ws.numel(), P, C, dt, act, slope, _stream()),
               "gc_batchnorm_act_bwd")
    return gx, gw, gb


@batchnorm_act_bwd.register_fake
def _(x, gy, weight, bias, mean, rstd, act, slope):
    shape = (x.shape[0], x.shape[1]) if weight is not None and weight.dim() == 2 else (x.shape[1],)
    return torch.empty_like(x), x.new_empty(shape, dtype=torch.float32), x.new_empty(shape, dtype=torch.float32)


def _bn_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)      # no zero tensors for the statistics outputs, whose gradients are ignored
    x, weight, bias, _, _, training, _, act, slope = inputs
    if not training:
        raise RuntimeError("ganconv: batchnorm_act is differentiable in training mode only")
    _, mean, rstd, _ = output
    ctx.save_for_backward(x, weight, bias, mean, rstd)
    ctx.cfg = (act, slope)


def _bn_backward(ctx, gy, g_mean, g_rstd, g_var):
    x, weight, bias, mean, rstd = ctx.saved_tensors
    # (mean / rstd are returned for the backward pass only; gradients arriving at them are ignored)
    if gy is None:
        return (None,) * 9
    if gy.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("ganconv: batchnorm_act has no second-order backward")
    gx, gw, gb = batchnorm_act_bwd(x, gy, weight, bias, mean, rstd, *ctx.cfg)
    return gx, gw if weight is not None else None, gb if bias is not None else None, None, None, None, None, None, None


torch.library.register_autograd("ganconv::batchnorm_act", _bn_backward, setup_context=_bn_setup)


class _BatchNormActTrain(torch.autograd.Function):
    """Training-mode batchnorm_act that also updates the running statistics in place, in the launches of the forward
    pass (an operator registered through torch.library may not both mutate an argument and carry an autograd formula)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, act, slope):
        y, mean, rstd, _ = _bn_forward_launch(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, True, eps, act, slope)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.cfg = (act, slope)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        if gy.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("ganconv: batchnorm_act has no second-order backward")
        gx, gw, gb = batchnorm_act_bwd(x, gy, weight, bias, mean, rstd, *ctx.cfg)
        return gx, gw if weight is not None else None, gb if bias is not None else None, None, None, None, None, None, None, None


def batchnorm_act_train(x, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, act, slope):
    """act(batch_norm(x)) on batch statistics; running_mean / running_var / num_batches_tracked (all or none) are updated
    as torch.nn.BatchNorm2d updates them.  weight / bias: [C] parameters, per-sample [N, C] gains and shifts, or None."""
    return _BatchNormActTrain.apply(x, weight, bias, running_mean, running_var, num_batches_tracked, float(momentum), float(eps),
                                    int(act), float(slope))


# ---- BatchNorm with statistics over all ranks of a process group ---------------------------------------------------
class _SyncBatchNormAct(torch.autograd.Function):
    """act(batch_norm(x)) where x is this rank's shard of the batch and the statistics are those of the whole batch:
    the result (and the gradients) a single device would compute on the concatenated shards.  Two small all-reduces per
    direction ([2C + 1] and [2C] floats).  First order only.  Returns (y, mean, unbiased variance) of the whole batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, act, slope, group):
        import torch.distributed as dist
        _check_cuda(x, weight, bias)
        x = _nhwc(x)
        N, C, H, W = x.shape
        P = N * H * W
        dt = _dtype_code(x)
        L = _lib.lib()
        dev = x.device
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        ws = torch.empty(max(L.gc_batchnorm_workspace_bytes(max(P, 1), C), 16), dtype=torch.uint8, device=dev)
        m, v = torch.zeros(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev)
        if P > 0:
            _lib.check(L.gc_batchnorm_stats(_ptr(x), _ptr(m), _ptr(v), _ptr(ws), ws.numel(), P, C, dt, _stream()), "gc_batchnorm_stats")
        if world > 1:
            # the shard means go first, so that each rank can centre its squares on the mean of the whole batch before
            # they are added (no cancellation for channels with |mean| >> std)
            from .ddp import all_reduce_sum
            head = torch.cat([m * P, torch.full((1,), float(P), dtype=torch.float32, device=dev)])
            all_reduce_sum(head, group)
            count = head[C]
            mean = head[:C] / count
            m2 = v * max(P - 1, 0) + P * (m - mean) ** 2
            all_reduce_sum(m2, group)
        else:
            count = torch.full((), float(P), dtype=torch.float32, device=dev)
            mean, m2 = m, v * max(P - 1, 0)
        var_b = m2 / count
        var_u = m2 / torch.clamp(count - 1, min=1)
        y = torch.empty_like(x, memory_format=torch.channels_last)
        save_mean, rstd = torch.empty(C, dtype=torch.float32, device=dev), torch.empty(C, dtype=torch.float32, device=dev)
        mean, var_b = mean.contiguous(), var_b.contiguous()
        if P > 0:
            _lib.check(L.gc_batchnorm_act_fwd(_ptr(x), _ptr(y), _optr(_bf16_twin(y)), _optr(weight), _optr(bias), _ptr(mean),
                                              _ptr(var_b), None, _ptr(save_mean), _ptr(rstd), None, _ptr(ws), ws.numel(), P, C,
                                              dt, eps, 0.0, act, slope, 0, _stream()), "gc_batchnorm_act_fwd")
        else:
            save_mean.copy_(mean)
            rstd.copy_(torch.rsqrt(var_b + eps))
        ctx.save_for_backward(x, weight, bias, save_mean, rstd, count)
        ctx.cfg = (act, slope, group)
        ctx.mark_non_differentiable(mean, var_u)
        return y, mean, var_u

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        import torch.distributed as dist
        x, weight, bias, mean, rstd, count = ctx.saved_tensors
        act, slope, group = ctx.cfg
        if gy.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("ganconv: batchnorm has no second-order backward")
        gy = _nhwc(gy)
        if gy.dtype != x.dtype:
            raise RuntimeError(f"ganconv: norm kernels take tensors of one dtype (got {x.dtype} and {gy.dtype})")
        N, C, H, W = x.shape
        P = N * H * W
        dt = _dtype_code(x)
        L = _lib.lib()
        sums = torch.zeros(2, C, dtype=torch.float32, device=x.device)        # [sum g | sum g * xh]
        if P > 0:
            ws = torch.empty(max(L.gc_batchnorm_workspace_bytes(P, C), 16), dtype=torch.uint8, device=x.device)
            _lib.check(L.gc_batchnorm_act_bwd_sums(_ptr(x), _ptr(gy), _optr(weight), _optr(bias), _ptr(mean), _ptr(rstd), _ptr(sums[1]),
                                                   _ptr(sums[0]), _ptr(ws), ws.numel(), P, C, dt, act, slope, _stream()),
                       "gc_batchnorm_act_bwd_sums")
        gb, gw = sums[0].clone(), sums[1].clone()      # parameter gradients stay per rank: the gradient reducer averages them
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            from .ddp import all_reduce_sum
            all_reduce_sum(sums, group)
        means = (sums / count).contiguous()
        gx = torch.empty_like(x, memory_format=torch.channels_last)
        if P > 0:
            _lib.check(L.gc_batchnorm_act_bwd_apply(_ptr(x), _ptr(gy), _optr(weight), _optr(bias), _ptr(mean), _ptr(rstd), _ptr(means),
                                                    _ptr(gx), _optr(_bf16_twin(gx)), P, C, dt, act, slope, _stream()),
                       "gc_batchnorm_act_bwd_apply")
        return gx, (gw if weight is not None else None), (gb if bias is not None else None), None, None, None, None


def sync_batchnorm_act(x, weight, bias, eps, act, slope, process_group=None):
    """Training-mode act(batch_norm(x)) with the statistics of the batch held by all ranks of `process_group`
    (see _SyncBatchNormAct).  Returns (y, mean, unbiased variance)."""
    return _SyncBatchNormAct.apply(x, weight, bias, float(eps), int(act), float(slope), process_group)


# ---- self-attention over image positions (BigGAN/model.py:97-103) -----------------------------------------------
def _attn_args(k, q, v):
    _check_cuda(k, q, v)
    if k.dim() != 3 or q.shape != k.shape or v.dim() != 3 or v.shape[:2] != k.shape[:2]:
        raise RuntimeError(f"ganconv: attention takes k, q [B, N, d] and v [B, N, C] (got {tuple(k.shape)}, {tuple(q.shape)}, {tuple(v.shape)})")
    for t in (k, q, v):
        if t.dtype != torch.bfloat16:
            raise RuntimeError("ganconv: attention operands are bf16")
    return k.shape[0], k.shape[1], k.shape[2], v.shape[2]


@torch.library.custom_op("ganconv::attention", mutates_args=())
def attention(k: Tensor, q: Tensor, v: Tensor) -> tuple[Tensor, Tensor]:
    """softmax(k q^T) v per sample (softmax over the last index, unit scale) without the [N, N] weights in memory.
    k, q: [B, N, d] bf16 (d 8 or 16); v: [B, N, C] bf16 (C 64 or 128); N a multiple of 128.  Returns the [B, N, C] bf16
    product and the fp32 log-sum-exp of every row, which the backward pass reads."""
    B, N, d, C = _attn_args(k, q, v)
    k, q, v = k.contiguous(), q.contiguous(), v.contiguous()
    out = torch.empty_like(v)
    lse = torch.empty(B, N, dtype=torch.float32, device=v.device)
    _lib.check(_lib.lib().gc_attention_fwd(_ptr(k), _ptr(q), _ptr(v), _ptr(out), _ptr(lse), B, N, d, C, _stream()), "gc_attention_fwd")
    return out, lse


@attention.register_fake
def _(k, q, v):
    return torch.empty_like(v), v.new_empty(v.shape[0], v.shape[1], dtype=torch.float32)


@torch.library.custom_op("ganconv::attention_bwd", mutates_args=())
def attention_bwd(k: Tensor, q: Tensor, v: Tensor, out: Tensor, dout: Tensor, lse: Tensor) -> tuple[Tensor, Tensor, Tensor]:
    B, N, d, C = _attn_args(k, q, v)
    k, q, v, out, dout = k.contiguous(), q.contiguous(), v.contiguous(), out.contiguous(), dout.to(torch.bfloat16).contiguous()
    dk, dq, dv = torch.empty_like(k), torch.empty_like(q), torch.empty_like(v)
    delta = torch.empty(B, N, dtype=torch.float32, device=v.device)
    _lib.check(_lib.lib().gc_attention_bwd(_ptr(k), _ptr(q), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dk), _ptr(dq), _ptr(dv),
                                           _ptr(delta), B, N, d, C, _stream()), "gc_attention_bwd")
    return dk, dq, dv


@attention_bwd.register_fake
def _(k, q, v, out, dout, lse):
    return torch.empty_like(k), torch.empty_like(q), torch.empty_like(v)


def _attn_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)      # (no zero tensor for the log-sum-exp output, whose gradient is ignored)
    ctx.save_for_backward(*inputs, *output)


def _attn_backward(ctx, g_out, g_lse):
    k, q, v, out, lse = ctx.saved_tensors
    if g_out is None:
        return None, None, None
    if g_out.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("ganconv: attention has no second-order backward")
    return attention_bwd(k, q, v, out, g_out, lse)


torch.library.register_autograd("ganconv::attention", _attn_backward, setup_context=_attn_setup)


# ---- 2x2 pooling / 2x nearest replication (an adjoint pair; each is the other's backward) -------------------------
# ---- the tail of the self-attention block: x + gamma * (the [B, N, C] product read as [B, C, H, W]) --------------------
@torch.library.custom_op("ganconv::attention_residual", mutates_args=())
def attention_residual(x: Tensor, mixed: Tensor, gamma: Tensor) -> Tensor:
    """x + gamma * mixed.reshape(x.shape) for the attention product `mixed` [B, N, C] (contiguous) and the block input x
    [B, C, H, W] in channels-last memory: the reference reinterprets the product's buffer as [B, C, H, W] without a
    transpose (BigGAN/model.py:97-105), which is a transposition against channels-last memory; one tiled launch instead of a
    strided addcmul.  N and C multiples of 64; gamma: one float."""
    _check_cuda(x, mixed, gamma)
    B, C, H, W = x.shape
    N = H * W
    if tuple(mixed.shape) != (B, N, C) or mixed.dtype != x.dtype:
        raise RuntimeError(f"ganconv: attention_residual expects the product as [B, N, C] = [{B}, {N}, {C}] in the input's dtype")
    x, mixed = _nhwc(x), mixed.contiguous()
    gm = gamma.detach().to(torch.float32).reshape(1).contiguous()
    out = torch.empty_like(x, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_attention_residual_fwd(_ptr(x), _ptr(mixed), _ptr(gm), _ptr(out), B, N, C, _dtype_code(x), _stream()),
               "gc_attention_residual_fwd")
    return out


@attention_residual.register_fake
def _(x, mixed, gamma):
    return torch.empty_like(x, memory_format=torch.channels_last)


@torch.library.custom_op("ganconv::attention_residual_bwd", mutates_args=())
def attention_residual_bwd(g: Tensor, mixed: Tensor, gamma: Tensor) -> tuple[Tensor, Tensor]:
    """(d mixed [B, N, C], d gamma [1]) of attention_residual from the gradient g of its output."""
    _check_cuda(g, mixed, gamma)
    B, C, H, W = g.shape
    N = H * W
    g = _nhwc(g if g.dtype == mixed.dtype else g.to(mixed.dtype))
    mixed = mixed.contiguous()
    gm = gamma.detach().to(torch.float32).reshape(1).contiguous()
    dm = torch.empty_like(mixed)
    dgamma = torch.empty(1, dtype=torch.float32, device=g.device)
    L = _lib.lib()
    ws = torch.empty(max(L.gc_attention_residual_workspace_bytes(B, N, C), 16), dtype=torch.uint8, device=g.device)
    _lib.check(L.gc_attention_residual_bwd(_ptr(g), _ptr(mixed), _ptr(gm), _ptr(dm), _ptr(dgamma), _ptr(ws), ws.numel(), B, N, C,
                                           _dtype_code(g), _stream()), "gc_attention_residual_bwd")
    return dm, dgamma


@attention_residual_bwd.register_fake
def _(g, mixed, gamma):
    return torch.empty_like(mixed), mixed.new_empty(1, dtype=torch.float32)


def _attn_res_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[1], inputs[2])


def _attn_res_backward(ctx, g):
    mixed, gamma = ctx.saved_tensors
    dm, dgamma = attention_residual_bwd(g, mixed, gamma)
    return g, dm, dgamma.to(gamma.dtype).reshape(gamma.shape)


torch.library.register_autograd("ganconv::attention_residual", _attn_res_backward, setup_context=_attn_res_setup)


def _resample(fn_name, x, scale, H, W, shape):
    _check_cuda(x)
    x = _nhwc(x)
    N, C = x.shape[0], x.shape[1]
    y = torch.empty(shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(getattr(_lib.lib(), fn_name)(_ptr(x), _ptr(y), N, H, W, C, _dtype_code(x), scale, _stream()), fn_name)
    return y


@torch.library.custom_op("ganconv::pool2x2", mutates_args=())
def pool2x2(x: Tensor, scale: float) -> Tensor:
    """scale * (sum over 2x2 blocks); H and W even, C a multiple of 8.  scale 0.25: AvgPool2d(2)."""
    N, C, H, W = x.shape
    if H % 2 or W % 2:
        raise RuntimeError("ganconv: pool2x2 needs even extents")
    return _resample("gc_pool2x2", x, scale, H // 2, W // 2, (N, C, H // 2, W // 2))


@pool2x2.register_fake
def _(x, scale):
    N, C, H, W = x.shape
    return x.new_empty(N, C, H // 2, W // 2)


@torch.library.custom_op("ganconv::replicate2x", mutates_args=())
def replicate2x(x: Tensor, scale: float) -> Tensor:
    """scale * (every pixel repeated 2x2); C a multiple of 8.  scale 1: nearest Upsample(scale_factor=2)."""
    N, C, H, W = x.shape
    return _resample("gc_replicate2x", x, scale, H, W, (N, C, 2 * H, 2 * W))


@replicate2x.register_fake
def _(x, scale):
    N, C, H, W = x.shape
    return x.new_empty(N, C, 2 * H, 2 * W)


def _resample_add(fn_name, x, addend, scale, H, W):
    _check_cuda(x, addend)
    x, addend = _nhwc(x), _nhwc(addend)
    if addend.dtype != x.dtype:
        raise RuntimeError("ganconv: the addend must have the input's dtype")
    N, C = x.shape[0], x.shape[1]
    y = torch.empty_like(addend, memory_format=torch.channels_last)
    _lib.check(getattr(_lib.lib(), fn_name)(_ptr(x), _ptr(addend), _ptr(y), N, H, W, C, _dtype_code(x), scale, _stream()), fn_name)
    return y


@torch.library.custom_op("ganconv::pool2x2_add", mutates_args=())
def pool2x2_add(x: Tensor, addend: Tensor, scale: float) -> Tensor:
    """scale * (sum over 2x2 blocks of x) + addend in one sweep (a residual block's sum behind its pooling)."""
    N, C, H, W = x.shape
    if H % 2 or W % 2 or tuple(addend.shape) != (N, C, H // 2, W // 2):
        raise RuntimeError("ganconv: pool2x2_add needs even extents and an addend of the pooled shape")
    return _resample_add("gc_pool2x2_add", x, addend, scale, H // 2, W // 2)


@pool2x2_add.register_fake
def _(x, addend, scale):
    return torch.empty_like(addend)


@torch.library.custom_op("ganconv::replicate2x_add", mutates_args=())
def replicate2x_add(x: Tensor, addend: Tensor, scale: float) -> Tensor:
    """scale * (every pixel of x repeated 2x2) + addend in one sweep (a residual block's sum behind its up-sampled shortcut)."""
    N, C, H, W = x.shape
    if tuple(addend.shape) != (N, C, 2 * H, 2 * W):
        raise RuntimeError("ganconv: replicate2x_add needs an addend of the up-sampled shape")
    return _resample_add("gc_replicate2x_add", x, addend, scale, H, W)


@replicate2x_add.register_fake
def _(x, addend, scale):
    return torch.empty_like(addend)


def _scale_ctx(ctx, inputs, output):
    ctx.scale = inputs[1]


def _scale_add_ctx(ctx, inputs, output):
    ctx.scale = inputs[2]


torch.library.register_autograd("ganconv::pool2x2_add", lambda ctx, g: (replicate2x(g, ctx.scale), g, None), setup_context=_scale_add_ctx)
torch.library.register_autograd("ganconv::replicate2x_add", lambda ctx, g: (pool2x2(g, ctx.scale), g, None), setup_context=_scale_add_ctx)


torch.library.register_autograd("ganconv::pool2x2", lambda ctx, g: (replicate2x(g, ctx.scale), None), setup_context=_scale_ctx)
torch.library.register_autograd("ganconv::replicate2x", lambda ctx, g: (pool2x2(g, ctx.scale), None), setup_context=_scale_ctx)


# ---- weight-space prologue of the up-sampling convolutions (StyleGAN/model.py:396-402, BigGAN/model.py:116-120) ------
def _fold_launch(src, dst, n0, n1, s0, s1, adjoint, fold):
    _lib.check(_lib.lib().gc_fused_upsample_weight(_ptr(src), _ptr(dst), n0, n1, s0, s1, 1.0, 1 if adjoint else 0, fold, _stream()),
               "gc_fused_upsample_weight")


@torch.library.custom_op("ganconv::upsample_weight_fold", mutates_args=())
def upsample_weight_fold(w: Tensor, fold: int, swap: bool) -> Tensor:
    """The [A, B, 4, 4] filter of the stride-2 transposed convolution that equals "nearest x2, then the 3 x 3 convolution w":
    fold 0 = the four-shift average of StyleGAN's FusedUpsample, fold 1 = the exact fold of the replication.  swap=False:
    w is [A, B, 3, 3] (a transposed convolution's layout); swap=True: w is [B, A, 3, 3] (a convolution's [Cout, Cin, 3, 3],
    read as its transpose).  One launch; fp32."""
    _check_cuda(w)
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
        raise RuntimeError(f"ganconv: upsample_weight_fold takes a [., ., 3, 3] weight, got {tuple(w.shape)}")
    w = w.to(torch.float32).contiguous()
    n0, n1 = (w.shape[1], w.shape[0]) if swap else (w.shape[0], w.shape[1])
    out = torch.empty(n0, n1, 4, 4, dtype=torch.float32, device=w.device)
    _fold_launch(w, out, n0, n1, 9 if swap else n1 * 9, n0 * 9 if swap else 9, False, fold)
    return out


@upsample_weight_fold.register_fake
def _(w, fold, swap):
    n0, n1 = (w.shape[1], w.shape[0]) if swap else (w.shape[0], w.shape[1])
    return w.new_empty(n0, n1, 4, 4, dtype=torch.float32)


@torch.library.custom_op("ganconv::upsample_weight_unfold", mutates_args=())
def upsample_weight_unfold(g: Tensor, fold: int, swap: bool) -> Tensor:
    """The adjoint of upsample_weight_fold: a [A, B, 4, 4] gradient back to the 3 x 3 weight's layout."""
    _check_cuda(g)
    g = g.to(torch.float32).contiguous()
    n0, n1 = g.shape[0], g.shape[1]
    out = torch.empty((n1, n0, 3, 3) if swap else (n0, n1, 3, 3), dtype=torch.float32, device=g.device)
    _fold_launch(g, out, n0, n1, 9 if swap else n1 * 9, n0 * 9 if swap else 9, True, fold)
    return out


@upsample_weight_unfold.register_fake
def _(g, fold, swap):
    n0, n1 = g.shape[0], g.shape[1]
    return g.new_empty((n1, n0, 3, 3) if swap else (n0, n1, 3, 3), dtype=torch.float32)


def _fold_ctx(ctx, inputs, output):
    ctx.cfg = (inputs[1], inputs[2], inputs[0].dtype)


torch.library.register_autograd("ganconv::upsample_weight_fold",
                                lambda ctx, g: (upsample_weight_unfold(g, ctx.cfg[0], ctx.cfg[1]).to(ctx.cfg[2]), None, None),
                                setup_context=_fold_ctx)
torch.library.register_autograd("ganconv::upsample_weight_unfold",
                                lambda ctx, g: (upsample_weight_fold(g, ctx.cfg[0], ctx.cfg[1]).to(ctx.cfg[2]), None, None),
                                setup_context=_fold_ctx)


# ---- per-sample channel scaling and its adjoint reduction: the halves of a modulated convolution (an extension: the
# reference's StyleGAN is version 1, StyleGAN/model.py:179-261, and has none) ------------------------------------------
@torch.library.custom_op("ganconv::channel_scale", mutates_args=())
def channel_scale(x: Tensor, s: Tensor) -> Tensor:
    """x[n, c, h, w] * s[n, c]; x fp32 or bf16 with C a multiple of 4, s float [N, C]."""
    _check_cuda(x, s)
    N, C, H, W = x.shape
    if s.shape != (N, C):
        raise RuntimeError("ganconv: channel_scale expects s of shape [N, C]")
    x = _nhwc(x)
    s = s.to(torch.float32).contiguous()
    y = torch.empty_like(x, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_channel_scale(_ptr(x), _ptr(s), _ptr(y), N, H * W, C, _dtype_code(x), _stream()), "gc_channel_scale")
    return y


@channel_scale.register_fake
def _(x, s):
    return torch.empty_like(x, memory_format=torch.channels_last)


@torch.library.custom_op("ganconv::rowdot", mutates_args=())
def rowdot(a: Tensor, b: Tensor) -> Tensor:
    """float [N, C]: sum over h, w of a[n, c, h, w] * b[n, c, h, w] (fixed summation order); a and b of one dtype."""
    _check_cuda(a, b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise RuntimeError("ganconv: rowdot expects two tensors of one shape and dtype")
    N, C, H, W = a.shape
    a, b = _nhwc(a), _nhwc(b)
    out = torch.empty(N, C, dtype=torch.float32, device=a.device)
    L = _lib.lib()
    ws = torch.empty(max(L.gc_rowdot_workspace_bytes(N, H * W, C), 16), dtype=torch.uint8, device=a.device)
    _lib.check(L.gc_rowdot(_ptr(a), _ptr(b), _ptr(out), _ptr(ws), ws.numel(), N, H * W, C, _dtype_code(a), _stream()), "gc_rowdot")
    return out


@rowdot.register_fake
def _(a, b):
    return a.new_empty(a.shape[0], a.shape[1], dtype=torch.float32)


def _channel_scale_backward(ctx, g):
    x, s = ctx.saved_tensors
    gx = channel_scale(g, s) if ctx.needs_input_grad[0] else None
    gs = rowdot(g, x).to(s.dtype) if ctx.needs_input_grad[1] else None
    return gx, gs


def _rowdot_backward(ctx, g):
    a, b = ctx.saved_tensors
    return (channel_scale(b, g) if ctx.needs_input_grad[0] else None,
            channel_scale(a, g) if ctx.needs_input_grad[1] else None)


torch.library.register_autograd("ganconv::channel_scale", _channel_scale_backward,
                                setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1]))
torch.library.register_autograd("ganconv::rowdot", _rowdot_backward,
                                setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0], inputs[1]))


# ---- the fused form: both per-sample scales in the convolution's output stage ---------------------------------------
_MOD_SLOPES = {_lib.GC_ACT_NONE: 1.0, _lib.GC_ACT_RELU: 0.0}


@torch.library.custom_op("ganconv::modconv_fprop", mutates_args=())
def modconv_fprop(x: Tensor, w: Tensor, bias: Optional[Tensor], pre: Optional[Tensor], post: Optional[Tensor], stride: int, padding: int,
                  precision: int, wkey: int, act: int = 0, slope: float = 0.0, out_dtype: int = -1, alpha: float = 1.0) -> Tensor:
    """post[n, o] * act(pre[n, o] * conv2d(x, alpha * w)[n, o] + bias[o]) as ONE launch: pre (the demodulation factor) and post
    (the style of the modulated convolution that reads this output next) are float [N, Cout] tables applied in the kernel's
    output stage.  act: none, ReLU or LeakyReLU.  An extension: the reference has no modulated convolution
    (StyleGAN/model.py:179-261 is version 1)."""
    _check_cuda(x, w, bias, pre, post)
    N, Cin, H, W = x.shape
    Cout, Cin_w, kh, kw = w.shape
    if Cin_w != Cin:
        raise RuntimeError(f"ganconv: weight expects {Cin_w} input channels, input has {Cin}")
    if act not in (_lib.GC_ACT_NONE, _lib.GC_ACT_RELU, _lib.GC_ACT_LRELU):
        raise RuntimeError("ganconv: the fused modulated convolution takes no activation, ReLU or LeakyReLU")
    if Cout % 4:
        raise RuntimeError("ganconv: the fused modulated convolution needs a multiple of 4 output channels")
    for t in (pre, post):
        if t is not None and t.shape != (N, Cout):
            raise RuntimeError(f"ganconv: per-sample output scales must be [N, Cout] = [{N}, {Cout}], got {tuple(t.shape)}")
    Ho, Wo = (H + 2 * padding - kh) // stride + 1, (W + 2 * padding - kw) // stride + 1
    x = _nhwc(x)
    y = torch.empty((N, Cout, Ho, Wo), dtype=_out_dtype(x, out_dtype), device=x.device, memory_format=torch.channels_last)
    if _dma_ok(x, Cin, Cout, precision):
        x = _to_bf16(x)
    d = _desc(N, H, W, Cin, Ho, Wo, Cout, kh, kw, stride, padding, _dtype_code(x), precision, _dtype_code(y))
    wp = _packed(d, w, _lib.GC_PASS_FPROP, wkey, alpha)
    ep = Epilogue()
    ep.act, ep.slope = act, slope
    keep = [t if t is None else t.to(torch.float32).contiguous() for t in (bias, pre, post)]
    if keep[0] is not None:
        ep.bias = keep[0].data_ptr()
    if keep[1] is not None:
        ep.sample_scale = keep[1].data_ptr()
    if keep[2] is not None:
        ep.sample_scale_out = keep[2].data_ptr()
    _lib.check(_lib.lib().gc_conv2d_fprop(ctypes.byref(d), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                          ctypes.c_void_p(y.data_ptr()), ctypes.byref(ep), _stream()), "gc_conv2d_fprop")
    return y


@modconv_fprop.register_fake
def _(x, w, bias, pre, post, stride, padding, precision, wkey, act=0, slope=0.0, out_dtype=-1, alpha=1.0):
    N, _, H, W = x.shape
    Cout, _, kh, kw = w.shape
    return x.new_empty((N, Cout, (H + 2 * padding - kh) // stride + 1, (W + 2 * padding - kw) // stride + 1),
                       dtype=_out_dtype(x, out_dtype))


@torch.library.custom_op("ganconv::mod_act_backward", mutates_args=())
def mod_act_backward(g: Tensor, out: Tensor, pre: Optional[Tensor], post: Optional[Tensor], slope: float) -> tuple[Tensor, Tensor, Tensor]:
    """The backward sweep of modconv_fprop's output stage from the incoming gradient g and the stored output (slope: 1 for no
    activation, 0 for ReLU):  (gc, dot, sum) with gc = pre * m * post * g the gradient of the raw convolution output
    (m = 1 where post * out > 0, else slope), dot[n, c] = sum_p g * out and sum[n, c] = sum_p m * post * g."""
    _check_cuda(g, out, pre, post)
    N, C, H, W = out.shape
    if g.shape != out.shape:
        raise RuntimeError("ganconv: gradient and output shapes differ")
    out = _nhwc(out)
    g = _nhwc(g if g.dtype == out.dtype else g.to(out.dtype))
    pre = None if pre is None else pre.to(torch.float32).contiguous()
    post = None if post is None else post.to(torch.float32).contiguous()
    gc = torch.empty_like(out, memory_format=torch.channels_last)
    dot = torch.empty(N, C, dtype=torch.float32, device=out.device)
    sm = torch.empty(N, C, dtype=torch.float32, device=out.device)
    L = _lib.lib()
    ws = torch.empty(max(L.gc_modulated_act_backward_workspace_bytes(N, H * W, C), 16), dtype=torch.uint8, device=out.device)
    _lib.check(L.gc_modulated_act_backward(_ptr(g), _ptr(out), None if pre is None else _ptr(pre), None if post is None else _ptr(post),
                                           _ptr(gc), _ptr(dot), _ptr(sm), _ptr(ws), ws.numel(), N, H * W, C, _dtype_code(out), slope,
                                           _stream()), "gc_modulated_act_backward")
    return gc, dot, sm


@mod_act_backward.register_fake
def _(g, out, pre, post, slope):
    N, C = out.shape[0], out.shape[1]
    return (torch.empty_like(out, memory_format=torch.channels_last), out.new_empty(N, C, dtype=torch.float32),
            out.new_empty(N, C, dtype=torch.float32))


def _modconv_setup(ctx, inputs, output):
    x, w, bias, pre, post, stride, padding, precision, wkey, act, slope, _, alpha = inputs
    ctx.edges = _edges(x, w, bias)
    ctx.save_for_backward(x, w, output, bias, pre, post)
    ctx.cfg = (stride, padding, precision, wkey, _MOD_SLOPES.get(act, slope), alpha)


def _modconv_backward(ctx, g):
    """out = post * z, z = act(u), u = pre * c + bias, c = conv(x, w).  Only out is kept: for these activations
    act'(u) * u = z, hence sum_p (dL/du) * u = sum_p g * out, which gives dL/dpre = sum_p (dL/du) * c without c."""
    stride, padding, precision, wkey, slope, alpha = ctx.cfg
    x, w, out, bias, pre, post = ctx.saved_tensors
    gc, dot, sm = mod_act_backward(g, out, pre, post, slope)
    need_x, need_w = _wanted(ctx, 0), _wanted(ctx, 1)
    gx, gw, _ = _input_weight_grads(x, w, gc, stride, padding, precision, wkey, alpha, need_x, need_w)
    gb = gpre = gpost = None
    if bias is not None and ctx.needs_input_grad[2]:
        gb = sm.sum(0).to(bias.dtype)
    if pre is not None and ctx.needs_input_grad[3]:
        gpre = ((dot if bias is None else dot - bias.to(torch.float32) * sm) / pre.to(torch.float32)).to(pre.dtype)
    if post is not None and ctx.needs_input_grad[4]:
        pf = post.to(torch.float32)
        gpost = torch.where(pf != 0, dot / torch.where(pf != 0, pf, torch.ones_like(pf)), torch.zeros_like(pf)).to(post.dtype)
    return gx, gw, gb, gpre, gpost, None, None, None, None, None, None, None, None


torch.library.register_autograd("ganconv::modconv_fprop", _modconv_backward, setup_context=_modconv_setup)


def _mod_bwd_setup(ctx, inputs, output):
    g, out, pre, post, slope = inputs
    ctx.save_for_backward(g, out, pre, post)
    ctx.slope = slope


def _mod_bwd_backward(ctx, v_gc, v_dot, v_sum):
    """Second order (a penalty on gradients taken through a modulated convolution): the sweep is linear in g for a fixed
    activation pattern; written with differentiable torch operators."""
    g, out, pre, post = ctx.saved_tensors
    f = torch.float32
    gf, of = g.to(f), out.to(f)
    one = torch.ones((), dtype=f, device=out.device)
    pr = (one if pre is None else pre.to(f)[:, :, None, None])
    po = (one if post is None else post.to(f)[:, :, None, None])
    m = torch.where(po * of > 0, one, one * ctx.slope)
    vg = torch.zeros_like(of) if v_gc is None else v_gc.to(f)
    vd = None if v_dot is None else v_dot.to(f)[:, :, None, None]
    vs = None if v_sum is None else v_sum.to(f)[:, :, None, None]
    inner = pr * vg if vs is None else pr * vg + vs            # cotangent of m * post * g
    g_g = po * m * inner
    if vd is not None:
        g_g = g_g + of * vd
    g_out = None if vd is None else (gf * vd).to(out.dtype)
    g_pre = None if pre is None else (po * m * gf * vg).sum((2, 3)).to(pre.dtype)
    g_post = None if post is None else (m * gf * inner).sum((2, 3)).to(post.dtype)
    return g_g.to(g.dtype), g_out, g_pre, g_post, None


torch.library.register_autograd("ganconv::mod_act_backward", _mod_bwd_backward, setup_context=_mod_bwd_setup)


# ---- channel concatenation fused with the activation behind it (U-Net skip connections, Pix2Pix/models.py:128-145) ----
@torch.library.custom_op("ganconv::act_concat", mutates_args=())
def act_concat(a: Tensor, b: Tensor, negative_slope: float) -> Tensor:
    """leaky_relu(torch.cat((a, b), dim=1), negative_slope) as one launch (0 = ReLU, 1 = plain concatenation); a and b
    of one dtype (fp32 / bf16) and one N, H, W, channels multiples of 4 / 8."""
    _check_cuda(a, b)
    if a.dtype != b.dtype or a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
        raise RuntimeError("ganconv: act_concat expects two tensors of one dtype, batch and image size")
    N, Ca, H, W = a.shape
    Cb = b.shape[1]
    a, b = _nhwc(a), _nhwc(b)
    y = torch.empty((N, Ca + Cb, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_act_concat_fwd(_ptr(a), _ptr(b), _ptr(y), N * H * W, Ca, Cb, _dtype_code(a), float(negative_slope), _stream()),
               "gc_act_concat_fwd")
    return y


@act_concat.register_fake
def _(a, b, negative_slope):
    return a.new_empty((a.shape[0], a.shape[1] + b.shape[1], a.shape[2], a.shape[3]), memory_format=torch.channels_last)


@torch.library.custom_op("ganconv::act_concat_backward", mutates_args=())
def act_concat_backward(g: Tensor, y: Tensor, ca: int, negative_slope: float) -> tuple[Tensor, Tensor]:
    """The two input gradients of act_concat from the output gradient g and the forward output y (first-order only)."""
    _check_cuda(g, y)
    N, C, H, W = y.shape
    if g.shape != y.shape:
        raise RuntimeError("ganconv: act_concat_backward expects a gradient of the output's shape")
    y = _nhwc(y)
    g = _nhwc(g.to(y.dtype))
    ga = torch.empty((N, ca, H, W), dtype=y.dtype, device=y.device, memory_format=torch.channels_last)
    gb = torch.empty((N, C - ca, H, W), dtype=y.dtype, device=y.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().gc_act_concat_bwd(_ptr(g), _ptr(y), _ptr(ga), _ptr(gb), N * H * W, ca, C - ca, _dtype_code(y), float(negative_slope),
                                            _stream()), "gc_act_concat_bwd")
    return ga, gb


@act_concat_backward.register_fake
def _(g, y, ca, negative_slope):
    N, C, H, W = y.shape
    return (y.new_empty((N, ca, H, W), memory_format=torch.channels_last), y.new_empty((N, C - ca, H, W), memory_format=torch.channels_last))


def _act_concat_setup(ctx, inputs, output):
    ctx.ca, ctx.slope = inputs[0].shape[1], inputs[2]
    ctx.save_for_backward(output)


def _act_concat_backward(ctx, g):
    (y,) = ctx.saved_tensors
    ga, gb = act_concat_backward(g, y, ctx.ca, ctx.slope)
    return ga, gb, None


torch.library.register_autograd("ganconv::act_concat", _act_concat_backward, setup_context=_act_concat_setup)


# ---- reflection padding (Pix2Pix/models.py:162,185,204,210) and its adjoint, each other's gradient ---------------
def _reflect(fn_name, x, pad, H, W, shape):
    _check_cuda(x)
    x = _nhwc(x)
    y = torch.empty(shape, dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(getattr(_lib.lib(), fn_name)(_ptr(x), _ptr(y), x.shape[0], H, W, x.shape[1], pad, _dtype_code(x), _stream()), fn_name)
    return y


@torch.library.custom_op("ganconv::reflect_pad", mutates_args=())
def reflect_pad(x: Tensor, pad: int) -> Tensor:
    """torch.nn.ReflectionPad2d(pad)(x); C a multiple of 8, pad smaller than both extents."""
    N, C, H, W = x.shape
    return _reflect("gc_reflect_pad", x, pad, H, W, (N, C, H + 2 * pad, W + 2 * pad))


@reflect_pad.register_fake
def _(x, pad):
    N, C, H, W = x.shape
    return x.new_empty(N, C, H + 2 * pad, W + 2 * pad)


@torch.library.custom_op("ganconv::reflect_pad_adjoint", mutates_args=())
def reflect_pad_adjoint(gy: Tensor, pad: int) -> Tensor:
    """The gradient of reflect_pad: every unpadded pixel collects the padded positions that read it."""
    N, C, HP, WP = gy.shape
    return _reflect("gc_reflect_pad_adjoint", gy, pad, HP - 2 * pad, WP - 2 * pad, (N, C, HP - 2 * pad, WP - 2 * pad))


@reflect_pad_adjoint.register_fake
def _(gy, pad):
    N, C, HP, WP = gy.shape
    return gy.new_empty(N, C, HP - 2 * pad, WP - 2 * pad)


def _pad_ctx(ctx, inputs, output):
    ctx.pad = inputs[1]


torch.library.register_autograd("ganconv::reflect_pad", lambda ctx, g: (reflect_pad_adjoint(g, ctx.pad), None), setup_context=_pad_ctx)
torch.library.register_autograd("ganconv::reflect_pad_adjoint", lambda ctx, g: (reflect_pad(g, ctx.pad), None), setup_context=_pad_ctx)


# ---- StyleGAN: noise injection -> LeakyReLU -> InstanceNorm -> style affine as one op (first-order backward) -----
def _adain_args(x, noise, sf, ys, yb):
    _check_cuda(x, noise, sf, ys, yb)
    N, C, H, W = x.shape
    if noise.numel() != N * H * W or sf.numel() != C or tuple(ys.shape) != (N, C) or tuple(yb.shape) != (N, C):
        raise RuntimeError("ganconv: noise_lrelu_adain takes noise [N,1,H,W], a scale per channel and styles [N,C]")
    f = lambda t: t.float().contiguous()
    return N, C, H * W, f(noise), f(sf).reshape(-1), f(ys), f(yb)


def _adain_workspace(N, HW, C, device):
    n = _lib.lib().gc_noise_lrelu_adain_workspace_bytes(N, HW, C)
    return torch.empty(n, dtype=torch.uint8, device=device) if n > 0 else None


@torch.library.custom_op("ganconv::noise_lrelu_adain", mutates_args=())
def noise_lrelu_adain(x: Tensor, noise: Tensor, sf: Tensor, ys: Tensor, yb: Tensor, eps: float, slope: float) -> tuple[Tensor, Tensor, Tensor]:
    """ys * InstanceNorm(leaky_relu(x + sf * noise)) + yb (reference StyleGAN/model.py:249-261).  Returns (out, mean,
    rstd); the last two are [N, C] statistics kept for the backward pass."""
    x = _nhwc(x)
    N, C, HW, noise, sf, ys, yb = _adain_args(x, noise, sf, ys, yb)
    out = torch.empty_like(x, memory_format=torch.channels_last)
    mu, r = (torch.empty(N, C, dtype=torch.float32, device=x.device) for _ in range(2))
    ws = _adain_workspace(N, HW, C, x.device)
    _lib.check(_lib.lib().gc_noise_lrelu_adain_fwd(_ptr(x), _ptr(noise), _ptr(sf), _ptr(ys), _ptr(yb), _ptr(out), _ptr(mu), _ptr(r),
                                                   _optr(ws), 0 if ws is None else ws.numel(), N, HW, C, _dtype_code(x), eps, slope,
                                                   _stream()), "gc_noise_lrelu_adain_fwd")
    return out, mu, r


@noise_lrelu_adain.register_fake
def _(x, noise, sf, ys, yb, eps, slope):
    f = dict(dtype=torch.float32)
    return torch.empty_like(x), x.new_empty(x.shape[0], x.shape[1], **f), x.new_empty(x.shape[0], x.shape[1], **f)


@torch.library.custom_op("ganconv::noise_lrelu_adain_bwd", mutates_args=())
def noise_lrelu_adain_bwd(x: Tensor, noise: Tensor, sf: Tensor, ys: Tensor, yb: Tensor, mu: Tensor, r: Tensor, gout: Tensor,
                          slope: float) -> tuple[Tensor, Tensor, Tensor, Tensor]:
    x, gout = _nhwc(x), _nhwc(gout)
    if gout.dtype != x.dtype:
        gout = gout.to(x.dtype)
    N, C, HW, noise, sf, ys, yb = _adain_args(x, noise, sf, ys, yb)
    gx = torch.empty_like(x, memory_format=torch.channels_last)
    g_sf, g_ys, g_yb = (torch.empty(N, C, dtype=torch.float32, device=x.device) for _ in range(3))
    ws = _adain_workspace(N, HW, C, x.device)
    _lib.check(_lib.lib().gc_noise_lrelu_adain_bwd(_ptr(x), _ptr(noise), _ptr(sf), _ptr(ys), _ptr(yb), _ptr(mu), _ptr(r), _ptr(gout), _ptr(gx),
                                                   _ptr(g_sf), _ptr(g_ys), _ptr(g_yb), _optr(ws), 0 if ws is None else ws.numel(), N, HW, C,
                                                   _dtype_code(x), slope, _stream()), "gc_noise_lrelu_adain_bwd")
    return gx, g_sf, g_ys, g_yb


@noise_lrelu_adain_bwd.register_fake
def _(x, noise, sf, ys, yb, mu, r, gout, slope):
    f = dict(dtype=torch.float32)
    N, C = x.shape[0], x.shape[1]
    return torch.empty_like(x), x.new_empty(N, C, **f), x.new_empty(N, C, **f), x.new_empty(N, C, **f)


def _adain_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)      # (no zero tensors for the statistics outputs, whose gradients are ignored)
    x, noise, sf, ys, yb, _, slope = inputs
    ctx.save_for_backward(x, noise, sf, ys, yb, output[1], output[2])
    ctx.slope = slope


def _adain_backward(ctx, gout, g_mu, g_r):
    x, noise, sf, ys, yb, mu, r = ctx.saved_tensors
    if gout is None:
        return (None,) * 7
    if gout.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("ganconv: noise_lrelu_adain has no second-order backward")
    gx, g_sf, g_ys, g_yb = noise_lrelu_adain_bwd(x, noise, sf, ys, yb, mu, r, gout, ctx.slope)
    return gx, None, g_sf.sum(dim=0).reshape(sf.shape).to(sf.dtype), g_ys.to(ys.dtype), g_yb.to(yb.dtype), None, None


torch.library.register_autograd("ganconv::noise_lrelu_adain", _adain_backward, setup_context=_adain_setup)


# ---- the scalar parts of a WGAN-GP critic iteration (BasicGANs/loss_functions.py:7-27) ---------------------------
def _same_layout(*ts):
    a = ts[0]
    return all(t.shape == a.shape and t.stride() == a.stride() and t.dtype == a.dtype for t in ts)


def _dense(t):
    """t if its elements fill its storage range without gaps in some dimension order, else a contiguous copy."""
    return t if (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous()


@torch.library.custom_op("ganconv::wgan_mix", mutates_args=())
def wgan_mix(real: Tensor, fake: Tensor, eps: Tensor) -> Tensor:
    """real * eps + fake * (1 - eps), eps one value per sample, in one launch.  The result is a fresh tensor that takes no
    gradient from its operands (the reference mixes `.data`); mark it requires_grad to differentiate a critic by it."""
    _check_cuda(real, fake, eps)
    real, fake = _dense(real.float()), fake.float()
    if not _same_layout(real, fake):
        fake = fake.contiguous(memory_format=torch.channels_last if real.is_contiguous(memory_format=torch.channels_last) else torch.contiguous_format)
        if not _same_layout(real, fake):
            real = real.contiguous()
            fake = fake.contiguous()
    N = real.shape[0]
    if eps.numel() != N:
        raise RuntimeError(f"ganconv: eps must hold one value per sample ({N}), got {tuple(eps.shape)}")
    e = eps.detach().reshape(N).float().contiguous()
    out = torch.empty_like(real)
    _lib.check(_lib.lib().gc_wgan_mix(_ptr(real), _ptr(fake), _ptr(e), _ptr(out), N, real.numel() // N, _stream()), "gc_wgan_mix")
    return out


@wgan_mix.register_fake
def _(real, fake, eps):
    return torch.empty_like(real, dtype=torch.float32)


@torch.no_grad()
def spectral_sigma_(weight, u, n_iterations=1):
    """SpectralNorm._power_iterations (BigGAN/model.py:27-35, p = 2) on the library's kernels: advances the left vector
    `u` ([1, out] or [out], fp32) in place and returns sigma as a 0-dim tensor.  `weight`: any fp32 parameter with `out`
    as its first dimension (read as [out, rest]).  No gradient flows through it, as in the reference."""
    _check_cuda(weight)
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    rows = w.shape[0]
    cols = w.numel() // rows
    if u.dtype != torch.float32 or u.numel() != rows or not u.is_contiguous():
        raise RuntimeError(f"ganconv: the left vector must be contiguous fp32 with {rows} elements, got {tuple(u.shape)} {u.dtype}")
    L = _lib.lib()
    ws = torch.empty(L.gc_spectral_norm_workspace_bytes(rows, cols) // 4, dtype=torch.float32, device=w.device)
    sigma = torch.empty((), dtype=torch.float32, device=w.device)
    _lib.check(L.gc_spectral_norm_power_iteration(ctypes.c_void_p(w.data_ptr()), rows, cols, ctypes.c_void_p(u.data_ptr()),
                                                  ctypes.c_void_p(sigma.data_ptr()), int(n_iterations), ctypes.c_void_p(ws.data_ptr()),
                                                  ws.numel() * 4, _stream()), "gc_spectral_norm_power_iteration")
    return sigma


_SN_JOB = None
_SN_PLAN = None


def _sn_dtypes():
    global _SN_JOB, _SN_PLAN
    if _SN_JOB is None:
        import numpy as np
        _SN_JOB = np.dtype([("w", "<u8"), ("u", "<u8"), ("sigma", "<u8"), ("rows", "<i4"), ("pad", "<i4"), ("cols", "<i8")], align=True)
        _SN_PLAN = np.dtype([("w", "<u8"), ("u", "<u8"), ("sigma", "<u8"), ("cols", "<i8"), ("ws_part", "<i8"), ("ws_t", "<i8"),
                             ("ws_s", "<i8"), ("ws_scal", "<i8"), ("rows", "<i4"), ("vec", "<i4"), ("splits", "<i4"),
                             ("rows_per_split", "<i4"), ("col_blocks", "<i4"), ("col_block0", "<i4"), ("row_block0", "<i4"),
                             ("pad", "<i4")], align=True)
        assert _SN_JOB.itemsize == 40 and _SN_PLAN.itemsize == 96       # gc_spectral_job, gc_spectral_plan
    return _SN_JOB, _SN_PLAN


class SpectralNormSet:
    """The power iterations of ALL spectrally normalised weights of a network as one set of four launches per iteration
    (gc_spectral_norm_multi_run) instead of four per weight.  The reference advances each layer's U_vector inside that
    layer's forward call (BigGAN/model.py:22-25); the weights do not change inside a forward pass, so one refresh() in front
    of it gives every layer the same sigma and the same advanced vector.  weights: fp32 CUDA parameters (read as
    [out, rest]); us: their left vectors, contiguous fp32 with `out` elements each, advanced in place."""

    def __init__(self, weights, us, n_iterations=1):
        self.weights, self.us, self.n_iterations = list(weights), list(us), int(n_iterations)
        self._key = None

    @torch.no_grad()
    def refresh(self):
        """Advances every left vector and returns the sigmas as one new [n] tensor (sigma of weights[i] at index i)."""
        import numpy as np
        key = tuple(w.data_ptr() for w in self.weights) + tuple(u.data_ptr() for u in self.us)
        L = _lib.lib()
        if key != self._key:          # first call, or a parameter / buffer was re-created: plan and upload again
            job_t, plan_t = _sn_dtypes()
            n = len(self.weights)
            dev = self.weights[0].device
            self.sigma = torch.empty(n, dtype=torch.float32, device=dev)
            jobs = np.zeros(n, job_t)
            for i, (w, u) in enumerate(zip(self.weights, self.us)):
                _check_cuda(w, u)
                if w.dtype != torch.float32 or not w.is_contiguous() or u.dtype != torch.float32 or not u.is_contiguous() or u.numel() != w.shape[0]:
                    raise RuntimeError("ganconv: SpectralNormSet takes contiguous fp32 weights and left vectors of `out` elements")
                jobs[i] = (w.data_ptr(), u.data_ptr(), self.sigma.data_ptr() + 4 * i, w.shape[0], 0, w.numel() // w.shape[0])
            plan = np.zeros(n, plan_t)
            ws_bytes = ctypes.c_size_t(0)
            self._grids = (ctypes.c_int32 * 2)()
            _lib.check(L.gc_spectral_norm_multi_plan(jobs.ctypes.data_as(ctypes.c_void_p), n, plan.ctypes.data_as(ctypes.c_void_p),
                                                     ctypes.byref(ws_bytes), self._grids), "gc_spectral_norm_multi_plan")
            self._plan = torch.from_numpy(plan.view(np.uint8)).to(dev)
            self._ws = torch.empty(max(ws_bytes.value // 4, 4), dtype=torch.float32, device=dev)
            self._key = key
        _lib.check(L.gc_spectral_norm_multi_run(ctypes.c_void_p(self._plan.data_ptr()), len(self.weights), self._grids, self.n_iterations,
                                                ctypes.c_void_p(self._ws.data_ptr()), _stream()), "gc_spectral_norm_multi_run")
        # (a fresh tensor per call: a graph that divides by an earlier call's values keeps them for its backward pass)
        return self.sigma.clone()


@torch.no_grad()
def orthogonal_regularization_(parameters, beta=1e-4, precision=None):
    """BigGAN/train.py:115-127: grad += 2 beta (W W^T with its diagonal removed) W for every parameter with more than two
    dimensions that has a gradient, W = the parameter flattened to [out, rest].  Both products run on the convolution
    kernels: W in memory IS a one-sample NHWC image of `out` pixels x `rest` channels, so W W^T is a 1x1 convolution of that
    image with W as the filter bank, and (W W^T) W is the data gradient of the same 1x1 convolution for the Gram matrix as the
    output gradient -- no transposed copy of W.  `precision`: None (the mode the convolutions run in, set_precision), "fp32x3"
    (split fp32 operands, fp32-like accuracy: the reference computes this in fp32) or "bf16" (operands rounded to bf16: the
    added term, 2 beta times a product of weights, is then accurate to about 1e-2 of itself)."""
    prec = get_precision() if precision is None else _PRECISION[precision]
    f32 = _code_of(torch.float32)
    for p in parameters:
        if p.dim() <= 2 or p.grad is None:
            continue
        rows = p.shape[0]
        w2 = p.detach().reshape(rows, -1).float()
        K = w2.shape[1]
        image = w2.reshape(1, rows, 1, K).permute(0, 3, 1, 2)            # [1, K, rows, 1] over W's own memory
        bank = w2.reshape(rows, K, 1, 1)
        gram = conv_fprop(image, bank, None, 1, 0, prec, 0, 0, 0.0, f32)     # [1, rows, rows, 1]: (0, j, i, 0) = <w_i, w_j>
        gram.permute(0, 2, 3, 1).reshape(rows, rows).fill_diagonal_(0.0)
        r = conv_dgrad(gram, bank, None, rows, 1, 1, 0, prec, 0, 0, 0.0, f32)     # [1, K, rows, 1]: (0, k, i, 0) = sum_j gram[i, j] W[j, k]
        p.grad.add_(r.permute(0, 2, 3, 1).reshape(p.shape).to(p.grad.dtype), alpha=2.0 * beta)


@torch.library.custom_op("ganconv::gradient_penalty", mutates_args=())
def gradient_penalty(gradient: Tensor, c_lambda: float) -> tuple[Tensor, Tensor, Tensor]:
    """(c_lambda * mean_n (||gradient_n||_2 - 1)^2 as a 0-dim tensor, the per-sample norms, the per-sample factor of
    gradient_n in the penalty's derivative).  Only the first output is differentiable."""
    _check_cuda(gradient)
    g = _dense(gradient)
    N = g.shape[0]
    per = g.numel() // N
    L = _lib.lib()
    penalty = torch.empty((), dtype=torch.float32, device=g.device)
    norms = torch.empty(N, dtype=torch.float32, device=g.device)
    coef = torch.empty(N, dtype=torch.float32, device=g.device)
    nws = L.gc_gradient_penalty_workspace_bytes(N, per)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=g.device)
    _lib.check(L.gc_gradient_penalty_fwd(_ptr(g), _dtype_code(g), N, per, float(c_lambda), _ptr(norms), _ptr(coef), _ptr(penalty),
                                         _ptr(ws), ws.numel(), _stream()), "gc_gradient_penalty_fwd")
    return penalty, norms, coef


@gradient_penalty.register_fake
def _(gradient, c_lambda):
    N = gradient.shape[0]
    return gradient.new_empty((), dtype=torch.float32), gradient.new_empty(N, dtype=torch.float32), gradient.new_empty(N, dtype=torch.float32)


@torch.library.custom_op("ganconv::gradient_penalty_bwd", mutates_args=())
def gradient_penalty_bwd(gradient: Tensor, coef: Tensor, g_penalty: Tensor) -> Tensor:
    """g_penalty * coef[n] * gradient_n, in gradient's layout and dtype."""
    _check_cuda(gradient, coef, g_penalty)
    g = _dense(gradient)
    N = g.shape[0]
    out = torch.empty_like(g)
    _lib.check(_lib.lib().gc_gradient_penalty_bwd(_ptr(g), _dtype_code(g), N, g.numel() // N, _ptr(coef.contiguous()),
                                                  _ptr(g_penalty.float().reshape(1)), _ptr(out), _stream()), "gc_gradient_penalty_bwd")
    return out


@gradient_penalty_bwd.register_fake
def _(gradient, coef, g_penalty):
    return torch.empty_like(gradient)


def _gp_setup(ctx, inputs, output):
    ctx.set_materialize_grads(False)
    ctx.save_for_backward(inputs[0], output[2])


def _gp_backward(ctx, g_penalty, g_norms, g_coef):
    gradient, coef = ctx.saved_tensors
    if g_penalty is None:
        return None, None
    return gradient_penalty_bwd(gradient, coef, g_penalty), None


torch.library.register_autograd("ganconv::gradient_penalty", _gp_backward, setup_context=_gp_setup)


@torch.library.custom_op("ganconv::wgan_critic_loss", mutates_args=())
def wgan_critic_loss(scores: Tensor, drift: float) -> tuple[Tensor, Tensor]:
    """scores: the critic's outputs for B real samples followed by B generated ones.  Returns ([mean(real) - drift *
    mean(real^2), mean(generated)], the gradient of (second - first) with respect to the scores): what the two backward
    passes of BasicGANs/loss_functions.py:7-15 start from.  Not differentiable itself."""
    _check_cuda(scores)
    s = scores.detach().reshape(-1).float().contiguous()
    if s.numel() % 2:
        raise RuntimeError("ganconv: wgan_critic_loss expects B real scores followed by B generated scores")
    out = torch.empty(2, dtype=torch.float32, device=s.device)
    grad = torch.empty_like(s)
    _lib.check(_lib.lib().gc_wgan_critic_loss(_ptr(s), s.numel() // 2, float(drift), _ptr(out), _ptr(grad), _stream()), "gc_wgan_critic_loss")
    return out, grad.view(scores.shape)


@wgan_critic_loss.register_fake
def _(scores, drift):
    return scores.new_empty(2, dtype=torch.float32), torch.empty_like(scores, dtype=torch.float32)
