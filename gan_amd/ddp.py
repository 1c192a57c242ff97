"""Data-parallel gradient reduction: one process per GPU, parameters replicated, gradients summed over ranks with
bucketed all-reduce (RCCL over xGMI when the backend is "nccl") overlapped with the backward pass.

The reference's only multi-GPU mechanism is single-process torch.nn.DataParallel (StyleGAN/train.py:121-125,
BigGAN/train.py:150-154), which re-broadcasts every parameter on each forward and reduces gradients onto
device 0.  Here every rank keeps its parameters resident and only gradients cross the links:

  * all gradients of a network live in a few flat buckets (`param.grad` is a view into its bucket), built
    in reverse registration order because the last layers' gradients are ready first;
  * zero_grad() is one memset per bucket;
  * a GAN phase may call .backward() several times (WGAN-GP: real, fake, penalty) -- gradients accumulate
    locally and `arm()` is called before the LAST backward; from then on a post-accumulate hook counts the
    parameters of each bucket and launches that bucket's asynchronous all-reduce as soon as it is complete,
    while the rest of the backward pass is still running;
  * finish() launches whatever was not triggered (parameters without gradient in this phase), waits for the
    collectives on the current stream and averages;
  * all_reduce() / average_() are the plain form for a step whose forward/backward is replayed as a HIP graph: the
    graphs contain no collective, the buckets are reduced between two replays.

Batch-coupled layers: BatchNorm statistics stay per rank (as they do under the reference's DataParallel).
"""
import torch
import torch.distributed as dist


def all_reduce_sum(tensor, group=None, async_op=False):
    """Sum of `tensor` over the ranks, in place; returns the work handle of an asynchronous RCCL call, else None.

    RCCL orders its collectives on the GPU.  gloo (the CPU rig of the tests: several ranks on one GPU) is given host
    tensors only: its own staging of GPU tensors -- a stream of its own that waits for the current one, copies and host
    waits -- was seen not to return on ROCm 7.2 with two ranks on one GPU.  Both ranks sat inside dist.all_reduce,
    behind a stream synchronisation that had returned, so the GPU work in front of it was finished and the wait was
    gloo's own.  Here the copy to the host and back are ordinary torch copies on the current stream."""
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        host = tensor.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        tensor.copy_(host)
        return None
    return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast(tensor, src=0, group=None):
    """dist.broadcast with the same host staging for gloo as all_reduce_sum."""
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        host = tensor.cpu()
        dist.broadcast(host, src, group=group)
        tensor.copy_(host)
    else:
        dist.broadcast(tensor, src, group=group)


class GradReducer:
    def __init__(self, module, bucket_bytes=32 << 20, process_group=None, average=True):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.average = average
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        self.dtype = params[0].dtype
        for p in params:
            if p.dtype != self.dtype:
                raise ValueError("GradReducer expects all parameters of a module to share one dtype")
        self.params = params
        self.buckets = []          # flat tensors
        self.bucket_of = {}        # id(param) -> bucket index
        self._offset = {}          # id(param) -> element offset inside its bucket
        self._members = []         # number of params per bucket
        cur, cur_bytes = [], 0
        groups = []
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for bi, group in enumerate(groups):
            n = sum(p.numel() for p in group)
            flat = torch.zeros(n, dtype=self.dtype, device=group[0].device)
            off = 0
            for p in group:
                self.bucket_of[id(p)] = bi
                self._offset[id(p)] = off
                off += p.numel()
            self.buckets.append(flat)
            self._members.append(len(group))
            for p in group:
                self._reattach(p)
        self._armed = False
        self._pending = [0] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        for p in params:
            p.register_post_accumulate_grad_hook(self._hook)

    # -- public ---------------------------------------------------------------------------------------------
    def zero_grad(self):
        for flat in self.buckets:
            flat.zero_()
        for p in self.params:      # re-attach views in case something replaced .grad (e.g. zero_grad(set_to_none=True))
            if p.grad is None or p.grad.data_ptr() != self.buckets[self.bucket_of[id(p)]].data_ptr() + p.element_size() * self._offset[id(p)]:
                self._reattach(p)

    def arm(self):
        """Call right before the last backward pass of the phase."""
        self._armed = True
        self._pending = list(self._members)
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def finish(self):
        """Reduce what is left, wait (stream-ordered) and average.  After this the gradients are the mean over
        ranks of the locally accumulated gradients."""
        self._armed = False
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        for w in self._works:
            if w is not None:
                w.wait()
        # (a phase that never armed the hooks -- gradients computed with torch.autograd.grad and added by the caller --
        #  must find every bucket unlaunched again the next time)
        self._works = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        if self.average and self.world > 1:
            for flat in self.buckets:
                flat.div_(self.world)

    def all_reduce(self):
        """Blocking-in-stream-order sum of every bucket over the ranks, without the hook machinery: for callers that
        run forward/backward as a captured HIP graph and reduce between graph replays."""
        if dist.is_available() and dist.is_initialized():
            for flat in self.buckets:
                all_reduce_sum(flat, self.pg)

    def average_(self):
        if self.average and self.world > 1:
            for flat in self.buckets:
                flat.div_(self.world)

    # -- internals -------------------------------------------------------------------------------------------
    def _reattach(self, p):
        off = self._offset[id(p)]
        p.grad = self.buckets[self.bucket_of[id(p)]][off:off + p.numel()].view(p.shape)

    def _launch(self, bi):
        self._launched[bi] = True
        if self.world > 1:
            self._works[bi] = all_reduce_sum(self.buckets[bi], self.pg, async_op=True)

    def _hook(self, p):
        if not self._armed:
            return
        bi = self.bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and not self._launched[bi]:
            self._launch(bi)
