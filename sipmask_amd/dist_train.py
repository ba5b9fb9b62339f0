"""Data-parallel training plumbing for the SipMask head: bucketed gradient all-reduce over
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests) overlapped with backward, and
the reference's SGD (M/mmdet/apis/train.py:92-139, cfg optimizer :108-113) on the HIP update kernel (one fused
multi-tensor launch per step).

The reference wraps the model in MMDistributedDataParallel and lets torch DDP reduce 25 MB buckets.  Here the buckets
are sized for xGMI: it is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound and
wants FEW, LARGE messages -- the whole head is ~20 MB of f32 gradients, the whole detector ~131 MB, so the default is
one 64 MB bucket per ~16 M parameters, each launched as soon as its last gradient has been produced.
"""
import torch
import torch.distributed as dist


class GradBucketer:
    """Flat f32 gradient buckets in reverse parameter order (the order backward produces them).  The gradients LIVE in
    the buckets: every parameter's `.grad` is a view of its bucket slice for the lifetime of the bucketer (what torch
    DDP's gradient_as_bucket_view gives MMDistributedDataParallel, M/mmdet/apis/train.py:135-139), the HIP kernels that
    produce parameter gradients write into those views directly (hip_ops.GRAD_SINK), a bucket's asynchronous all-reduce
    starts when its last gradient has been produced and reduces the bucket IN PLACE, and the optimizer reads the same
    memory -- no copy into the buckets, none back, and a pointer table that never changes for HipSGD.
    Round 2 copied every gradient in (hook) and out (finish): +9 ms per 4-image step on one rank.

    Per step: zero_grad() (one memset per bucket) -> forward / backward -> finish() (wait, turn sums into means).

    Collective ORDER is fixed: bucket k is launched only after buckets 0..k-1 have been launched, on every rank.
    Which parameters receive a gradient can differ between ranks (an image batch without positives gives
    loss_mask = area.sum()*0 with no graph, so sip_cof / sip_mask_lat get no gradient on that rank only): a rank
    whose bucket never fills defers it -- and every later bucket -- to finish(), which launches the rest in index
    order, so all ranks issue the same sequence of equally sized all-reduces (no RCCL hang, no mispaired buffers).
    Parameters without a gradient contribute zeros, as DDP(find_unused_parameters=True) would."""

    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, force=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force: issue the all-reduces even in a one-rank group (exercises the RCCL path on a 1-GPU box)
        self.force = bool(force) and dist.is_initialized()
        self.buckets = []                 # dict(flat, params, views, pending, work)
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self._where = {}                  # id(param) -> (bucket index, slot): tensors must not be compared with ==
        for bi, b in enumerate(self.buckets):
            for i, p in enumerate(b["params"]):
                self._where[id(p)] = (bi, i)
        self._next = 0                    # first bucket whose all-reduce has not been launched this step
        self._seen = set()                # parameters whose gradient has been counted this step
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        if self.params and self.params[0].is_cuda:
            from . import hip_ops as H
            H.GRAD_SINK.attach({p.data_ptr(): p._sm_grad_view for p in self.params
                                if p.dtype == torch.float32 and p.is_contiguous()}, owner=self)
            # the overlap of a bucket of directly written gradients with backward relies on the post-accumulate hook firing
            # for a leaf whose backward op returned None (torch >= 2.1 registers the hook; verified on 2.10) -- on an older
            # torch such buckets would silently be reduced in finish() only
            ver = tuple(int(x) for x in torch.__version__.split("+")[0].split(".")[:2])
            if ver < (2, 4):
                import warnings
                warnings.warn("GradBucketer: torch %s may not fire post-accumulate hooks for undefined gradients; buckets of "
                              "directly written gradients are then reduced in finish() (correct, no overlap)" % torch.__version__)
        self._zeroed = False              # zero_grad() seen since the last finish(): gradients ACCUMULATE in the views

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        views, o = [], 0
        for p in ps:
            v = flat[o:o + p.numel()].view_as(p)
            o += p.numel()
            views.append(v)
            p.grad = v                    # the gradient lives here from now on
            p._sm_grad_view = v           # (HipSGD.zero_grad leaves such parameters alone)
        self.buckets.append(dict(flat=flat, params=list(ps), views=views, pending=len(ps), work=None))

    def zero_grad(self):
        """start of a step: gradients are zeroed in place, one memset per bucket (`.grad` stays the bucket view)"""
        for b in self.buckets:
            b["flat"].zero_()
        self._seen.clear()
        self._zeroed = True
        if self.params and self.params[0].is_cuda:
            from . import hip_ops as H
            H.GRAD_SINK.begin_step()

    def _launch_ready(self, force=False):
        """launch, in index order, every bucket that is full (all of them when force)"""
        while self._next < len(self.buckets) and (force or self.buckets[self._next]["pending"] == 0):
            b = self.buckets[self._next]
            if self.world > 1 or self.force:
                b["work"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._next += 1

    def _on_grad(self, p):
        """post-accumulate hook = "this parameter's gradient is final for this step".  The engine runs a leaf's
        AccumulateGrad node once per backward, after EVERY op that uses the leaf has run -- also when those ops wrote the
        gradient straight into the view and handed autograd `None` (the hook fires for an undefined gradient as well; torch
        2.10) -- so readiness is counted here and nowhere else.  If something reset `.grad` to None meanwhile, autograd
        installed a fresh tensor: it is folded back into the view."""
        if not self._zeroed:
            raise RuntimeError("GradBucketer: backward without bucketer.zero_grad() at the start of the step -- the gradients "
                               "live in the buckets and would accumulate across steps (optimizer.zero_grad() leaves them alone)")
        bi, i = self._where[id(p)]
        b = self.buckets[bi]
        v = b["views"][i]
        if p.grad is None:
            p.grad = v
        elif p.grad is not v and p.grad.data_ptr() != v.data_ptr():
            v.add_(p.grad.reshape(v.shape))
            p.grad = v
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch_ready()

    def finish(self):
        """Wait for every bucket and turn sums into means, in place.  Parameters that received no gradient this step
        contribute (and keep) zeros, as DDP with find_unused_parameters would."""
        self._launch_ready(force=True)                     # buckets with a missing gradient: reduce what there is
        self._next = 0
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                if self.world > 1:
                    b["flat"].div_(self.world)
            b["pending"], b["work"] = len(b["params"]), None
        self._zeroed = False

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self.params and self.params[0].is_cuda:
            from . import hip_ops as H
            H.GRAD_SINK.detach(self)
        for p in self.params:
            if getattr(p, "_sm_grad_view", None) is not None:
                p._sm_grad_view = None
                p.grad = None


class _SgdItem(__import__("ctypes").Structure):
    """sm_sgd_multi's item (include/sipmask_hip.h)"""
    _fields_ = [("p", __import__("ctypes").c_void_p), ("g", __import__("ctypes").c_void_p),
                ("buf", __import__("ctypes").c_void_p), ("n", __import__("ctypes").c_int64),
                ("lr", __import__("ctypes").c_float), ("wd", __import__("ctypes").c_float)]


class HipSGD:
    """torch.optim.SGD(momentum, weight_decay) with mmdet's paramwise options (bias_lr_mult, bias_decay_mult:
    M/mmdet/apis/train.py:92-133), the update as ONE sm_sgd_multi launch over all parameter tensors."""

    def __init__(self, named_params, lr=0.01, momentum=0.9, weight_decay=1e-4, bias_lr_mult=2.0, bias_decay_mult=0.0):
        self.items = []
        for name, p in named_params:
            if not p.requires_grad:
                continue
            is_bias = name.endswith(".bias")
            self.items.append(dict(p=p, lr=lr * (bias_lr_mult if is_bias else 1.0),
                                   wd=weight_decay * (bias_decay_mult if is_bias else 1.0), buf=None))
        self.momentum = momentum
        self._steps, self._blocks, self._keep = 0, None, None

    def zero_grad(self):
        """gradients that live in all-reduce buckets (GradBucketer) stay where they are: the bucketer zeroes them"""
        for it in self.items:
            if getattr(it["p"], "_sm_grad_view", None) is None:
                it["p"].grad = None

    @torch.no_grad()
    def step(self):
        """ONE launch for every parameter tensor (sm_sgd_multi): the per-tensor pointers go to the device as a small
        table (40 bytes per tensor, rebuilt per step because autograd hands out fresh .grad tensors), the
        (tensor, 4096-element chunk) list per thread block is built once."""
        import ctypes as C
        from . import _lib
        live = [it for it in self.items if it["p"].grad is not None]
        if not live:
            return
        dev = live[0]["p"].device
        first = self._steps == 0
        for it in live:
            if it["buf"] is None:
                it["buf"] = torch.zeros_like(it["p"], dtype=torch.float32)
        key = tuple(id(it) for it in live)
        if self._blocks is None or self._blocks[0] != key:
            bl = []
            for i, it in enumerate(live):
                bl.extend((i, c) for c in range((it["p"].numel() + 4095) // 4096))
            self._blocks = (key, torch.tensor(bl, dtype=torch.int32, device=dev).contiguous(), len(bl))
        grads = [it["p"].grad.detach().float().contiguous() for it in live]     # keep alive until the launch is queued
        tab = (_SgdItem * len(live))()
        for t, it, g in zip(tab, live, grads):
            t.p, t.g, t.buf, t.n = it["p"].data_ptr(), g.data_ptr(), it["buf"].data_ptr(), it["p"].numel()
            t.lr, t.wd = it["lr"], it["wd"]
        # the table goes up through PINNED host memory with a non-blocking copy: a pageable H2D copy is stream-ordered behind
        # the whole queued backward AND blocks the host until it ran, i.e. a device sync per step that kept the host from
        # enqueueing the next step's forward while this step's backward is still running.  Two alternating buffers, each
        # guarded by an event recorded behind its copy (the copy of step n may not have executed when step n+2 wants the
        # buffer back -- ADVICE r2).  With gradients living in all-reduce buckets the table never changes and is uploaded once.
        nbytes = C.sizeof(tab)
        raw = bytes(tab)
        if getattr(self, "_pin", None) is None or self._pin[0].numel() < nbytes:
            self._pin = [torch.empty(max(nbytes, 64), dtype=torch.uint8).pin_memory() for _ in range(2)]
            self._dev_items = [torch.empty(max(nbytes, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
            self._pin_ev, self._tab_raw, self._tab_k = [None, None], None, 0
        if raw != self._tab_raw:
            k = self._tab_k = (self._tab_k + 1) & 1
            if self._pin_ev[k] is not None:
                self._pin_ev[k].synchronize()
            C.memmove(self._pin[k].data_ptr(), C.addressof(tab), nbytes)
            self._dev_items[k][:nbytes].copy_(self._pin[k][:nbytes], non_blocking=True)
            if dev.type == "cuda":
                self._pin_ev[k] = torch.cuda.Event()
                self._pin_ev[k].record()
            self._tab_raw = raw
        items = self._dev_items[self._tab_k]
        lib = _lib.load()
        _lib.check(lib.sm_sgd_multi(_lib.ptr(items), _lib.ptr(self._blocks[1]), self._blocks[2], float(self.momentum),
                                    int(first), _lib.stream_ptr()), "sm_sgd_multi")
        self._steps += 1
        self._keep = (items, grads)          # until the next step: the launch above is asynchronous
        for it in live:
            # the kernel wrote through raw pointers: tell autograd / the launch-plan caches (plan_cache.py) that this
            # parameter changed
            torch.autograd.graph.increment_version(it["p"])


def head_train_step(head, feats, gt_bboxes, gt_labels, gt_masks, img_metas, optimizer, bucketer=None, train_cfg=None):
    """One data-parallel training step of the SipMask head on this rank's images: forward_train (HIP autograd ops)
    -> loss -> backward (bucketed all-reduce overlapped) -> SGD.  Returns the loss dict (detached floats)."""
    head.train()
    optimizer.zero_grad()
    if bucketer is not None:
        bucketer.zero_grad()
    if feats[0].is_cuda:
        from .ops_rows import begin_step
        begin_step()
    out = head(feats)
    losses = head.loss(*out, gt_bboxes, gt_labels, img_metas, train_cfg, gt_masks_list=gt_masks)
    total = sum(losses.values())
    total.backward()
    if bucketer is not None:
        bucketer.finish()
    optimizer.step()
    return {k: float(v.detach()) for k, v in losses.items()}


def detector_train_step(det, img, img_metas, gt_bboxes, gt_labels, gt_masks, optimizer, bucketer=None, sync=True):
    """One data-parallel training step of the whole detector (BASELINE config #4): SipMask.forward_train (HIP
    autograd ops for backbone stages 2-4, FPN and head; BN and stage 1 frozen as in the config) -> backward with the
    bucketed all-reduce overlapped -> SGD.  Returns the loss dict: floats, or with sync=False detached 0-d device tensors
    -- reading a loss value is a device sync, and a training loop that logs every N iterations need not pay one per step
    (the host then enqueues the next step's forward while this step's backward is still running)."""
    optimizer.zero_grad()
    if bucketer is not None:
        bucketer.zero_grad()
    if img.is_cuda:
        from .ops_rows import begin_step
        begin_step()
    losses = det.forward_train(img, img_metas, gt_bboxes, gt_labels, gt_masks=gt_masks)
    sum(losses.values()).backward()
    if bucketer is not None:
        bucketer.finish()
    optimizer.step()
    return {k: (float(v.detach()) if sync else v.detach()) for k, v in losses.items()}
