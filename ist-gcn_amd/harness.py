"""Counterpart of the reference's training harness body (processor/recognition.py:152-176, 249-296) for the drop-in
Models: optimizer construction, step-LR schedule and one training iteration.  The reference's own
`REC_Processor` runs unchanged on these Models; this module exists because `processor/` needs
h5py / tensorboard / prettytable to import and so that bench.py and the parity tests share one definition of a step.
"""
import numpy as np
import os

import torch
import torch.nn.functional as F


class FlatSGD:
    """torch.optim.SGD(lr, momentum, nesterov, weight_decay) of processor/recognition.py:154-159,289 on three FLAT fp32
    buffers: every live parameter's `.data` is a view of `P`, its gradient lands in the matching view of `G`, its
    momentum in `M`, and ONE launch of `istgcn_sgd_step` updates the whole model.  `G` is also the buffer the
    data-parallel all-reduce runs on, in place (`attach_sync`): no pack / unpack copies, and the 1/world of the SUM as
    well as the 1/loss_scale of float16 training are folded into the update kernel.

    * live = parameters that have a gradient after the first backward.  Parameters whose gradient stays None (the
      reference's dead `linear.*`, `gcn.branch.bn.*`) are skipped entirely -- no weight decay either -- exactly as
      torch.optim.SGD skips `p.grad is None`.  A parameter that gets its first gradient later is updated by the same
      rule with plain tensor ops (slow path, never taken by the shipped models).
    * gradients that autograd delivered as tensors of their own are gathered into `G` by one multi-tensor copy; those
      the engine wrote straight into their `G` view cost nothing.
    * interface used by the reference's processor: `param_groups[i]['lr']`, `zero_grad()`, `step()`.
    `update_fn(P, G, M, lr, momentum, weight_decay, nesterov, grad_scale)` defaults to the HIP kernel (GPU tensors only);
    the CPU gloo test of the bucket logic passes a torch restatement.

    Overlapped exchange (`overlap=True`, data parallel only).  The flat buffers are laid out in the order in which the
    gradients became FINAL in the first backward pass (recorded by post-accumulate hooks; rank 0's order is adopted by
    every rank), so "the gradients that are ready early" are one contiguous prefix of `G`.  From the second step on the
    hook of the last parameter of that prefix (`early_fraction` of the bytes: the deep, wide blocks, which backward
    reaches first) launches its all-reduce asynchronously while the rest of the backward pass is still running;
    `step()` reduces the remaining suffix, waits for the early one and updates.  Two collectives instead of one, the
    first of them hidden behind ~2/3 of the backward pass.  One backward per step is assumed (a second backward
    before `step()` raises); under hipGraph capture the early launch is skipped and `step()` reduces everything."""

    def __init__(self, params, lr=0.1, momentum=0.9, nesterov=True, weight_decay=1e-4, loss_scale=1.0, update_fn=None,
                 overlap=True, early_fraction=0.75, skip_nonfinite=None, poll_every=32):
        self.params = [p for p in params if p.requires_grad]
        self.param_groups = [dict(params=self.params, lr=lr, momentum=momentum, nesterov=nesterov,
                                  weight_decay=weight_decay)]
        self.loss_scale = float(loss_scale)
        self.sync = None
        self._update = update_fn
        self.P = self.G = self.M = None
        self.found_inf = None              # int32[1] on the device: raised by a non-finite gradient (see check_overflow)
        # GradScaler's rule for a scaled backward pass: ANY inf / NaN in the (all-reduced) gradients skips the WHOLE update
        # (a pre-pass over the flat gradient buffer raises a device flag, the update kernel is a no-op when it is set; no
        # host sync).  Default: on whenever a loss scale is in use (float16 storage).  Off: only the non-finite elements
        # themselves are left out.  `poll_overflow()` -- called by train_step / GraphedStep -- reads the flag back every
        # `poll_every` steps and backs the loss scale off.
        self.skip_nonfinite = (self.loss_scale != 1.0) if skip_nonfinite is None else bool(skip_nonfinite)
        self.poll_every = int(poll_every)
        self._nf = None                    # int32[2]: the skip flag of this step / of the previous one (cleared a step later)
        self._steps = 0
        self._pending_M = None             # momentum loaded before the layout existed (load_state_dict on a fresh optimizer)
        self._pending_order = None         # ... and the layout order it was saved in
        self._live, self._gviews, self._late = [], [], {}
        # ISTGCN_OVERLAP=0: one bucket, all-reduced in step() (the A/B switch for the multi-GPU scaling run)
        self.overlap = bool(overlap) and os.environ.get('ISTGCN_OVERLAP', '1') != '0'
        self.early_fraction = float(early_fraction)
        self._arrival, self._seen = [], set()      # first backward: indices into self.params in the order the gradients became final
        self._order = None                 # layout order (indices into self.params) of the live parameters
        self._pos = {}                     # index into self.params -> position in the layout
        self._early_n = self._early_end = 0    # live parameters / flat elements of the early bucket (0: one bucket)
        self._arrived, self._work = 0, None
        self.early_launches = 0            # diagnostics: early all-reduces launched so far
        self._hooks = []
        if hasattr(torch.Tensor, 'register_post_accumulate_grad_hook'):
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(param):
            self._on_grad(i)
        return hook

    def _on_grad(self, i):
        if self.P is None:                 # before the layout exists: record the order of the first backward
            if i not in self._seen:
                self._seen.add(i)
                self._arrival.append(i)
            return
        if not self._early_n or self.sync is None or self.sync.world <= 1:
            return
        pos = self._pos.get(i)
        if pos is None or pos >= self._early_n:
            return
        if self._work is not None:
            raise RuntimeError('FlatSGD(overlap=True): a gradient of the early bucket arrived after its all-reduce was '
                               'launched (second backward before step()?); use overlap=False for gradient accumulation')
        self._arrived += 1
        if self._arrived == self._early_n:
            if self.P.is_cuda and torch.cuda.is_current_stream_capturing():
                return                     # hipGraph capture: no collective inside the graph, step() reduces everything
            self._gather(0, self._early_n)
            self._work = self.sync.all_reduce_flat_(self.G[:self._early_end], async_op=True)
            self.early_launches += 1

    def _gather(self, lo, hi):
        """gradients of the live parameters [lo, hi) that autograd delivered as tensors of their own -> their `G` views"""
        src, dst = [], []
        for p, gv in zip(self._live[lo:hi], self._gviews[lo:hi]):
            g = p.grad
            if g is None:
                raise RuntimeError('FlatSGD: a parameter had a gradient on the first step and has none now')
            if g.data_ptr() != gv.data_ptr():
                src.append(g if g.dtype == torch.float32 else g.float())
                dst.append(gv)
                p.grad = gv
        if src:
            torch._foreach_copy_(dst, src)

    def attach_sync(self, sync):
        """dp.FlatGradSync: all-reduce `G` in place inside step() (and, with overlap, its early prefix from the backward pass)."""
        self.sync = sync
        if self.P is not None and sync is not None and sync.world > 1:
            mine = list(self._order)
            if sync.agree_on(mine, self.P.device) != mine:
                raise RuntimeError('FlatSGD.attach_sync: the flat layout was fixed before the ranks could agree on it; '
                                   'attach the sync before the first step')
        return self

    def zero_grad(self, set_to_none=True):
        if not set_to_none and self.G is not None:
            # in place: the gradient tensors stay the views of the flat buffer (a step recorded in a hipGraph accumulates
            # into the same addresses on every replay)
            self.G.zero_()
            for p, gv in zip(self._live, self._gviews):
                p.grad = gv
            self._arrived = 0
            return
        for p in self.params:
            p.grad = None
        self._arrived = 0

    def _build(self):
        live_idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not live_idx:
            raise RuntimeError('FlatSGD.step(): no parameter has a gradient')
        # layout order = the order in which the gradients became final (what the hooks saw), then whatever they missed
        lset = set(live_idx)
        order = [i for i in self._arrival if i in lset]
        order += [i for i in live_idx if i not in set(order)]
        dev = self.params[order[0]].device
        if self.sync is not None and self.sync.world > 1:
            order = self.sync.agree_on(order, dev)            # rank 0's order: one layout on every rank
            if set(order) != lset:
                raise RuntimeError('FlatSGD: the ranks disagree on which parameters have gradients')
        self._order = order
        self._pos = {i: n for n, i in enumerate(order)}
        live = [self.params[i] for i in order]
        offs, total = [], 0
        for p in live:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError('FlatSGD: parameters must be fp32 tensors on one device')
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4                      # every view 16-byte aligned
        self.P = torch.zeros(total, dtype=torch.float32, device=dev)
        self.G = torch.zeros(total, dtype=torch.float32, device=dev)
        self.M = torch.zeros(total, dtype=torch.float32, device=dev)
        def view(buf, o, p):
            flat = buf[o:o + p.numel()]
            if getattr(p, '_istgcn_flat_layout', None) == 'tap_major' and p.dim() == 4 and p.shape[3] == 1:
                # (Cout, Cin, k, 1) temporal-conv weight stored [k][Cout][Cin]: the layout its gradient is computed in, so
                # AccumulateGrad takes the gradient as it is (same strides) instead of copying it into the Conv2d layout
                co, ci, k, _ = p.shape
                return flat.view(k, co, ci).permute(1, 2, 0).unsqueeze(-1)
            return flat.view(p.shape)
        with torch.no_grad():
            pviews = [view(self.P, o, p) for p, o in zip(live, offs)]
            for p, v in zip(live, pviews):
                v.copy_(p.data)
                p.data = v
        self._gviews = [view(self.G, o, p) for p, o in zip(live, offs)]
        self._live = live
        self._pptrs = [p.data_ptr() for p in live]
        self._live_ids = {id(p) for p in live}
        # early bucket: the shortest prefix of the arrival order that holds `early_fraction` of the elements -- only when the
        # hooks really saw every live parameter arrive (otherwise the prefix is not known to be final early: one bucket)
        self._early_n = self._early_end = 0
        if self.overlap and self._hooks and lset <= self._seen and len(live) > 1 and 0.0 < self.early_fraction < 1.0:
            for n, o in enumerate(offs[1:], 1):
                if o >= self.early_fraction * total:
                    self._early_n, self._early_end = n, o
                    break
        if self.sync is not None and self.sync.world > 1:
            # whether there IS an early bucket depends on rank-local state (which hooks fired): rank 0's cut on every rank,
            # or one rank would launch two collectives per step and another one (mismatched collectives hang)
            self._early_n, self._early_end = self.sync.agree_on([self._early_n, self._early_end], dev)

    @property
    def bucket_bytes(self):
        return 0 if self.G is None else self.G.numel() * 4

    @torch.no_grad()
    def step(self):
        if self.P is None:
            self._build()
            if self._pending_M is not None:        # exact resume: the momentum goes in BEFORE the first update
                self._install_momentum(self._pending_M, self._pending_order)
                self._pending_M = self._pending_order = None
        for p, pp in zip(self._live, self._pptrs):
            if p.data_ptr() != pp:
                # .half()/.float()/.to(device)/`p.data = ...` after the layout was fixed: the update would go to an orphaned
                # flat buffer and the model would silently stop learning
                raise RuntimeError('FlatSGD: a parameter no longer aliases the flat buffer (was the model moved or cast '
                                   'after the first step?); create a new optimizer')
        early, self._work = self._work, None
        self._arrived = 0
        self._gather(self._early_n if early is not None else 0, len(self._live))
        scale = 1.0 / self.loss_scale
        if self.sync is not None and self.sync.world > 1:
            # SUM, in place; the mean's 1/world goes into the update.  With the early prefix already on its way only the
            # suffix is left (every rank launched the early one: the layout and the hook count are the same everywhere)
            if early is not None:
                self.sync.all_reduce_flat_(self.G[self._early_end:])
                early.wait()
            else:
                self.sync.all_reduce_flat_(self.G)
            scale /= self.sync.world
        grp = self.param_groups[0]
        upd = self._update
        if upd is None:
            from . import ops
            if self.found_inf is None:
                self.found_inf = torch.zeros(1, dtype=torch.int32, device=self.P.device)
            skip = None
            if self.skip_nonfinite:
                if self._nf is None:
                    self._nf = torch.zeros(2, dtype=torch.int32, device=self.P.device)
                par = self._steps & 1
                skip = self._nf[par:par + 1]
                skip.zero_()                                         # (last read by the update of two steps ago)
                ops.grad_nonfinite(self.G, skip)                     # after the all-reduce: every rank reaches the same verdict
            ops.sgd_step(self.P, self.G, self.M, grp['lr'], grp['momentum'], grp['weight_decay'], grp['nesterov'], scale,
                         found_inf=self.found_inf, skip_if=skip)
        else:
            upd(self.P, self.G, self.M, grp['lr'], grp['momentum'], grp['weight_decay'], grp['nesterov'], scale)
        self._steps += 1
        for p in self.params:                                        # slow path: first gradient after the layout was fixed
            if p.grad is not None and id(p) not in self._live_ids:
                g = p.grad.float() * scale
                if self.sync is not None and self.sync.world > 1:
                    import torch.distributed as dist
                    dist.all_reduce(g, group=self.sync.group)
                g = g + grp['weight_decay'] * p.data
                buf = self._late.get(id(p))
                buf = g.clone() if buf is None else buf.mul_(grp['momentum']).add_(g)
                self._late[id(p)] = buf
                p.data.add_(g + grp['momentum'] * buf if grp['nesterov'] else buf, alpha=-grp['lr'])

    def poll_overflow(self, backoff=0.5):
        """check_overflow() every `poll_every` steps (one host sync each time), nothing in between: what train_step and
        GraphedStep call after every step.  Steps with non-finite gradients in between were skipped on the device."""
        if self.skip_nonfinite and self.poll_every > 0 and self._steps % self.poll_every == 0:
            return self.check_overflow(backoff)
        return False

    def check_overflow(self, backoff=0.5):
        """Host poll (one sync) of the non-finite flag the update kernel raises: True if any gradient element since the
        last poll was inf / NaN (with `skip_nonfinite` those steps were skipped as a whole, otherwise the non-finite
        elements were; parameters and momentum stayed finite either way).  With a loss scale in use (float16 storage)
        the scale is multiplied by `backoff`, as a dynamic loss scaler would."""
        if self.found_inf is None or int(self.found_inf.item()) == 0:
            return False
        self.found_inf.zero_()
        if self.loss_scale != 1.0:
            self.loss_scale = max(1.0, self.loss_scale * backoff)
        return True

    def state_dict(self):
        return {'momentum': None if self.M is None else self.M.clone(), 'param_groups': [
            {k: v for k, v in g.items() if k != 'params'} for g in self.param_groups], 'loss_scale': self.loss_scale,
            'order': None if self._order is None else list(self._order)}     # layout of `momentum` (indices into the parameter list)

    def load_state_dict(self, sd):
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)
        self.loss_scale = sd.get('loss_scale', 1.0)
        if sd.get('momentum') is not None:
            if self.M is None:
                # the flat layout is fixed by the first step's gradients: keep the momentum and put it in place right after
                # the layout is built, before that step's update (no throw-away step, the resume is exact)
                self._pending_M = sd['momentum'].detach().clone()
                self._pending_order = sd.get('order')
            else:
                self._install_momentum(sd['momentum'], sd.get('order'))

    def _install_momentum(self, saved, order):
        """saved flat momentum (laid out in `order`: indices into the parameter list; None = this optimizer's order) -> M"""
        if order is None or list(order) == list(self._order):
            if saved.numel() != self.M.numel():
                raise RuntimeError('FlatSGD.load_state_dict: the saved momentum does not match this model\'s live parameters')
            self.M.copy_(saved)
            return
        if sorted(order) != sorted(self._order):
            raise RuntimeError('FlatSGD.load_state_dict: the saved momentum covers other parameters than this model\'s live ones')
        offs, o = {}, 0
        for i in order:                                  # offsets in the SAVED layout (same 4-element alignment rule)
            offs[i] = o
            o += (self.params[i].numel() + 3) // 4 * 4
        if o != saved.numel():
            raise RuntimeError('FlatSGD.load_state_dict: the saved momentum does not match this model\'s live parameters')
        o = 0
        for i in self._order:
            n = (self.params[i].numel() + 3) // 4 * 4
            self.M[o:o + n].copy_(saved[offs[i]:offs[i] + n])
            o += n


def make_optimizer(model, optimizer='SGD', base_lr=0.1, nesterov=True, weight_decay=1e-4, loss_scale=1.0):
    """recognition.py:152-166.  SGD on GPU parameters -> FlatSGD (one update launch, flat all-reduce bucket)."""
    if optimizer == 'SGD':
        params = list(model.parameters())
        if params and all(p.is_cuda for p in params):
            return FlatSGD(params, lr=base_lr, momentum=0.9, nesterov=nesterov, weight_decay=weight_decay,
                           loss_scale=loss_scale)
        return torch.optim.SGD(params, lr=base_lr, momentum=0.9, nesterov=nesterov, weight_decay=weight_decay)
    if optimizer == 'Adam':
        return torch.optim.Adam(model.parameters(), lr=base_lr, weight_decay=weight_decay)
    raise ValueError()


def adjust_lr(optimizer, base_lr, epoch, step):
    """recognition.py:168-176: lr = base_lr * 0.1 ** #(epoch >= step)."""
    lr = base_lr * (0.1 ** int(np.sum(epoch >= np.array(step)))) if step else base_lr
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr


def train_step(model, optimizer, data, label, grad_sync=None):
    """recognition.py:258-289: forward, CrossEntropy, zero_grad, backward, (gradient all-reduce,) step.
    Returns the loss tensor (no host sync here; the reference's `.item()` at :292 is the caller's choice)."""
    data = data.float()
    label = label.long()
    output = model(data)
    loss = F.cross_entropy(output, label)
    optimizer.zero_grad()
    ls = getattr(optimizer, 'loss_scale', 1.0)
    (loss * ls if ls != 1.0 else loss).backward()       # float16 storage: scaled backward, un-scaled inside the update
    if grad_sync is not None and getattr(optimizer, 'sync', None) is not grad_sync:
        grad_sync()                                       # (FlatSGD with attach_sync all-reduces its own flat buffer)
    optimizer.step()
    if hasattr(optimizer, 'poll_overflow'):
        optimizer.poll_overflow()                         # float16 storage: every poll_every steps, back the loss scale off
    return loss.detach()


class GraphedStep:
    """train_step with forward + backward recorded ONCE in a hipGraph (torch.cuda.CUDAGraph) and replayed per step: the
    ~250 kernel launches of a step leave the host as one graph launch, which removes the per-launch gaps (1.3 ms of a
    19.5 ms bf16 step at batch 64).  Outside the graph, eagerly: the copy of the batch into the static input buffers, the
    gradient all-reduce (RCCL, in place on the optimizer's flat buffer) and the optimizer's one-launch update.

    Requirements: a FlatSGD optimizer (static gradient addresses), fixed batch shape, no host-side randomness in the model's
    forward (the GPU augmentation's draws are host-side: not supported here).  Dropout masks: the host seed is frozen at
    capture time, so the kernels add a device counter (`Model.device_seed_epoch`) that the graph itself advances."""

    def __init__(self, model, optimizer, data, label, warmup=2, stream=None):
        if not isinstance(optimizer, FlatSGD):
            raise RuntimeError('GraphedStep needs harness.FlatSGD (gradients at fixed addresses)')
        if getattr(model, 'gpu_augment', None) is not None:
            raise RuntimeError('GraphedStep: the GPU augmentation draws on the host per step; use the eager train_step')
        self.model, self.opt = model, optimizer
        dev = data.device
        # one side stream for the eager warm-up, the capture and (by default) the replays: autograd's gradient
        # accumulators remember the stream they were created on, and capture is not allowed on the default stream
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            self.data = data.float().clone()
            self.label = label.long().clone()
            self.epoch = model.device_seed_epoch(dev) if hasattr(model, 'device_seed_epoch') else None
            self.ls = getattr(optimizer, 'loss_scale', 1.0)
            # eager steps: the optimizer moves the parameters into its flat buffer at the end of its FIRST step, and the
            # step after that rebuilds the packed-weight plans for the new addresses -- both must precede the capture
            for _ in range(max(2 if optimizer.P is None else 1, warmup)):
                train_step(model, optimizer, self.data, self.label)
            optimizer.zero_grad(set_to_none=False)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.loss = self._fwd_bwd()
        torch.cuda.current_stream(dev).wait_stream(self.stream)

    def _fwd_bwd(self):
        if self.epoch is not None:
            self.epoch.add_(1)
        output = self.model(self.data)
        loss = F.cross_entropy(output, self.label)
        self.opt.zero_grad(set_to_none=False)
        (loss * self.ls if self.ls != 1.0 else loss).backward()
        return loss.detach()

    def __call__(self, data=None, label=None):
        if data is not None and data.data_ptr() != self.data.data_ptr():
            self.data.copy_(data, non_blocking=True)
        if label is not None and label.data_ptr() != self.label.data_ptr():
            self.label.copy_(label, non_blocking=True)
        cur = torch.cuda.current_stream(self.data.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.graph.replay()
            self.opt.step()                                # (all-reduce of the flat gradient buffer,) one-launch update
            if self.opt.poll_overflow():
                # the scale is baked into the captured backward pass: the graph keeps the OLD scale, so the update must too
                self.opt.loss_scale = self.ls
        from . import ops
        ops.bump_weights_epoch()                           # the replay wrote BatchNorm running statistics (no Python ran)
        if cur != self.stream:
            cur.wait_stream(self.stream)
        return self.loss


def _host_copy(dst, src):
    """dst (pinned, contiguous) <- src on the host.  Same type and contiguous: ONE memmove -- `Tensor.copy_` fans an 11.5 MB
    copy out over every host core torch sees, which on the MI355X boxes (256 cores visible, 16 usable) took 16.9 ms against
    0.18 ms single-threaded (tools/h2d_probe.py); else torch's converting copy."""
    if src.dtype == dst.dtype and src.is_contiguous() and not src.is_cuda and src.numel() == dst.numel():
        import ctypes
        ctypes.memmove(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size())
    else:
        dst.copy_(src)


class DeviceStager:
    """Pinned host -> device staging of the training batches (first half): the reference moves every batch with
    a blocking `data.float().to(dev)` from pageable memory (processor/recognition.py:258; `pin_memory` is commented out in
    processor/processor.py:72).  Here the batches of any iterable of (data, label) host tensors -- a torch DataLoader over
    feeder.Feeder included -- go through `depth` (3) pinned host buffers and as many device buffers on a side stream: while step k
    computes on the current stream, batch k+1 is copied host -> pinned -> device (`non_blocking`), and the consumer only
    waits on an event.  Yields (data fp32 [N,C,T,V,M] on `device`, label int64 on `device`).

        for data, label in DeviceStager(loader, dev):
            loss = train_step(model, opt, data, label)

    A yielded pair stays valid until `depth - 1` further batches have been requested.  Batches of another shape (the last,
    ragged one without drop_last) are staged through buffers of their own."""

    def __init__(self, batches, device, depth=3):
        self.batches, self.device, self.depth = batches, torch.device(device), max(2, int(depth))
        self.stream = torch.cuda.Stream(device=self.device)
        self._slots = {}
        self.timers = {'wait_slot': 0.0, 'host_copy': 0.0, 'enqueue': 0.0, 'batches': 0, 'blocked': 0}

    def _slot(self, i, data, label):
        key = (i, tuple(data.shape), tuple(label.shape))
        sl = self._slots.get(key)
        if sl is None:
            # the device buffers come out of the SIDE stream's allocator pool: a block handed out on the consumer's stream may
            # still be the scratch of a kernel queued there, and the copy on the side stream would race with it (found by the
            # test: a slot allocated mid-run -- the ragged last batch -- arrived corrupted)
            with torch.cuda.stream(self.stream):
                dx = torch.empty(data.shape, dtype=torch.float32, device=self.device)
                dy = torch.empty(label.shape, dtype=torch.int64, device=self.device)
            sl = self._slots[key] = (torch.empty(data.shape, dtype=torch.float32).pin_memory(),
                                     torch.empty(label.shape, dtype=torch.int64).pin_memory(), dx, dy,
                                     torch.cuda.Event(), torch.cuda.Event())
        return sl

    def _stage(self, i, data, label):
        import time
        t0 = time.perf_counter()
        hx, hy, dx, dy, ready, free = self._slot(i % self.depth, data, label)
        # Two hazards, two waits.  The pinned buffer may be overwritten once the PREVIOUS copy out of it has run: a host check of
        # that copy's event (done long ago).  The device buffer may be overwritten once the step that read it has run: a
        # DEVICE-side wait of the copy stream on `free`.  (The first use of a slot allocates it -- ~50 ms of hipHostMalloc per
        # 11.5 MB pinned buffer: a one-off that a 20-step measurement must not average in, bench.py --h2d starts its clock
        # after `depth` batches.)
        if not ready.query():                              # (normally complete two steps ago: no host wait)
            ready.synchronize()
            self.timers['blocked'] += 1
        t1 = time.perf_counter()
        _host_copy(hx, data)                               # (host-side cast to fp32 + copy into pinned memory)
        _host_copy(hy, label)
        t2 = time.perf_counter()
        self.stream.wait_event(free)
        with torch.cuda.stream(self.stream):
            dx.copy_(hx, non_blocking=True)
            dy.copy_(hy, non_blocking=True)
            ready.record(self.stream)
        t3 = time.perf_counter()
        tm = self.timers                                   # host seconds per phase, summed over the batches staged so far
        tm['wait_slot'] += t1 - t0; tm['host_copy'] += t2 - t1; tm['enqueue'] += t3 - t2; tm['batches'] += 1
        return dx, dy, ready, free

    def __iter__(self):
        # batch i+1 is staged AFTER batch i has been handed over, i.e. while the consumer's step i runs on the GPU; its slot
        # (depth 2: the one of batch i-1) is free once step i-1 has run -- `free` is recorded when the consumer comes back,
        # behind the work it issued on the slot's buffers
        pending, i = None, 0
        for data, label in self.batches:
            if pending is not None:
                dx, dy, ready, free = pending
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ready)                      # device-side wait: the host runs ahead
                yield dx, dy
                free.record(torch.cuda.current_stream(self.device))
            pending = self._stage(i, data, label)
            i += 1
        if pending is not None:
            dx, dy, ready, free = pending
            torch.cuda.current_stream(self.device).wait_event(ready)
            yield dx, dy
            free.record(torch.cuda.current_stream(self.device))


def weights_init(m):
    """recognition.py:31-44 (applied by REC_Processor.load_model via model.apply)."""
    classname = m.__class__.__name__
    if classname.find('Conv1d') != -1 or type(m) is torch.nn.Conv2d:
        m.weight.data.normal_(0.0, 0.02)
        if m.bias is not None:
            m.bias.data.fill_(0)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


# ---------------------------------------------------------------------------------------------------------------
# checkpoints and evaluation: counterparts of torchlight/torchlight/io.py:57-90,101-107 and of
# processor/recognition.py:178-183,312-385, so accuracy parity can be shown with upstream `.pt` files both ways
# ---------------------------------------------------------------------------------------------------------------
def load_weights(model, weights, ignore_weights=None, log=None):
    """torchlight IO.load_weights (io.py:57-90): `weights` is a path to a torch-saved state dict or the dict itself;
    a leading `module.` (nn.DataParallel checkpoints) is stripped from every key, keys starting with any of
    `ignore_weights` are dropped, and when the strict load fails the missing entries keep the model's own values."""
    from collections import OrderedDict
    if ignore_weights is None:
        ignore_weights = []
    if isinstance(ignore_weights, str):
        ignore_weights = [ignore_weights]
    if not isinstance(weights, dict):
        weights = torch.load(weights, map_location='cpu')
    weights = OrderedDict([[k.split('module.')[-1], v.cpu()] for k, v in weights.items()])
    for i in ignore_weights:
        for n in [w for w in weights if w.find(i) == 0]:
            weights.pop(n)
            if log:
                log('Filter [{}] remove weights [{}].'.format(i, n))
    try:
        model.load_state_dict(weights)
    except (KeyError, RuntimeError):
        state = model.state_dict()
        if log:
            for d in set(state.keys()).difference(set(weights.keys())):
                log('Can not find weights [{}].'.format(d))
        state.update(weights)
        model.load_state_dict(state)
    return model


def save_model(model, path):
    """torchlight IO.save_model (io.py:101-107): CPU state dict with every `module.` removed from the keys -- a file the
    reference's own load_weights (and therefore its processors and demos) reads back."""
    from collections import OrderedDict
    # (.contiguous(): parameters living in a flat optimizer buffer may be stored tap-major; the file holds plain tensors)
    weights = OrderedDict([[''.join(k.split('module.')), v.cpu().contiguous()] for k, v in model.state_dict().items()])
    torch.save(weights, path)
    return path


def topk_accuracy(result, label, k):
    """recognition.py:178-183 (argsort, hit if the label is among the k largest scores)."""
    rank = np.asarray(result).argsort()
    hit = [l in rank[i, -k:] for i, l in enumerate(np.asarray(label))]
    return sum(hit) * 1.0 / len(hit)


@torch.no_grad()
def evaluate(model, batches, device=None, show_topk=(1, 5)):
    """recognition.py:312-385 without the plotting: eval mode, no_grad forward per batch, mean CrossEntropy loss over
    batches, top-k accuracy and the confusion matrix (rows = true class, columns = argmax) over all samples.
    `batches` yields (data [N,C,T,V,M], label [N]).  Runs the Models' inference path (folded BatchNorms)."""
    model.eval()
    results, labels, losses = [], [], []
    for data, label in batches:
        data = data.float()
        label = label.long()
        if device is not None:
            data, label = data.to(device, non_blocking=True), label.to(device)
        out = model(data)
        losses.append(float(F.cross_entropy(out, label)))
        results.append(out.float().cpu().numpy())
        labels.append(label.cpu().numpy())
    result, lab = np.concatenate(results), np.concatenate(labels)
    nc = result.shape[1]
    conf = np.zeros((nc, nc), dtype=np.int64)
    np.add.at(conf, (lab, result.argmax(1)), 1)
    return {'mean_loss': float(np.mean(losses)), 'topk': {k: topk_accuracy(result, lab, k) for k in show_topk},
            'confusion': conf, 'result': result, 'label': lab}
