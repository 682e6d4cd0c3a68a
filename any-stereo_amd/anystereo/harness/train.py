"""cfg 4: the data-parallel training step — one process per GPU, DDP over RCCL.

Mirrors the `multi_training` branch of train_continuous_IGEV.py:201-239 (model.train() + freeze_bn, AdamW +
linear OneCycleLR, sequence_loss_multiscale, clip 1.0); the reference shards the batch with nn.DataParallel
(:184), here every rank owns batch/world samples and the only collective is DDP's bucketed fp32 gradient
all-reduce (45.6 MB for IGEV), overlapped with backward.  Synthetic batches have the shapes of
stereo_datasets.py:71,116-123,190-193: 160x320 network input, 51 200 random HR queries per sample, one scale
per sample in [1, 2.95].
"""
from __future__ import annotations

import os

import torch
import torch.distributed as td

from ..nn.liif import make_coord
from .metrics import fetch_optimizer, train_step
from .synthetic import det_uniform, synthetic_pair


def synthetic_train_batch(batch: int, height: int = 160, width: int = 320, n_query: int | None = None, seed: int = 0,
                          scale_min: float = 1.0, scale_max: float = 2.95, device="cpu"):
    """(image1, image2, hr_coord [B,Q,2], hr_disp [B,1,Q], scale [B,1]) — Q = height*width by default
    (`sample_q = inp_size[0]*inp_size[1]`, stereo_datasets.py:71), queries drawn without replacement from the
    cell-centre grid of the round(size*scale) HR image (stereo_datasets.py:190-193)."""
    q = n_query or height * width
    img1, img2 = synthetic_pair(batch, height, width, shift=8, seed=1000 + seed)
    scale = det_uniform((batch, 1), 2000 + seed, scale_min, scale_max)
    gen = torch.Generator().manual_seed(3000 + seed)
    rows = []
    for b in range(batch):
        s = float(scale[b, 0])
        grid = make_coord([round(height * s), round(width * s)])
        n = grid.shape[0]
        idx = torch.randperm(n, generator=gen)[:q] if n >= q else torch.randint(n, (q,), generator=gen)
        rows.append(grid[idx])
    hr_coord = torch.stack(rows).contiguous()
    hr_disp = det_uniform((batch, 1, q), 4000 + seed, 0.5, 64.0)
    return tuple(t.to(device) for t in (img1, img2, hr_coord, hr_disp, scale))


def shard_batch(batch, rank: int, world: int):
    """Rank's contiguous slice of a global batch (the role of DataParallel's scatter, train_continuous_IGEV.py:184)."""
    n = batch[0].shape[0]
    if n % world:
        raise ValueError(f"global batch {n} is not divisible by world size {world}")
    per = n // world
    return tuple(t[rank * per:(rank + 1) * per].contiguous() for t in batch)


class Trainer:
    """model.train() + freeze_bn + gradient exchange when a process group exists + AdamW/OneCycleLR; `step(batch)` runs
    harness.metrics.train_step — by default with its gradient half replayed as a captured hipGraph (see __init__) — and returns
    (loss, metrics).

    Eager multi-rank steps (`graph=False`) go through DDP:

    DDP without `find_unused_parameters`: in the `multi_training` branch the loss sees only the GRU predictions
    (train_continuous_IGEV.py:219), and the loop detaches `disp` at every iteration (continuous_IGEVstereo.py:285), so the
    parameters that only feed `init_disp` (the IGEV `classifier`) never receive a gradient — in the reference as well, where
    AdamW simply skips them.  The reducer must not wait for them: the FIRST step therefore runs forward + backward once on the
    bare module, freezes (`requires_grad_(False)`) every parameter that came back without a gradient, and only then wraps the
    module — a plain DDP whose autograd graph is the same every step, with no per-step graph walk (find_unused_parameters cost
    10 ms of a 129 ms step in round 1) and no dependence on `static_graph`."""

    def __init__(self, model, lr: float = 2e-4, wdecay: float = 1e-5, num_steps: int = 100000, train_iters: int = 16,
                 max_disp: int = 192, lr_fixed: bool = False, mixed_precision: bool = False, bucket_cap_mb: int | None = None,
                 force_ddp: bool = False, loss_scale: float | None = None, graph: bool | None = None, ddp_impl: str | None = None):
        model.train()
        model.freeze_bn()  # train_continuous_IGEV.py:203
        self.model = model
        self.module = model
        self._want_ddp = td.is_available() and td.is_initialized() and (td.get_world_size() > 1 or force_ddp)
        self._bucket_cap_mb = int(os.environ.get("ANYSTEREO_DDP_BUCKET_MB", "25")) if bucket_cap_mb is None else bucket_cap_mb
        self.ddp_mode = "none"
        self.frozen_unused = []
        # The step as a captured hipGraph (default on a CUDA model without GradScaler; ANYSTEREO_TRAIN_GRAPH=0 / graph=False = eager):
        # zero_grad + forward + loss + backward + unscale replayed as ONE graph per step, then — eagerly — the gradient exchange
        # between ranks, clip and AdamW (scope "grads", the default), or clip + a capturable AdamW inside the graph too (scope
        # "step", one rank).  The eager step is host-bound (~4 900 launches, 60-83 ms by host load); the replayed one
        # takes the GPU's 57 ms on any host.  `graph_warmup` eager steps come first.
        if graph is None:
            graph = os.environ.get("ANYSTEREO_TRAIN_GRAPH", "1") == "1"
        p0 = next(model.parameters())
        # Gradient exchange between ranks.  "ddp" (default): torch's DistributedDataParallel — bucketed all-reduce overlapped with
        # the backward pass.  "flat": the bare module computes its local gradients, then ONE all-reduce of the flattened gradient
        # vector (50 MB for IGEV: ~1 ms over xGMI at 8 ranks) and the update — no reducer hooks, nothing of torch.distributed inside
        # the forward / backward, so the gradient half of the step can be a captured hipGraph on every rank (the eager step is
        # host-bound and eight ranks share one host).  Forced when a graph is asked for with more than one rank.
        self.ddp_impl = os.environ.get("ANYSTEREO_DDP_IMPL", "ddp") if ddp_impl is None else ddp_impl
        if self.ddp_impl not in ("ddp", "flat"):
            raise ValueError(f"Trainer: ddp_impl={self.ddp_impl!r} (ddp | flat)")
        # (ANYSTEREO_OVERFLOW_POLICY=skip decides per step on the host whether the update runs: eager only)
        # the graphed step computes the loss without host synchronisation (a boolean-mask loss cannot be captured): a caller who
        # turns that form off (ANYSTEREO_SYNC_FREE_LOSS=0) gets the eager step
        self.use_graph = (bool(graph) and p0.is_cuda and not mixed_precision and os.environ.get("ANYSTEREO_OVERFLOW_POLICY", "poll") != "skip"
                          and os.environ.get("ANYSTEREO_SYNC_FREE_LOSS", "1") != "0")
        if self.use_graph and self._want_ddp:
            self.ddp_impl = "flat"
        self.graph_warmup = int(os.environ.get("ANYSTEREO_TRAIN_GRAPH_WARMUP", "3"))
        # "step": the whole step is one graph; "grads": zero_grad .. unscaled gradients are the graph, clip + AdamW + scheduler eager
        # (with more than one rank: the all-reduce sits between the two halves)
        self.graph_scope = "grads" if (self.use_graph and self._want_ddp) else os.environ.get("ANYSTEREO_TRAIN_GRAPH_SCOPE", "grads")
        if self._want_ddp and self.ddp_impl == "flat":
            nbytes = 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)
            self.ddp_mode = f"flat: one all-reduce of the concatenated gradients ({nbytes / 1e6:.0f} MB) between backward and the update"
        self._graph = None
        self.graph_cache_size = int(os.environ.get("ANYSTEREO_TRAIN_GRAPH_CACHE", "2"))
        self.optimizer, self.scheduler = fetch_optimizer(lr, wdecay, num_steps, model.parameters(), lr_fixed,
                                                         capturable=self.use_graph and self.graph_scope == "step")
        self.scaler = torch.amp.GradScaler("cuda", enabled=True) if mixed_precision else None
        self.train_iters, self.max_disp = train_iters, max_disp
        # Power-of-two loss scale (exact in fp32) for the split-precision dgrad kernels only: it keeps the activation gradients
        # (1e-6 .. 1e-9 at cfg-4 scale) out of the fp16 subnormal range of x = hi + lo/2048.  It buys nothing in fp32 mode or under
        # GradScaler, so it is 1 there.  The other end of the range is WATCHED, not assumed: the split kernels saturate |x| >= 65504
        # and count the event (ops.split_overflow_count); `step` polls the counters every `overflow_check_every` steps (one device
        # synchronisation, off the per-step path), halves the scale on an event and records it in `overflow_events`.
        # overflow_policy="skip" checks EVERY step before the optimizer runs and drops a step whose gradients saw a saturated
        # operand (GradScaler's policy; costs a synchronisation per step).
        from .. import ops
        split = ops.get_precision() == "split"
        if loss_scale is None:
            loss_scale = float(os.environ.get("ANYSTEREO_LOSS_SCALE", "4096")) if split else 1.0
        self.loss_scale = 1.0 if (mixed_precision or not split) else float(loss_scale)
        self.overflow_check_every = int(os.environ.get("ANYSTEREO_OVERFLOW_CHECK_EVERY", "20"))
        self.overflow_policy = os.environ.get("ANYSTEREO_OVERFLOW_POLICY", "poll")  # poll | skip | off
        self.overflow_events = []   # (step index, waves that saturated, loss scale before, loss scale after)
        self.skipped_steps = 0
        self.steps_done = 0
        # the loss as masked sums without host synchronisation (harness/metrics.py); ANYSTEREO_SYNC_FREE_LOSS=0 = the reference's
        # boolean-mask statement (16 + 3 synchronisations per step)
        self.sync_free_loss = os.environ.get("ANYSTEREO_SYNC_FREE_LOSS", "1") != "0"

    def _sync_module_state(self):
        """Rank 0's parameters and buffers on every rank — what DistributedDataParallel's constructor does for the "ddp" exchange
        and nothing did for the "flat" one (a bare module): ranks that built the model under different RNG state, or where only
        rank 0 loaded a checkpoint, would otherwise average gradients taken at different weights.  Once, before the first step."""
        if self.__dict__.get("_state_synced") or not (td.is_available() and td.is_initialized() and td.get_world_size() > 1):
            return
        # One flat buffer per dtype goes through the collective and every tensor is then overwritten with `copy_` through
        # `t.detach()` — NOT `t.data`, and not by broadcasting into the tensor itself: a c10d collective does not bump a tensor's
        # VERSION COUNTER and `.data` has one of its own, while every cache of the library (PackedConv / BN folds, grad.py's
        # transposed packs, the inference graph's fingerprint) is keyed on (data_ptr, _version).  A rank that ran a forward
        # before its first step must drop the packs of its pre-broadcast weights, frozen BatchNorm statistics included (no
        # optimizer step ever bumps those); `copy_` on a detached alias bumps the shared counter.
        with torch.no_grad():
            groups = {}
            for t in list(self.model.parameters()) + [b for b in self.model.buffers() if b is not None]:
                if t.numel():
                    groups.setdefault((t.dtype, t.device), []).append(t)
            for ts in groups.values():
                flat = torch.cat([t.detach().reshape(-1) for t in ts])
                td.broadcast(flat, 0)
                off = 0
                for t in ts:
                    n = t.numel()
                    t.detach().copy_(flat[off:off + n].view(t.shape))
                    off += n
        self._state_synced = True

    def _collective_max(self, n: int) -> int:
        """max over the ranks of a per-rank event count: every decision that changes the loss scale, drops an update or drops a
        captured graph is taken from this number, so all ranks take the same branch at the same step."""
        if not (self._want_ddp and td.get_world_size() > 1):
            return int(n)
        dev = next(self.model.parameters()).device if td.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([int(n)], dtype=torch.int64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return int(t.item())

    def _wrap_ddp(self, batch):
        """Probe pass on the bare module -> freeze gradient-less parameters -> wrap (see the class docstring)."""
        from .metrics import sequence_loss_multiscale
        model = self.model
        mode = os.environ.get("ANYSTEREO_DDP", "probe")  # probe (default) | find_unused | static
        if mode == "probe":
            # the probe runs in the mode every later step runs in (train + frozen BatchNorm2d), whatever the caller left
            model.train()
            model.freeze_bn()
            self._requires_grad_before = {n: p.requires_grad for n, p in model.named_parameters()}
            model.zero_grad(set_to_none=True)
            image1, image2, hr_coord, gt, scale = batch
            res = model(image1, image2, iters=min(2, self.train_iters), hr_coord=hr_coord.clone(), scale=scale)
            preds = res[1] if isinstance(res, tuple) else res
            loss, _ = sequence_loss_multiscale(preds, gt, (gt < 512) & (gt > 0.0), max_disp=self.max_disp)
            loss.backward()
            for n, p in model.named_parameters():
                if p.requires_grad and p.grad is None:
                    p.requires_grad_(False)
                    self.frozen_unused.append(n)
            model.zero_grad(set_to_none=True)
            if td.get_world_size() > 1:
                # every rank must wrap the same parameter set, or the reducers' buckets disagree and the first all-reduce hangs
                mine = sorted(self.frozen_unused)
                sets = [None] * td.get_world_size()
                td.all_gather_object(sets, mine)
                if any(s_ != mine for s_ in sets):
                    raise RuntimeError(f"Trainer: the probe pass froze different parameter sets on different ranks: "
                                       f"{[len(s_) for s_ in sets]} tensors per rank")
            if self.frozen_unused:
                import warnings
                warnings.warn("anystereo Trainer: requires_grad_(False) on %d parameter tensors the loss never reaches (%s%s); "
                              "Trainer.restore_requires_grad() undoes it" % (
                                  len(self.frozen_unused), ", ".join(self.frozen_unused[:4]), ", ..." if len(self.frozen_unused) > 4 else ""),
                              RuntimeWarning)
        p = next(model.parameters())
        ids = [p.device.index] if p.is_cuda else None
        self.module = torch.nn.parallel.DistributedDataParallel(
            model, device_ids=ids, find_unused_parameters=(mode == "find_unused"), static_graph=(mode == "static"),
            bucket_cap_mb=self._bucket_cap_mb, gradient_as_bucket_view=True,
            # BatchNorm2d is frozen for the whole run (train_continuous_IGEV.py:203) and the constructor has already broadcast rank
            # 0's module state: re-broadcasting ~270 unchanged buffers before every forward is pure overhead.  The hourglass'
            # BatchNorm3d layers normalise with batch statistics in training; their running statistics then evolve per rank, and
            # rank 0's — the ones a checkpoint holds, as under the reference's nn.DataParallel — are what they were with the broadcast
            broadcast_buffers=os.environ.get("ANYSTEREO_DDP_BROADCAST_BUFFERS", "0") == "1")
        self.ddp_mode = {"probe": f"plain DDP, {len(self.frozen_unused)} gradient-less parameter tensors frozen after a probe pass",
                         "find_unused": "find_unused_parameters", "static": "static_graph"}[mode]

    def restore_requires_grad(self):
        """Undo the probe pass's requires_grad_(False) on the bare module (e.g. before training it with another loss, or outside
        this Trainer).  The DDP wrapper built around the frozen set is dropped; the next `step` probes and wraps again."""
        before = getattr(self, "_requires_grad_before", None)
        if before:
            for n, p in self.model.named_parameters():
                if n in before:
                    p.requires_grad_(before[n])
        self.frozen_unused = []
        self.module = self.model
        self.ddp_mode = "none"

    def step(self, batch, sync_grads: bool = True):
        """One optimisation step.  sync_grads=False (measurement only): DDP's reducer is bypassed (`no_sync`), every rank
        steps on its local gradient — what a step costs without the all-reduce."""
        flat = self._want_ddp and self.ddp_impl == "flat"
        if flat:
            self._sync_module_state()
        if self._want_ddp and not flat and self.module is self.model:
            self._wrap_ddp(batch)
        # BatchNorm2d stays frozen whatever the caller did in between (validation's .eval(), a bare .train())
        if not self.module.training:
            self.module.train()
        self.model.freeze_bn()
        if self.use_graph:
            return self._step_graphed(batch, sync_grads)
        if flat:
            return self._step_flat(batch, sync_grads)
        watch = (self.loss_scale != 1.0 or self.overflow_events) and self.overflow_policy != "off" and self._on_gpu()
        gate = self._overflow_gate if (watch and self.overflow_policy == "skip") else None
        kw = dict(max_disp=self.max_disp, loss_scale=self.loss_scale, sync_free_loss=self.sync_free_loss, should_step=gate)
        if not sync_grads and self.module is not self.model:
            with self.module.no_sync():
                out = train_step(self.module, self.optimizer, self.scheduler, self.scaler, batch, self.train_iters, **kw)
        else:
            out = train_step(self.module, self.optimizer, self.scheduler, self.scaler, batch, self.train_iters, **kw)
        self.steps_done += 1
        if watch and gate is None and self.overflow_check_every > 0 and self.steps_done % self.overflow_check_every == 0:
            self._poll_overflow()
        return out

    def allreduce_gradients_flat(self):
        """Average the gradients over the ranks with ONE all-reduce of their concatenation (ddp_impl "flat").  Parameters without a
        gradient are the same set on every rank (same model, same autograd graph), so the vectors line up."""
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        if not grads or not (td.is_available() and td.is_initialized()):
            return 0
        world = td.get_world_size()
        flat = torch.cat([g.reshape(-1) for g in grads])
        td.all_reduce(flat)
        if world > 1:
            flat.div_(world)
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
        return flat.numel()

    def _step_flat(self, batch, sync_grads=True):
        """Eager step with the flat gradient exchange: local gradients -> one all-reduce -> clip + AdamW + schedule."""
        watch = (self.loss_scale != 1.0 or self.overflow_events) and self.overflow_policy != "off" and self._on_gpu()
        gate = self._overflow_gate if (watch and self.overflow_policy == "skip") else None
        out = train_step(self.model, self.optimizer, None, self.scaler, batch, self.train_iters, max_disp=self.max_disp,
                         loss_scale=self.loss_scale, sync_free_loss=self.sync_free_loss, phase="grads")
        if sync_grads:
            self.allreduce_gradients_flat()
            if self.scaler is not None and td.get_world_size() > 1:
                # unscale_ ran before the exchange: its per-rank inf/NaN flags become the group's, or some ranks would step (and
                # shrink their scale) while others do not
                st = self.scaler._per_optimizer_states.get(id(self.optimizer), {})
                for f in st.get("found_inf_per_device", {}).values():
                    td.all_reduce(f, op=td.ReduceOp.MAX)
        train_step(self.model, self.optimizer, self.scheduler, self.scaler, None, self.train_iters, should_step=gate, phase="update")
        self.steps_done += 1
        if watch and gate is None and self.overflow_check_every > 0 and self.steps_done % self.overflow_check_every == 0:
            self._poll_overflow()
        return out

    def _step_graphed(self, batch, sync_grads=True):
        """The step as a hipGraph replay: inputs are copied into static buffers, the captured graph holds zero_grad (gradients
        re-materialise at fixed addresses in the graph's pool), forward, the synchronisation-free loss, backward and the loss-scale
        division; the gradient exchange, clip_grad_norm_ and AdamW run eagerly behind it (scope "grads") or — scope "step", one
        rank — clip and a capturable AdamW are part of the graph and the OneCycleLR scheduler writes the device-side learning rate
        between replays.  Returns clones of the static loss / metric tensors."""
        # one captured graph per batch shape (LRU of `graph_cache_size`: each holds a memory pool with a whole step); a shape seen for
        # the first time runs eagerly first — MIOpen's solver search (timed trial launches) and the allocator's growth cannot happen
        # inside a capture
        graphs = self.__dict__.setdefault("_graphs", {})
        warm = self.__dict__.setdefault("_warm", {})
        # ... and per state of the module's BUFFERS: a captured step holds constants derived from them (grad.conv_frozen_bn caches
        # 1/sqrt(var + eps) of the frozen BatchNorm layers per buffer version), so load_state_dict() / a resume in the same process
        # selects a fresh capture instead of replaying the old statistics.  Parameters are re-packed from live storage inside the
        # graph and need no such key.  (Edits through `.data` bump no version counter: call `drop_graphs()` after those.)
        key = tuple(tuple(t.shape) if torch.is_tensor(t) else ("scalar", float(t)) for t in batch) + (self._buffers_fingerprint(),)
        ent = graphs.pop(key, None)
        if ent is not None:
            graphs[key] = ent  # most recently used last
        self._graph = ent
        if ent is None:
            # warm-up steps and the capture run on ONE side stream (PyTorch's whole-network capture recipe): autograd binds a
            # parameter's AccumulateGrad node to the stream of the forward that created it
            if getattr(self, "_gstream", None) is None:
                self._gstream = torch.cuda.Stream(device=batch[0].device)
            cur = torch.cuda.current_stream(batch[0].device)
            first = not graphs and all(k == key for k in warm)  # the first shape also initialises the optimizer state and the weight packs
            need = self.graph_warmup if first else 1
            if warm.get(key, 0) < need:  # eager steps first: solver searches, weight packs, allocator, optimizer state
                warm[key] = warm.get(key, 0) + 1
                self._gstream.wait_stream(cur)
                single = os.environ.get("ANYSTEREO_TRAIN_GRAPH_SINGLE_THREAD", "1") != "0"
                # warm-up on the thread (and stream) the capture will use: the BLAS / MIOpen handles are per thread and are
                # created — device allocations, not capturable — at a thread's first call
                with torch.autograd.set_multithreading_enabled(not single), torch.cuda.stream(self._gstream):
                    if self._want_ddp:  # the ranks stay in step during the warm-up too
                        out = train_step(self.module, self.optimizer, None, None, batch, self.train_iters, max_disp=self.max_disp,
                                         loss_scale=self.loss_scale, sync_free_loss=True, phase="grads")
                    else:
                        out = train_step(self.module, self.optimizer, self.scheduler, None, batch, self.train_iters, max_disp=self.max_disp,
                                         loss_scale=self.loss_scale, sync_free_loss=True)
                if self._want_ddp:
                    # the collective is issued from the CALLER's stream, as after a replay: the process group records its
                    # synchronisation events on the issuing stream, and its watchdog thread may not query an event whose last
                    # recording stream is capturing (hipErrorCapturedEvent) — the capture stream must never have issued one
                    cur.wait_stream(self._gstream)
                    if sync_grads:
                        self.allreduce_gradients_flat()
                    self._gstream.wait_stream(cur)
                    with torch.cuda.stream(self._gstream):
                        train_step(self.module, self.optimizer, self.scheduler, None, None, self.train_iters, phase="update")
                cur.wait_stream(self._gstream)
                for t in (out[0], *out[1].values()):
                    t.record_stream(cur)
                self.steps_done += 1
                self.__dict__["_active_graph"] = None  # an eager step re-created the gradients
                return out
            batch = tuple(t if torch.is_tensor(t) else torch.full((batch[0].shape[0], 1), float(t), device=batch[0].device) for t in batch)
            static = tuple(t.detach().clone() for t in batch)
            from .. import grad as G
            G.begin_forward()  # no anchor (and no autograd node) of the warm-up steps survives into the capture
            self.optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            # the hipGraph_t is kept: its memset nodes are rewritten before instantiation (below; tools/train_graph_nodes.py reads it)
            g = torch.cuda.CUDAGraph(keep_graph=True)
            # the backward pass is captured from THIS thread (autograd's device worker thread launching into a stream another
            # thread put into capture mode loses nodes / dependencies of the graph's tail on this ROCm stack: the last
            # gradients of a replay came back as garbage, tools/train_graph_check.py)
            single = os.environ.get("ANYSTEREO_TRAIN_GRAPH_SINGLE_THREAD", "1") != "0"
            # with a process group alive its watchdog thread polls events of finished collectives while this thread captures: the
            # default "global" capture mode turns that query into an error that aborts the process
            mode = "thread_local" if (td.is_available() and td.is_initialized()) else "global"
            with torch.autograd.set_multithreading_enabled(not single), torch.cuda.graph(g, stream=self._gstream, capture_error_mode=mode):
                loss, metrics = train_step(self.module, self.optimizer, None, None, static, self.train_iters, max_disp=self.max_disp,
                                           loss_scale=self.loss_scale, sync_free_loss=True,
                                           phase="all" if self.graph_scope == "step" else "grads")
            # Memset nodes (ATen's reduction semaphores, MIOpen's split-K zero-fills: 34 in the IGEV step) -> fill kernel nodes.  Inside
            # a chain this long the runtime does not reliably order them with the kernels around them: from the second replay on
            # the reductions behind them returned stale values (valid-pixel count, metrics) and gradients came back zero
            # (csrc/graph.hip).  ANYSTEREO_TRAIN_GRAPH_FILL=0 leaves the graph as captured (diagnostics).
            self.graph_memsets = (0, 0)
            if os.environ.get("ANYSTEREO_TRAIN_GRAPH_FILL", "1") != "0":
                from .. import ops
                self.graph_memsets = ops.graph_replace_memsets(g)
            left = self._collective_max(self.graph_memsets[1])
            if left:
                # a memset node that could not be rewritten (unreadable parameters, an element size the fill kernel does not serve)
                # would be replayed unordered: stale loss / zero gradients from the second replay on.  No graph then, on any rank.
                import warnings
                warnings.warn(f"anystereo Trainer: {left} memset node(s) of the captured step could not be rewritten as kernels; "
                              "the step stays eager", RuntimeWarning)
                self.use_graph = False
                self._graph = None
                graphs.clear()
                del g
                self.optimizer.zero_grad(set_to_none=True)
                return self.step(batch, sync_grads)
            g.instantiate()
            # the capture itself executed nothing: the step below is the first replay
            # the gradients this graph's kernels write (they were materialised at capture time, in this graph's pool): with more
            # than one cached graph the parameters' .grad must point at the replayed graph's tensors before the eager update
            ent = self._graph = {"graph": g, "batch": static, "loss": loss, "metrics": metrics,
                                 "grads": [(p_, p_.grad) for p_ in self.model.parameters()]}
            self.__dict__["_active_graph"] = ent
            while len(graphs) >= max(1, self.graph_cache_size):
                graphs.pop(next(iter(graphs)))
            graphs[key] = ent
        # Replays are stream-ordered like any other work since the graph holds kernel nodes only where libraries issued memsets
        # (see the capture above): no synchronisation is needed between steps and the host runs ahead.
        # ANYSTEREO_TRAIN_GRAPH_SYNC=device ends every step with torch.cuda.synchronize() (diagnostics).
        if self.__dict__.get("_active_graph") is not ent:
            for p_, g_ in ent["grads"]:
                p_.grad = g_
            self.__dict__["_active_graph"] = ent
        for dst, src in zip(ent["batch"], batch):
            dst.copy_(src)
        ent["graph"].replay()
        if self.graph_scope == "step":
            # the replayed AdamW wrote the parameters behind autograd's back: bump their version counters so that every
            # (data_ptr, _version)-keyed cache (weight packs, BatchNorm folds, the inference graph's fingerprint) sees the update —
            # an evaluation pass between training steps would otherwise run on the packs of its first call
            ps = self.__dict__.get("_all_params")
            if ps is None:
                ps = self.__dict__["_all_params"] = [p_ for p_ in self.model.parameters() if p_.requires_grad]
            try:
                torch._C._autograd._unsafe_set_version_counter(ps, [p_._version + 1 for p_ in ps])
            except (AttributeError, TypeError):
                with torch.no_grad():
                    torch._foreach_mul_(ps, 1.0)
        if self.graph_scope != "step":
            if self._want_ddp and sync_grads:
                self.allreduce_gradients_flat()
            train_step(self.module, self.optimizer, None, None, None, self.train_iters, phase="update")
        if self.scheduler is not None:
            self.scheduler.step()
        out = ent["loss"].clone(), {k: v.clone() for k, v in ent["metrics"].items()}
        if os.environ.get("ANYSTEREO_TRAIN_GRAPH_SYNC", "none") == "device":
            torch.cuda.synchronize(batch[0].device)
        self.steps_done += 1
        if (self.loss_scale != 1.0 and self.overflow_policy != "off" and self.overflow_check_every > 0
                and self.steps_done % self.overflow_check_every == 0):
            self._poll_overflow()  # a changed scale takes effect at the next capture only: drop the graph then
            if self.overflow_events and self.overflow_events[-1][0] == self.steps_done:
                self._graph = None
                self.__dict__.get("_graphs", {}).clear()
        return out

    def _buffers_fingerprint(self):
        bufs = self.__dict__.get("_all_buffers")
        if bufs is None:
            # the FROZEN layers' statistics (BatchNorm2d, eval for the whole run: freeze_bn): the hourglass' BatchNorm3d layers
            # normalise with batch statistics and update their running buffers every step — inside the graph once captured
            bufs = self.__dict__["_all_buffers"] = [b for m in self.model.modules() if isinstance(m, torch.nn.BatchNorm2d)
                                                    for b in (m.running_mean, m.running_var) if b is not None]
        ver = ptr = 0
        for i, t in enumerate(bufs):
            ver += t._version
            ptr ^= (t.data_ptr() + 0x9E3779B97F4A7C15 * i) & 0xFFFFFFFFFFFFFFFF
        return (len(bufs), ver, ptr)

    def drop_graphs(self):
        """Forget every captured step (after edits the version counters do not see, e.g. through `.data`)."""
        self._graph = None
        self.__dict__.get("_graphs", {}).clear()
        self.__dict__.get("_warm", {}).clear()

    def _on_gpu(self) -> bool:
        return next(self.model.parameters()).is_cuda

    def _poll_overflow(self) -> int:
        """Read and reset the split kernels' saturation counters (synchronises); on an event halve the loss scale (never
        below 1) and record it.  Returns the number of waves that saturated since the last poll."""
        from .. import ops
        n = self._collective_max(ops.split_overflow_count(reset=True))  # every rank polls at the same step index
        if n:
            before = self.loss_scale
            self.loss_scale = max(1.0, self.loss_scale * 0.5)
            self.overflow_events.append((self.steps_done, n, before, self.loss_scale))
            import warnings
            warnings.warn(f"anystereo Trainer: {n} waves saturated an fp16-split operand (|x| >= 65504 or NaN) by step "
                          f"{self.steps_done}; loss scale {before:g} -> {self.loss_scale:g}.  Gradients of the affected steps "
                          "were clipped at the fp16 range; use set_precision('fp32') if this persists.", RuntimeWarning)
        return n

    def _overflow_gate(self) -> bool:
        """should_step hook of train_step (policy "skip"): False drops the optimizer step of a backward pass that saturated."""
        if self._poll_overflow():
            self.skipped_steps += 1
            return False
        return True
