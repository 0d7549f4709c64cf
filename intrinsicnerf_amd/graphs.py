"""HIP graphs for the launch-bound part of the path: the reference's TRAINING STEP (run_nerf.py:918-1027, trainer.py:876-991).

A step through the staged path is ~250 launches (sampling, two fused network forwards, compositing, their backward kernels,
the weight-gradient products, re-packing, ~115 small torch kernels of the loss / Adam tail) for ~9.9 ms of GPU work; issued one
by one the step takes 10.6 ms (profiles/r03_train_step.txt).  ``GraphedTrainStep`` records the step
ONCE into two HIP graphs and replays them:

    graph A:  zero grads -> loss_fn(*static inputs) -> loss.backward()        (every HIP kernel of the library and every
                                                                               torch kernel between them, on one stream)
    host   :  ONE read of the step's f16 range words (the split-precision kernels' guard)
    graph B:  optimizer.step()

The library's launches go through ``hipLaunchKernelGGL`` on the stream torch hands over - during capture that is the
capturing stream, so they become kernel nodes like torch's own; nothing in the C ABI changes.  The range words cannot be
read during capture; they are collected (kernels.captured_status) and read between the two graphs, so a batch that leaves
f16's range never reaches the optimizer: it is re-run eagerly (where the front-end's own fallback evaluates it with torch
layers) from the same RNG state.  Random draws inside the step (jitter, noise) use torch's graph-safe generator and advance
on every replay.

Learning-rate schedules: both reference trainers write ``param_group['lr'] = new_lrate`` every iteration (run_nerf.py:1024-1027,
trainer.py:1006-1009).  A Python float would be baked into graph B's kernels at capture time, so every group's rate lives in a
DEVICE tensor that the captured Adam reads; ``__call__`` notices a rate the trainer wrote into the group (any float, or another
tensor), copies its value into that tensor and puts the tensor back - the reference's decay lines work unchanged.
"""
import torch

from . import _capi, kernels


class GraphedTrainStep:
    """``step = GraphedTrainStep(loss_fn, example_inputs, optimizer)``; then ``loss = step(*batch)`` per iteration.

    ``loss_fn(*inputs) -> scalar loss`` renders and compares (device ops only: no ``.item()``, no data-dependent Python
    control flow); ``example_inputs`` fix the shapes - every later batch is copied into static tensors of those shapes.
    ``optimizer`` must support ``capturable=True`` (torch.optim.Adam, as both trainers use: run_nerf.py:307, trainer.py:841).
    The returned loss is a static tensor that the next call overwrites.
    """

    def __init__(self, loss_fn, example_inputs, optimizer, warmup=2):
        self.loss_fn, self.opt = loss_fn, optimizer
        self.static = [t.detach().clone() for t in example_inputs]
        dev = self.static[0].device
        for group in optimizer.param_groups:
            if "capturable" not in group:
                raise ValueError(f"{type(optimizer).__name__} has no capturable mode; HIP-graph capture needs one")
            group["capturable"] = True
        for st in optimizer.state.values():               # an optimizer that already stepped keeps `step` on the host
            if isinstance(st.get("step"), torch.Tensor) and st["step"].device != dev:
                st["step"] = st["step"].to(dev)
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        # one device scalar per group: what the captured optimizer kernels read as the learning rate (see the module docstring)
        self._lr = []
        for group in optimizer.param_groups:
            lr = group["lr"]
            t = lr.detach().to(dev, torch.float32).clone() if isinstance(lr, torch.Tensor) else torch.tensor(float(lr), dtype=torch.float32, device=dev)
            group["lr"] = t
            self._lr.append(t)
        # Eager warm-up on a side stream (allocator pools, the optimizer's lazily created state - a state created DURING capture
        # would be re-zeroed by every replay).  It must leave no trace: parameters, optimizer state and the RNG are put back.
        keep_p = [p.detach().clone() for p in self.params]
        keep_s = {p: {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in optimizer.state[p].items()}
                  for p in self.params if p in optimizer.state and optimizer.state[p]}
        rng = torch.cuda.get_rng_state(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for p, old in zip(self.params, keep_p):
                p.copy_(old)
                for k, v in optimizer.state.get(p, {}).items():
                    if isinstance(v, torch.Tensor):
                        if p in keep_s:
                            v.copy_(keep_s[p][k])
                        else:
                            v.zero_()                      # a fresh optimizer: zeroed moments and step count ARE its initial state
        torch.cuda.set_rng_state(rng, dev)
        torch.cuda.synchronize(dev)
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        saved, kernels.captured_status = kernels.captured_status, []
        try:
            optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph_a):
                self.loss = self.loss_fn(*self.static)
                self.loss.backward()
                words = kernels.captured_status
                # inside the graph: the step's range words folded into one, so that the host reads 4 bytes per step
                self.status = torch.cat(words).max().reshape(1) if words else None
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                optimizer.step()
        finally:
            kernels.captured_status = saved
        # the gradient tensors graph A writes (owned by its pool): what p.grad must point at whenever the caller looks
        self._grads = [p.grad for p in self.params]
        self.fallbacks = 0

    def close(self):
        """Hand the optimizer back to eager use: every group's learning rate becomes a Python float again (its current value) and
        ``capturable`` is switched off.  Call it before ``optimizer.state_dict()`` goes into a checkpoint that other code will read
        (run_nerf.py:1035-1043 saves it; a ``param_group['lr']`` that is a tensor on THIS device would travel with the file), or
        use ``optimizer_state_dict()``, which leaves the live optimizer untouched.  The graphs must not be replayed afterwards."""
        for group, t in zip(self.opt.param_groups, self._lr):
            group["lr"] = float(t.item())
            group["capturable"] = False
        for st in self.opt.state.values():                # an eager Adam keeps its step counts on the host
            if isinstance(st.get("step"), torch.Tensor):
                st["step"] = st["step"].detach().cpu()
        self.graph_a = self.graph_b = None

    def optimizer_state_dict(self):
        """``optimizer.state_dict()`` as an eager optimizer would have written it: float learning rates, ``capturable`` off, the
        per-parameter state (moments, step counts) as it is.  Loads into a fresh ``torch.optim.Adam`` on any device."""
        sd = self.opt.state_dict()
        groups = []
        for g in sd["param_groups"]:
            g = dict(g)
            if isinstance(g.get("lr"), torch.Tensor):
                g["lr"] = float(g["lr"].item())
            g["capturable"] = False
            groups.append(g)
        state = {k: {n: (v.detach().cpu() if n == "step" and isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                 for k, st in sd["state"].items()}
        return {"state": state, "param_groups": groups}

    def set_lr(self, lr, group=None):
        """Learning rate of ``group`` (index; default: every group) for the following steps.  Equivalent to the reference's
        ``param_group['lr'] = lr``, which ``__call__`` also honours."""
        for i, t in enumerate(self._lr):
            if group is None or group == i:
                t.fill_(float(lr))

    def _sync_lr(self):
        for group, t in zip(self.opt.param_groups, self._lr):
            cur = group["lr"]
            if cur is not t:                              # the trainer assigned a new rate (run_nerf.py:1026-1027, trainer.py:1008-1009)
                if isinstance(cur, torch.Tensor):
                    t.copy_(cur.detach().reshape(()))
                else:
                    t.fill_(float(cur))
                group["lr"] = t

    def _eager_step(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(*self.static)
        loss.backward()
        self.opt.step()
        return loss

    def __call__(self, *inputs):
        if len(inputs) != len(self.static):
            raise ValueError(f"expected {len(self.static)} inputs, got {len(inputs)}")
        for dst, src in zip(self.static, inputs):
            if dst.shape != src.shape:
                raise ValueError(f"input of shape {tuple(src.shape)}, the graph was captured for {tuple(dst.shape)}")
            dst.copy_(src, non_blocking=True)
        if self.graph_a is None:
            raise RuntimeError("this GraphedTrainStep was closed")
        self._sync_lr()
        dev = self.static[0].device
        rng = torch.cuda.get_rng_state(dev) if self.status is not None else None
        self.graph_a.replay()
        if self.status is not None and int(self.status.item()) & _capi.STATUS_F16_RANGE:
            # this batch left the split-precision kernels' range: its gradients are invalid and were NOT applied.  Same batch,
            # same random draws, eagerly - the front-end's own handler evaluates it with torch layers (object_level.render_rays).
            torch.cuda.set_rng_state(rng, dev)
            self.fallbacks += 1
            loss = self._eager_step()                     # (re-binds every p.grad to a fresh eager tensor)
            self.loss.copy_(loss.detach())
            with torch.no_grad():                         # ... back to the graph's own gradient tensors, holding this step's values: a caller
                for p, g in zip(self.params, self._grads):      # that clips or logs p.grad keeps seeing what the last step used
                    if g is not None:
                        if p.grad is not None:
                            g.copy_(p.grad)
                        else:
                            g.zero_()
                        p.grad = g
            return self.loss
        self.graph_b.replay()
        return self.loss
