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
        self.fallbacks = 0

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
        dev = self.static[0].device
        rng = torch.cuda.get_rng_state(dev) if self.status is not None else None
        self.graph_a.replay()
        if self.status is not None and int(self.status.item()) & _capi.STATUS_F16_RANGE:
            # this batch left the split-precision kernels' range: its gradients are invalid and were NOT applied.  Same batch,
            # same random draws, eagerly - the front-end's own handler evaluates it with torch layers (object_level.render_rays).
            torch.cuda.set_rng_state(rng, dev)
            self.fallbacks += 1
            loss = self._eager_step()
            self.loss.copy_(loss.detach())
            return self.loss
        self.graph_b.replay()
        return self.loss
