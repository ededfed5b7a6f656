"""The reference's training step (`train()`, vv_recon.py:461-550) on this stack -- row f2's caller.

One step = `full_process` forward (:467), the loss block (:474-500), backward and an Adam update
(:504: `tf.train.AdamOptimizer(alpha0).minimize(loss)`), with the reference's piecewise-constant
schedules for the learning rate `alpha0` and the weight `alpha1` of the decline-factor term (:479-482).
The data pipeline (lmdb + tensorpack, `data_util.py`) is outside the hot path: callers hand over device
tensors `(B, 3000, 3)` partial / `(B, 16384, 3)` ground truth; `python -m rfnet_amd.trainrun` drives it
with synthetic clouds.

MI355X specifics:
  * forward + loss + backward have static shapes and ~1400 kernel launches, a third of the eager step's
    wall time being host work: they are captured ONCE into a HIP graph and replayed (30 -> 23.7 ms at
    B = 32, tools/experiments/train_graph.py).  The learning rate and `alpha1` are device scalars, so the
    schedules move without re-capturing.
  * ROCm 7's graph "packet capture" replays a captured hipMemsetAsync of a small buffer with garbage
    from the second replay on (tools/experiments/graph_memset_probe.py) -- which is how torch's reduction
    kernels clear their semaphores, i.e. every captured `sum` / `max` goes stale.  This module's `main()`
    (like evalrun's and bench.py) calls `rfnet_amd.enable_graph_safe_runtime()`, which sets
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before the HIP runtime starts (no cost in replay time; importing the
    package does not touch the environment), and the library never issues a memset.  An application may have started HIP first
    (`torch.cuda.is_available()` is enough), so `TrainStep` asks `_host.graph_replay_ok()` -- a memset node
    replayed from a test graph -- and stays eager when the runtime fails it; the captured step is also
    checked against eager gradients on a second batch (which catches what a capture froze, though not
    this fault: it shows on later replays only).
  * data parallel: one process per GPU, each on its batch shard; after the backward the gradients are
    all-reduced in one flat bucket (RCCL over xGMI, `shard.allreduce_gradients`), outside the graph.
"""
import argparse
import os
import sys
import time

import torch

from . import shard
from .rfnet import RFNet, training_loss

# vv_recon.py:479-482
LR_BOUNDARIES, LR_VALUES = (50000, 100000, 150000, 200000), (0.0005, 0.0002, 0.0002, 0.0001, 0.00001)
A1_BOUNDARIES, A1_VALUES = (50000, 150000), (0.01, 0.01, 0.001)


def piecewise_constant(step, boundaries, values):
    """tf.train.piecewise_constant: values[0] for step <= boundaries[0], values[i] for
    boundaries[i-1] < step <= boundaries[i], values[-1] beyond the last boundary."""
    for b, v in zip(boundaries, values):
        if step <= b:
            return v
    return values[-1]


class TfAdam:
    """tf.train.AdamOptimizer's update (beta1 0.9, beta2 0.999, epsilon 1e-8), which places epsilon
    differently from torch.optim.Adam:  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);
    m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2;  p -= lr_t * m / (sqrt(v) + epsilon).
    Multi-tensor (`torch._foreach_*`) updates: a handful of launches for all 240 parameters."""

    def __init__(self, params, beta1=0.9, beta2=0.999, eps=1e-8):
        self.params = [p for p in params]
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self, lr):
        self.t += 1
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not idx:
            return
        ps = [self.params[i] for i in idx]
        gs = [self.params[i].grad for i in idx]
        ms = [self.m[i] for i in idx]
        vs = [self.v[i] for i in idx]
        torch._foreach_mul_(ms, self.beta1)
        torch._foreach_add_(ms, gs, alpha=1.0 - self.beta1)
        torch._foreach_mul_(vs, self.beta2)
        torch._foreach_addcmul_(vs, gs, gs, value=1.0 - self.beta2)
        lr_t = lr * (1.0 - self.beta2 ** self.t) ** 0.5 / (1.0 - self.beta1 ** self.t)
        den = torch._foreach_sqrt(vs)
        torch._foreach_add_(den, self.eps)
        torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)


class TrainStep:
    """`step(partial, gt) -> loss` for a fixed batch shape.  graph=True captures forward + loss + backward
    into a HIP graph (checked against the eager gradients, eager fallback with a note on stderr);
    the gradient all-reduce over `group` and the Adam update follow eagerly.  The graph path is OPT-IN at
    process level: call `rfnet_amd.enable_graph_safe_runtime()` before anything starts the HIP runtime,
    otherwise `_host.graph_replay_ok()` reports False and every step stays eager (`self.mode` says which)."""

    def __init__(self, net, batch, npartial=3000, ngt=16384, graph=True, group=None, optimizer=None,
                 check_tol=1e-3):
        self.net, self.group = net, group
        self.params = list(net.parameters())
        self.opt = optimizer if optimizer is not None else TfAdam(self.params)
        self.global_step = 0
        dev = self.params[0].device
        self.partial = torch.zeros(batch, npartial, 3, device=dev)
        self.gt = torch.zeros(batch, ngt, 3, device=dev)
        self.alpha1 = torch.zeros((), device=dev)
        self.graph = None
        self.graph_note = "eager (graph not requested)"
        self._loss = None
        if graph and dev.type == "cuda":
            self._capture(check_tol)

    # forward + loss block + backward on the static buffers
    def _fwd_bwd(self):
        collect = {}
        outs = self.net(self.partial, collect=collect)
        # (the ground truth's FPS as a forked branch of the graph under the forward, which pays in the
        # forward-only C5 step -- 8.49 -> 8.05 ms -- measured SLOWER here: 18.7 -> 19.3 ms per step)
        loss = training_loss(self.net, outs, collect, self.gt, self.alpha1)
        loss.backward()
        return loss.detach()

    def _capture(self, tol):
        from ._host import graph_replay_ok
        if not graph_replay_ok(self.partial.device):
            self.graph_note = ("eager (torch reductions do not replay from a HIP graph in this process: the HIP "
                               "runtime was started without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)")
            sys.stderr.write(f"rfnet_amd.trainrun: training step not captured -- {self.graph_note}\n")
            return
        g = torch.Generator(device="cpu").manual_seed(1234)
        first = (torch.rand(self.partial.shape, generator=g) - 0.5, torch.rand(self.gt.shape, generator=g) - 0.5)
        second = (torch.rand(self.partial.shape, generator=g) - 0.5, torch.rand(self.gt.shape, generator=g) - 0.5)
        self.alpha1.fill_(A1_VALUES[0])
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        try:
            with torch.cuda.stream(side):
                # eager on the SECOND pair of clouds (also the warm-up: library handles, scratch, autotuning --
                # none of it capturable): the reference the replays are checked against
                for _ in range(2):
                    self.partial.copy_(second[0])
                    self.gt.copy_(second[1])
                    self.net.zero_grad(set_to_none=True)
                    self._fwd_bwd()
                ref = [None if p.grad is None else p.grad.clone() for p in self.params]
                # captured on the FIRST pair ...
                self.partial.copy_(first[0])
                self.gt.copy_(first[1])
                self.net.zero_grad(set_to_none=True)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    loss = self._fwd_bwd()
            cur.wait_stream(side)
            worst = 0.0
            graph.replay()
            # ... and the second replay (where the runtime's replay fault shows: stale or garbage) runs on
            # the second pair: anything the graph froze at capture, or fails to recompute, shows up here
            self.partial.copy_(second[0])
            self.gt.copy_(second[1])
            graph.replay()
            torch.cuda.synchronize()
            for p, r in zip(self.params, ref):
                if (p.grad is None) != (r is None):
                    raise RuntimeError("captured step reaches a different set of parameters")
                if r is not None:
                    worst = max(worst, float((p.grad - r).abs().max()) / (float(r.abs().max()) + 1e-20))
            if not worst <= tol:
                raise RuntimeError(f"replayed gradients differ from the eager ones by {worst:.2e} of their maximum "
                                   "(ROCm graph memset replay fault? DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be set "
                                   "before the HIP runtime starts)")
            self.graph, self._loss = graph, loss
            self.graph_note = f"HIP graph (replayed gradients within {worst:.1e} of eager)"
        except Exception as exc:  # noqa: BLE001 -- any capture problem means: run eagerly
            cur.wait_stream(side)
            torch.cuda.synchronize()
            self.graph = None
            self.net.zero_grad(set_to_none=True)
            self.graph_note = f"eager ({type(exc).__name__}: {exc})"
            sys.stderr.write(f"rfnet_amd.trainrun: training step not captured -- {self.graph_note}\n")

    def __call__(self, partial, gt):
        lr = piecewise_constant(self.global_step, LR_BOUNDARIES, LR_VALUES)
        self.alpha1.fill_(piecewise_constant(self.global_step, A1_BOUNDARIES, A1_VALUES))
        self.partial.copy_(partial)
        self.gt.copy_(gt)
        if self.graph is not None:
            self.graph.replay()
            loss = self._loss.clone()
        else:
            self.net.zero_grad(set_to_none=True)
            loss = self._fwd_bwd()
        shard.allreduce_gradients(self.params, group=self.group)
        self.opt.step(lr)
        self.global_step += 1
        return loss


def main(argv=None):
    ap = argparse.ArgumentParser(description="RFNet training steps on synthetic clouds (one process per GPU)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU (the reference trains with 32)")
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    from . import enable_graph_safe_runtime
    enable_graph_safe_runtime()  # before anything starts the HIP runtime (the next line does)
    if not torch.cuda.is_available():
        raise SystemExit("rfnet_amd.trainrun needs a HIP device (MI355X); there is no CPU fallback")
    rank, world, _ = shard.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(args.seed)  # the same initial weights on every rank
    net = RFNet().to(dev)
    shard.broadcast_parameters(net.parameters())
    step = TrainStep(net, args.batch, graph=not args.eager)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)  # synthetic clouds, made on the device
    t0 = None
    losses = []
    for i in range(args.steps):
        partial = torch.rand(args.batch, 3000, 3, generator=g, device=dev) - 0.5
        gt = torch.rand(args.batch, 16384, 3, generator=g, device=dev) - 0.5
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        losses.append(step(partial, gt))  # no host read-back inside the loop
    torch.cuda.synchronize()
    dt = None if t0 is None else time.perf_counter() - t0
    if rank == 0:
        for i in sorted(set(list(range(0, args.steps, max(1, args.steps // 10))) + [args.steps - 1])):
            print(f"step {i} loss {float(losses[i]):.6f}", flush=True)
    if rank == 0 and dt is not None and args.steps > 2:
        ms = dt / (args.steps - 2) * 1e3
        print(f"{ms:.2f} ms per step, {world * args.batch / ms * 1e3:.0f} samples/s over {world} GPU(s); {step.graph_note}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
