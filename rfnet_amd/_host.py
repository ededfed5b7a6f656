"""Host-side plumbing shared by the op wrappers: argument staging, output allocation,
workspace cache, stream hand-off.  PyTorch is used only for device memory and streams.

Inputs may be torch tensors on the GPU (zero-copy), or CPU tensors / numpy arrays (staged to
the current GPU and the results copied back in the same kind).  There is no CPU execution
path: without a HIP device every op raises.
"""
import functools

import numpy as np
import torch

from . import _lib


def on_input_device(fn):
    """Run `fn` with the device of its first GPU tensor argument current: the C ABI launches on the
    calling thread's current HIP device with the stream it is handed, so outputs, scratch, the
    stream and (when profiling) the events must all belong to the device that holds the inputs."""
    @functools.wraps(fn)
    def wrapper(*args, **kw):
        dev = None
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                dev = a.device
                break
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapper


class Staged:
    """Remembers how the caller passed its arrays so results come back the same way."""

    def __init__(self):
        self.kind = "cuda"  # "cuda" | "cpu" | "numpy"
        self.device = None

    def take(self, x, dtype):
        """Normalise one argument to a contiguous tensor of `dtype` WITHOUT touching the GPU,
        so shape validation (and its reference-worded errors) works before any staging."""
        if isinstance(x, torch.Tensor):
            t = x.detach()
            if t.is_cuda:
                if self.device is None:
                    self.device = t.device
                elif t.device != self.device:
                    raise ValueError(f"all GPU inputs of one op must live on the same device: "
                                     f"got {self.device} and {t.device}")
            elif self.kind == "cuda":
                self.kind = "cpu"
        else:
            self.kind = "numpy"
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
        if t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()

    def up(self, *ts):
        """Stage validated arguments onto the GPU (no-op for tensors already there)."""
        dev = self._dev()
        return tuple(t if t.is_cuda else t.to(dev) for t in ts)

    def _dev(self):
        if self.device is None:
            if not torch.cuda.is_available():
                raise _lib.RfopsError(
                    "rfnet_amd ops run on an MI355X (gfx950) only: no HIP device is visible "
                    "and there is no CPU fallback"
                )
            self.device = torch.device("cuda", torch.cuda.current_device())
        return self.device

    def device_(self):
        return self._dev()

    def give(self, t):
        if self.kind == "cuda":
            return t
        if self.kind == "cpu":
            return t.cpu()
        return t.cpu().numpy()


def empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


def zeros(shape, dtype, dev):
    return torch.zeros(shape, dtype=dtype, device=dev)


_ws_cache = {}


def workspace(nbytes, dev, tag):
    """A cached per-(device, stream, op) scratch buffer, grown on demand (caller-owned scratch,
    like the reference's allocate_temp).  Keyed by the current stream as well: two streams running
    the same op concurrently must not share scratch."""
    if nbytes == 0:
        return None, 0
    if torch.cuda.is_current_stream_capturing():
        # inside a HIP-graph capture the scratch must belong to the graph's own memory pool and to
        # nobody else: a fresh buffer, not cached (a cached one would later be handed to eager calls)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        return buf, int(buf.numel())
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf, int(buf.numel())


def ptr(t):
    return None if t is None else t.data_ptr()


def stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def invalid(msg):
    """The reference raises tf.errors.InvalidArgumentError with this wording (OP_REQUIRES in
    the OpKernels); here it is a ValueError with the same text."""
    return ValueError(msg)


_replay_ok = {}


def graph_replay_ok(dev=None):
    """True when HIP graphs that contain small memset nodes -- i.e. graphs that contain torch reductions,
    which clear their semaphores with cudaMemsetAsync -- replay correctly in this process.

    On ROCm 7 a captured hipMemsetAsync of a small buffer replays garbage from the second launch on unless
    the runtime was STARTED with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; a captured `sum` / `max` then returns
    stale or wrong values on some later replay (tools/experiments/graph_bug_probe2.py, train_graph.py) --
    not reliably on the first few, so checking a captured graph's outputs is no substitute.
    Nothing in this package sets the switch at import: the host opts in with
    `rfnet_amd.enable_graph_safe_runtime()` before anything touches the HIP runtime (`torch.cuda.is_available()`
    already does).  This asks the runtime itself: a graph of
    [rf_probe_memset_async; += 1] replayed four times must read 1 every time (measured: True exactly when
    the switch took effect).  Callers that capture graphs holding torch reductions (trainrun.TrainStep) go
    eager when it says no.  Cached per device; not callable during a capture."""
    from ._lib import lib
    dev = torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)
    key = dev.index
    if key not in _replay_ok:
        ok = True
        try:
            with torch.cuda.device(dev), torch.no_grad():
                buf = torch.full((64,), 5.0, device=dev)

                def body():
                    rc = lib.rf_probe_memset_async(buf.data_ptr(), 256, torch.cuda.current_stream().cuda_stream)
                    if rc != 0:
                        raise RuntimeError(f"rf_probe_memset_async: {rc}")
                    buf.add_(1.0)

                cur = torch.cuda.current_stream()
                side = torch.cuda.Stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    body()
                cur.wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    body()
                for _ in range(4):
                    graph.replay()
                    torch.cuda.synchronize()
                    # (the reductions in between are part of the probe: ordinary launches between replays)
                    ok = ok and float(buf.min()) == 1.0 and float(buf.max()) == 1.0
                del graph
        except Exception:  # noqa: BLE001 -- a runtime that cannot capture at all
            ok = False
        _replay_ok[key] = ok
    return _replay_ok[key]
