"""Host-thread safety of the C ABI ("thread-safe, no globals"; TensorFlow calls an op's Compute from any
inter-op thread, /root/reference's tf_ops/CD/tf_nndistance.cpp:172-204 is stateless).  Four Python threads, each with its own
HIP stream, inputs, outputs and workspace, hammer rf_nn_distance, rf_approxmatch + rf_matchcost, rf_farthestpointsampling and
rf_queryballpoint_boxes through ctypes (which releases the GIL for the duration of a call) while a fifth toggles the
per-kernel profiling hook and drains it: every result must be bit-equal to the same call made alone beforehand."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _make_case(seed):
    rng = np.random.RandomState(seed)
    dev = "cuda"
    case = {
        "a": torch.from_numpy(rng.randn(4, 2500 + 64 * seed, 3).astype(np.float32)).to(dev),
        "c": torch.from_numpy(rng.randn(4, 5000, 3).astype(np.float32)).to(dev),
        "u": torch.from_numpy((rng.rand(2, 600, 3) - 0.5).astype(np.float32)).to(dev),
        "v": torch.from_numpy((rng.rand(2, 500 + 16 * seed, 3) - 0.5).astype(np.float32)).to(dev),
        "cloud": torch.from_numpy(rng.rand(4, 4096, 3).astype(np.float32)).to(dev),
    }
    return case


def _run_all(lib, case, stream):
    """One pass over the four ops on `stream` with freshly allocated outputs / workspaces -> list of output tensors."""
    from rfnet_amd._lib import check
    s = C.c_void_p(stream.cuda_stream)
    outs = []
    with torch.cuda.stream(stream):
        a, c = case["a"], case["c"]
        b, n, m = a.shape[0], a.shape[1], c.shape[1]
        d1 = torch.empty(b, n, device="cuda"); i1 = torch.empty(b, n, dtype=torch.int32, device="cuda")
        d2 = torch.empty(b, m, device="cuda"); i2 = torch.empty(b, m, dtype=torch.int32, device="cuda")
        wsz = lib.rf_nn_distance_workspace_bytes(b, n, m)
        ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device="cuda")
        check(lib.rf_nn_distance(b, n, m, _p(a), _p(c), _p(d1), _p(i1), _p(d2), _p(i2), _p(ws), wsz, s), "rf_nn_distance")
        outs += [d1, i1, d2, i2]
        u, v = case["u"], case["v"]
        b, n, m = u.shape[0], u.shape[1], v.shape[1]
        match = torch.empty(b, m, n, device="cuda")
        wsz = lib.rf_approxmatch_workspace_bytes(b, n, m, 10)
        ws2 = torch.empty(max(wsz, 16), dtype=torch.uint8, device="cuda")
        check(lib.rf_approxmatch(b, n, m, _p(u), _p(v), _p(match), _p(ws2), wsz, s), "rf_approxmatch")
        cost = torch.empty(b, device="cuda")
        wsz3 = lib.rf_matchcost_workspace_bytes(b, n, m)
        ws3 = torch.empty(max(wsz3, 16), dtype=torch.uint8, device="cuda")
        check(lib.rf_matchcost(b, n, m, _p(u), _p(v), _p(match), _p(cost), _p(ws3), wsz3, s), "rf_matchcost")
        outs += [match, cost]
        x = case["cloud"]
        b, n = x.shape[0], x.shape[1]
        fi = torch.empty(b, 128, dtype=torch.int32, device="cuda")
        check(lib.rf_farthestpointsampling(b, n, 128, _p(x), None, _p(fi), s), "rf_farthestpointsampling")
        qp = torch.empty(b, 128, 3, device="cuda")
        check(lib.rf_gatherpoint(b, n, 128, _p(x), _p(fi), _p(qp), s), "rf_gatherpoint")
        bi = torch.zeros(b, 128, 16, dtype=torch.int32, device="cuda")
        bc = torch.empty(b, 128, dtype=torch.int32, device="cuda")
        wsz4 = lib.rf_queryballpoint_boxes_workspace_bytes(b, n)
        ws4 = torch.empty(wsz4, dtype=torch.uint8, device="cuda")
        check(lib.rf_queryballpoint_boxes(b, n, 128, C.c_float(0.12), None, 16, _p(x), _p(qp), None, _p(bi), _p(bc), _p(ws4),
                                          wsz4, s), "rf_queryballpoint_boxes")
        outs += [fi, bi, bc]
        keep = [ws, ws2, ws3, ws4, qp]  # alive until the stream has drained
    stream.synchronize()
    del keep
    return outs


def test_four_host_threads_and_a_profiler_toggle():
    from rfnet_amd import _lib
    lib = _lib.lib
    nthreads, rounds = 4, 6
    cases = [_make_case(k) for k in range(nthreads)]
    torch.cuda.synchronize()
    solo = [_run_all(lib, cases[k], torch.cuda.Stream()) for k in range(nthreads)]
    results = [[] for _ in range(nthreads)]
    errors = []
    stop = threading.Event()
    start = threading.Barrier(nthreads + 1)

    def worker(k):
        try:
            st = torch.cuda.Stream()
            start.wait()
            for _ in range(rounds):
                results[k].append(_run_all(lib, cases[k], st))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    def toggler():
        start.wait()
        flip = False
        while not stop.is_set():
            flip = not flip
            _lib.profile_enable(flip)
            _lib.profile_collect()
        _lib.profile_enable(False)
        _lib.profile_collect()

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(nthreads)]
    tg = threading.Thread(target=toggler)
    for t in ths + [tg]:
        t.start()
    for t in ths:
        t.join(timeout=120)
    stop.set()
    tg.join(timeout=30)
    assert not errors, errors
    assert all(not t.is_alive() for t in ths), "a worker thread hung"
    for k in range(nthreads):
        assert len(results[k]) == rounds
        for r, outs in enumerate(results[k]):
            for j, (x, y) in enumerate(zip(outs, solo[k])):
                assert torch.equal(x, y), f"thread {k} round {r} output {j} differs from the single-threaded call"
