"""Which op, captured into a HIP graph with input A and replayed with input B, differs from eager on B?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from rfnet_amd import _raw as R, glue
from rfnet_amd.rfnet import RFNet, linear_relu
rng = np.random.RandomState(0)
def t(*s): return torch.from_numpy((rng.rand(*s) - 0.5).astype(np.float32)).cuda()
def check(name, fn, make):
    a, b = make(), make()
    static = [x.clone() for x in a]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        fn(*static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        out = fn(*static)
    out = out if isinstance(out, (tuple, list)) else (out,)
    res = []
    for rep in range(4):
        b = make()
        for x, y in zip(static, b): x.copy_(y)
        g.replay(); torch.cuda.synchronize()
        with torch.no_grad():
            ref = fn(*b)
        ref = ref if isinstance(ref, (tuple, list)) else (ref,)
        res.append(all(torch.equal(o, r) for o, r in zip(out, ref) if o is not None))
    print(f"{name:40s} {res}")
check("fps 3000->32", lambda x: R.farthest_point_sample(32, x), lambda: [t(1, 3000, 3)])
check("gather", lambda x: R.gather_point(x, R.farthest_point_sample(32, x)), lambda: [t(1, 3000, 3)])
check("nn_distance dense 3000x64", lambda x, y: R.nn_distance(x, y), lambda: [t(1, 3000, 3), t(1, 64, 3)])
check("nn_distance 3000x16384", lambda x, y: R.nn_distance(x, y), lambda: [t(1, 3000, 3), t(1, 16384, 3)])
check("nn_distance culled", lambda x, y: R.nn_distance(x, y, mode="culled"), lambda: [t(1, 3000, 3), t(1, 16384, 3)])
check("nn_distance_dir dir2", lambda x, y: R.nn_distance_dir(x, y, False, True)[2:], lambda: [t(1, 3000, 3), t(1, 16384, 3)])
dec = torch.tensor([0.1], device="cuda")
check("merge_layer fused 64", lambda x, y: glue.merge_layer(x, y, dec), lambda: [t(1, 3000, 3), t(1, 64, 3)])
check("merge_layer fused 16384", lambda x, y: glue.merge_layer(x, y, dec), lambda: [t(1, 3000, 3), t(1, 16384, 3)])
check("merge_layer + handle", lambda x, y: glue.merge_layer(x, y, dec, sorted_raw=glue.sort_if_large(x)), lambda: [t(1, 3000, 3), t(1, 16384, 3)])
dec = torch.tensor([0.1], device="cuda")
w, b = t(259, 128), t(128)
check("linear_relu", lambda x: linear_relu(x, w, b), lambda: [t(1, 3000, 259)])
w3, r = t(3, 128), t(1, 1, 128)
check("point_affine", lambda p: R.point_affine(None, p, w3, r, "relu"), lambda: [t(1, 3000, 3)])
torch.manual_seed(0); net = RFNet().cuda().eval()
check("RFNet forward", lambda x: net(x), lambda: [t(1, 3000, 3)])
check("amax", lambda x: x.amax(1, keepdim=True), lambda: [t(1, 3000, 128)])
# ---- bisect the RFNet graph: cells captured alone
code = t(1, 1, 256)
check("global_mlp", lambda x: net.global_mlp("init_mlp", x), lambda: [t(1, 3000, 3)])
check("encode_cell", lambda x, st: net.encode_cell(x, st, 0), lambda: [t(1, 3000, 3), t(1, 1, 256)])
check("recover_cell", lambda c, x: net.recover_cell("recover1", c, x), lambda: [t(1, 1, 256), t(1, 3000, 3)])
check("init_move_layer", lambda s, c: net.init_move_layer(s, c), lambda: [t(1, 32, 3), t(1, 1, 256)])
check("init_decode_layer", lambda f: net.init_decode_layer(f), lambda: [t(1, 1, 256)])
check("refine_layer 64", lambda p, f, f2: net.refine_layer("refine_layer1", p, f, f2), lambda: [t(1, 64, 3), t(1, 1, 256), t(1, 64, 128)])
check("refine_layer 16384", lambda p, f, f2: net.refine_layer("refine_layer_final", p, f, f2), lambda: [t(1, 16384, 3), t(1, 1, 256), t(1, 16384, 128)])
check("decode_cell 64", lambda c, p, s: net.decode_cell(c, p, s, 0), lambda: [t(1, 1, 256), t(1, 64, 3), t(1, 64, 128)])
check("decode_cell 1024", lambda c, p, s: net.decode_cell(c, p, s, 1), lambda: [t(1, 1, 256), t(1, 1024, 3), t(1, 1024, 128)])
check("sampling 32", lambda x: glue.sampling(32, x)[1], lambda: [t(1, 3000, 3)])
import torch.nn.functional as F
for (n, k, c) in ((3000, 3, 64), (3000, 64, 128), (3000, 128, 256), (3000, 256, 384), (4024, 259, 256), (16384, 128, 128)):
    wk, bk = t(k, c), t(c)
    check(f"addmm_activation {n}x{k}x{c}", lambda x: torch._addmm_activation(bk, x.reshape(-1, k), wk), lambda: [t(1, n, k)])
    check(f"F.linear+relu    {n}x{k}x{c}", lambda x: F.relu(F.linear(x, wk.t(), bk)), lambda: [t(1, n, k)])
    check(f"matmul           {n}x{k}x{c}", lambda x: x @ wk, lambda: [t(1, n, k)])
w0, b0, w1, b1, w2, b2 = t(3, 64), t(64), t(64, 128), t(128), t(128, 256), t(256)
def chain(x):
    y = torch._addmm_activation(b0, x.reshape(-1, 3), w0)
    y = torch._addmm_activation(b1, y, w1)
    y = torch._addmm_activation(b2, y, w2)
    return y
check("chain of 3 addmm_activation", chain, lambda: [t(1, 3000, 3)])
check("chain + amax", lambda x: chain(x).reshape(1, 3000, 256).amax(1, keepdim=True), lambda: [t(1, 3000, 3)])
def chain2(x):
    y = torch._addmm_activation(b0, x.reshape(-1, 3), w0)
    return torch._addmm_activation(b1, y, w1)
check("chain of 2", chain2, lambda: [t(1, 3000, 3)])
def chainF(x):
    y = F.relu(F.linear(x, w0.t(), b0)); y = F.relu(F.linear(y, w1.t(), b1)); return F.relu(F.linear(y, w2.t(), b2))
check("chain of 3 F.linear+relu", chainF, lambda: [t(1, 3000, 3)])
check("net.d x1", lambda x: net.d("init_mlp", "ini_layer0", x), lambda: [t(1, 3000, 3)])
check("net.mlp x2", lambda x: net.mlp("init_mlp", "ini_layer", 2, x), lambda: [t(1, 3000, 3)])
print("---- amax combos")
check("1 addmm_act + amax", lambda x: torch._addmm_activation(b0, x.reshape(-1, 3), w0).reshape(1, 3000, 64).amax(1, keepdim=True), lambda: [t(1, 3000, 3)])
check("matmul + amax", lambda x: (x @ w0).amax(1, keepdim=True), lambda: [t(1, 3000, 3)])
check("relu + amax (no gemm)", lambda x: torch.relu(x).amax(1, keepdim=True), lambda: [t(1, 3000, 64)])
check("chainF + amax", lambda x: chainF(x).amax(1, keepdim=True), lambda: [t(1, 3000, 3)])
check("chain + max.values", lambda x: chain(x).reshape(1, 3000, 256).max(1, keepdim=True).values, lambda: [t(1, 3000, 3)])
check("chain + sum", lambda x: chain(x).reshape(1, 3000, 256).sum(1, keepdim=True), lambda: [t(1, 3000, 3)])
check("chain + clone + amax", lambda x: chain(x).reshape(1, 3000, 256).clone().amax(1, keepdim=True), lambda: [t(1, 3000, 3)])
check("amax of 256 ch", lambda x: x.amax(1, keepdim=True), lambda: [t(1, 3000, 256)])
check("amax 16384x128", lambda x: x.amax(1, keepdim=True), lambda: [t(1, 16384, 128)])
