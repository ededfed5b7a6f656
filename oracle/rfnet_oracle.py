"""TEST INFRASTRUCTURE ONLY -- numpy float64 restatement of the RFNet generator graph and of the
training loss block of the reference, for the C5 / row-f2 parity tests.  Never imported by the
product (rfnet_amd/), only by tests/.

Restates, statement by statement, /root/reference/vv_recon.py:
    conv2d :47-65 (1x1 convolution = dense layer on the channel axis + bias [+ activation])
    sampling :67-83, global_mlp :84-91, encode_cell :93-112, recover_cell :124-131,
    merge_layer :132-139, init_move_layer :140-159, feat_trans :160-163, re_chamfer :171-193,
    full_process :194-244, init_decode_layer :246-272, refine_layer :273-308, decode_cell :310-364,
    chamfer_big :381-385, earth_mover :392-399, groupin_near / zero_groupnear :405-419,
    the loss block of train() :474-500.
Tensors are (batch, points, channels) float64 (the reference's (batch, points, 1, channels) with
the unit axis dropped).  The point-cloud operators inside the graph are the C oracle's
(oracle/rfops_oracle.c: FPS, gather, nn_distance, approx_match, match_cost -- fp32, the reference
CUDA ops' arithmetic); everything dense is float64 here, so a comparison against the fp32 GPU
graph carries fp32 GEMM rounding: the tests use rel 1e-4.

Variables come in as {TensorFlow variable name: array} with the reference's names (checked against
the checkpoint index, tests/golden/rfnet_variables.json): kernels `<scope>/<layer>/weights`
[1,1,cin,cout] shared between re-applications of a cell (tf.get_variable under reuse=True), biases
`<scope>[_k]/<layer>/Variable` fresh per application (tf.Variable lands in a uniquified name
scope: cell, cell_1, cell_2; decode_cell, decode_cell_1) -- vv_recon.py:34-43.

Index-producing operators (FPS, the Chamfer argmin inside merge_layer) are discontinuous: a last-bit
difference in a coordinate can flip an index and move a point by a finite amount.  `forward` can
therefore be handed the indices the GPU run took (`shared`), and always reports how many of its own
indices agree with them, so that the dense arithmetic is compared on the same discrete choices.

Parity status: this file restates the reference graph from its source; TensorFlow is not
installable here and the checkpoint's weight blob is missing, so it is NOT
pinned against an execution of the reference -- "parity unpinned by a reference run".  What pins it:
the variable inventory (names, shapes, sharing quirk) equals the reference checkpoint index.
"""
import numpy as np


def _relu(x):
    return np.maximum(x, 0.0)


def _leaky_relu(x):  # tf.nn.leaky_relu default alpha = 0.2 (vv_recon.py:351)
    return np.where(x > 0, x, 0.2 * x)


_ACT = {"relu": _relu, "tanh": np.tanh, "leaky_relu": _leaky_relu, None: lambda x: x}


class RFNetOracle:
    def __init__(self, variables, orc):
        """variables: {tf name: array}; orc: oracle.oracle.Oracle (the C operator oracle)."""
        self.v = {k: np.asarray(v, np.float64) for k, v in variables.items()}
        self.orc = orc
        self.used = set()

    # ---- conv2d, vv_recon.py:47-65 (kernel [1,1,cin,cout], padding VALID, stride 1) ----------
    def conv(self, scope, name, x, act="relu", call=0):
        base = f"{scope}/{name}" if scope else name
        sc = scope if call == 0 else f"{scope}_{call}"
        bname = f"{sc}/{name}" if sc else name
        w = self.v[base + "/weights"][0, 0]
        b = self.v[bname + "/Variable"]
        self.used.update((base + "/weights", bname + "/Variable"))
        assert x.shape[-1] == w.shape[0], (base, x.shape, w.shape)
        return _ACT[act](x @ w + b)

    # ---- :84-91 --------------------------------------------------------------------------------
    def global_mlp(self, scope, xyz, mlp):
        t = xyz
        for i, _ in enumerate(mlp):
            t = self.conv(scope, f"ini_layer{i}", t)
        return t.max(axis=1, keepdims=True)

    # ---- :93-112 -------------------------------------------------------------------------------
    def encode_cell(self, input_tensor, state_tensor, call, mlp=(256, 384), mlpout=(256, 256)):
        n = input_tensor.shape[1]
        new_state = np.concatenate([input_tensor, np.repeat(state_tensor, n, axis=1)], -1)  # :101
        for i, _ in enumerate(mlp):
            new_state = self.conv("cell", f"state{i}", new_state, call=call)
        new_state = self.conv("cell", "state_end", new_state, call=call).max(axis=1, keepdims=True)  # :106-107
        codeout = new_state
        for i, _ in enumerate(mlpout):
            codeout = self.conv("cell", f"codemlp{i}", codeout, call=call)
        return codeout, new_state

    # ---- :124-131 (the output layer's name uses the loop variable AFTER the loop: recover2out1) -
    def recover_cell(self, scope, input_tensor, con_tensor, mlp2=(256, 256)):
        n = con_tensor.shape[1]
        t = np.concatenate([np.repeat(input_tensor, n, axis=1), con_tensor], -1)
        i = 0
        for i, _ in enumerate(mlp2):
            t = self.conv(scope, f"recover2{i}", t)
        t = t.max(axis=1, keepdims=True)
        return self.conv(scope, f"recover2out{i}", t, act=None)

    # ---- :132-139 ------------------------------------------------------------------------------
    def merge_layer(self, rawpts, newpts, decfactor, shared_idx2=None, report=None):
        own = self.orc.nn_distance(rawpts.astype(np.float32), newpts.astype(np.float32))[3]  # idx2, :134
        if report is not None:
            report.append(1.0 if shared_idx2 is None else float((own == shared_idx2).mean()))
        idx2 = own if shared_idx2 is None else shared_idx2
        grouped = np.take_along_axis(rawpts, idx2[..., None].astype(np.int64), 1)  # group_point, nsample 1 :135
        diff = grouped - newpts
        dismat = (diff * diff).sum(-1, keepdims=True)  # :136
        ratio = np.exp(-dismat / (1e-8 + float(decfactor) ** 2))  # :137
        return newpts + ratio * diff  # :138

    # ---- :140-159 ------------------------------------------------------------------------------
    def init_move_layer(self, startpts, codeword, mlp=(256, 256, 256), mlp1=(256, 128), mlp2=(256, 128, 64)):
        n = startpts.shape[1]
        tensor1 = np.concatenate([startpts, np.repeat(codeword.reshape(-1, 1, codeword.shape[-1]), n, 1)], -1)
        t = tensor1
        for i, _ in enumerate(mlp):
            t = self.conv("", f"ini_layer{i}", t)
        t = np.concatenate([tensor1, np.repeat(t.max(axis=1, keepdims=True), n, 1)], -1)
        outfeats = t
        for i, _ in enumerate(mlp1):
            outfeats = self.conv("", f"ini_featout{i}", outfeats)
        outfeats = self.conv("", "inimove_featout", outfeats)
        for i, _ in enumerate(mlp2):
            t = self.conv("", f"ini_ptsout{i}", t)
        outpts = startpts + self.conv("", "inimove_ptsout", t, act="tanh")
        return outpts, outfeats

    # ---- :160-163 ------------------------------------------------------------------------------
    def feat_trans(self, feat, mlp=(256, 256)):
        for i, _ in enumerate(mlp):
            feat = self.conv("", f"partfeat{i}", feat)
        return feat

    # ---- :246-272 with state_tensor=None ---------------------------------------------------------
    def init_decode_layer(self, input_tensor, ptnum=32, mlp=(256, 256), mlp2=(256, 256)):
        sc = "init_cell"
        new_state = self.conv(sc, "input_trans", input_tensor)
        for i, _ in enumerate(mlp):
            new_state = self.conv(sc, f"basic_state{i}", new_state)
        po = self.conv(sc, "points_out", new_state, act=None)  # (B,1,3*ptnum+12)
        transmat = po[..., -12:-3].reshape(-1, 3, 3)
        movemat = po[..., -3:].reshape(-1, 1, 3)
        pts = np.tanh(po[..., :-12]).reshape(-1, ptnum, 3)
        pts = pts @ transmat + movemat  # :260
        so = self.conv(sc, "state_out", new_state).reshape(-1, ptnum, 16)
        so = np.concatenate([so, np.repeat(new_state, ptnum, 1)], -1)
        for i, _ in enumerate(mlp2):
            so = self.conv(sc, f"state{i}", so)
        return pts, self.conv(sc, "state_outo", so)

    # ---- :273-308 ------------------------------------------------------------------------------
    def refine_layer(self, scope, ptcoor, feat, feat2, mlp=(128, 64, 64), mlp2=(128, 128), mlpself=(128, 128)):
        n = ptcoor.shape[1]
        t = np.concatenate([ptcoor, np.repeat(feat, n, 1)], -1)
        for i, _ in enumerate(mlpself):
            t = self.conv(scope, f"ini_layer{i}", t)
        t = np.concatenate([ptcoor, np.repeat(t.max(axis=1, keepdims=True), n, 1)], -1)
        for i, _ in enumerate(mlp):
            t = self.conv(scope, f"refine_layers{i}", t)
        newvec = self.conv(scope, "refine_layer_final", t, act="tanh")
        newcoor = newvec + ptcoor
        t = np.concatenate([newcoor, feat2, np.repeat(feat, feat2.shape[1], 1)], -1)
        for i, _ in enumerate(mlp2):
            t = self.conv(scope, f"feat_refine{i}", t)
        newfeat = self.conv(scope, "feat_refine_final", t, act="tanh")
        return newcoor, newfeat + feat2, newvec

    # ---- :310-364 ------------------------------------------------------------------------------
    def decode_cell(self, input_tensor, center, state_tensor, call, up_ratio=16, mlp=(256, 256), mlp1=(128, 64),
                    mlp2=(128, 128), mlp_mask=(128, 128), mlp_expand=(128,)):
        sc = "decode_cell"
        n = state_tensor.shape[1]
        mask = np.concatenate([center, np.repeat(input_tensor, n, 1)], -1)
        for i, _ in enumerate(mlp_mask):
            mask = self.conv(sc, f"mlp_mask{i}", mask, call=call)
        mask = self.conv(sc, "mask_tensor", mask, call=call)  # relu, :320
        input_info = self.conv(sc, "input_trans", mask * input_tensor, call=call)
        state_info = self.conv(sc, "state_trans", state_tensor, call=call)
        new_state = np.concatenate([input_info, state_info], -1)
        for i, _ in enumerate(mlp):
            new_state = self.conv(sc, f"basic_state{i}", new_state, call=call)
        po = new_state
        for i, _ in enumerate(mlp1):
            po = self.conv(sc, f"points{i}", po, call=call)
        po = self.conv(sc, "points_out", po, act="tanh", call=call)
        points_move = po.reshape(-1, n, up_ratio, 3)
        points_out = (center[:, :, None, :] + points_move).reshape(-1, n * up_ratio, 3)
        new_state = np.concatenate([new_state, np.repeat(input_tensor, n, 1)], -1)
        for i, _ in enumerate(mlp2):
            new_state = self.conv(sc, f"state{i}", new_state, call=call)
        newnew = new_state
        parts = []
        for i in range(up_ratio):  # a chain: expansion i feeds expansion i+1 (:347-354)
            for j, _ in enumerate(mlp_expand):
                newnew = self.conv(sc, f"state_expand{i}_{j}", newnew, call=call)
            newnew = self.conv(sc, f"state_expand{i}", newnew, act="leaky_relu", call=call)
            parts.append(newnew)
        state_move = np.stack(parts, 2)  # (B, n, up_ratio, state_len)
        new_state = (state_tensor[:, :, None, :] + state_move).reshape(-1, n * up_ratio, state_tensor.shape[-1])
        return points_out, new_state, points_move

    # ---- :194-244 ------------------------------------------------------------------------------
    def forward(self, pointcloud, shared=None):
        """-> dict with points1, points2, points3, points_final and the collections the loss block
        reads (points1_pre = 'points1', points2_pre = 'points2', refinemove3, decode_move64/1024),
        plus `agreement`: fraction of this oracle's own indices equal to the shared ones per
        index-producing op (1.0 where nothing was shared)."""
        shared = shared or {}
        agree = {}
        pc32 = np.ascontiguousarray(pointcloud, np.float32)
        pc = pc32.astype(np.float64)
        state0 = self.global_mlp("init_mlp", pc, (64, 128, 256))
        code1, state = self.encode_cell(pc, state0, 0)
        code1 = self.recover_cell("recover1", code1, pc)
        fidx = self.orc.farthest_point_sample(32, pc32)  # sampling(32, pointcloud, 'f'), :204
        if "fps32" in shared:
            agree["fps32"] = float((fidx == shared["fps32"]).mean())
            fidx = shared["fps32"]
        start = np.take_along_axis(pc, fidx[..., None].astype(np.int64), 1)
        points1, dstate = self.init_move_layer(start, code1)
        partfeat = self.global_mlp("part_mlp", np.concatenate([pc, points1], 1), (64, 128, 256))
        points0, dstate0 = self.init_decode_layer(self.feat_trans(np.concatenate([partfeat, code1], -1)))
        points1, dstate = np.concatenate([points0, points1], 1), np.concatenate([dstate0, dstate], 1)
        points1_pre = points1  # collection 'points1', :210
        rep = []
        points1 = self.merge_layer(pc, points1, self.v["decline_factor0"][0], shared.get("merge1"), rep)
        points1, dstate, _ = self.refine_layer("refine_layer1", points1, code1, dstate)

        pin = np.concatenate([pc, points1], 1)
        code2, state = self.encode_cell(pin, state, 1)
        code2 = code1 + self.recover_cell("recover2", code2, pin)
        points2, dstate, move64 = self.decode_cell(code2, points1, dstate, 0)
        points2_pre = points2  # collection 'points2', :222
        points2 = self.merge_layer(pc, points2, self.v["decline_factor1"][0], shared.get("merge2"), rep)
        points2, dstate, _ = self.refine_layer("refine_layer2", points2, code2, dstate)

        pin = np.concatenate([pc, points2], 1)
        code3, state = self.encode_cell(pin, state, 2)
        code3 = code2 + self.recover_cell("recover3", code3, pin)
        points3, dstate, move1024 = self.decode_cell(code3, points2, dstate, 1)
        final = self.merge_layer(pc, points3, self.v["decline_factor"][0], shared.get("merge3"), rep)
        final, _, refinemove3 = self.refine_layer("refine_layer_final", final, code3, dstate)
        agree.update({"merge1": rep[0], "merge2": rep[1], "merge3": rep[2]})
        self.used.update(("decline_factor0", "decline_factor1", "decline_factor"))
        return {"points1": points1, "points2": points2, "points3": points3, "points_final": final,
                "points1_pre": points1_pre, "points2_pre": points2_pre, "refinemove3": refinemove3,
                "decode_move64": move64, "decode_move1024": move1024, "agreement": agree}

    # ---- loss helpers ----------------------------------------------------------------------------
    def chamfer_big(self, pcd1, pcd2):  # :381-385
        d1, _, d2, _ = self.orc.nn_distance(pcd1.astype(np.float32), pcd2.astype(np.float32))
        return (np.sqrt(d1.astype(np.float64)).mean() + np.sqrt(d2.astype(np.float64)).mean()) / 2

    def earth_mover(self, pcd1, pcd2):  # :392-399
        a, c = pcd1.astype(np.float32), pcd2.astype(np.float32)
        cost = self.orc.match_cost(a, c, self.orc.approx_match(a, c)).astype(np.float64)
        return (cost / float(pcd1.shape[1])).mean()

    def re_chamfer(self, gt, pred, part=8):  # :171-193
        interval = int(gt.shape[1] / 8)
        return sum(self.chamfer_big(pred[:, i * interval:(i + 1) * interval], gt[:, i * interval:(i + 1) * interval])
                   for i in range(part)) / part

    def zero_groupnear(self, ptcens, rawpts, outmat):  # :405-419
        dist = self.orc.nn_distance(ptcens.astype(np.float32), rawpts.astype(np.float32))[2].astype(np.float64)
        outval = (outmat * outmat).sum(-1).mean(-1).mean(-1).mean()
        return max(outval - 0.4 * dist.mean(), 0.0)

    def training_loss(self, out, gt, shared=None, alpha1=0.01):
        """train()'s loss, vv_recon.py:474-500, at global_step 0 (alpha1 = 0.01, :482-483)."""
        shared = shared or {}
        gt32 = np.ascontiguousarray(gt, np.float32)
        g = gt32.astype(np.float64)
        i64 = shared.get("gt_fps64", self.orc.farthest_point_sample(64, gt32))      # :474
        i1024 = shared.get("gt_fps1024", self.orc.farthest_point_sample(1024, gt32))  # :475
        gt1 = np.take_along_axis(g, i64[..., None].astype(np.int64), 1)
        gt2 = np.take_along_axis(g, i1024[..., None].astype(np.int64), 1)
        terms = {}
        terms["cd1"] = self.earth_mover(gt1, out["points1_pre"])      # :489
        terms["cd2"] = self.earth_mover(gt2, out["points2_pre"])      # :490
        terms["cd3"] = self.chamfer_big(g, out["points3"])            # :491
        terms["cd4"] = self.chamfer_big(g, out["points_final"])       # :492 (= chamfer_loss :484)
        terms["recd3"] = self.re_chamfer(g, out["points3"], 8)        # :493
        terms["moveloss"] = (out["refinemove3"] ** 2).sum(-1).mean()  # :487-488
        terms["loss_d1"] = 0.05 * self.zero_groupnear(gt1, gt2, out["decode_move64"])    # :497
        terms["loss_d2"] = 0.05 * self.zero_groupnear(gt2, g, out["decode_move1024"])    # :498
        terms["loss_dec"] = sum(float(self.v[k][0]) ** 2 for k in ("decline_factor0", "decline_factor1", "decline_factor"))
        terms["loss"] = (0.2 * (terms["cd1"] + terms["cd2"]) + terms["cd3"] + terms["cd4"] + 0.2 * terms["recd3"]
                         + 0.1 * terms["moveloss"] + terms["loss_d1"] + terms["loss_d2"] + alpha1 * terms["loss_dec"])
        return terms
