"""CPU checks of the culled Chamfer sweep's host-side logic (no GPU): the Hilbert state-machine
table compiled into nn_pruned.hip, the workspace planner and the size rule of RF_NN_AUTO."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))


def _lut():
    src = open(os.path.join(ROOT, "rfnet_amd", "csrc", "nn_pruned.hip")).read()
    m = re.search(r"kHilbertLut\[192\] = \{([^}]*)\}", src)  # (the two-level table is built from it in LDS)
    return np.array([int(v) for v in m.group(1).split(",")], dtype=np.int64).reshape(24, 8)


def test_hilbert_table_is_skillings_curve():
    """The 24-state octant table walks exactly Skilling's 3-D Hilbert curve (5 bits per axis): every
    one of the 32768 cells gets the index of the reference transform, and consecutive indices are
    face-adjacent cells (the property the culling's box tightness rests on)."""
    from cull_model_orders import hilbert_index
    lut = _lut()
    g = np.stack(np.meshgrid(np.arange(32), np.arange(32), np.arange(32), indexing="ij"), -1).reshape(-1, 3)
    st = np.zeros(len(g), np.int64)
    key = np.zeros(len(g), np.int64)
    for lvl in range(4, -1, -1):
        octant = (((g[:, 0] >> lvl) & 1) << 2) | (((g[:, 1] >> lvl) & 1) << 1) | ((g[:, 2] >> lvl) & 1)
        e = lut[st, octant]
        key = (key << 3) | (e & 7)
        st = e >> 3
    ref = hilbert_index(g, 5).astype(np.int64)
    assert np.array_equal(key, ref)
    order = np.argsort(key)
    assert len(np.unique(key)) == 32768
    assert np.abs(np.diff(g[order], axis=0)).sum(1).max() == 1


def test_workspace_planner_and_auto_rule():
    """rf_nn_distance_mode_workspace_bytes: the culled plan holds both sorted clouds and their boxes;
    RF_NN_AUTO sizes the workspace for the sweep it will take (rule measured in tools/ab_modes.py)."""
    from rfnet_amd._lib import lib
    AUTO, DENSE, CULLED = 0, 1, 2
    ws = lib.rf_nn_distance_mode_workspace_bytes

    def takes_culled(b, n, m):
        a, d, c = ws(b, n, m, AUTO), ws(b, n, m, DENSE), ws(b, n, m, CULLED)
        assert a in (d, c) and d != c
        return a == c

    # sorted xyz (12 B) + original index (4 B) per padded record, boxes 96 + 32 B per 64 records; a cloud
    # sorted by two workgroups (8192 < points <= 16384) carries one more superblock of padding
    b, n, m = 32, 2048, 16384
    need = b * (n + m) * 16 + b * ((n + m) // 64) * 128
    assert need <= ws(b, n, m, CULLED) <= need + b * (64 * 16 + 128) + 16 * 1024
    assert ws(1, 70000, 100, CULLED) == 0  # beyond the culled sweep's 65536-point limit
    assert ws(0, 5, 5, AUTO) == 0
    for shape in [(32, 2048, 16384), (32, 16384, 16384), (1, 4096, 4096), (4, 3000, 16384), (32, 3000, 1024),
                  (1, 65536, 65536), (256, 2048, 2048)]:
        assert takes_culled(*shape), shape
    for shape in [(4, 1024, 1024), (128, 1024, 1024), (32, 512, 16384), (2, 65536, 4096), (32, 3000, 64),
                  (1, 70000, 3000), (16, 700, 20000)]:
        assert not takes_culled(*shape), shape
