"""CPU checks of the culled Chamfer sweep's host-side logic (no GPU): the workspace planner and the size rule of
RF_NN_AUTO.  (The 24-state Hilbert table of the rounds 1-2 register sort left the product with that order:
tools/experiments/nn_pruned_decided_knobs.patch.txt; larger clouds use Skilling's transform directly.)"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "experiments"))


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
                  (1, 65536, 65536), (256, 2048, 2048), (32, 512, 16384)]:  # (the last one since round 4: 0.060 vs 0.083 ms)
        assert takes_culled(*shape), shape
    for shape in [(4, 1024, 1024), (128, 1024, 1024), (32, 400, 16384), (2, 65536, 4096), (32, 3000, 64),
                  (1, 70000, 3000), (16, 700, 20000)]:
        assert not takes_culled(*shape), shape
