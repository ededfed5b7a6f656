"""CPU check of the compiled kernels' resource figures (hipcc -Rpass-analysis=kernel-resource-usage, kept by
rfnet_amd/build.py next to the objects): no kernel of the library touches scratch, and the hot kernels keep the residency
they were tuned for.  Round 4 found a batched re-scan and a histogram loop that had pushed two kernels into scratch without
any test noticing (profiles/r04_rescan.txt); a spill inside nnp_sweep's traversal costs 10 % of the headline."""
from rfnet_amd.build import kernel_resources


def _find(res, *parts):
    hits = {k: v for k, v in res.items() if all(p in k for p in parts)}
    assert hits, parts
    return hits


def test_no_kernel_uses_scratch():
    res = kernel_resources()
    assert len(res) >= 60  # every .hip of csrc reported
    bad = {k: v for k, v in res.items() if v.get("scratch", 0) or v.get("vgpr_spill", 0)}
    assert not bad, bad


def test_hot_kernels_keep_their_residency():
    res = kernel_resources()
    # the culled sweep: 7 waves per SIMD (RFP_WPE; 8 spills into the traversal loop and measures slower)
    for k, v in _find(res, "nnp_sweep_kernel").items():
        assert v["occupancy"] == 7 and v["vgprs"] <= 72, (k, v)
    # the register-resident sort: 1024 threads = 4 waves per SIMD = 128 registers at most
    for k, v in _find(res, "nnp_sort_reg_kernel").items():
        assert v["vgprs"] <= 128 and v["occupancy"] >= 4, (k, v)
    for k, v in _find(res, "nnp_grad_sorted_kernel").items():
        assert v["occupancy"] == 8, (k, v)
    # match_cost_grad: the 40 KB tile allows 4 workgroups per CU; 32 prefetched rows must fit 128 registers
    for k, v in _find(res, "mcg_kernel").items():
        assert v["vgprs"] <= 128 and v["lds"] <= 40 * 1024 + 256, (k, v)
    # approx_match's level sweeps and the match kernel: full residency or one below it
    for k, v in {**_find(res, "am_rowk_kernel"), **_find(res, "am_rowl_kernel"), **_find(res, "am_match_kernel")}.items():
        assert v["occupancy"] >= 7, (k, v)
    # the dense sweep's own-side register blocking: forward-only form at 7 waves
    for k, v in _find(res, "nn_sweep_kernel", "Lb0E").items():
        assert v["occupancy"] >= 7, (k, v)
