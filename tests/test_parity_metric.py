"""The parity yardstick itself (oracle.yolov9_oracle.match_detections_strict / parity_summary): a detection counts as matched
only with the same class, IoU >= 0.9 AND all four coordinates within the pixel tolerance; unmatched rows within the score
tolerance of the 0.25 threshold are reported as borderline.  Synthetic detections, CPU only."""
import numpy as np

from oracle.yolov9_oracle import match_detections, match_detections_strict, parity_summary


def _dets(n, seed=0):
    rng = np.random.default_rng(seed)
    d = np.zeros((300, 6), np.float32)
    xy = rng.uniform(0, 300, (n, 2)); wh = rng.uniform(80, 300, (n, 2))
    d[:n, :2] = xy; d[:n, 2:4] = xy + wh
    d[:n, 4] = np.sort(rng.uniform(0.3, 0.9, n))[::-1]
    d[:n, 5] = np.arange(n) % 7                           # classes spread so that boxes of one class rarely overlap at IoU 0.9
    return d


def test_identical_detections_match_exactly():
    a = _dets(50)
    m = match_detections_strict(a, a.copy(), 0.64)
    assert (m["n_ref"], m["n_got"], m["n_iou"], m["n_strict"]) == (50, 50, 50, 50)
    assert m["box_err"].max() == 0 and m["score_err_max"] == 0 and m["borderline_ref"] == 0 == m["borderline_got"]


def test_box_tolerance_separates_small_errors_from_selection_flips():
    a = _dets(40, 1)
    b = a.copy()
    b[:40, :4] += np.float32(0.3)                          # a rigid 0.3 px shift: inside the 0.64 px bar
    b[3, 0] += 5.0                                         # one large box moved by 5 px in x: still IoU >= 0.9, NOT a small error
    assert (a[3, 2] - a[3, 0]) * (a[3, 3] - a[3, 1]) > 80 * 80
    m = match_detections_strict(a, b, 0.64)
    assert m["n_iou"] >= 39 and m["n_strict"] == m["n_iou"] - 1
    old = match_detections(a, b, 0.9)                      # the IoU-only matcher counts it as matched and reports the 5.3 px as "box error"
    assert old[2] == m["n_iou"] and old[3] > 5.0
    s = parity_summary(a[None], b[None], 0.64)
    assert abs(s["match_frac"] - m["n_strict"] / 40) < 1e-9 and s["box_err_px_max_strict"] <= 0.64 < s["box_err_px_max_iou_pairs"]


def test_rows_at_the_threshold_are_borderline_not_errors():
    a = _dets(30, 2)
    a[29, 4] = 0.2508                                      # a detection 8e-4 over the 0.25 threshold ...
    b = a.copy(); b[29] = 0                                # ... that the other side scored just under it (row zeroed, as postprocess does)
    c = _dets(1, 9); c[0, 4] = 0.2512; c[0, 5] = 6
    b[29] = c[0]                                           # and one the other side has, just over the threshold, the reference has not
    s = parity_summary(a[None], b[None], 0.64, score_margin=2e-3)
    assert s["n_ref"] == 30 and s["n_got"] == 30 and s["n_strict"] == 29
    assert s["borderline_ref"] == 1 and s["borderline_got"] == 1
    assert abs(s["match_frac"] - 29 / 30) < 1e-9 and s["match_frac_clear_of_threshold"] == 1.0
    t = parity_summary(a[None], b[None], 0.64, score_margin=1e-4)      # a tighter score tolerance: they count as misses
    assert t["borderline_ref"] == 0 and t["match_frac_clear_of_threshold"] == t["match_frac"]


def test_per_anchor_statistics_use_only_anchors_both_sides_score():
    rng = np.random.default_rng(3)
    dec_a = np.zeros((2, 100, 6), np.float32); dec_b = np.zeros((2, 100, 6), np.float32)
    dec_a[..., :4] = rng.uniform(0, 600, (2, 100, 4)); dec_b[..., :4] = dec_a[..., :4] + np.float32(0.25)
    dec_a[:, :40, 4] = 0.5; dec_b[:, 20:60, 4] = 0.5001     # anchors 20..39 are over the threshold on both sides
    dec_b[0, 25, 0] += 3.0
    s = parity_summary(np.zeros((2, 300, 6), np.float32), np.zeros((2, 300, 6), np.float32), 0.64, dec_a, dec_b)
    assert s["anchors_both_over_thr"] == 40 and abs(s["anchor_box_err_px_p50"] - 0.25) < 1e-3
    assert abs(s["anchor_box_err_px_max"] - 3.25) < 1e-3 and abs(s["anchor_score_err_max"] - 1e-4) < 1e-6


def test_tolerance_bars_levels():
    """tolerance_bars: detections (strict matches, scores), 99.9 % of the anchors within the tolerance, no anchor beyond 1.5x."""
    from oracle.yolov9_oracle import tolerance_bars
    rng = np.random.default_rng(4)
    a = np.stack([_dets(60, s) for s in range(4)])
    dec_a = np.zeros((4, 3000, 6), np.float32)
    dec_a[..., :4] = rng.uniform(0, 600, (4, 3000, 4)); dec_a[..., 4] = 0.5

    def bars(shift, outliers=(), n_big=0, score=0.0):
        b = a.copy(); b[..., :4] += np.float32(shift); b[:, :60, 4] += np.float32(score)
        dec_b = dec_a.copy(); dec_b[..., :4] += np.float32(shift); dec_b[..., 4] += np.float32(score)
        for i, e in enumerate(outliers):
            dec_b[0, i, 0] += np.float32(e)
        dec_b[1, :n_big, 1] += np.float32(0.7)
        return tolerance_bars(parity_summary(a, b, 0.64, dec_a, dec_b))

    assert bars(0.2)["all"]
    assert bars(0.2, outliers=(0.5, 0.6))["all"]                          # two anchors of 12000 at 0.7-0.8 px: inside the tail bound
    t = bars(0.2, outliers=(0.9,))
    assert not t["tail"] and t["anchors"] and t["detections"] and not t["all"]      # one anchor at 1.1 px
    t = bars(0.2, n_big=30)
    assert not t["anchors"] and t["tail"] and not t["all"]                # 0.25 % of the anchors at 0.9 px
    t = bars(0.7)
    assert not t["detections"] and not t["all"]                           # every row off by more than the tolerance
    assert not bars(0.2, score=3e-3)["detections"]                        # scores off by 3e-3
