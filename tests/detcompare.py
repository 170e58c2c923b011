"""Comparison of two detection sets whose predictions differ in the last bits (shared by the GPU parity tests; the
logic itself is tested on the CPU in test_detcompare.py)."""
import numpy as np

from oracle import darknet_ref as O

TOL = 1e-4


def assert_detections_equivalent(d, gd, conf, thr, tol=TOL, eps_obj=2e-4, eps_iou=2e-3):
    """Two detection sets from forwards that differ in the last bits (GPU arithmetic vs the reference's CPU ops).
    Rows are matched by (image, class, box); every matched pair must agree within the coordinate / score tolerance,
    and every unmatched row must be explainable by a threshold-adjacent decision:
      * its objectness is within eps_obj of `conf` (the other side dropped it at the strict `>`), or
      * it overlaps a row of its (image, class) group with IoU within eps_iou of `thr` (a suppression that flipped), or
      * an unmatched row of the other set with HIGHER objectness overlaps it with IoU >= thr - eps_iou (it was suppressed
        there by a row that is itself a flipped decision: cascade)."""
    d = np.asarray(d, dtype=np.float32).reshape(-1, 8)
    gd = np.asarray(gd, dtype=np.float32).reshape(-1, 8)
    used = np.zeros(len(gd), bool)
    unmatched_d = []
    for i, r in enumerate(d):
        cand = np.where((gd[:, 0] == r[0]) & (gd[:, 7] == r[7]) & ~used)[0]
        best, best_err = -1, np.inf
        for j in cand:
            scale = max(1.0, float(np.abs(gd[j, 1:5]).max()))
            err = float(np.abs(r[1:5].astype(np.float64) - gd[j, 1:5]).max()) / scale
            if err < best_err:
                best, best_err = j, err
        if best >= 0 and best_err <= tol and np.abs(r[5:7].astype(np.float64) - gd[best, 5:7]).max() <= tol:
            used[best] = True
        else:
            unmatched_d.append(i)
    unmatched_g = list(np.where(~used)[0])

    def explain(row, own, other, other_unmatched):
        if abs(float(row[5]) - conf) <= eps_obj:
            return True
        for S in (own, other):
            grp = S[(S[:, 0] == row[0]) & (S[:, 7] == row[7])]
            if len(grp):
                iou = O.bbox_iou_np(row[None, 1:5], grp[:, 1:5])
                iou = iou[iou < 0.999999]                            # not itself
                if len(iou) and np.abs(iou - thr).min() <= eps_iou:
                    return True
        grp = other_unmatched[(other_unmatched[:, 0] == row[0]) & (other_unmatched[:, 7] == row[7])] if len(other_unmatched) else other_unmatched
        grp = grp[grp[:, 5] > row[5]] if len(grp) else grp                 # only a higher-objectness row can have suppressed it
        if len(grp) and (O.bbox_iou_np(row[None, 1:5], grp[:, 1:5]) >= thr - eps_iou).any():
            return True
        return False

    for i in unmatched_d:
        assert explain(d[i], d, gd, gd[unmatched_g]), f"GPU row {i} {d[i]} has no counterpart and is not threshold-adjacent"
    for j in unmatched_g:
        assert explain(gd[j], gd, d, d[unmatched_d]), f"reference row {j} {gd[j]} has no counterpart and is not threshold-adjacent"
    assert len(unmatched_d) + len(unmatched_g) <= max(4, len(gd) // 25), (len(unmatched_d), len(unmatched_g))
    return len(unmatched_d), len(unmatched_g)
