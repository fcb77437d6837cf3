"""Shared parity helpers: golden loading and the comparison contract of SURVEY.md 8(c).

Contract (written out so the tolerances are visible in the tests):
  * top-k VALUES: |v - v_ref| <= VAL_ATOL (float32 dot products of unit vectors, any add order).
  * top-k INDEX SETS: equal on rows whose (k-th, k+1-th) similarity gap exceeds GAP_TOL and that
    hold no exact ties at the boundary; other rows are compared by value only.
  * vote SCORES: relative 1e-5 (+ tiny absolute floor for zero-score classes).
  * top-1 LABEL: equal wherever the reference's top-2 score gap exceeds 1e-5 relative.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KNN_CASES = [
    "knn_rand_tiny", "knn_rand_cifar100", "knn_rand_c10_4096", "knn_rand_k_eq_n",
    "knn_rand_one_class", "knn_rand_ragged", "knn_bf16_inputs", "knn_clustered",
    "knn_dyadic_wide", "knn_dyadic_ties",
]
EXACT_CASES = ["knn_dyadic_wide", "knn_dyadic_ties"]     # exact arithmetic in every add order
SCORE_RTOL = 1e-5


def load_case(name):
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "feature_bf16" in z:
        z["feature"] = (z["feature_bf16"].astype(np.uint32) << 16).view(np.float32)
        z["bank_rows"] = (z["bank_rows_bf16"].astype(np.uint32) << 16).view(np.float32)
    if "feature_i8" in z:
        z["feature"] = z["feature_i8"].astype(np.float32) * z["scale"]
        z["bank_rows"] = z["bank_rows_i8"].astype(np.float32) * z["scale"]
    z["classes"], z["knn_k"], z["knn_t"] = int(z["classes"]), int(z["knn_k"]), float(z["knn_t"])
    z["labels"] = z["labels"].astype(np.int64)
    return z


def val_atol(D):
    # |sum of D products of unit-vector entries| <= 1; rounding of any order <= ~D*eps*sum|ab| <= D*eps
    return 4.0 * D * np.finfo(np.float32).eps


def check_topk(val, idx, ref_val, ref_idx, kth_gap, D, exact=False, what=""):
    """val/idx: [B, k] candidate result; ref_*: reference (torch.topk) result."""
    val, ref_val = np.asarray(val, np.float32), np.asarray(ref_val, np.float32)
    idx, ref_idx = np.asarray(idx, np.int64), np.asarray(ref_idx, np.int64)
    assert val.shape == ref_val.shape, what
    atol = 0.0 if exact else val_atol(D)
    assert np.all(np.diff(val, axis=1) <= 0), f"{what}: values not sorted descending"
    assert np.max(np.abs(val - ref_val)) <= atol, f"{what}: top-k values differ by {np.max(np.abs(val - ref_val))}"
    n_checked = 0
    for b in range(val.shape[0]):
        if not decided_rows(kth_gap[b], D, exact):
            continue      # boundary decided by rounding or by a tie: compared by value above
        assert set(idx[b].tolist()) == set(ref_idx[b].tolist()), f"{what}: row {b} index set differs"
        n_checked += 1
    return n_checked


def check_lower_index_ties(val, idx, what=""):
    """Within runs of equal values the build's rule orders indices ascending."""
    same = val[:, 1:] == val[:, :-1]
    assert np.all(idx[:, 1:][same] > idx[:, :-1][same]), f"{what}: tie order is not lower-index-first"


def decided_rows(kth_gap, D, exact=False):
    """Rows whose top-k index SET is determined (no tie / rounding ambiguity at the k-th boundary).
    On the other rows the reference may have kept a different, equally valid neighbour (possibly
    with another label), so only the top-k values are comparable there."""
    return np.asarray(kth_gap) > (0.0 if exact else 2 * val_atol(D))


def check_scores(scores, ref_scores, what="", rtol=SCORE_RTOL, rows=None):
    scores, ref_scores = np.asarray(scores, np.float64), np.asarray(ref_scores, np.float64)
    if rows is not None:
        scores, ref_scores = scores[rows], ref_scores[rows]
    err = np.abs(scores - ref_scores)
    tol = rtol * np.abs(ref_scores) + 1e-30
    assert np.all(err <= tol), f"{what}: vote scores off by rel {np.max(err / (np.abs(ref_scores) + 1e-30))}"


def check_top1(pred_labels, ref_pred_labels, ref_scores, what="", rtol=SCORE_RTOL, rows=None):
    """Column 0 (the only column the caller consumes, ssltrainer.py:187)."""
    if rows is not None:
        pred_labels, ref_pred_labels = np.asarray(pred_labels)[rows], np.asarray(ref_pred_labels)[rows]
        ref_scores = np.asarray(ref_scores)[rows]
    s = np.sort(np.asarray(ref_scores, np.float64), axis=1)[:, ::-1]
    if s.shape[1] == 1:
        decided = np.ones(s.shape[0], bool)
    else:
        decided = (s[:, 0] - s[:, 1]) > 4 * rtol * np.abs(s[:, 0])
    got, ref = np.asarray(pred_labels)[:, 0], np.asarray(ref_pred_labels)[:, 0]
    assert np.array_equal(got[decided], ref[decided]), f"{what}: top-1 labels differ"
    return int(decided.sum())


def check_ranking_prefix(pred_labels, scores, what=""):
    """The ranking is a permutation, sorted by score descending, lower class first among ties."""
    pred_labels = np.asarray(pred_labels, np.int64)
    B, C = pred_labels.shape
    assert np.array_equal(np.sort(pred_labels, axis=1), np.tile(np.arange(C), (B, 1))), f"{what}: not a permutation"
    s = np.take_along_axis(np.asarray(scores, np.float32), pred_labels, axis=1)
    assert np.all(np.diff(s, axis=1) <= 0), f"{what}: ranking not descending"
    same = s[:, 1:] == s[:, :-1]
    assert np.all(pred_labels[:, 1:][same] > pred_labels[:, :-1][same]), f"{what}: class tie order"
