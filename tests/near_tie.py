"""TEST INFRASTRUCTURE: what a finite-precision implementation of the reference's R2D2 math may legitimately get wrong, as proofs
instead of quantiles.

A TD error / priority at (t, b) reads Q_target(s', argmax_a Q_online(s', a)).  Where the REFERENCE's own two best legal Q_online
values at s' are closer than the implementation's Q tolerance, either action is a legitimate argmax and the value may move by
O(Q_target spread); everywhere else it has to be within tolerance.  So: every outlier must come with such a near-tie of the fp32
reference -- anything else fails."""
import torch


def top2_gap(q, legal):
    """gap between the two best LEGAL entries of q along the last axis (rows with fewer than two legal entries: +inf)"""
    score = (1 + q - q.min()) * legal
    top = score.topk(2, dim=-1).values
    gap = top[..., 0] - top[..., 1]
    return torch.where(legal.sum(-1) >= 2, gap, torch.full_like(gap, float("inf")))


def assert_sequence_outliers_are_near_ties(prio, rprio, loss, rloss, gap, n, tol_q, tol_loss, tag=""):
    """[T, B] priorities and [B] losses of a learner against the fp32 reference; `gap` [T, B] = top2_gap of the reference's Q_online.
    -> (number of tie-excused priorities, max error outside them, max loss error outside tie sequences)"""
    T = prio.shape[0]
    dp = (prio - rprio).abs()
    out_p = torch.nonzero(dp > tol_q)
    for t, b in out_p.tolist():
        assert t + n < T and float(gap[t + n, b]) < 2 * tol_q, (
            tag, "priority outlier without a near-tie of the reference", t, b, float(dp[t, b]), float(gap[min(t + n, T - 1), b]))
    dl = (loss - rloss).abs()
    tie_seq = set(b for _, b in out_p.tolist())
    for b in torch.nonzero(dl > tol_loss).flatten().tolist():
        assert b in tie_seq, (tag, "loss outlier without a near-tie in its sequence", b, float(dl[b]))
    clean_p, clean_l = dp.clone(), dl.clone()
    clean_p[dp > tol_q] = 0
    clean_l[dl > tol_loss] = 0
    return int(out_p.shape[0]), float(clean_p.max()), float(clean_l.max())
