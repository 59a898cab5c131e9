"""CPU: the chroma comparison helper itself — it must accept a justified tie swap and REJECT a permutation of pitch classes."""
import numpy as np
import pytest

from chroma_check import check_chroma


def _case(seed=0, gap=0.05):
    r = np.random.default_rng(seed)
    medians = np.sort(r.random(12) * 0.5 + np.arange(12) * gap)
    order = r.permutation(12)  # pitch class of every delivered column
    cols = np.abs(r.standard_normal((40, 12))) * 0.05 + medians[None, :]
    cols = cols / cols.sum(1)[:, None]
    return cols, order, medians


def test_identical_results_pass_and_a_permutation_of_pitch_classes_fails():
    cols, order, med = _case()
    assert check_chroma(cols, order, med, cols, order, med) == 0
    wrong = order.copy()
    wrong[[3, 9]] = wrong[[9, 3]]  # same columns, but two of them claim each other's pitch class
    with pytest.raises(AssertionError, match="pitch-class order"):
        check_chroma(cols, wrong, med, cols, order, med)
    # the comparison round 4 used (both sides re-sorted by their column means) accepts exactly that
    shuffled = cols[:, np.random.default_rng(1).permutation(12)]
    np.testing.assert_allclose(shuffled[:, np.argsort(shuffled.mean(0))], cols[:, np.argsort(cols.mean(0))])


def test_a_swap_is_accepted_only_between_tied_medians():
    cols, order, med = _case(gap=0.05)
    med_t = med.copy()
    med_t[5] = med_t[4] + 1e-5  # columns 4 and 5 tie in the oracle
    got_cols, got_order, got_med = cols.copy(), order.copy(), med_t.copy()
    got_cols[:, [4, 5]] = got_cols[:, [5, 4]]
    got_order[[4, 5]] = got_order[[5, 4]]
    got_med[[4, 5]] = got_med[[5, 4]] + np.array([-2e-5, 2e-5])  # the product saw them the other way round, by 2e-5
    assert check_chroma(got_cols, got_order, got_med, cols, order, med_t) == 2
    # the same swap between columns whose oracle medians are 0.05 apart is an ordering error, not a tie
    got_cols, got_order, got_med = cols.copy(), order.copy(), med.copy()
    got_cols[:, [7, 8]] = got_cols[:, [8, 7]]
    got_order[[7, 8]] = got_order[[8, 7]]
    got_med[[7, 8]] = got_med[[8, 7]]  # (every pitch class keeps its column and its median: only the delivery ORDER is wrong)
    with pytest.raises(AssertionError, match="although the oracle's medians"):
        check_chroma(got_cols, got_order, got_med, cols, order, med)
