"""Test helper: compare a product chromagram with the oracle's PITCH CLASS FOR PITCH CLASS.

``chroma()`` (reference audioreactive/signal.py:136-156) delivers its columns ordered by their median, so two implementations whose
medians nearly tie may deliver the same chromagram with two columns swapped.  Round 4 side-stepped that by re-sorting both results by
their column means — under which ANY permutation of the pitch classes passes.  Here the product's column order is recomputed from the
product's own stages (harmonic -> raw_chroma -> resample -> median), both results are put back into pitch-class order (C = 0 ... B = 11)
and compared there; the delivered ORDER may then differ from the oracle's only where the oracle's medians are closer together than the
measured error of the product's medians."""
import numpy as np


def to_pitch_classes(columns, order, n_classes=12):
    """[T, notes] columns delivered in ``order`` (pitch class of every column) -> [T, n_classes] with NaN for classes not delivered."""
    out = np.full((columns.shape[0], n_classes), np.nan, dtype=np.float64)
    out[:, np.asarray(order)] = columns
    return out


def check_chroma(got, got_order, got_medians, want, want_order, want_medians, atol=2e-3, median_atol=5e-3):
    """got / want: [T, notes] as delivered; *_order: the pitch class of every delivered column; *_medians: the medians the columns were
    ordered by (same column order).  Returns the number of positions whose pitch class differs (justified ties)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    got_order, want_order = np.asarray(got_order), np.asarray(want_order)
    assert got.shape == want.shape and sorted(got_order.tolist()) == sorted(want_order.tolist()), "different pitch classes were selected"
    pc_got, pc_want = to_pitch_classes(got, got_order), to_pitch_classes(want, want_order)
    np.testing.assert_allclose(pc_got, pc_want, atol=atol, equal_nan=True, err_msg="chromagram differs in pitch-class order")
    med_got, med_want = np.full(12, np.nan), np.full(12, np.nan)
    med_got[got_order], med_want[want_order] = got_medians, want_medians
    err = float(np.nanmax(np.abs(med_got - med_want)))
    assert err <= median_atol, f"column medians differ by {err:.2e}"
    swapped = 0
    for j in range(len(want_order)):  # a different class at position j is a tie only if the oracle's medians of the two classes are that close
        if got_order[j] != want_order[j]:
            gap = abs(med_want[got_order[j]] - med_want[want_order[j]])
            assert gap <= 2 * err + 1e-7, (f"column {j} is pitch class {got_order[j]} (oracle: {want_order[j]}) although the oracle's medians "
                                           f"differ by {gap:.2e} (measured median error {err:.2e})")
            swapped += 1
    return swapped
