"""Shared parity reporting for the argmax / occupancy comparisons: print what was achieved, then assert
(a) agreement >= floor (0.9999) and (b) every disagreeing element is a near-tie in the REFERENCE logits
(reference top-1 minus the reference's score of the class we chose <= tie_tol), so no flip goes unexplained.
PW_PARITY_REPORT_ONLY=1 prints without asserting (used once to set the tolerances)."""
import os

import numpy as np

REPORT_ONLY = os.environ.get('PW_PARITY_REPORT_ONLY', '0') == '1'
FLOOR = 0.9999       # SURVEY section 7; measured agreement on MI355X is >= 0.99998 everywhere (profiles/r02_parity_*.md)


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)


def check_argmax(name, got, want, ref_logits=None, tie_tol=None, floor=FLOOR):
    """got / want: integer class arrays of one shape; ref_logits: want's logits (..., n_classes)."""
    got, want = _np(got), _np(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    diff = got != want
    n = int(diff.sum())
    agree = 1.0 - n / max(diff.size, 1)
    margin = 0.0
    if n and ref_logits is not None:
        lg = _np(ref_logits)[diff]
        margin = float((lg.max(-1) - np.take_along_axis(lg, got[diff][:, None].astype(np.int64), 1)[:, 0]).max())
    print('[parity] %-44s agreement %.6f (%d / %d differ)%s' % (
        name, agree, n, diff.size, '' if ref_logits is None else '; largest reference margin among flips %.3e (tie tol %.1e)'
        % (margin, tie_tol or 0.0)))
    if REPORT_ONLY:
        return agree
    if ref_logits is not None and tie_tol is not None:
        assert margin <= tie_tol, (name, 'a flipped element is not a near-tie', margin, tie_tol)
        # tiny grids: one explained flip may already be below the floor
        assert agree >= floor or n <= max(1, int(diff.size * (1 - floor) + 0.5)) + 2, (name, agree)
    else:
        assert agree >= floor, (name, agree)
    return agree


def check_close(name, got, want, rtol, atol=0.0):
    """max |got - want| <= atol + rtol * max |want|, printed."""
    got, want = _np(got), _np(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    scale = float(np.abs(want).max()) if want.size else 0.0
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
    print('[parity] %-44s max|err| %.3e  max|ref| %.3e  rel %.2e (bound %.1e)' % (name, err, scale, err / max(scale, 1e-30), rtol))
    assert REPORT_ONLY or err <= atol + rtol * scale, (name, err, scale, rtol)
    return err
