"""Host-side glue of the drop-in classes that needs no GPU: the substring dispatch of the model-Hessian names with their modifiers
(reference approx_hessian.py:35-110), the charges of the short-range correction (shortrange.py:147-186) against the oracle's restatement, the
unit handling of the well potentials, and the D&C tree shapes of the two leaf kinds (dc_logic.h restated)."""
import numpy as np
import pytest


def test_model_name_dispatch_and_modifier_flags():
    from multioptpy_amd.ModelHessian import approx_hessian as ah
    k = ah.model_kind
    assert k("fischerd3old") == ah.MODEL_FISCHERD3OLD and k("FischerD3") == ah.MODEL_FISCHERD3 and k("fischer") == ah.MODEL_FISCHER
    assert k("schlegel") == ah.MODEL_SCHLEGEL and k("morse") == ah.MODEL_MORSE and k("lindh2007d3") == ah.MODEL_LINDH2007D3
    assert k("something_unknown") == ah.MODEL_LINDH2007D3            # the dispatcher's fall-back (approx_hessian.py:88-91)
    for name in ("gfnff", "swart", "lindh", "lindh2007d4", "schlegeld3", "fischerd4"):
        with pytest.raises(NotImplementedError):
            k(name)
    assert k("fischerd3old_ts", with_flags=True) == (ah.MODEL_FISCHERD3OLD, ah.FLAG_TS)
    assert k("schlegel_sr_clip", with_flags=True) == (ah.MODEL_SCHLEGEL, ah.FLAG_SR | ah.FLAG_CLIP)
    assert k("fischerd3_ts_sr_clip", with_flags=True) == (ah.MODEL_FISCHERD3, ah.FLAG_TS | ah.FLAG_SR | ah.FLAG_CLIP)
    assert k("morse", with_flags=True) == (ah.MODEL_MORSE, 0)


def test_sr_charges_match_the_oracle():
    from multioptpy_amd.ModelHessian.approx_hessian import sr_charges, _PAULING
    from oracle import model_hessian as mh
    assert _PAULING == {k: v for k, v in mh._PAULING.items()}
    el = ["C", "H", "H", "O", "Cl", "Xx", "Ar"]
    en = [mh._PAULING.get(e, 2.0) for e in el]
    ref = np.array([0.2 * (sum(en) / len(en) - v) for v in en])
    assert np.array_equal(sr_charges(el), ref)


def test_well_potential_units():
    from multioptpy_amd.Potential import switching_potential as sp
    from oracle import well_pot as owp
    assert sp.BOHR2ANG == owp.BOHR2ANG == 0.52917721067 and sp.HARTREE2KJMOL == owp.HARTREE2KJMOL == 2625.5
    # the well function itself: continuity of value and slope at the four limits (quintic switches, linear walls)
    lim, e0 = [1.0, 2.0, 4.0, 5.5], 0.03
    for x in lim:
        lo, hi = owp.well(x - 1e-9, lim, e0), owp.well(x + 1e-9, lim, e0)
        assert abs(lo[0] - hi[0]) < 1e-9 and abs(lo[1] - hi[1]) < 1e-6, x
    assert owp.well(3.0, lim, e0) == (0.0, 0.0, 0.0) and abs(owp.well(2.0, lim, e0)[0]) < 1e-15 and abs(owp.well(1.0, lim, e0)[0] - e0) < 1e-15


def test_dc_tree_shapes_of_both_leaf_kinds():
    """dc_logic.h: dc_levels (leaves <= 32 rows, one-wave QL) and dc_levels_wg (leaves <= 40 rows, one workgroup each), bounds lb(i) = (i n) >> L."""
    def levels(n, leaf):
        L = 0
        while ((n + (1 << L) - 1) >> L) > leaf:
            L += 1
        return L
    for n, lq, lw in ((300, 4, 3), (600, 5, 4), (120, 2, 2), (73, 2, 1), (320, 4, 3), (321, 4, 4)):
        assert levels(n, 32) == lq and levels(n, 40) == lw, n
        for L in (lq, lw):
            b = [(i * n) >> L for i in range((1 << L) + 1)]
            assert b[0] == 0 and b[-1] == n and all(0 < b[i + 1] - b[i] <= (32 if L == lq else 40) for i in range(1 << L))
