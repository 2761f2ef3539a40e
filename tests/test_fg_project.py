"""CPU tests of the project layer (host logic of the drop-in boundary): XML path syntax,
expression evaluation, material constants, phase normalisation and the native voxeliser.
No GPU compute is called."""
import math

import numpy as np
import pytest

from fibergen_amd import materials
from fibergen_amd.fg import FG, _Fiber, _normalize_phi
from fibergen_amd.xmlproject import XMLProject
from oracle.ls_oracle import material_from_pair

XML = """
<settings>
  <title>Title</title>
  <variables><dx type="float" value="2" /><n type="int" value="8" /></variables>
  <dx>dx</dx>
  <solver n="16">
    <tol>1e-6</tol>
    <materials>
      <matrix E="1" nu="0.3" />
      <fiber  E="2" nu="0.3" />
    </materials>
  </solver>
  <actions>
    <select_material name="fiber" />
    <place_fiber R="0.5" />
    <run_load_case e11="1" />
    <run_load_case e22="1" />
  </actions>
</settings>
"""


def test_xml_path_syntax_like_reference():
    """get_path / set / get / erase  F:26632-26736 and SetParameters F:26854-26901
    (the calls of demo/python/pure_python/project.py:34-40)."""
    fg = FG()
    fg.set_xml(XML)
    fg.set("solver..n", 32)
    fg.set("solver.tol", 1e-8)
    fg.set("title", "New Title")
    fg.set("solver.materials.fiber.", E=10, nu=0.35)
    fg.set("actions.run_load_case[0].", e11=2)
    fg.set("actions.run_load_case[1].", e22=0, e33=1)
    assert fg.get("solver..n") == "32"
    assert float(fg.get("solver.tol")) == 1e-8
    assert fg.get("title") == "New Title"
    assert fg.get("solver.materials.fiber..E") == "10"
    assert fg.get("actions.run_load_case[1]..e33") == "1"
    assert fg.get("actions.run_load_case[0]..e11") == "2"
    with pytest.raises(RuntimeError, match="not found"):
        fg.get("solver.nonexistent")
    fg.set("solver.maxiter", 5)                # creates the element
    assert fg.get("solver.maxiter") == "5"
    fg.set("actions.run_load_case[3]..e12", 0.5)   # creates cases [2] and [3]
    assert len(fg._project.root.find("actions").findall("run_load_case")) == 4
    fg.erase("actions.run_load_case[3]")
    fg.erase("actions.run_load_case[2]")
    assert len(fg._project.root.find("actions").findall("run_load_case")) == 2
    fg.erase("solver..n")
    with pytest.raises(RuntimeError):
        fg.get("solver..n")
    xml = fg.get_xml()
    assert xml.startswith("<?xml") and "New Title" in xml
    fg2 = FG()
    fg2.set_xml(xml)
    assert fg2.get("solver.materials.fiber..nu") == "0.35"
    fg.set_xml_precision(3)
    fg.set("solver.tol", 1.23456789e-5)
    assert fg.get("solver.tol") == "1.23e-05"
    assert fg.get_xml_precision() == 3


def test_expression_evaluation_and_grid():
    fg = FG()
    fg.set_xml(XML)
    fg._init_python()
    assert fg._eval("0.2*dx") == pytest.approx(0.4)
    assert fg._eval("n*2", int) == 16
    (nx, ny, nz), (dx, dy, dz), x0 = fg._grid()
    assert (nx, ny, nz) == (16, 16, 16) and dx == 2.0 and dy == 1.0
    fg.set("solver..mult", 0.5)
    fg.set("solver..nz", 4)
    assert fg._grid()[0] == (8, 8, 2)
    fg.set_variable("variable", [1, 2, 3])
    assert fg.get_variable("variable") == [1, 2, 3]
    assert fg.get_rve_dims() == [0.0, 0.0, 0.0, 2.0, 1.0, 1.0]
    fg.set_py_enabled(False)
    with pytest.raises(Exception):
        fg._eval("0.2*dx")
    assert fg._eval("0.25") == 0.25


def test_material_constants_match_oracle_and_errors():
    ref = material_from_pair(E=100.0, nu=0.4)
    for a, b in materials.PAIRS:
        got = materials.material_constants({a: repr(ref[a]), b: repr(ref[b]), "law": "iso"})
        chk = material_from_pair(**{a: ref[a], b: ref[b]})
        for k in chk:
            assert got[k] == chk[k], (a, b, k)   # same formulas, same operation order
    with pytest.raises(RuntimeError, match="Incomplete"):
        materials.material_constants({"E": "1"})
    with pytest.raises(RuntimeError, match="Ambiguous"):
        materials.material_constants({"E": "1", "nu": "0.3", "mu": "2"})


def test_normalize_phi_last_material_wins():
    """normalizePhi  F:17613-17626"""
    phi = np.array([[1.0, 1.0, 1.0, 1.0], [0.0, 0.6, 1.0, 0.3], [0.0, 0.0, 0.5, 0.9]])[:, :, None, None]
    out = _normalize_phi(phi)[:, :, 0, 0]
    assert np.allclose(out[2], [0, 0, 0.5, 0.9])
    assert np.allclose(out[1], [0, 0.6, 0.5, 0.1])
    assert np.allclose(out[0], [1, 0.4, 0, 0])
    assert np.allclose(out.sum(axis=0), 1)


def test_unknown_settings_raise_like_reference(monkeypatch):
    fg = FG()
    fg.set_xml(XML)
    fg._init_python()
    fg.set("solver.mode", "hyperelasticity")

    class Dummy:
        def __init__(self, *a, **k):
            raise AssertionError("solver must not be created")
    import fibergen_amd.fg as fgmod
    monkeypatch.setattr(fgmod, "LSSolver", Dummy)
    with pytest.raises(RuntimeError, match="mode"):
        fg.init_lss()
    fg.set("solver.mode", "elasticity")
    fg.set("solver.gamma_scheme", "willot")
    with pytest.raises(RuntimeError, match="gamma scheme"):
        fg.init_lss()
    fg.set("solver.gamma_scheme", "auto")
    fg.set("solver.method", "nesterov2")
    with pytest.raises(RuntimeError, match="solver method"):
        fg.init_lss()


def test_orientation_moments_of_placed_fibres():
    """get_A2 / get_A4 (F:25155-25180): moments of the placed fibres' axes, FiberGenerator::updateMoments F:6263-6275 with the
    trace normalisations of getA2 / getA4 F:6683-6707 (place_fiber needs the solver, i.e. a GPU: the fibres are set directly)."""
    from fibergen_amd import FG
    from fibergen_amd.fg import _Fiber
    fg = FG()
    with pytest.raises(RuntimeError, match="no fibres"):
        fg.get_A2()
    fg._fibers = [_Fiber("capsule", [0.5, 0.5, 0.5], ax, 0.2, 0.1, 1) for ax in ([1, 0, 0], [0, 2, 0], [1, 1, 0])]
    A2 = np.array(fg.get_A2())
    axes = np.array([[1, 0, 0], [0, 1, 0], [2 ** -0.5, 2 ** -0.5, 0]])
    ref2 = sum(np.outer(a, a) for a in axes)
    assert np.allclose(A2, ref2 / np.trace(ref2)) and abs(np.trace(A2) - 1) < 1e-15
    A4 = np.array(fg.get_A4())
    ref4 = sum(np.einsum("i,j,k,l->ijkl", a, a, a, a) for a in axes)
    assert A4.shape == (3, 3, 3, 3) and np.allclose(A4, ref4 / 3.0)      # the contraction of each unit axis' term has trace 1
    assert np.allclose(np.einsum("iikl->kl", A4), A2)                     # A4_iikl = A2 for unit axes


def test_calc_hs_bounds_action_and_the_hashin_demo_constant():
    """<calc_HS_bounds> (F:25730-25742, HashinBounds::get F:7463-7484).  Reference-held pin: the Hashin demo's matrix is chosen so
    that its bulk modulus equals the coated sphere's -- `k_star = 4.305343511446667 (theoretical)` in
    demo/elasticity/hashin/project.xml:32 -- and the coated-sphere assemblage attains the Hashin-Shtrikman bound of (core,
    coating) at the core's volume share (0.2 / 0.4)^3."""
    fg = FG.__new__(FG)        # the action needs no solver and no device
    fg._project = None
    import xml.etree.ElementTree as ET
    act = ET.fromstring('<calc_HS_bounds mu1="5" lambda1="4" phi1="0.125" mu2="3" lambda2="2" phi2="0.875" />')
    fg._eval = lambda text, typ=float: typ(float(text))
    assert fg._run_action(act) is None
    lo, hi = fg._hs_bounds["lower"], fg._hs_bounds["upper"]
    assert lo["K"] <= hi["K"] and lo["mu"] <= hi["mu"]
    assert lo["K"] == pytest.approx(4.305343511446667, rel=1e-11)          # the demo's k_star (its lambda is given to 12 digits)
    assert lo["K"] == pytest.approx(3.63867684478 + 2.0 / 3.0, rel=1e-11)   # = lambda_matrix + 2/3 mu_matrix of the demo
    assert lo["lambda"] == pytest.approx(lo["K"] - 2.0 / 3.0 * lo["mu"], rel=1e-15)
    # one phase only: both bounds are that material; other constant pairs are read like any material (E, nu)
    act = ET.fromstring('<calc_HS_bounds E1="10" nu1="0.2" phi1="1" mu2="3" lambda2="2" phi2="0" />')
    fg._run_action(act)
    c = materials.material_constants({"E": 10, "nu": 0.2})
    for b in ("lower", "upper"):
        assert fg._hs_bounds[b]["K"] == pytest.approx(c["K"], rel=1e-14) and fg._hs_bounds[b]["mu"] == pytest.approx(c["mu"], rel=1e-14)
    with pytest.raises(RuntimeError, match="Incomplete material definition"):
        fg._run_action(ET.fromstring('<calc_HS_bounds mu1="5" phi1="0.5" mu2="3" lambda2="2" phi2="0.5" />'))
