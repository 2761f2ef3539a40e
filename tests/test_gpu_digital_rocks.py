"""The reference's own three-phase data set on the product path: demo/elasticity/digital_rocks/project.xml with its two
raw files (grosmont_stanford_128x128x128_{1,2}.raw.gz, 8-bit volume fractions of quartz and calcite at 128^3; kept as data
fixtures under tests/golden/digital_rocks/).  Pore space / quartz / calcite, contrast 1000, general Voigt mixing of three
phases with fractional voxels -- the sweep that reads the two effective-moduli arrays (no complementary-phase shortcut).

The project holds no expected numbers; checked are the reader (read_raw_data F:25494-25573, normalizePhi F:17613-17626),
three passes of the loop against oracle/c on this data, and the effective stiffness against the elementary bounds."""
import gzip
import os

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "digital_rocks")
FILES = [os.path.join(DATA, "grosmont_stanford_128x128x128_%d.raw.gz" % i) for i in (1, 2)]
# <materials> of the project: K, mu
MATS = {"matrix": (0.037, 0.044), "quartz": (37.0, 44.0), "calcite": (68.3, 28.4)}
XML = """<settings>
  <solver n="128"><tol>1e-5</tol>
    <materials><matrix K="0.037" mu="0.044" /><quartz K="37.0" mu="44.0" /><calcite K="68.3" mu="28.4" /></materials>
    <gamma_scheme>staggered</gamma_scheme></solver>
  <actions>
    <read_raw_data material="quartz" filename="%s" />
    <read_raw_data material="calcite" filename="%s" />
    %s
  </actions>
</settings>"""


def _raw(i):
    with gzip.open(FILES[i], "rb") as f:
        a = np.frombuffer(f.read(), dtype=np.uint8)
    return a.reshape(128, 128, 128).transpose(2, 1, 0) / 255.0     # x fastest in the file  F:16946-16966


def test_reader_and_phase_normalisation():
    from fibergen_amd import FG
    fg = FG()
    fg.set_xml(XML % (FILES[0], FILES[1], "<init_phase />"))
    assert fg.run() == 0
    phi = fg.get_field("phi")
    assert phi.shape == (3, 128, 128, 128) and fg.get_phase_names() == ["matrix", "quartz", "calcite"]
    q, c = _raw(0), _raw(1)
    # normalizePhi: the later material wins where the fractions exceed one, the matrix takes the rest
    assert np.abs(phi[2] - c).max() < 1e-15
    assert np.abs(phi[1] - np.minimum(q, 1.0 - c)).max() < 1e-15
    assert np.abs(phi.sum(axis=0) - 1.0).max() < 1e-15 and phi.min() >= 0.0
    assert fg.get_volume_fraction("quartz") == pytest.approx(float(phi[1].mean()), rel=1e-12)
    assert 0.70 < phi[1].mean() < 0.73 and 0.003 < phi[2].mean() < 0.0032 and 0.26 < phi[0].mean() < 0.29
    assert ((phi[1] > 0) & (phi[1] < 1)).mean() > 0.3       # a grey-scale data set: most voxels are mixtures


def _lame(K, mu):
    return mu, K - 2.0 * mu / 3.0


def test_three_passes_match_c_oracle_on_the_data_set():
    from fibergen_amd import LSSolver
    from oracle.c_oracle import CRef
    q, c = _raw(0), _raw(1)
    phis = [1.0 - np.minimum(q, 1.0 - c) - c, np.minimum(q, 1.0 - c), c]
    mats = [_lame(*MATS[k]) for k in ("matrix", "quartz", "calcite")]
    n = 128
    s = LSSolver(n, n, n)
    s.set_num_phases(3)
    for p in range(3):
        s.set_phase(p, mats[p][0], mats[p][1], phis[p])
    mu_0, lam_0 = s.calc_ref_material()
    eig = [v for mu, lam in mats for v in (2 * mu, 2 * mu + 3 * lam)]
    assert mu_0 == pytest.approx(0.25 * (min(eig) + max(eig)), rel=1e-14)
    E = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 0.5])
    x = (np.arange(n) + 0.5) / n
    w = np.sin(2 * np.pi * x)[:, None, None] * np.cos(4 * np.pi * x)[None, :, None] + 0.5 * np.sin(6 * np.pi * x)[None, None, :]
    eps0 = np.stack([E[k] + 0.05 * (k + 1) * phis[1] + 0.02 * (6 - k) * w for k in range(6)])
    s.set_field("epsilon", eps0)
    s.iterate(E, 3)                       # pass 1: strain-state pipeline, passes 2-3: displacement loop
    got = s.get_field("epsilon")
    sig = s.get_field("sigma")
    s.close()
    ref = CRef((n, n, n), (1.0, 1.0, 1.0), mats, phis, None, "voigt", threads=min(16, os.cpu_count() or 1))
    eps = eps0
    for _ in range(3):
        eps = ref.basic_scheme(E, eps, mu_0, lam_0)
    assert rel_err(got, eps) < 1e-11
    assert rel_err(sig, ref.calc_stress(0.0, 0.0, eps)) < 1e-11


def test_project_effective_stiffness_within_bounds():
    from fibergen_amd import FG
    fg = FG()
    fg.set_xml(XML % (FILES[0], FILES[1], "<calc_effective_properties />"))
    assert fg.run() == 0
    C = np.array(fg.get_effective_property())
    phi = fg.get_field("phi")
    f = phi.mean(axis=(1, 2, 3))
    K = np.array([MATS[k][0] for k in ("matrix", "quartz", "calcite")])
    G = np.array([MATS[k][1] for k in ("matrix", "quartz", "calcite")])
    # the project solves six load cases to tol 1e-5 with the reference's default method (cg)
    assert np.abs(C - C.T).max() < 2e-3 * np.abs(C).max()
    assert np.all(np.linalg.eigvalsh(0.5 * (C + C.T)) > 0)
    K_eff = C[:3, :3].sum() / 9.0
    G_eff = (C[3, 3] + C[4, 4] + C[5, 5]) / 3.0
    # Voigt / Reuss bounds on the voxel-wise Voigt-mixed medium: per-voxel moduli are arithmetic means already
    Kv, Gv = (phi * K[:, None, None, None]).sum(axis=0), (phi * G[:, None, None, None]).sum(axis=0)
    assert 1.0 / (1.0 / Kv).mean() < K_eff < Kv.mean() and 1.0 / (1.0 / Gv).mean() < G_eff < Gv.mean()
    assert Kv.mean() == pytest.approx(float(f @ K), rel=1e-12)
    # a rock sample is not isotropic, but close: the three shear moduli and the three axial moduli within 15 %
    assert max(C[3, 3], C[4, 4], C[5, 5]) < 1.15 * min(C[3, 3], C[4, 4], C[5, 5])
    assert max(C[0, 0], C[1, 1], C[2, 2]) < 1.15 * min(C[0, 0], C[1, 1], C[2, 2])
