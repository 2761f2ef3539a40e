"""NumPy restatement of fibergen's Lippmann-Schwinger basic scheme (staggered grid).

TEST INFRASTRUCTURE ONLY -- this is the *checker*, never the product path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.

Every function restates one reference routine and cites it as ``F:<lines>``
where ``F`` = ``/root/reference/src/fibergen.cpp`` (fospald/fibergen @ 2024_08_07).
Nothing is imported from the reference (it is C++ and cannot be built here:
boost / FFTW / LAPACK are absent).

Pinning (see tests/test_oracle_pins.py): the reference holds no numeric
fixtures; the oracle is pinned against the reference's own known answers:
  * "staggered epsG0div identity" self-test          F:24129-24151 (tol sqrt(eps))
  * "collocated epsG0div identity" self-test         F:24085-24105 (tol sqrt(eps))
  * closed-form isotropic laminate                   F:26405-26474 / demo/elasticity/laminate
  * Hashin coated sphere <sigma> = 12.9152 I         demo/elasticity/hashin/project.xml:30-32
  * homogeneous medium => eps == E after one pass
The third-party arithmetic on the path is FFTW3's r2c/c2r (system package,
unpinned; docker/Dockerfile:17 => 3.3.8): it computes the unnormalised DFT in
half-spectrum layout, which is what numpy.fft.rfftn / irfftn (pocketfft)
compute; the 1/N goes on the forward transform as in F:18486-18506.

Conventions: fields are float64 arrays ``[ncomp, nx, ny, nz]``
(no z padding here -- padding is a storage detail of the reference);
component order 0=11 1=22 2=33 3=23 4=13 5=12; shear entries are plain tensor
components.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

EPS = np.finfo(np.float64).eps
SMALLEST = np.finfo(np.float64).tiny  # boost::numeric::bounds<double>::smallest()


# --------------------------------------------------------------------------
# Material constants                                              F:7293-7455
# --------------------------------------------------------------------------

_PAIRS = [("K", "E"), ("K", "lambda"), ("K", "mu"), ("K", "nu"), ("E", "mu"),
          ("E", "nu"), ("lambda", "mu"), ("lambda", "nu"), ("mu", "nu"), ("mu", "M")]


def material_from_pair(**kw):
    """Convert any supported pair of elastic constants to (lambda, mu, ...).

    Restates Material::readSettings / calc_from_* F:7333-7454 (formulas verbatim,
    same operation order).  Raises like the reference on incomplete/ambiguous input.
    """
    keys = {k for k in kw if k in ("K", "E", "lambda", "mu", "nu", "M")}
    icalc = None
    for a, b in _PAIRS:
        if a in keys and b in keys:
            icalc = (a, b)
    if icalc is None:
        raise RuntimeError("Incomplete material definition")
    if keys - set(icalc):
        raise RuntimeError("Ambiguous material definition")
    v = {k: float(kw[k]) for k in icalc}
    if icalc == ("K", "E"):
        K, E = v["K"], v["E"]
        lam = (3 * K * (3 * K - E)) / (9 * K - E)
        mu = (3 * K * E) / (9 * K - E)
        nu = (3 * K - E) / (6 * K)
        M = (3 * K * (3 * K + E)) / (9 * K - E)
    elif icalc == ("K", "lambda"):
        K, lam = v["K"], v["lambda"]
        E = (9 * K * (K - lam)) / (3 * K - lam)
        mu = (3 * (K - lam)) / 2
        nu = lam / (3 * K - lam)
        M = 3 * K - 2 * lam
    elif icalc == ("K", "mu"):
        K, mu = v["K"], v["mu"]
        E = (9 * K * mu) / (3 * K + mu)
        lam = K - (2 * mu) / 3
        nu = (3 * K - 2 * mu) / (2 * (3 * K + mu))
        M = K + (4 * mu) / 3
    elif icalc == ("K", "nu"):
        K, nu = v["K"], v["nu"]
        E = 3 * K * (1 - 2 * nu)
        lam = (3 * K * nu) / (1 + nu)
        mu = (3 * K * (1 - 2 * nu)) / (2 * (1 + nu))
        M = (3 * K * (1 - nu)) / (1 + nu)
    elif icalc == ("E", "mu"):
        E, mu = v["E"], v["mu"]
        K = (E * mu) / (3 * (3 * mu - E))
        lam = (mu * (E - 2 * mu)) / (3 * mu - E)
        nu = E / (2 * mu) - 1
        M = (mu * (4 * mu - E)) / (3 * mu - E)
    elif icalc == ("E", "nu"):
        E, nu = v["E"], v["nu"]
        K = E / (3 * (1 - 2 * nu))
        lam = (E * nu) / ((1 + nu) * (1 - 2 * nu))
        mu = E / (2 * (1 + nu))
        M = (E * (1 - nu)) / ((1 + nu) * (1 - 2 * nu))
    elif icalc == ("lambda", "mu"):
        lam, mu = v["lambda"], v["mu"]
        K = lam + (2 * mu) / 3
        E = (mu * (3 * lam + 2 * mu)) / (lam + mu)
        nu = lam / (2 * (lam + mu))
        M = lam + 2 * mu
    elif icalc == ("lambda", "nu"):
        lam, nu = v["lambda"], v["nu"]
        K = (lam * (1 + nu)) / (3 * nu)
        E = (lam * (1 + nu) * (1 - 2 * nu)) / nu
        mu = (lam * (1 - 2 * nu)) / (2 * nu)
        M = (lam * (1 - nu)) / nu
    elif icalc == ("mu", "nu"):
        mu, nu = v["mu"], v["nu"]
        K = (2 * mu * (1 + nu)) / (3 * (1 - 2 * nu))
        E = 2 * mu * (1 + nu)
        lam = (2 * mu * nu) / (1 - 2 * nu)
        M = (2 * mu * (1 - nu)) / (1 - 2 * nu)
    else:  # mu, M
        mu, M = v["mu"], v["M"]
        K = M - (4 * mu) / 3
        E = (mu * (3 * M - 4 * mu)) / (M - mu)
        lam = M - 2 * mu
        nu = (M - 2 * mu) / (2 * M - 2 * mu)
    return {"K": K, "E": E, "lambda": lam, "mu": mu, "nu": nu, "M": M}


# --------------------------------------------------------------------------
# Voigt helpers                                                     F:494-598
# --------------------------------------------------------------------------

def voigt_id4():
    """Voigt::Id4(6): diag(1,1,1,1/2,1/2,1/2)                         F:501-512"""
    return np.diag([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])


def voigt_ii4():
    """Voigt::II4(6)                                                   F:517-525"""
    m = np.zeros((6, 6))
    m[:3, :3] = 1.0
    return m


def voigt_dyad4_mv(M, v):
    """M : v with the factor 2 on the shear entries of v              F:563-575"""
    vc = np.array(v, dtype=np.float64).copy()
    vc[3:6] *= 2
    return M @ vc


def voigt_dyad4_mm(A, B):
    """A : B column by column                                          F:582-597"""
    C = np.empty((6, 6))
    for i in range(6):
        C[:, i] = voigt_dyad4_mv(A, B[:, i])
    return C


def voigt_norm2(v):
    """Voigt::norm_2 (shear counted twice)                             F:530-537"""
    v = np.asarray(v, dtype=np.float64)
    return math.sqrt(float(v @ v) + v[3] * v[3] + v[4] * v[4] + v[5] * v[5])


# --------------------------------------------------------------------------
# Constitutive laws
# --------------------------------------------------------------------------

def hooke(E, mu, lam, alpha=1.0):
    """LinearIsotropicMaterialLaw::PK1  F:11375-11396.

    ``E`` is ``[6, ...]``; returns ``S`` of the same shape with
    S_i = E_i*(2*alpha*mu) + (alpha*lam)*(E_0+E_1+E_2) for i<3, S_i = E_i*(2*alpha*mu) else.
    ``alpha`` may be an array broadcastable to the voxel shape (phase fraction).
    """
    two_mu = 2 * alpha * mu
    lam_tr = alpha * lam * (E[0] + E[1] + E[2])
    S = np.empty_like(E)
    S[0] = E[0] * two_mu + lam_tr
    S[1] = E[1] * two_mu + lam_tr
    S[2] = E[2] * two_mu + lam_tr
    S[3] = E[3] * two_mu
    S[4] = E[4] * two_mu
    S[5] = E[5] * two_mu
    return S


VOIGT_THRESHOLD = 10 * EPS  # F:12736


def pk1_voigt(eps, phis, mats, alpha=1.0):
    """VoigtMixedMaterialLaw::PK1  F:12752-12761: sum_p phi_p*Hooke_p(eps),
    phases with phi <= 10*eps skipped (first contributing phase assigns,
    later ones accumulate -- identical to adding to zero)."""
    P = np.zeros_like(eps)
    first = np.ones(eps.shape[1:], dtype=bool)
    for phi, (mu, lam) in zip(phis, mats):
        use = phi > VOIGT_THRESHOLD
        S = hooke(eps, mu, lam, phi * alpha)
        # gamma==false on the first contributing phase => plain assignment
        P = np.where(use & first, S, np.where(use, P + S, P))
        first = first & ~use
    return P


# index maps of the 9-component tensor 11,22,33,23,13,12,32,31,21  F:13186-13188
_ROW = (0, 1, 2, 1, 0, 0, 2, 2, 1)
_COL = (0, 1, 2, 2, 2, 1, 1, 0, 0)

LAMINATE_EPS_G = EPS               # F:13111
LAMINATE_EPS_A = EPS ** (2.0 / 3.0)  # F:13110


def _fix_dim(t6):
    """MixedMaterialLawBase::fix_dim for dim()==6: mirror 3,4,5 -> 6,7,8  F:12115-12125"""
    return [t6[0], t6[1], t6[2], t6[3], t6[4], t6[5], t6[3], t6[4], t6[5]]


def _dot9(A, B):
    """Tensor3x3::dot, nine products summed in index order            F:9332-9335"""
    s = B[0] * A[0]
    for i in range(1, 9):
        s = s + B[i] * A[i]
    return s


def _hooke_list(E, mu, lam, alpha):
    two_mu = 2 * alpha * mu
    lam_tr = alpha * lam * (E[0] + E[1] + E[2])
    return [E[0] * two_mu + lam_tr, E[1] * two_mu + lam_tr, E[2] * two_mu + lam_tr,
            E[3] * two_mu, E[4] * two_mu, E[5] * two_mu]


def sym3_det(H):
    """SymTensor3x3::det  F:9373-9382, components 11, 22, 33, 23, 13, 12 (element-wise on arrays)."""
    return (H[0] * (H[1] * H[2] - H[3] * H[3])
            - H[5] * (H[5] * H[2] - H[3] * H[4])
            + H[4] * (H[5] * H[3] - H[1] * H[4]))


def sym3_inv(H):
    """SymTensor3x3::inv  F:9483-9488: cofactors times 1/det, the operation order of the reference (the laminate rule's
    Hessian solve; pinned by the reference's 'symmetric left / right inverse' and 'sym determinant' self-tests F:23669-23736)."""
    invdet = 1 / sym3_det(H)
    return [(H[1] * H[2] - H[3] * H[3]) * invdet,
            (H[0] * H[2] - H[4] * H[4]) * invdet,
            (H[0] * H[1] - H[5] * H[5]) * invdet,
            -(H[0] * H[3] - H[4] * H[5]) * invdet,
            (H[5] * H[3] - H[4] * H[1]) * invdet,
            -(H[5] * H[2] - H[3] * H[4]) * invdet]


def laminate_split(Fbar, n, c1, c2, mat1, mat2, eps_g=LAMINATE_EPS_G, eps_a=LAMINATE_EPS_A):
    """LaminateMixedMaterialLaw::solve_newton for DIM==6  F:13157-13371.

    One Newton step from a=0 for the rank-one jump a (x) n, then return
    (F:13366-13371).  Works element-wise on arrays: ``Fbar`` is a list/array of 6
    arrays, ``n`` of 3 arrays, ``c1``/``c2`` arrays; mat = (mu, lam) scalars or arrays.
    Returns (F1, F2) as lists of 6 arrays (the first six of the symmetrised 9).
    Early exits (||g||<=eps_g, ||H^-1 g||<=eps_a) leave F1=F2=Fbar (F:13262,13305).
    """
    mu1, lam1 = mat1
    mu2, lam2 = mat2
    Fb = _fix_dim([np.asarray(Fbar[i], dtype=np.float64) for i in range(6)])
    zero = np.zeros_like(Fb[0])
    n = [np.asarray(n[i], dtype=np.float64) + zero for i in range(3)]
    c1 = np.asarray(c1, dtype=np.float64) + zero
    c2 = np.asarray(c2, dtype=np.float64) + zero

    # dF1/da_k = -c2 sym(e_k (x) n), dF2/da_k = +c1 sym(e_k (x) n)   F:13232-13243 (RT = identity)
    dF1 = []
    dF2 = []
    for k in range(3):
        d1 = [(-c2 * n[_COL[i]]) if _ROW[i] == k else zero for i in range(9)]
        d2 = [(c1 * n[_COL[i]]) if _ROW[i] == k else zero for i in range(9)]
        for d in (d1, d2):  # fix_sym F:12128-12138
            d[6] = d[3] = 0.5 * (d[3] + d[6])
            d[7] = d[4] = 0.5 * (d[4] + d[7])
            d[8] = d[5] = 0.5 * (d[5] + d[8])
        dF1.append(d1)
        dF2.append(d2)

    # gradient g = dW/da  F:13246-13253
    P1 = _fix_dim(_hooke_list(Fb, mu1, lam1, 1.0))
    P2 = _fix_dim(_hooke_list(Fb, mu2, lam2, 1.0))
    g = [c1 * _dot9(P1, dF1[k]) + c2 * _dot9(P2, dF2[k]) for k in range(3)]
    g_norm = np.sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2])

    # Hessian  F:13268-13274
    H = []
    for i in range(6):
        k, l = _ROW[i], _COL[i]
        dP1 = _fix_dim(_hooke_list(dF1[l], mu1, lam1, 1.0))
        dP2 = _fix_dim(_hooke_list(dF2[l], mu2, lam2, 1.0))
        H.append(c1 * _dot9(dP1, dF1[k]) + c2 * _dot9(dP2, dF2[k]))

    with np.errstate(divide="ignore", invalid="ignore"):
        Hi = sym3_inv(H)
        # Tensor3::mult(SymTensor3x3, Tensor3)  F:9516-9521
        da = [Hi[0] * g[0] + Hi[5] * g[1] + Hi[4] * g[2],
              Hi[5] * g[0] + Hi[1] * g[1] + Hi[3] * g[2],
              Hi[4] * g[0] + Hi[3] * g[1] + Hi[2] * g[2]]
        da_norm = np.sqrt(da[0] * da[0] + da[1] * da[1] + da[2] * da[2])

    # a_next = a - t*da with a=0, t=1  F:13338-13340
    a = [0.0 - 1.0 * da[i] for i in range(3)]
    F1 = list(Fb)
    F2 = list(Fb)
    for i in range(9):  # F:13348-13351
        F1[i] = F1[i] - c2 * a[_ROW[i]] * n[_COL[i]]
        F2[i] = F2[i] + c1 * a[_ROW[i]] * n[_COL[i]]
    for F in (F1, F2):  # fix_sym F:13352
        F[6] = F[3] = 0.5 * (F[3] + F[6])
        F[7] = F[4] = 0.5 * (F[4] + F[7])
        F[8] = F[5] = 0.5 * (F[5] + F[8])

    # `!(x > eps)` keeps NaN on the "step taken" side exactly like the C++ `<=` tests
    stop = (g_norm <= eps_g) | (da_norm <= eps_a)
    F1 = [np.where(stop, Fb[i], F1[i]) for i in range(6)]
    F2 = [np.where(stop, Fb[i], F2[i]) for i in range(6)]
    return F1, F2


def _get_mix(eps, phis, mats, normals):
    """LaminateMixedMaterialLaw::get_mix  F:13456-13525 on whole fields: walks the phases in <materials> order.
    Returns (p1, c1, single, idx, m1, m2, cc1, cc2, F1, F2): phase index and fraction of the first phase per voxel, the
    mask of voxels with one phase only, and for the two-phase voxels `idx` the materials, fractions (c2 := 1 - c1,
    F:13523) and the two strains of the laminate split (None when there are no such voxels)."""
    shape = eps.shape[1:]
    nph = len(phis)
    # walk phases like get_mix does
    p1 = np.full(shape, -1, dtype=np.int64)
    p2 = np.full(shape, -1, dtype=np.int64)
    c1 = np.zeros(shape)
    pure = np.zeros(shape, dtype=bool)
    for p in range(nph):
        phi = phis[p]
        live = ~pure & (phi != 0)
        is_pure = live & (phi == 1)
        # phi==1: c1=phi, p1=p, p2 reset, return
        p1 = np.where(is_pure, p, p1)
        p2 = np.where(is_pure, -1, p2)
        c1 = np.where(is_pure, phi, c1)
        pure |= is_pure
        live &= ~is_pure
        take1 = live & (p1 < 0)
        take2 = live & ~take1 & (p2 < 0)
        third = live & ~take1 & ~take2
        if np.any(third):
            raise RuntimeError("The laminate mixing rule supports only two phase mixtures")
        p1 = np.where(take1, p, p1)
        c1 = np.where(take1, phi, c1)
        p2 = np.where(take2, p, p2)
    if np.any(p1 < 0):
        raise RuntimeError("The laminate mixing rule supports only two phase mixtures")
    mus = np.array([m[0] for m in mats], dtype=np.float64)
    lams = np.array([m[1] for m in mats], dtype=np.float64)
    single = p2 < 0
    mixed = ~single
    if not np.any(mixed):
        return p1, c1, single, None, None, None, None, None, None, None
    idx = np.nonzero(mixed)
    e = [eps[i][idx] for i in range(6)]
    nn = [normals[i][idx] for i in range(3)]
    cc1 = c1[idx]
    cc2 = 1.0 - cc1  # F:13523
    m1 = (mus[p1[idx]], lams[p1[idx]])
    m2 = (mus[p2[idx]], lams[p2[idx]])
    F1, F2 = laminate_split(e, nn, cc1, cc2, m1, m2)
    return p1, c1, single, idx, m1, m2, cc1, cc2, np.array(F1), np.array(F2)


def pk1_laminate(eps, phis, mats, normals, alpha=1.0):
    """LaminateMixedMaterialLaw::PK1 + get_mix  F:13456-13558.

    Pure voxels (some phi==1) -> that phase's Hooke law; voxels whose first
    non-zero phase is the only one -> c1*Hooke(eps); two-phase voxels ->
    c1 = phi of the first phase with 0<phi<1 (materials order), c2 := 1-c1,
    laminate split, P = c1*Hooke_1(F1) + c2*Hooke_2(F2).  More than two
    non-zero phases raise like the reference (F:13473).
    """
    p1, c1, single, idx, m1, m2, cc1, cc2, F1, F2 = _get_mix(eps, phis, mats, normals)
    mus = np.array([m[0] for m in mats], dtype=np.float64)
    lams = np.array([m[1] for m in mats], dtype=np.float64)
    # single-phase (pure, or lone partial phase): P = Hooke_p1(eps, c1*alpha)
    P_single = hooke(eps, mus[p1], lams[p1], c1 * alpha)
    if idx is not None:
        S1 = hooke(F1, m1[0], m1[1], cc1 * alpha)
        S2 = hooke(F2, m2[0], m2[1], cc2 * alpha)
        Pm = S1 + S2
        for i in range(6):
            P_single[i][idx] = Pm[i]
    return P_single


def energy_hooke(E, mu, lam):
    """LinearIsotropicMaterialLaw::W  F:11368-11373: 0.5 * S.dot(E) with S = PK1(E, 1) and SymTensor3x3::dot
    F:9414-9417 (shear products doubled)."""
    S = hooke(E, mu, lam, 1.0)
    return 0.5 * (S[0] * E[0] + S[1] * E[1] + S[2] * E[2] + 2 * (S[3] * E[3] + S[4] * E[4] + S[5] * E[5]))


def energy_voigt(eps, phis, mats):
    """VoigtMixedMaterialLaw::W  F:12739-12750: sum_p phi_p W_p(eps), phases with phi <= 10 eps skipped."""
    W = np.zeros(eps.shape[1:])
    for phi, (mu, lam) in zip(phis, mats):
        use = phi > VOIGT_THRESHOLD
        W = np.where(use, W + phi * energy_hooke(eps, mu, lam), W)
    return W


def energy_laminate(eps, phis, mats, normals):
    """LaminateMixedMaterialLaw::W  F:13527-13540: c1 W_1(F1) (+ c2 W_2(F2) at two-phase voxels) with the strains of
    get_mix."""
    p1, c1, single, idx, m1, m2, cc1, cc2, F1, F2 = _get_mix(eps, phis, mats, normals)
    mus = np.array([m[0] for m in mats], dtype=np.float64)
    lams = np.array([m[1] for m in mats], dtype=np.float64)
    W = c1 * energy_hooke(eps, mus[p1], lams[p1])
    if idx is not None:
        Wm = cc1 * energy_hooke(F1, m1[0], m1[1])
        Wm = Wm + cc2 * energy_hooke(F2, m2[0], m2[1])
        W[idx] = Wm
    return W


# --------------------------------------------------------------------------
# Solver
# --------------------------------------------------------------------------

@dataclass
class LSOracle:
    """Restatement of LSSolver<double,double,3> for mode=elasticity,
    method=basic, gamma_scheme=staggered (F:14641-24740, the rows of SURVEY section 8a)."""

    nx: int
    ny: int
    nz: int
    dx: float = 1.0
    dy: float = 1.0
    dz: float = 1.0
    mats: list = field(default_factory=list)      # [(mu, lam)] in <materials> order
    phis: list = field(default_factory=list)      # [ndarray[nx,ny,nz]]
    normals: np.ndarray | None = None             # [3,nx,ny,nz]
    mixing_rule: str = "voigt"
    tol: float = 1e-4                             # F:14800-14862 defaults
    abs_tol: float = EPS
    bc_tol: float = 1e-3
    maxiter: int = 10000
    ref_scale: float = 1.0
    bc_relax: float = 1.0
    mu_0: float = float("nan")
    lambda_0: float = 0.0
    update_ref: str = "loadstep"
    gamma_scheme: str = "staggered"               # or "collocated" (GammaOperatorCollocated F:20302-20310)
    error_estimator: str = "epsilon"              # "residual" (method cg only, F:14382-14405), "sigma", "energy", "none"
    loadstep_extrapolation_order: int = 0         # 0 = none, 1 = linear, ...  (F:14696, F:14830; method "polynomial")

    def __post_init__(self):
        self.N = self.nx * self.ny * self.nz
        self.eps = np.zeros((6, self.nx, self.ny, self.nz))
        self.residuals = []
        self.BC_P = voigt_id4()
        self.E = np.zeros(6)
        self.S = np.zeros(6)
        self.callback = None
        self.error = None
        self._F00 = np.zeros(6)
        self._set_bc_projector(self.BC_P)

    # -- constitutive ----------------------------------------------------
    def pk1(self, eps, alpha=1.0):
        """_mat->PK1 dispatch on mixing_rule  F:15129 (create_mixing_rule)"""
        if self.mixing_rule == "voigt":
            return pk1_voigt(eps, self.phis, self.mats, alpha)
        if self.mixing_rule == "laminate":
            return pk1_laminate(eps, self.phis, self.mats, self.normals, alpha)
        raise RuntimeError("Unknown mixing rule '%s'" % self.mixing_rule)

    def calc_stress(self, mu_0, lambda_0, eps, alpha=1.0):
        """calcStress  F:18134-18184: tau = P(eps) - 2 mu0 eps - lambda0 tr(eps) I"""
        beta = -alpha * 2 * mu_0
        gamma = -alpha * lambda_0
        P = self.pk1(eps, alpha)
        if beta != 0:
            P = P + beta * eps
        if gamma != 0:
            tr = eps[0] + eps[1] + eps[2]
            P[0] = P[0] + gamma * tr
            P[1] = P[1] + gamma * tr
            P[2] = P[2] + gamma * tr
        return P

    def calc_stress_const(self, mu_0, lambda_0, eps):
        """calcStressConst  F:17973-18020"""
        return hooke(eps, mu_0, lambda_0, 1.0)

    def mean_stress(self, eps=None):
        """calcMeanStress -> meanPK1  F:17793-17811, F:12312-12351 (C0 = 0)."""
        eps = self.eps if eps is None else eps
        P = self.pk1(eps, 1.0 / self.N)
        return P.reshape(6, -1).sum(axis=1)

    def mean_energy(self, eps=None):
        """meanW  F:12239-12262: sum of the voxels' energies / nxyz (mixing rule's W: F:12739-12750, F:13527-13540)."""
        eps = self.eps if eps is None else eps
        if self.mixing_rule == "voigt":
            W = energy_voigt(eps, self.phis, self.mats)
        elif self.mixing_rule == "laminate":
            W = energy_laminate(eps, self.phis, self.mats, self.normals)
        else:
            raise RuntimeError("Unknown mixing rule '%s'" % self.mixing_rule)
        return float(W.sum()) / self.N

    def create_error_estimator(self, method):
        """create_error_estimator  F:14940-14972 on the current strain field (the constructors measure it).  Returns
        update(gamma, gamma0) -> (abs_err, rel_err): ErrorEstimator::update / update_cg."""
        name = self.error_estimator
        if name == "epsilon":    # EpsilonErrorEstimator  F:14591-14637
            st = {"prev": self._norm9(self.component_norm(self.eps))}

            def update(gamma=None, gamma0=None):
                cur = self._norm9(self.component_norm(self.eps))
                abs_err = abs(st["prev"] - cur)
                st["prev"] = cur
                return abs_err, abs_err / (SMALLEST + cur)
            return update
        if name == "residual":   # ResidualErrorEstimator  F:14382-14405: update() is the throwing base-class one (F:14353)
            if method != "cg":
                raise RuntimeError("Selected error estimator is not compatible with the selected solution method")
            return lambda gamma, gamma0: (math.sqrt(gamma), math.sqrt(gamma / gamma0))
        if name == "none":       # NoneErrorEstimator  F:14370-14378
            return lambda gamma=None, gamma0=None: (1.0, 1.0)
        if name == "energy":     # EnergyErrorEstimator  F:14410-14468
            st = {"prev": self.mean_energy()}

            def update(gamma=None, gamma0=None):
                W = self.mean_energy()
                abs_err = abs(st["prev"] - W)
                st["prev"] = W
                return abs_err, abs_err / (SMALLEST + abs(W))
            return update
        if name == "sigma":      # SigmaErrorEstimator  F:14514-14587, created with _mode = 2 for every method (F:14949)
            m0 = self.mean_stress()
            st = {"prev": m0, "pp": m0, "iter": 0}

            def update(gamma=None, gamma0=None):
                m = self.mean_stress()
                if st["iter"] > 1:
                    abs_err = 0.5 * (self._norm9(st["pp"] - m) + self._norm9(st["prev"] - m))
                else:
                    abs_err = self._norm9(st["prev"] - m)
                rel_err = abs_err / (SMALLEST + self._norm9(m))
                st["pp"], st["prev"] = st["prev"], m
                st["iter"] += 1
                return abs_err, rel_err
            return update
        raise RuntimeError("Unknown error estimator '%s'" % name)

    def mean_strain(self):
        """TensorField::average  F:10171-10210"""
        return self.eps.reshape(6, -1).sum(axis=1) / self.N

    # -- staggered difference operators ---------------------------------
    def div_staggered(self, x):
        """divOperatorStaggered  F:18853-18908 (periodic offsets F:14867-14891)."""
        hx, hy, hz = self.nx / self.dx, self.ny / self.dy, self.nz / self.dz
        fx = lambda a: np.roll(a, -1, axis=0)
        fy = lambda a: np.roll(a, -1, axis=1)
        fz = lambda a: np.roll(a, -1, axis=2)
        bx = lambda a: np.roll(a, 1, axis=0)
        by = lambda a: np.roll(a, 1, axis=1)
        bz = lambda a: np.roll(a, 1, axis=2)
        y = np.empty((3,) + x.shape[1:])
        y[0] = (x[0] - bx(x[0])) * hx + (fy(x[5]) - x[5]) * hy + (fz(x[4]) - x[4]) * hz
        y[1] = (fx(x[5]) - x[5]) * hx + (x[1] - by(x[1])) * hy + (fz(x[3]) - x[3]) * hz
        y[2] = (fx(x[4]) - x[4]) * hx + (fy(x[3]) - x[3]) * hy + (x[2] - bz(x[2])) * hz
        return y

    def eps_staggered(self, E, u):
        """epsOperatorStaggered  F:18614-18692."""
        hx, hy, hz = self.nx / self.dx, self.ny / self.dy, self.nz / self.dz
        fx = lambda a: np.roll(a, -1, axis=0)
        fy = lambda a: np.roll(a, -1, axis=1)
        fz = lambda a: np.roll(a, -1, axis=2)
        bx = lambda a: np.roll(a, 1, axis=0)
        by = lambda a: np.roll(a, 1, axis=1)
        bz = lambda a: np.roll(a, 1, axis=2)
        y = np.empty((6,) + u.shape[1:])
        y[3] = E[3] + 0.5 * ((u[2] - by(u[2])) * hy + (u[1] - bz(u[1])) * hz)
        y[4] = E[4] + 0.5 * ((u[2] - bx(u[2])) * hx + (u[0] - bz(u[0])) * hz)
        y[5] = E[5] + 0.5 * ((u[1] - bx(u[1])) * hx + (u[0] - by(u[0])) * hy)
        y[0] = E[0] + (fx(u[0]) - u[0]) * hx
        y[1] = E[1] + (fy(u[1]) - u[1]) * hy
        y[2] = E[2] + (fz(u[2]) - u[2]) * hz
        return y

    # -- FFT wrappers ----------------------------------------------------
    def fft_vector(self, f):
        """fftVector  F:18481-18510: unnormalised r2c (FFTW, F:7232-7237) times 1/N."""
        scale = 1 / float(self.N)
        return np.fft.rfftn(f, axes=(1, 2, 3)) * scale

    def ifft_vector(self, fh):
        """fftInvVector  F:18513-18528: unnormalised c2r (F:7239-7244)."""
        return np.fft.irfftn(fh, s=(self.nx, self.ny, self.nz), axes=(1, 2, 3)) * float(self.N)

    # -- Green operator ---------------------------------------------------
    def g0_axis_tables(self):
        """Per-axis factors of G0OperatorFourierStaggeredGeneral  F:19838-19876.

        Returns for each axis (kpm[n], kp[n] complex) with xi = xi_0*m,
        m = idx if idx <= half else idx - n, half = n/2-1 (even) or n/2 (odd).
        The z table is cut to nzc entries.
        """
        out = []
        for n, d in ((self.nx, self.dx), (self.ny, self.dy), (self.nz, self.dz)):
            h = d / (2 * n)
            xi_0 = 2 * math.pi * h / d
            half = (n // 2 - 1) if (n % 2 == 0) else n // 2
            kpm = np.empty(n)
            kp = np.empty(n, dtype=np.complex128)
            for i in range(n):
                xi = xi_0 * (float(i) if i <= half else (float(i) - float(n)))
                s = math.sin(xi) / h
                # std::exp(std::complex(0, xi)) = (cos xi, sin xi)
                kpm[i] = s
                kp[i] = complex(s * math.cos(xi), s * math.sin(xi))
            out.append((kpm, kp))
        nzc = self.nz // 2 + 1
        out[2] = (out[2][0][:nzc], out[2][1][:nzc])
        return out

    def g0_apply(self, mu_0, lambda_0, fh, alpha=-1.0):
        """G0OperatorFourierStaggered(+General)  F:19749-19755, F:19834-19927.

        u_j = c1 f_j + c2 (sum_a f_a k+_a) k-_j ; zero frequency set to 0.
        """
        c10 = -alpha / mu_0
        c20 = -alpha / (mu_0 * (1 + mu_0 / (lambda_0 + mu_0)))
        (s0, kp0), (s1, kp1), (s2, kp2) = self.g0_axis_tables()
        kp = [kp0[:, None, None], kp1[None, :, None], kp2[None, None, :]]
        km = [-np.conj(k) for k in kp]  # (-Re, +Im)  F:19860
        norm2 = (s0 * s0)[:, None, None] + (s1 * s1)[None, :, None] + (s2 * s2)[None, None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            c1 = c10 / norm2
            c2 = c20 / (norm2 * norm2)
            c2_fkp = c2 * (fh[0] * kp[0] + fh[1] * kp[1] + fh[2] * kp[2])
            uh = np.empty_like(fh)
            for j in range(3):
                uh[j] = c1 * fh[j] + c2_fkp * km[j]
        uh[:, 0, 0, 0] = 0
        return uh

    def g0_staggered(self, mu_0, lambda_0, f, alpha=1.0):
        """G0OperatorStaggered (fft branch)  F:20101-20116"""
        return self.ifft_vector(self.g0_apply(mu_0, lambda_0, self.fft_vector(f), alpha))

    # -- boundary-condition projector -------------------------------------
    def _set_bc_projector(self, P):
        """setBCProjector  F:20599-20665 (pseudo inverse through the 9x9 extension)."""
        P = np.asarray(P, dtype=np.float64)
        se = math.sqrt(EPS)
        if P.shape != (6, 6) or np.linalg.norm(P - P.T) > se:
            raise RuntimeError("Projector is not symmetric")
        if np.linalg.norm(P - voigt_dyad4_mm(P, P)) > se:
            raise RuntimeError("Specified Projector is not a projector")
        mu_0 = self.mu_0
        C0 = 2 * mu_0 * voigt_id4() + self.lambda_0 * voigt_ii4()
        self.BC_P = P
        self.BC_Q = voigt_id4() - P
        if not np.any(self.BC_Q):
            # Q == 0 exactly (pure strain BC): Q:C0, M and MQ vanish for any C0,
            # also while mu_0 is still NaN (F:15340) -- the SVD of the zero matrix gives M = 0.
            self.BC_QC0 = np.zeros((6, 6))
            self.BC_M = np.zeros((6, 6))
            self.BC_MQ = np.zeros((6, 6))
            return
        self.BC_QC0 = voigt_dyad4_mm(self.BC_Q, C0)
        if math.isnan(mu_0):
            # run() calls setBCProjector before calcRefMaterial has replaced the NaN mu_0
            # (F:21354 vs F:21742); LAPACK just propagates NaN there, the values are
            # recomputed in calcRefMaterial before first use.
            self.BC_M = np.full((6, 6), np.nan)
            self.BC_MQ = np.full((6, 6), np.nan)
            return
        QC0Q = voigt_dyad4_mm(self.BC_QC0, self.BC_Q)
        A = np.empty((9, 9))
        for i in range(9):
            for j in range(i, 9):
                A[j, i] = A[i, j] = QC0Q[i if i < 6 else i - 3, j if j < 6 else j - 3]
        U, s, VT = np.linalg.svd(A)
        thr = math.sqrt(EPS) * np.linalg.norm(s)
        sinv = np.where(np.abs(s) > thr, 1.0 / np.where(s == 0, 1, s), 0.0)
        # reference: gesvd(A, s, U, VT) on a column-major view of a row-major
        # symmetric matrix, then M = VT*Sinv*U; for symmetric A this is A^+.
        M = (VT.T * sinv) @ U.T
        for i in range(3):
            for j in range(6):
                M[j, 3 + i] = 0.5 * (M[j, 3 + i] + M[j, 6 + i])
            for j in range(6):
                M[3 + i, j] = 0.5 * (M[3 + i, j] + M[6 + i, j])
        self.BC_M = M[:6, :6].copy()
        self.BC_MQ = voigt_dyad4_mm(self.BC_M, self.BC_Q)

    def set_bc_projector(self, P):
        self._set_bc_projector(P)

    def calc_bc_mean(self, E, S):
        """calcBCMean  F:20242-20245"""
        return E + self.bc_relax * voigt_dyad4_mv(self.BC_M, S - voigt_dyad4_mv(self.BC_QC0, E))

    # -- reference material -------------------------------------------------
    def tangent_eig_minmax(self):
        """getRefMaterial + eig  F:12153-12236, F:12472-12559 for isotropic phases.

        The per-voxel tangent (Voigt: F:12763-12771; laminate with
        tangent="approx": F:13611-13624) is isotropic with mu=sum w_p mu_p,
        lambda=sum w_p lambda_p, whose 6x6 matrix [[2mu I + lam 11^T, 0],[0, 2mu I]]
        has eigenvalues {2mu (x5), 2mu+3lam}.  The reference gets them from LAPACK
        dsyev per voxel; the closed form agrees to rounding (checked in tests
        against numpy.linalg.eigvalsh).
        """
        mu_bar, lam_bar = self._tangent_moduli()
        e1 = 2 * mu_bar
        e2 = 2 * mu_bar + 3 * lam_bar
        lo = min(float(np.min(e1)), float(np.min(e2)))
        hi = max(float(np.max(e1)), float(np.max(e2)))
        return lo, hi

    def _tangent_moduli(self):
        shape = (self.nx, self.ny, self.nz)
        if self.mixing_rule == "voigt":
            mu_bar = np.zeros(shape)
            lam_bar = np.zeros(shape)
            for phi, (mu, lam) in zip(self.phis, self.mats):
                w = np.where(phi > VOIGT_THRESHOLD, phi, 0.0)
                mu_bar = mu_bar + 2 * w * mu / 2  # two_mu = 2*alpha*mu with alpha=phi
                lam_bar = lam_bar + w * lam
            return mu_bar, lam_bar
        # laminate: weights c1 (first live phase) and 1-c1 (second), or phi for single
        nph = len(self.phis)
        p1 = np.full(shape, -1)
        p2 = np.full(shape, -1)
        c1 = np.zeros(shape)
        pure = np.zeros(shape, dtype=bool)
        for p in range(nph):
            phi = self.phis[p]
            live = ~pure & (phi != 0)
            is_pure = live & (phi == 1)
            p1 = np.where(is_pure, p, p1)
            p2 = np.where(is_pure, -1, p2)
            c1 = np.where(is_pure, phi, c1)
            pure |= is_pure
            live &= ~is_pure
            t1 = live & (p1 < 0)
            t2 = live & ~t1 & (p2 < 0)
            p1 = np.where(t1, p, p1)
            c1 = np.where(t1, phi, c1)
            p2 = np.where(t2, p, p2)
        mus = np.array([m[0] for m in self.mats])
        lams = np.array([m[1] for m in self.mats])
        c2 = np.where(p2 >= 0, 1.0 - c1, 0.0)
        mu_bar = c1 * mus[p1] + c2 * mus[np.maximum(p2, 0)]
        lam_bar = c1 * lams[p1] + c2 * lams[np.maximum(p2, 0)]
        return mu_bar, lam_bar

    def calc_ref_material(self):
        """calcRefMaterial  F:22283-22313: mu0 = 0.5*ref_scale*0.5*(lmin+lmax), lambda0 kept."""
        lo, hi = self.tangent_eig_minmax()
        if lo < 0:
            lo = 0.0  # F:12183-12223
        mu_0 = 0.5 * (lo + hi)
        mu_0 *= 0.5 * self.ref_scale
        self.mu_0 = mu_0
        self._set_bc_projector(self.BC_P)

    # -- the iteration -------------------------------------------------------
    def gamma_staggered(self, E, mu_0, lambda_0, tau, alpha=-1.0):
        """GammaOperatorStaggered  F:20288-20300."""
        if np.linalg.norm(self.BC_MQ) < EPS:   # initBCProjector F:20228-20239
            F0 = np.zeros(6)
        else:
            F0 = tau.reshape(6, -1).sum(axis=1) / self.N
        f = self.div_staggered(tau)
        u = self.g0_staggered(mu_0, lambda_0, f, alpha)
        eta = self.eps_staggered(E, u)
        # applyBCProjector F:20263-20270 (bc_relax == 1 => second term vanishes)
        R = alpha * (self.bc_relax * voigt_dyad4_mv(self.BC_MQ, F0)
                     - (1 - self.bc_relax) * voigt_dyad4_mv(self.BC_M, voigt_dyad4_mv(self.BC_QC0, self._F00)))
        eta = eta + R[:, None, None, None]
        return eta

    def gamma_collocated(self, E, mu_0, lambda_0, tau, alpha=-1.0, beta=0.0):
        """GammaOperatorCollocated  F:20302-20310 = fftTensor (1/N, F:18531-18560),
        GammaOperatorFourierCollocated  F:19381-19608 (freq_hack off), applyBCProjector on the zero frequency
        (F:20272-20279), fftInvTensor."""
        th = np.fft.rfftn(tau, axes=(1, 2, 3)) * (1 / float(self.N))
        F0 = th[:, 0, 0, 0].real.copy()   # initBCProjector(tau_hat)  F:20219-20225: the mean of tau, always
        xi = []
        for n, d in ((self.nx, self.dx), (self.ny, self.dy), (self.nz, self.dz)):
            half = (n // 2 - 1) if (n % 2 == 0) else n // 2
            xi.append(np.array([(1 / d) * (float(i) if i <= half else (float(i) - float(n))) for i in range(n)]))
        nzc = self.nz // 2 + 1
        xi0 = xi[0][:, None, None]
        xi1 = xi[1][None, :, None]
        xi2 = xi[2][None, None, :nzc]
        xi00, xi01, xi11 = xi0 * xi0, xi0 * xi1, xi1 * xi1
        xi02, xi12, xi22 = xi0 * xi2, xi1 * xi2, xi2 * xi2
        c10 = alpha / (4 * mu_0)
        c20 = -alpha / (mu_0 * (1 + mu_0 / (lambda_0 + mu_0)))
        with np.errstate(divide="ignore", invalid="ignore"):
            norm_xi2 = xi00 + xi11 + xi22
            c1 = c10 / norm_xi2
            c12 = c1 * 2
            c2 = c20 / (norm_xi2 * norm_xi2)
            c3, c4, c5 = c12 + c2 * xi00, c12 + c2 * xi11, c12 + c2 * xi22
            G = {}
            G[0, 0] = (c12 + c3) * xi00
            G[1, 0] = c2 * xi00 * xi11
            G[2, 0] = c2 * xi00 * xi22
            G[3, 0] = c2 * xi00 * xi12
            G[4, 0] = c3 * xi02
            G[5, 0] = c3 * xi01
            G[1, 1] = (c12 + c4) * xi11
            G[2, 1] = c2 * xi11 * xi22
            G[3, 1] = c4 * xi12
            G[4, 1] = c2 * xi11 * xi02
            G[5, 1] = c4 * xi01
            G[2, 2] = (c12 + c5) * xi22
            G[3, 2] = c5 * xi12
            G[4, 2] = c5 * xi02
            G[5, 2] = c2 * xi22 * xi01
            G[3, 3] = c1 * (xi11 + xi22) + c2 * xi11 * xi22
            G[4, 3] = (c1 + c2 * xi22) * xi01
            G[5, 3] = (c1 + c2 * xi11) * xi02
            G[4, 4] = c1 * (xi00 + xi22) + c2 * xi00 * xi22
            G[5, 4] = (c1 + c2 * xi00) * xi12
            G[5, 5] = c1 * (xi00 + xi11) + c2 * xi00 * xi11
            g = lambda i, j: G[(i, j)] if i >= j else G[(j, i)]
            eh = np.empty_like(th)
            for i in range(6):
                ey = th[0] * g(i, 0) + th[1] * g(i, 1) + th[2] * g(i, 2) + \
                    (th[3] * g(i, 3) + th[4] * g(i, 4) + th[5] * g(i, 5)) * 2.0
                eh[i] = ey + beta * th[i]
        eh[:, 0, 0, 0] = np.asarray(E, dtype=np.float64)   # F:19605-19607
        R = alpha * (self.bc_relax * voigt_dyad4_mv(self.BC_MQ, F0)
                     - (1 - self.bc_relax) * voigt_dyad4_mv(self.BC_M, voigt_dyad4_mv(self.BC_QC0, self._F00)))
        eh[:, 0, 0, 0] += R                                # applyBCProjector(eta_hat, alpha)  F:20272-20279
        return np.fft.irfftn(eh, s=(self.nx, self.ny, self.nz), axes=(1, 2, 3)) * float(self.N)

    def basic_scheme(self, E, eps):
        """basicScheme  F:20558-20578: eps <- E - Gamma0 : (C - C0) : eps"""
        self._F00 = eps.reshape(6, -1).sum(axis=1) / self.N if self.bc_relax != 1.0 else np.zeros(6)
        tau = self.calc_stress(self.mu_0, self.lambda_0, eps)
        if self.gamma_scheme == "collocated":
            return self.gamma_collocated(E, self.mu_0, self.lambda_0, tau, -1.0)
        return self.gamma_staggered(E, self.mu_0, self.lambda_0, tau, -1.0)

    def component_norm(self, eps):
        """TensorField::component_norm  F:10088-10138: sqrt(mean(eps_c^2))"""
        return np.sqrt((eps.reshape(6, -1) ** 2).sum(axis=1) / self.N)

    @staticmethod
    def _norm9(m6):
        """||fix_dim(m)||_2 over the 9 mirrored entries  F:14600-14609, F:14627"""
        m9 = np.concatenate([m6, m6[3:6]])
        return math.sqrt(float((m9 * m9).sum()))

    def bc_error(self, E_cur, S_cur):
        """bc_error  F:21129-21161"""
        Emean = self.mean_strain()
        Smean = self.mean_stress()
        P_Emean = voigt_dyad4_mv(self.BC_P, Emean)
        Q_Smean = voigt_dyad4_mv(self.BC_Q, Smean)
        PE = voigt_dyad4_mv(self.BC_P, E_cur)
        norm_E = voigt_norm2(PE)
        err_F = voigt_norm2(P_Emean - E_cur) / (1 if norm_E < self.bc_tol else norm_E)
        norm_S = voigt_norm2(S_cur)
        err_S = voigt_norm2(Q_Smean - S_cur) / (1 if norm_S < self.bc_tol else norm_S)
        return max(err_F, err_S)

    def run(self, E0, S0=None, P=None):
        """LSSolver::run -> runLoadsteppingSolver (one step, t=1) -> runBasic
        F:21247-21398, F:21584-21685, F:21716-21805, stop rule _converged F:21177-21244.
        Returns True on error like the reference (F:21674-21677)."""
        return self.run_load_steps(E0, S0, P, params=[0.0, 1.0])

    def run_load_steps(self, E0, S0=None, P=None, params=(0.0, 1.0), first=None, method="basic", step_callback=None):
        """run() with the <loadsteps> of the project  F:21247-21398 + runLoadsteppingSolver F:21584-21685: step i
        prescribes params[i] * (E0, S0), starts from the strain field of step i-1 (zeroed once, F:21379) and is followed
        by the load-step action (step_callback(i) -> True stops, F:21435-21447).  first_loadstep defaults to 1 for the
        standard two-entry list [0, 1] and to 0 otherwise (F:21591)."""
        E0 = np.asarray(E0, dtype=np.float64)
        S0 = np.zeros(6) if S0 is None else np.asarray(S0, dtype=np.float64)
        self.E, self.S = E0, S0
        self.residuals = []
        self.error = None
        if P is not None:
            self.BC_P = np.asarray(P, dtype=np.float64)
        self._set_bc_projector(self.BC_P)
        se = math.sqrt(EPS)
        if np.linalg.norm(voigt_dyad4_mv(self.BC_P, S0)) > se * np.linalg.norm(S0):
            raise RuntimeError("Incompatible stress boundary condition specified")
        if np.linalg.norm(voigt_dyad4_mv(self.BC_Q, E0)) > se * np.linalg.norm(E0):
            raise RuntimeError("Incompatible strain boundary condition specified")
        self.eps = np.zeros((6, self.nx, self.ny, self.nz))  # F:21379
        self._F00 = np.zeros(6)
        if first is None:
            first = 0 if len(params) > 2 else 1
        self.step_iterations = []
        last = []   # (parameter, converged strain field) of the previous steps  F:21586
        for istep in range(first, len(params)):
            t = float(params[istep])
            order = int(self.loadstep_extrapolation_order)
            if order > 0 and istep > first:   # F:21634-21650
                while len(last) > order:
                    last.pop(0)
                last.append((float(params[istep - 1]), self.eps.copy()))
                if len(last) >= 2:
                    self.eps = self.extrapolate_loadstep_polynomial(last, t)
            failed = self._run_cg_step(t * E0, t * S0) if method == "cg" else self._run_basic_step(t * E0, t * S0)
            self.step_iterations.append(self.iterations)
            if failed:
                return True
            if step_callback is not None and step_callback(istep):
                return True
        return False

    @staticmethod
    def extrapolate_loadstep_polynomial(last, t):
        """extrapolateLoadstepPolynomial  F:21468-21514: per voxel and component the polynomial through the values of the
        last steps, p = V^-1 f with the Vandermonde matrix V_ij = t_i^j, evaluated at t (sum_i t^i p_i)."""
        n = len(last)
        V = np.array([[lt ** j for j in range(n)] for lt, _ in last], dtype=np.float64)
        tpowers = np.array([t ** i for i in range(n)], dtype=np.float64)
        Vinv = np.linalg.solve(V, np.eye(n))   # lapack gesv on the identity  F:21485-21489
        f = np.stack([e for _, e in last])      # [n, 6, nx, ny, nz]
        p = np.tensordot(Vinv, f, axes=(1, 0))
        return np.tensordot(tpowers, p, axes=(0, 0))

    def _run_basic_step(self, E0, S0):
        """runBasic  F:21716-21805 for one load step"""
        # the estimator's constructor measures the field the step starts from  F:21722, F:14612-14618
        ee = self.create_error_estimator("basic")
        it = 1
        update_ref = self.update_ref != "never"
        E = E0
        while True:
            if update_ref:
                self.calc_ref_material()
                E = self.calc_bc_mean(E0, S0)
                update_ref = False
            self.eps = self.basic_scheme(E, self.eps)
            abs_err, rel_err = ee()   # ee->update()  F:21786
            # _converged F:21177-21244
            if math.isnan(rel_err):
                self.error = "NaN detected in solution. Aborting."
                return True
            self.residuals.append(rel_err)
            if self.callback is not None and self.callback():
                break
            if it >= self.maxiter:
                break
            if rel_err <= self.tol or abs_err <= self.abs_tol:
                if self.bc_error(E0, S0) <= self.bc_tol:
                    break
            it += 1
        self.iterations = it
        return False

    # -- CG on the same operator ---------------------------------------------------------
    def inner_l2(self, a, b, c=None):
        """innerProductL2  F:20871-20953 (3 arguments: a:(b-c)), F:20955-21038: sum a:b with the
        shear terms doubled, divided by N."""
        d = b if c is None else (b - c)
        s = a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + 2 * (a[3] * d[3] + a[4] * d[4] + a[5] * d[5])
        return float(s.sum()) / self.N

    def run_cg(self, E0, S0=None, P=None):
        """LSSolver::run with method=cg: runCGElasticity  F:23153-23247 (l2 inner product,
        epsilon error estimator, no residual re-initialisation)."""
        return self.run_load_steps(E0, S0, P, params=[0.0, 1.0], method="cg")

    def _run_cg_step(self, E0, S0):
        """runCGElasticity  F:23153-23247 for one load step; the error estimator (create_error_estimator) is updated
        through update_cg: 'residual' uses (gamma, gamma0), the others re-measure the strain field (F:14633, F:14465, F:14584)."""
        if self.update_ref != "never":
            self.calc_ref_material()
        E = self.calc_bc_mean(E0, S0)
        ee = self.create_error_estimator("cg")   # constructed on the field the step starts from  F:23168
        Z = np.zeros(6)
        eps = np.empty_like(self.eps)
        eps[:] = E[:, None, None, None]          # epsilon.setConstant(E)  F:23184
        r = self.basic_scheme(Z, eps)            # krylovOperator: -Gamma0 (C - C0) eps
        r = r + (E[:, None, None, None] - eps)   # adjustResidual  F:10012-10022
        gamma = self.inner_l2(r, r) + SMALLEST
        gamma0 = gamma
        p = r.copy()
        it = 0
        while True:
            w = self.basic_scheme(Z, p)
            alpha = self.inner_l2(p, p, w) + SMALLEST
            alpha = gamma / alpha
            eps = eps + alpha * p
            self.eps = eps
            abs_err, rel_err = ee(gamma, gamma0)   # ee->update_cg(gamma, gamma0)  F:23219
            if math.isnan(rel_err):
                self.error = "NaN detected in solution. Aborting."
                return True
            self.residuals.append(rel_err)
            if self.callback is not None and self.callback():
                break
            if it >= self.maxiter:
                break
            if rel_err <= self.tol or abs_err <= self.abs_tol:
                if self.bc_error(E0, S0) <= self.bc_tol:
                    break
            it += 1
            r = r + (-alpha) * (p - w)           # xpaymz  F:9993-10010
            delta = self.inner_l2(r, r) + SMALLEST
            beta = delta / gamma
            gamma = delta
            p = r + beta * p                     # xpay  F:9819-9838
        self.iterations = it
        return False

    def calc_effective_properties(self):
        """calc_effective_properties  F:26030-26088: six unit load cases,
        Ceff = S E^-1 (E = identity), last three columns halved."""
        S = np.empty((6, 6))
        iters = []
        for i in range(6):
            Ep = np.zeros(6)
            Ep[i] = 1.0
            if self.run(Ep):
                raise RuntimeError(self.error)
            S[:, i] = self.mean_stress()
            iters.append(self.iterations)
        Ceff = S @ np.linalg.inv(np.eye(6))
        Cv = Ceff.copy()
        Cv[:, 3:6] *= 0.5
        self.Ceff_voigt = Cv
        self.ceff_iterations = iters
        return Cv

    def get_field(self, name):
        """get_raw_field  F:15396-15684 (epsilon, sigma, u subset)."""
        if name == "epsilon":
            return self.eps.copy()
        if name == "sigma":
            return self.calc_stress(0.0, 0.0, self.eps)
        if name == "u":
            s = self.calc_stress_const(self.mu_0, self.lambda_0, self.eps)
            return self.g0_staggered(self.mu_0, self.lambda_0, self.div_staggered(s), 1.0)
        raise RuntimeError("Unknown field '%s'" % name)


def isotropic_laminate_ceff(layers):
    """Closed-form effective stiffness of an isotropic laminate stacked along x.

    Standard result (continuity of in-plane strain and of the traction on the
    lamination plane); used as an *independent* known answer for the pin test that
    mirrors demo/elasticity/laminate (the reference prints the same quantity via
    calc_isotropic_laminate F:26405-26474).  ``layers`` = [(phi, mu, lam)].
    Returns the conventional 6x6 Voigt matrix in fibergen's ordering 11,22,33,23,13,12.
    """
    # Backus-type averaging with lamination direction 1.
    phi = np.array([l[0] for l in layers])
    mu = np.array([l[1] for l in layers])
    lam = np.array([l[2] for l in layers])
    M = lam + 2 * mu
    avg = lambda a: float((phi * a).sum())
    C = np.zeros((6, 6))
    c11 = 1.0 / avg(1 / M)
    c12 = avg(lam / M) * c11
    c22 = avg(4 * mu * (lam + mu) / M) + avg(lam / M) ** 2 * c11
    c23 = avg(2 * mu * lam / M) + avg(lam / M) ** 2 * c11
    C[0, 0] = c11
    C[0, 1] = C[1, 0] = C[0, 2] = C[2, 0] = c12
    C[1, 1] = C[2, 2] = c22
    C[1, 2] = C[2, 1] = c23
    C[3, 3] = avg(mu)              # 23: in-plane shear -> Voigt average
    C[4, 4] = C[5, 5] = 1.0 / avg(1 / mu)  # 13, 12: out-of-plane shear -> Reuss
    return C
