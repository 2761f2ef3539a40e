"""Elastic-constant conversion of the project layer: any of the ten supported pairs
(K,E) (K,lambda) (K,mu) (K,nu) (E,mu) (E,nu) (lambda,mu) (lambda,nu) (mu,nu) (mu,M)
-> all six constants.  Mirrors Material::readSettings and calc_from_* (F:7333-7454),
including the "Incomplete" / "Ambiguous material definition" errors."""
from __future__ import annotations

PAIRS = [("K", "E"), ("K", "lambda"), ("K", "mu"), ("K", "nu"), ("E", "mu"),
         ("E", "nu"), ("lambda", "mu"), ("lambda", "nu"), ("mu", "nu"), ("mu", "M")]
NAMES = ("K", "E", "lambda", "mu", "nu", "M")


def _from_K_E(K, E):
    return dict(lam=(3 * K * (3 * K - E)) / (9 * K - E), mu=(3 * K * E) / (9 * K - E), nu=(3 * K - E) / (6 * K),
                M=(3 * K * (3 * K + E)) / (9 * K - E), K=K, E=E)


def _from_K_lambda(K, lam):
    return dict(E=(9 * K * (K - lam)) / (3 * K - lam), mu=(3 * (K - lam)) / 2, nu=lam / (3 * K - lam),
                M=3 * K - 2 * lam, K=K, lam=lam)


def _from_K_mu(K, mu):
    return dict(E=(9 * K * mu) / (3 * K + mu), lam=K - (2 * mu) / 3, nu=(3 * K - 2 * mu) / (2 * (3 * K + mu)),
                M=K + (4 * mu) / 3, K=K, mu=mu)


def _from_K_nu(K, nu):
    return dict(E=3 * K * (1 - 2 * nu), lam=(3 * K * nu) / (1 + nu), mu=(3 * K * (1 - 2 * nu)) / (2 * (1 + nu)),
                M=(3 * K * (1 - nu)) / (1 + nu), K=K, nu=nu)


def _from_E_mu(E, mu):
    return dict(K=(E * mu) / (3 * (3 * mu - E)), lam=(mu * (E - 2 * mu)) / (3 * mu - E), nu=E / (2 * mu) - 1,
                M=(mu * (4 * mu - E)) / (3 * mu - E), E=E, mu=mu)


def _from_E_nu(E, nu):
    return dict(K=E / (3 * (1 - 2 * nu)), lam=(E * nu) / ((1 + nu) * (1 - 2 * nu)), mu=E / (2 * (1 + nu)),
                M=(E * (1 - nu)) / ((1 + nu) * (1 - 2 * nu)), E=E, nu=nu)


def _from_lambda_mu(lam, mu):
    return dict(K=lam + (2 * mu) / 3, E=(mu * (3 * lam + 2 * mu)) / (lam + mu), nu=lam / (2 * (lam + mu)),
                M=lam + 2 * mu, lam=lam, mu=mu)


def _from_lambda_nu(lam, nu):
    return dict(K=(lam * (1 + nu)) / (3 * nu), E=(lam * (1 + nu) * (1 - 2 * nu)) / nu, mu=(lam * (1 - 2 * nu)) / (2 * nu),
                M=(lam * (1 - nu)) / nu, lam=lam, nu=nu)


def _from_mu_nu(mu, nu):
    return dict(K=(2 * mu * (1 + nu)) / (3 * (1 - 2 * nu)), E=2 * mu * (1 + nu), lam=(2 * mu * nu) / (1 - 2 * nu),
                M=(2 * mu * (1 - nu)) / (1 - 2 * nu), mu=mu, nu=nu)


def _from_mu_M(mu, M):
    return dict(K=M - (4 * mu) / 3, E=(mu * (3 * M - 4 * mu)) / (M - mu), lam=M - 2 * mu,
                nu=(M - 2 * mu) / (2 * M - 2 * mu), mu=mu, M=M)


_CALC = {("K", "E"): _from_K_E, ("K", "lambda"): _from_K_lambda, ("K", "mu"): _from_K_mu, ("K", "nu"): _from_K_nu,
         ("E", "mu"): _from_E_mu, ("E", "nu"): _from_E_nu, ("lambda", "mu"): _from_lambda_mu,
         ("lambda", "nu"): _from_lambda_nu, ("mu", "nu"): _from_mu_nu, ("mu", "M"): _from_mu_M}


def material_constants(attrs, evaluate=float):
    """attrs: mapping name -> string/number for a subset of K,E,lambda,mu,nu,M (+ anything else).
    Returns dict with K, E, lambda, mu, nu, M."""
    present = [k for k in NAMES if k in attrs]
    icalc = None
    for pair in PAIRS:  # the last complete pair in table order wins, as in F:7358-7364
        if pair[0] in attrs and pair[1] in attrs:
            icalc = pair
    if icalc is None:
        raise RuntimeError("Incomplete material definition")
    if any(k not in icalc for k in present):
        raise RuntimeError("Ambiguous material definition")
    r = _CALC[icalc](evaluate(attrs[icalc[0]]), evaluate(attrs[icalc[1]]))
    return {"K": r["K"], "E": r["E"], "lambda": r["lam"], "mu": r["mu"], "nu": r["nu"], "M": r["M"]}
