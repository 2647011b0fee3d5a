"""Generate the golden vectors in tests/golden/*.json.

Each fixture is DATA: inputs (A, theta_inv, regP, regD, xi_p, xi_d) of the augmented system
defined at /root/reference/src/KKT/KKT.jl:70-75

    [ -(Theta^-1 + Rp)   A' ] [dx]   [xi_d]
    [        A           Rd ] [dy] = [xi_p]

and the expected (dx, dy), obtained by a dense float64 numpy solve of that (n+m)x(n+m) system
(independent of every sparse code in this repository, oracle included).

* kat1  : the reference's own KKT fixture -- A = [1 0 1 0; 0 1 0 1] from
          /root/reference/test/KKT/Cholmod/cholmod.jl:3-6 with the all-ones data of
          /root/reference/src/KKT/Test/test.jl:26-36.  Closed form: S = 2I, dy = [1,1], dx = 0.
* kat2  : asymmetric 2x2 data, also carries S and chol(S).
* rand* : seeded random sparse instances, incl. free variables (theta_inv = 0) and the
          late-IPM scaling regime theta_inv in 10^[-8, 8].

Run:  python tests/golden/make_golden.py      (rewrites the JSON files deterministically)
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def dense_reference(A, th, rp, rd, xp, xd):
    m, n = A.shape
    K = np.zeros((n + m, n + m))
    K[:n, :n] = -np.diag(th + rp)
    K[:n, n:] = A.T
    K[n:, :n] = A
    K[n:, n:] = np.diag(rd)
    sol = np.linalg.solve(K, np.concatenate([xd, xp]))
    # one step of refinement in extended precision keeps the fixture at the rounding floor
    r = np.concatenate([xd, xp]).astype(np.longdouble) - K.astype(np.longdouble) @ sol.astype(np.longdouble)
    sol = sol + np.linalg.solve(K, r.astype(np.float64))
    return sol[:n], sol[n:]


def pack(name, A, th, rp, rd, xp, xd, extra=None):
    dx, dy = dense_reference(A, th, rp, rd, xp, xd)
    S = A @ np.diag(1.0 / (th + rp)) @ A.T + np.diag(rd)
    d = {
        "cond_S": float(np.linalg.cond(S)),     # sets the attainable accuracy of any K1 solver
        "name": name, "m": int(A.shape[0]), "n": int(A.shape[1]),
        "A": A.tolist(), "theta_inv": th.tolist(), "regP": rp.tolist(), "regD": rd.tolist(),
        "xi_p": xp.tolist(), "xi_d": xd.tolist(), "dx": dx.tolist(), "dy": dy.tolist(),
    }
    if extra:
        d.update(extra)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(d, f, indent=1)
    return d


def main():
    # KAT-1: reference fixture
    A = np.array([[1.0, 0, 1, 0], [0, 1, 0, 1]])
    pack("kat1", A, np.ones(4), np.ones(4), np.ones(2), np.ones(2), np.ones(4))
    # KAT-2
    A = np.array([[1.0, 1.0], [1.0, -1.0]])
    th, rp, rd = np.array([2.0, 3.0]), np.array([0.5, 0.25]), np.array([0.1, 0.2])
    D = 1.0 / (th + rp)
    S = A @ np.diag(D) @ A.T + np.diag(rd)
    pack("kat2", A, th, rp, rd, np.array([1.0, 2.0]), np.array([3.0, -1.0]),
         extra={"S": S.tolist(), "cholS": np.linalg.cholesky(S).tolist()})
    # seeded random instances
    rng = np.random.default_rng(20260927)
    for idx, (m, n, dens, regime) in enumerate([(7, 12, 0.4, "mid"), (15, 40, 0.2, "mid"),
                                                (30, 55, 0.1, "late"), (24, 24, 0.3, "free")]):
        A = rng.standard_normal((m, n)) * (rng.random((m, n)) < dens)
        for i in range(m):                       # no empty rows: keeps S nonsingular without Rd
            if not A[i].any():
                A[i, rng.integers(n)] = 1.0
        if regime == "mid":
            th = 10.0 ** rng.uniform(-3, 3, n); rp = np.full(n, 1e-4); rd = np.full(m, 1e-4)
        elif regime == "late":
            th = 10.0 ** rng.uniform(-8, 8, n); rp = np.full(n, 1.5e-8); rd = np.full(m, 1.5e-8)
        else:
            th = 10.0 ** rng.uniform(-2, 2, n); th[rng.random(n) < 0.3] = 0.0
            rp = np.full(n, 1e-3); rd = np.full(m, 1e-6)
        pack(f"rand{idx}", A, th, rp, rd, rng.standard_normal(m), rng.standard_normal(n),
             extra={"regime": regime})


if __name__ == "__main__":
    main()
