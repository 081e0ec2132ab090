"""Generate tests/golden/oracle_vectors.json: regression vectors of the CPU oracle.

These are NOT outputs of the reference (it cannot be built here: no Go toolchain,
gonum / infergo absent -- DESIGN.md section 2).  They are outputs of the pure
numpy/scipy restatement (oracle.FastOracle(use_c=False)), written only after the
independent faithful C restatement (oracle.Oracle, the dense-dK algorithm of
gp/gp.go:418-499) agreed with it to 1e-9 at every size it can reach.  They pin the
two restatements against each other and against drift, and give the GPU parity
tests committed expectations for every kernel family at N in {2, 20, 64, 256, 1024}
(SURVEY.md section 8c).  The reference's own known answers are in
gp_test_known_answers.json.

Inputs come from the repository's counter-based generator (gogp_amd/synth.py,
splitmix64), so a vector is (case name, N, seed) -> expected numbers; the inputs of
the N <= 20 vectors are stored as well.

    python tests/golden/make_oracle_vectors.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from cases import CASES  # noqa: E402
from gogp_amd import synth  # noqa: E402
from oracle.oracle import FastOracle, Oracle  # noqa: E402

SIZES = [2, 20, 64, 256, 1024]
M = 8


def inputs(n, D, seed):
    X, y = synth.make_inputs(n, D, seed)
    Z = synth.make_test_points(M, D, seed + 1)
    return X, y, Z


def main():
    out = []
    for ci, (name, D, simil, noise, ts, tn) in enumerate(CASES):
        x = np.log(np.array(list(ts) + list(tn)))
        for si, n in enumerate(SIZES):
            seed = 7000 + 10 * ci + si
            X, y, Z = inputs(n, D, seed)
            f = FastOracle(D, simil, noise, use_c=False)
            f.set_data(X, y)
            lml = f.Observe(x)
            grad = f.Gradient()
            mu, sigma = f.Produce(Z)
            if n <= 256:  # the faithful algorithm is 4P N^3 with P dense matrices
                o = Oracle(D, simil, noise)
                o.set_data(X, y)
                lml_o = o.Observe(x)
                grad_o = o.Gradient()
                mu_o, sigma_o = o.Produce(Z)
                assert abs(lml - lml_o) <= 1e-9 * max(1, abs(lml_o)), (name, n, lml, lml_o)
                assert np.abs(grad - grad_o).max() <= 1e-9 * max(1, np.abs(grad_o).max()), (name, n)
                assert np.abs(mu - mu_o).max() <= 1e-9 * max(1, np.abs(mu_o).max()), (name, n)
                assert np.abs(sigma - sigma_o).max() <= 1e-8, (name, n)
            v = {"case": name, "n": n, "ndim": D, "seed": seed, "m": M, "log_theta": x.tolist(),
                 "x_sum": float(X.sum()), "y_sum": float(y.sum()), "z_sum": float(Z.sum()),
                 "lml": float(lml), "grad": grad.tolist(), "mu": mu.tolist(), "sigma": sigma.tolist(),
                 "checked_by_faithful_oracle": n <= 256}
            if n <= 20:
                v["X"] = X.tolist()
                v["y"] = y.tolist()
                v["Z"] = Z.tolist()
            out.append(v)
            print("%-18s n=%-5d lml=%.9f" % (name, n, lml), flush=True)
    doc = {"_comment": "Oracle regression vectors -- see make_oracle_vectors.py (NOT reference outputs). "
                       "Inputs: gogp_amd.synth.make_inputs(n, ndim, seed), make_test_points(m, ndim, seed+1).",
           "vectors": out}
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as fh:
        json.dump(doc, fh, indent=0)
    print("wrote %d vectors" % len(out))


if __name__ == "__main__":
    main()
