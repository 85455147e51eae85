"""dev (GPU box): the eigensolver of two builds of the library on the same matrices, BIT FOR BIT:
    python scripts/dev_eig_ab.py libA.so libB.so [n ...]        ("-" = the in-tree library)
Each library decomposes in a child process of its own; B, D and the sampler's packed operand are
compared with array_equal."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib, ns, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bboptpy_amd import _ffi
    if lib != "-":
        _ffi.LIB_PATH = os.path.abspath(lib)
    import bboptpy_amd as hip
    res = {}
    for n in ns:
        for P, bit in ((1, 0), (1, 4194304), (40, 0)):
            rng = np.random.default_rng(n)
            X = rng.normal(size=(n, 3 * n)) * np.logspace(0, -3, n)[:, None]
            mats = [np.eye(n), np.eye(n) + 1e-3 * (X @ X.T) / (3 * n), X @ X.T / (3 * n)]
            for mi, Cm in enumerate(mats):
                Cm = 0.5 * (Cm + Cm.T)
                g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=max(4, 2 * n), seed=1, populations=P)
                g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros((P, n)))
                if bit:
                    g.set_state("dbg", [float(bit)])
                for p in range(P):
                    g.set_state("C", Cm * (1. + 0.01 * p), p)
                    g.set_state("fev", [10 ** 6], p)
                    g.set_state("eigenlastev", [0], p)
                g.phase(_ffi.PHASE_EIGEN)
                for p in (0, P - 1):
                    res["n%d_P%d_b%d_m%d_p%d_B" % (n, P, bit, mi, p)] = g.get_state("B", p)
                    res["n%d_P%d_b%d_m%d_p%d_D" % (n, P, bit, mi, p)] = g.get_state("D", p)
    np.savez(out, **res)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], [int(v) for v in sys.argv[4:]], sys.argv[3])
        sys.exit(0)
    la, lb = sys.argv[1], sys.argv[2]
    ns = sys.argv[3:] or ["20", "33", "64", "65", "96", "100", "127", "128", "160", "256", "300"]
    outs = []
    for k, lib in enumerate((la, lb)):
        out = "/tmp/eig_ab_%d.npz" % k
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", lib, out] + ns)
        outs.append(np.load(out))
    bad = 0
    for key in outs[0].files:
        same = np.array_equal(outs[0][key], outs[1][key])
        if not same:
            bad += 1
            d = np.abs(outs[0][key] - outs[1][key]).max()
            print("DIFFERS %-28s max |a - b| = %.3e" % (key, d))
    print("%d arrays compared, %d differ" % (len(outs[0].files), bad))
