"""Print an md5 of the final coordinates of a short fixed run (regression guard when kernels are restructured:
the value must not change unless the arithmetic order was meant to change)."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, ".")
from chromosome3d_amd import Solver, default_model, default_schedule  # noqa: E402
from tests.util import load_if  # noqa: E402


def main():
    for cid, nrep in (("chr21_1mb", 3), ("chr1_500kb", 5)):
        s = Solver(0)
        s.set_model(default_model())
        s.set_if_matrix(load_if(cid))
        s.set_schedule(default_schedule(300), None, 0.0, 250)
        for opt in sys.argv[1:]:
            k, v = opt.split("=")
            s.set_option(k, float(v))
        s.init_replicas(nrep, 82364, 0)
        s.run()
        x = s.coords()
        ms, steps, launches = s.last_timing()
        print(cid, nrep, hashlib.md5(np.ascontiguousarray(x).tobytes()).hexdigest(), f"{ms:.2f} ms {steps} steps {launches} launches")


if __name__ == "__main__":
    main()
