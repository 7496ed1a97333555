"""Build / launch variants that are not the default, run as child processes with their switch set -- `pytest -m gpu` on an MI355X.
(Last in file order on purpose: a variant that has never run on hardware must not stop `-x` before the default paths were seen.)

  * HOP_QUADS_HASH=1: stage 1 of the congruent-set extraction in the reference's own structure -- first pairs binned by the cell of their
    invariant point (a chained hash in LDS), a second pair meets the 27 cells around its own (IndexedNormalSet,
    src/OpenGR_4pcs/src/gr/accelerators/normalset.hpp:111-253, FunctorSuper4pcs.h:131-293) -- instead of the dense pair x pair predicate
    of k_quads.  The generator tests compare the emitted hypothesis multiset with the REFERENCE-BUILT goldens and with the oracle
    (bit-equal, whatever order the fit queue is filled in), so they decide whether k_quads_hash enumerates the same quadrilaterals.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generator_goldens_with_the_binned_quadrilateral_stage():
    env = dict(os.environ, HOP_QUADS_HASH="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_zz_objects_golden.py"),
                        os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
                        "-k", "generator"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=3000)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and " passed" in r.stdout, tail
    n_passed = int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n_passed >= 15, tail      # 9 reference-built cases x modes, the explicit-trials case, the stand-in objects, the C2-size case
