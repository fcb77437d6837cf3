"""A fixed-seed slice of tools/fuzz_gpu.py in the GPU suite: random shapes around every boundary of the kernels
(B, N, D, k incl. 1 / 63-65 / 256-257 / 1024), random / clustered (also bank-sorted) / ascending / all-ties data --
mi_knn_topk bit-exact against the chain oracle (fp32) and within the rounding bound (bf16), mi_knn_vote on its output within
1e-5 with a well-formed ranking; and of tools/fuzz_moco_gpu.py: MoCo loss / dq / dkey and the soft-kNN against the float64 oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixed_seed_fuzz_slice():
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tools", "fuzz_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(cases=48, seed=2, verbose=False) == 48


def test_fixed_seed_moco_soft_knn_fuzz_slice():
    spec = importlib.util.spec_from_file_location("fuzz_moco_gpu", os.path.join(ROOT, "tools", "fuzz_moco_gpu.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run(cases=60, seed=1, verbose=False) == 60
