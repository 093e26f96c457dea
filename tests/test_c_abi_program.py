"""The C ABI used from plain C99 (tests/c_abi/abi_check.c): header is valid C, the library links, and the
engine either solves the problem (GPU) or refuses loudly (no device) -- never a silent CPU path."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "admm-project_amd")


def _build(tmp_path):
    exe = str(tmp_path / "abi_check")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-o", exe, "-L", LIBDIR, "-ladmm_hip",
           f"-Wl,-rpath,{LIBDIR}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_program_builds_and_refuses_without_device(ap, tmp_path):
    ap._lib.load()
    if ap._lib.device_count() > 0:
        pytest.skip("a device is present: covered by the gpu-marked test")
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("NO_DEVICE rc=-3"), out.stdout


@pytest.mark.gpu
def test_c_program_solves_lasso(gpu, tmp_path):
    from oracle import solvers_ref as S

    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.match(r"OK steps=(\d+) x=\[(\S+) (\S+)\] objopt=(\S+)", out.stdout)
    assert m, out.stdout
    D = np.asfortranarray(np.array([0.5, -0.5, 0.5, 0.5, 0.1, 0.7, -0.7, 0.1]).reshape((4, 2), order="F"))
    s = np.array([1.0, -0.2, 0.3, 0.8])
    ref = S.lasso(D, s, 0.05, dict(objevals=1, maxiters=50))
    assert int(m.group(1)) == ref["steps"]
    np.testing.assert_allclose([float(m.group(2)), float(m.group(3))], ref["xopt"], rtol=1e-9)
    assert float(m.group(4)) == pytest.approx(ref["objopt"], rel=1e-9)
