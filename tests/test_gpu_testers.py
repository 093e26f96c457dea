"""The reference's testers (testers/*.m) run on the device solvers: each one's own pass criterion holds."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["lassotest", "ladtest", "huberfittest", "totalvariationtest", "basispursuittest",
                                  "modeltest"])
@pytest.mark.parametrize("seed", [0, 1])
def test_reference_tester_passes(gpu, name, seed):
    results, test = getattr(gpu.testers, name)(seed)
    assert test["failed"] == 0, {k: v for k, v in test.items() if not hasattr(v, "shape")}
    assert results["steps"] == test["steps"] >= 1


def test_linearsvmtest_both_losses(gpu):
    """linearsvmtest.m:153-190: the second run passes the string '0-1', i.e. hinge prox + 0-1 objective."""
    results, test = gpu.testers.linearsvmtest(0)
    assert set(results) == set(test) == {"hingeloss", "zoloss"}
    for key in ("hingeloss", "zoloss"):
        assert test[key]["relerror"] <= 0.05, (key, test[key]["relerror"])
    # same prox in both runs -> same iterates; only the recorded objective differs
    assert results["hingeloss"]["steps"] == results["zoloss"]["steps"]
    assert results["hingeloss"]["objopt"] != results["zoloss"]["objopt"]


def test_linearprogramtest_reports_the_reference_quantities(gpu):
    """linearprogramtest.m:122-134.  The planted point is only feasible, not optimal, so `failed` may be 1 --
    exactly as in the reference; the constraint residual criterion must hold."""
    results, test = gpu.testers.linearprogramtest(1)
    assert test["relerror"] <= 1e-3
    assert test["objopt"] <= test["trueobjopt"] * (1 + 1e-3)
