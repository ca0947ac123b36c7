import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common
from tests.test_verbatim_cl import tab_setup, ANGULAR
from clsim_amd import converter as CV, tabulator as TB, synthetic as S
from oracle import builders as B, capi
for case in ("tabulate", "tabulate360"):
    cfg, axes, tb, steps, x, a, ref, f = tab_setup(case)
    p_axes = TB.SphericalAxes([(TB.PowerAxis if ax["kind"] == "power" else TB.LinearAxis)(*((ax["min"], ax["max"], ax["n_bins"]) +
                              ((ax["power"],) if ax["kind"] == "power" else ()))) for ax in axes])
    x256, a256 = common.streams(256)
    padded = S.cascade_steps(64, seed=5, vertex=(3.0, -2.0, 10.0), photons_per_step=12, pad_to=256)
    tab = TB.I3CLSimStepToTableConverterHIP(0, p_axes, False, cfg["med_p"], np.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(),
                                            TB.I3CLSimFunctionPolynomial(ANGULAR), (x256, a256))
    tab.EnqueueSteps(padded, tuple(float(v) for v in ref[:7]))
    tab.Finish()
    sums = tab.GetBinSums().ravel()
    expect = np.zeros(int(f["n_bins"]), dtype=np.float64)
    expect[f["bins_nonzero"]] = f["bins_sum"]
    d = np.nonzero((sums > 0) != (expect > 0))[0]
    print(case, "bins", len(sums), "occupied dev", int((sums > 0).sum()), "expect", int((expect > 0).sum()), "pattern differs in", len(d), "sum dev %.9g expect %.9g" % (sums.sum(), expect.sum()))
    shape = tab.shape
    for i in d[:20]:
        print("   bin", i, np.unravel_index(i, shape), "dev", sums[i], "expect", expect[i])
    nz = (expect > 0) & (sums > 0)
    rel = np.abs(sums[nz] - expect[nz]) / expect[nz]
    print("   common bins: max rel diff", rel.max(), "n rel>1e-12:", int((rel > 1e-12).sum()))
    # the device's RNG states against the oracle's
    xs = tab.GetRNGState(256) if hasattr(tab, "GetRNGState") else None
    if xs is not None:
        print("   rng states equal:", np.array_equal(xs[:64], f["rng_x"]))
