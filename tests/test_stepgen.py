"""Step producer (the arithmetic inside the reference tree): the oracle's distributions against the closed
forms of the reference's formulas (CPU), and the GPU kernel against the oracle bit for bit (GPU)."""
import math

import numpy as np
import pytest
from scipy import stats

from clsim_amd import converter as CV
from oracle import capi
from tests import common


def requests():
    req = np.zeros(5, dtype=capi.REQUEST_DTYPE)
    d = np.array([0.3, -0.5, math.sqrt(1 - 0.09 - 0.25)])
    #          x    y    z    t     dx    dy    dz   len  pa    pb   kind id  pps last nsteps
    req[0] = (10., 20., 30., 5.0, d[0], d[1], d[2], 0.0, 3.2, 0.55, 0, 7, 200, 17, 40000)     # cascade, Cheng branch
    req[1] = (1., 2., 3., 10.0, 0.0, 0.0, 1.0, 0.0, 0.6, 0.9, 0, 8, 200, 0, 30000)           # cascade, Weibull branch, vertical
    req[2] = (1., 2., 3., 10.0, 0.6, 0.0, -0.8, 30.0, 0.0, 0.0, 1, 9, 150, 3, 20000)         # muon, cascade-like steps
    req[3] = (-5., 0., 9., 1.0, 0.0, 1.0, 0.0, 120.0, 0.0, 0.0, 2, 10, 200, 50, 10)          # muon-like steps
    req[4] = (0., 0., 0., 0.0, 0.0, 0.0, -1.0, 0.0, 2.0, 0.0, 0, 11, 100, 1, 0)              # no extension, only a last step
    return req


def direction(steps):
    st, ct = np.sin(steps["theta"].astype(np.float64)), np.cos(steps["theta"].astype(np.float64))
    return np.stack([st * np.cos(steps["phi"]), st * np.sin(steps["phi"]), ct], axis=1)


def test_step_distributions_follow_the_reference_formulas():
    req = requests()
    first, real, padded = capi.plan_generated_steps(req, 256)
    assert real == 40001 + 30000 + 20001 + 11 + 1 and padded % 256 == 0 and padded - real < 256
    steps = capi.generate_steps(req, 2024, granularity=256)
    assert len(steps) == padded
    # bookkeeping: photons per step, last steps, identifiers, no-op padding (Async.cxx:240-257)
    for r in range(5):
        s = steps[int(first[r]):int(first[r + 1])]
        assert np.all(s["id"] == req["identifier"][r]) and np.all(s["weight"] == 1.0) and np.all(s["beta"] == 1.0)
        assert np.all(s["num"][:int(req["num_steps"][r])] == req["photons_per_step"][r])
        if req["num_photons_in_last_step"][r]:
            assert s["num"][-1] == req["num_photons_in_last_step"][r]
    pad = steps[real:]
    assert np.all(pad["num"] == 0) and np.all(pad["weight"] == 0) and np.allclose(pad["theta"], math.pi)
    # FillStep(CascadeStepData_t) (PPC.cxx:524-537): distance along the axis = pb * Gamma(pa)
    for r in (0, 1):
        s = steps[int(first[r]):int(first[r + 1])]
        axis = np.array([req["dx"][r], req["dy"][r], req["dz"][r]], dtype=np.float64)
        rel = np.stack([s["x"] - req["x"][r], s["y"] - req["y"][r], s["z"] - req["z"][r]], axis=1).astype(np.float64)
        along = rel @ axis
        assert np.abs(rel - np.outer(along, axis)).max() < 1e-3                    # on the axis
        assert np.allclose(s["t"], req["time"][r] + along / 0.299792458, atol=2e-3)
        ks = stats.kstest(along, stats.gamma(a=float(req["pa"][r]), scale=float(req["pb"][r])).cdf)
        assert ks.pvalue > 1e-3, (r, ks)
        assert np.all(s["length"] == np.float32(0.001))
    # FillStep(MuonStepData_t, cascade-like) (:539-551): uniform along the track
    s = steps[int(first[2]):int(first[3])]
    axis = np.array([0.6, 0.0, -0.8])
    along = np.stack([s["x"] - 1., s["y"] - 2., s["z"] - 3.], axis=1).astype(np.float64) @ axis
    assert along.min() >= 0 and along.max() < 30.0 and stats.kstest(along / 30.0, "uniform").pvalue > 1e-3
    # FeederThread (:755-757): cos = 1 - (-ln(1 - u I)/b)^(1/a) <=> F(c) = (1 - exp(-b (1-c)^a)) / I, azimuth uniform
    a, b = 0.39, 2.61
    norm = 1.0 - math.exp(-b * 2.0 ** a)
    for r, axis in ((0, np.array([req["dx"][0], req["dy"][0], req["dz"][0]], dtype=np.float64)), (1, np.array([0., 0., 1.])), (2, np.array([0.6, 0., -0.8]))):
        s = steps[int(first[r]):int(first[r + 1])]
        cosang = np.clip(direction(s) @ axis, -1.0, 1.0)
        survival = lambda c: 1.0 - (1.0 - np.exp(-b * np.power(np.clip(1.0 - c, 0.0, 2.0), a))) / norm   # P(cos <= c)
        assert stats.kstest(cosang, survival).pvalue > 1e-3, r
        # azimuth around the axis
        ref = np.array([1.0, 0.0, 0.0]) if abs(axis[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
        e1 = np.cross(axis, ref); e1 /= np.linalg.norm(e1); e2 = np.cross(axis, e1)
        dvec = direction(s)
        az = (np.arctan2(dvec @ e2, dvec @ e1) + 2 * np.pi) % (2 * np.pi)
        assert stats.kstest(az / (2 * np.pi), "uniform").pvalue > 1e-3, r
    # GenerateStepForMuon (:821-842): the whole track, unsmeared
    s = steps[int(first[3]):int(first[4])]
    assert np.all(s["length"] == 120.0) and np.all(s["x"] == -5.0) and np.allclose(direction(s), [0, 1, 0], atol=1e-6)
    # no cascade extension (useCascadeExtension_ = false -> pb = 0): at the vertex
    s = steps[int(first[4]):int(first[5])]
    assert len(s) == 1 and s["num"][0] == 1 and s["x"][0] == 0 and s["z"][0] == 0
    # a different seed gives different steps, the same seed the same ones
    assert np.array_equal(capi.generate_steps(req, 2024, 256), steps) and not np.array_equal(capi.generate_steps(req, 2025, 256)["x"], steps["x"])


def test_request_validation():
    req = requests()[:1].copy()
    assert CV.CountGeneratedSteps(req, 256) == (40001, 40192)
    bad = req.copy(); bad["kind"] = 5
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="unknown step request kind"):
        CV.CountGeneratedSteps(bad)
    bad = req.copy(); bad["pa"] = 0.0
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="shape parameter"):
        CV.CountGeneratedSteps(bad)
    bad = req.copy(); bad["photons_per_step"] = 0
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="photonsPerStep"):
        CV.CountGeneratedSteps(bad)


@pytest.mark.gpu
def test_gpu_steps_equal_the_oracle_and_propagate():
    import torch
    req = requests()
    want = capi.generate_steps(req, 99, granularity=256)
    got = CV.GenerateSteps(req, 99, granularity=256)
    assert got.tobytes() == want.tobytes()
    # born in HBM and propagated from there: same photons as the same steps uploaded from the host
    cfg = common.config("mie")
    n = len(want)
    conv = common.product_converter(cfg, n)
    dev = torch.device("cuda", 0)
    d_steps = torch.zeros((n, 48), dtype=torch.uint8, device=dev)
    assert CV.GenerateStepsDevice(req, 99, d_steps.data_ptr(), n, granularity=256, stream=torch.cuda.current_stream().cuda_stream) == n
    assert d_steps.cpu().numpy().tobytes() == want.tobytes()
    d_out = torch.zeros((1 << 16, 80), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    conv.PropagateDevice(d_steps.data_ptr(), n, d_out.data_ptr(), 1 << 16, d_cnt.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cnt = int(d_cnt.item())
    conv2 = common.product_converter(cfg, n)
    conv2.EnqueueSteps(want, 1)
    _, ph = conv2.GetConversionResult()
    assert cnt == len(ph) and cnt > 100
