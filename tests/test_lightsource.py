"""Particle -> step requests (clsimhip_ppc_*, csrc/lightsource.cpp): the front end of
I3CLSimLightSourceToStepConverterPPC (private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx:94-132, 188-470).

What the reference tree defines is checked exactly (photon yield per metre against an independent integration, the step
split's integer arithmetic, the muon/cascade sharing, thresholds, error messages); what it takes from sim-services /
phys-services / GSL (shower parameters, the random service) is restated from published sources and is PARITY UNPINNED:
those parts are checked against their own closed forms and as distributions."""
import math

import numpy as np
import pytest
from scipy import integrate

from clsim_amd import converter as CV
from oracle import builders as B
from tests import common

PT = CV.ParticleType
DENSITY = 0.9216


def make_ppc(seed=2024):
    cfg = common.config("mie")
    p = CV.I3CLSimLightSourceToStepConverterPPC()
    p.SetWlenBias(CV.GetIceCubeDOMAcceptance())
    p.SetMediumProperties(cfg["med_p"])
    p.SetRandomSeed(seed)
    p.Initialize()
    return p


@pytest.fixture
def ppc():
    return make_ppc()


def particles(n, ptype, energy, length=np.nan, shape=CV.SHAPE_OTHER, first_id=0):
    p = np.zeros(n, dtype=CV.PARTICLE_DTYPE)
    p["type"], p["shape"], p["energy"], p["length"] = ptype, shape, energy, length
    p["dz"] = -1.0
    p["identifier"] = np.arange(first_id, first_id + n, dtype=np.uint32)
    return p


def photons_of(req):
    return req["num_steps"].astype(np.float64) * req["photons_per_step"] + req["num_photons_in_last_step"]


def test_photon_yield_per_metre_equals_an_independent_integration(ppc):
    """NumberOfPhotonsPerMeter (ConverterUtils.cxx:44-105): Frank-Tamm x wavelength bias over 1/lambda, here with scipy's
    adaptive quadrature on the oracle's refractive index and DOM acceptance (the reference integrates with
    gsl_integration_qag to 1e-5)."""
    med = common.config("mie")["med_o"]
    bias = B.icecube_dom_acceptance()
    n = med["n"]

    def index(w):
        x = w / 1e-6
        return n[0] + x * (n[1] + x * (n[2] + x * (n[3] + x * n[4])))

    def acceptance(w):
        q = (w - bias["start"]) / bias["step"]
        i = int(np.clip(math.floor(q), 0, len(bias["values"]) - 2))
        f = min(max(q - i, 0.0), 1.0)
        return bias["values"][i] * (1 - f) + bias["values"][i + 1] * f

    f = lambda e: acceptance(1.0 / e) * (2 * math.pi / 137.0) * (1.0 - 1.0 / index(1.0 / e) ** 2)
    lo, hi = 1.0 / med["max_wlen"], 1.0 / med["min_wlen"]
    edges = np.linspace(lo, hi, 90)
    want = sum(integrate.quad(f, a, b, epsrel=1e-10)[0] for a, b in zip(edges[:-1], edges[1:]))
    got = ppc.MeanPhotonsPerMeter(0)
    assert abs(got - want) / want < 1e-6
    assert 2000 < got < 3000                    # ~2450 detectable photons per metre of track
    assert ppc.MeanPhotonsPerMeter(170) == got  # the same refractive index function in every layer


def test_shower_parameters_follow_the_published_forms():
    """(a, b, emScale, emScaleSigma): a = alpha + beta ln E, b = Lrad / b0 with Lrad = 0.358 g/cm2 / density; no extension below
    1 GeV; hadrons: F = 1 - (E/E0)^-m (1 - f0), sigma = F rms0 (ln E)^-gamma.  (sim-services; parity unpinned)"""
    lrad = 0.358 / DENSITY
    a, b, f, s = CV.ShowerParameters(PT.EMinus, 40000.0, DENSITY)
    assert a == pytest.approx(2.01849 + 0.63176 * math.log(40000.0), rel=1e-12) and b == pytest.approx(lrad / 0.63207, rel=1e-12)
    assert (f, s) == (1.0, 0.0)
    assert CV.ShowerParameters(PT.EPlus, 100.0, DENSITY)[0] == pytest.approx(2.00035 + 0.63190 * math.log(100.0))
    assert CV.ShowerParameters(PT.Gamma, 100.0, DENSITY)[:2] == pytest.approx((2.83923 + 0.58209 * math.log(100.0), lrad / 0.64526))
    assert CV.ShowerParameters(PT.Brems, 100.0, DENSITY)[:2] == CV.ShowerParameters(PT.EMinus, 100.0, DENSITY)[:2]
    assert CV.ShowerParameters(PT.EMinus, 0.5, DENSITY)[:2] == (2.01849, 0.0)                  # ln E clamped at 0, b = 0 below 1 GeV
    a, b, f, s = CV.ShowerParameters(PT.Hadrons, 1000.0, DENSITY)
    F = 1.0 - (1000.0 / 0.18791678) ** -0.16267529 * (1.0 - 0.30974123)
    assert a == pytest.approx(1.58357292 + 0.41886807 * math.log(1000.0)) and b == pytest.approx(lrad / 0.33833116)
    assert f == pytest.approx(F) and s == pytest.approx(F * 0.95899551 * math.log(1000.0) ** -1.35589541)
    assert CV.ShowerParameters(99999, 1000.0, DENSITY) == (a, b, f, s)                          # unknown PDG code: "probably a hadron"
    assert CV.ShowerParameters(PT.Hadrons, 1000.0, 2 * DENSITY)[1] == pytest.approx(b / 2)        # radiation length scales with 1/density


def test_electron_cascades_split_like_the_reference(ppc):
    """:284-371: mean = photons/m x 5.21 m/GeV x (0.924/density) x E, Poisson; steps of photonsPerStep + one shorter step."""
    n = 4000
    req = ppc.EnqueueLightSources(particles(n, PT.EMinus, 10.0))
    assert len(req) == n and np.all(req["kind"] == CV.STEPS_CASCADE)
    assert np.all(req["photons_per_step"] == 200) and np.all(req["num_photons_in_last_step"] < 200)
    assert np.array_equal(req["identifier"], np.arange(n))
    a, b, _, _ = CV.ShowerParameters(PT.EMinus, 10.0, DENSITY)
    assert np.all(req["pa"] == np.float32(a)) and np.all(req["pb"] == np.float32(b))
    mean = ppc.MeanPhotonsPerMeter(0) * 5.21 * 0.924 / DENSITY * 10.0
    ph = photons_of(req)
    z = (ph - mean) / math.sqrt(mean)
    assert abs(z.mean()) < 4 / math.sqrt(n) and abs(z.var() - 1.0) < 0.12          # Poisson: variance = mean
    # the same light source gives the same numbers whatever else is enqueued with it, another one different numbers
    other = make_ppc()
    assert np.array_equal(photons_of(other.EnqueueLightSources(particles(10, PT.EMinus, 10.0, first_id=100))), ph[100:110])
    assert len(np.unique(ph)) > n // 10
    # an identifier that comes back (identifiers restarting per frame) does not replay its fluctuations: the reference's one
    # sequential random service never does; the n-th occurrence is reproducible in its turn
    again = photons_of(ppc.EnqueueLightSources(particles(10, PT.EMinus, 10.0, first_id=100)))
    assert not np.array_equal(again, ph[100:110])
    assert np.array_equal(photons_of(other.EnqueueLightSources(particles(10, PT.EMinus, 10.0, first_id=100))), again)


def test_small_and_huge_means(ppc):
    per_gev = ppc.MeanPhotonsPerMeter(0) * 5.21 * 0.924 / DENSITY
    # mean ~ 12.8 photons: Poisson by inversion, every photon in the short last step
    req = ppc.EnqueueLightSources(particles(20000, PT.EMinus, 1e-3))
    ph = photons_of(req)
    assert np.all(req["num_steps"] == 0) and abs(ph.mean() - per_gev * 1e-3) < 0.1 and abs(ph.var() / ph.mean() - 1.0) < 0.05
    assert np.all(req["pb"] == 0.0)                                             # E < 1 GeV: no cascade extension
    # mean 5.1e8 > 1e7: Gaussian approximation (:300-313), still 200 photons per step (< 1e9)
    req = ppc.EnqueueLightSources(particles(2000, PT.EMinus, 40000.0))
    ph = photons_of(req)
    mean = per_gev * 40000.0
    z = (ph - mean) / math.sqrt(mean)
    assert abs(z.mean()) < 0.1 and abs(z.std() - 1.0) < 0.06 and np.all(req["photons_per_step"] == 200)
    # above useHighPhotonsPerStepStartingFromNumPhotons = 1e9 photons: 2000 per step (:333-335)
    req = ppc.EnqueueLightSources(particles(4, PT.EMinus, 100000.0))
    assert np.all(req["photons_per_step"] == 2000) and np.all(photons_of(req) > 1e9)
    # zero energy: no photons, one empty request
    req = ppc.EnqueueLightSources(particles(1, PT.EMinus, 0.0))
    assert photons_of(req)[0] == 0


def test_hadronic_cascades_carry_the_electromagnetic_fraction(ppc):
    n = 4000
    req = ppc.EnqueueLightSources(particles(n, PT.Hadrons, 1000.0))
    _, _, F, sigma = CV.ShowerParameters(PT.Hadrons, 1000.0, DENSITY)
    full = ppc.MeanPhotonsPerMeter(0) * 5.21 * 0.924 / DENSITY * 1000.0
    frac = photons_of(req) / full
    assert np.all(frac > 0) and frac.max() < 1.0 + 5 / math.sqrt(full)          # f is redrawn until it lies in [0, 1] (:290-295)
    # f ~ N(F, sigma) truncated to [0, 1]: F = 0.83, sigma = 0.058 -- truncation at 1 is 3 sigma away
    assert abs(frac.mean() - F) < 0.005 and abs(frac.std() - sigma) < 0.004


def test_muons_make_two_requests_and_keep_the_reference_modulo(ppc):
    """:373-462: bare muon light and secondary-cascade light as separate requests over the track length; the short last step of
    the cascade-like request is numSteps % photonsPerStep as the reference writes it (:455)."""
    n = 3000
    E, L = 1000.0, 400.0
    req = ppc.EnqueueLightSources(particles(n, PT.MuMinus, E, length=L))
    assert len(req) == 2 * n
    mu, ca = req[0::2], req[1::2]
    assert np.all(mu["kind"] == CV.STEPS_MUON) and np.all(ca["kind"] == CV.STEPS_MUON_CASCADE)
    assert np.all(mu["length"] == np.float32(L)) and np.all(ca["length"] == np.float32(L))
    assert np.array_equal(mu["identifier"], ca["identifier"])
    extr = 1.0 + max(0.0, 0.1880 + 0.0206 * math.log(E))
    total = ppc.MeanPhotonsPerMeter(0) * L * extr
    assert abs(photons_of(mu).mean() / (total / extr) - 1.0) < 0.002
    assert np.array_equal(ca["num_photons_in_last_step"], (ca["num_steps"] % ca["photons_per_step"]).astype(np.uint32))
    cascade_photons = ca["num_steps"].astype(np.float64) * ca["photons_per_step"]     # up to the < 200 photons of the last step
    assert abs(cascade_photons.mean() / (total * (1.0 - 1.0 / extr)) - 1.0) < 0.005
    # taus are treated like muons; a muon without a length gets 2000 m (:377-380)
    tau = ppc.EnqueueLightSources(particles(2, PT.TauPlus, E, length=L))
    assert len(tau) == 4
    nolen = ppc.EnqueueLightSources(particles(1, PT.MuPlus, E))
    assert np.all(nolen["length"] == np.float32(2000.0))


def test_cascade_segments_and_errors(ppc):
    seg = ppc.EnqueueLightSources(particles(3, PT.EMinus, 50.0, length=7.5, shape=CV.SHAPE_CASCADE_SEGMENT))
    assert np.all(seg["kind"] == CV.STEPS_MUON_CASCADE) and np.all(seg["length"] == np.float32(7.5)) and np.all(seg["pa"] == 0)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="cascade segment with length"):
        ppc.EnqueueLightSources(particles(1, PT.EMinus, 50.0, length=0.0, shape=CV.SHAPE_CASCADE_SEGMENT))
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="not initialized"):
        CV.I3CLSimLightSourceToStepConverterPPC().EnqueueLightSources(particles(1, PT.EMinus, 1.0))
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="photonsPerStep may not be <= 0"):
        CV.I3CLSimLightSourceToStepConverterPPC(photonsPerStep=0)
    p = CV.I3CLSimLightSourceToStepConverterPPC()
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="WlenBias not set"):
        p.Initialize()


def test_requests_feed_the_step_counter(ppc):
    """the requests are what clsimhip_count_generated_steps / clsimhip_generate_steps consume: a 40 TeV electron of the
    reference's benchmark is 2.56 million steps"""
    req = ppc.EnqueueLightSources(particles(1, PT.EMinus, 40000.0))
    steps, padded = CV.CountGeneratedSteps(req, granularity=512)
    assert steps == int(req["num_steps"][0]) + (1 if req["num_photons_in_last_step"][0] else 0)
    assert padded % 512 == 0 and 0 <= padded - steps <= 512
    assert 2.4e6 < steps < 2.7e6


@pytest.mark.gpu
def test_benchmark_event_born_and_propagated_on_the_device(ppc):
    """particle -> requests -> steps born in HBM -> propagated there (bench.py --workload benchmark), against the oracle fed
    with the downloaded steps"""
    import torch
    from oracle import capi
    from clsim_amd.synthetic import PHOTON_DTYPE, STEP_DTYPE
    cfg = common.config("mie")
    req = ppc.EnqueueLightSources(particles(1, PT.EMinus, 60.0))                 # ~3800 steps
    n = CV.CountGeneratedSteps(req, granularity=256)[1]
    dev = torch.device("cuda", 0)
    d_steps = torch.zeros((n, 48), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    got = CV.GenerateStepsDevice(req, 99, d_steps.data_ptr(), n, granularity=256, device=0, stream=stream)
    assert got == n
    conv = common.product_converter(cfg, n)
    d_out = torch.zeros((1 << 16, 80), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    conv.PropagateDevice(d_steps.data_ptr(), n, d_out.data_ptr(), 1 << 16, d_cnt.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    steps = np.frombuffer(d_steps.cpu().numpy().tobytes(), dtype=STEP_DTYPE).copy()
    assert int(steps["num"].sum()) == int(photons_of(req)[0])
    cnt = int(d_cnt.item())
    ph = np.frombuffer(d_out[:cnt].cpu().numpy().tobytes(), dtype=PHOTON_DTYPE).copy()
    x, a = common.streams(n)
    ph_o, cnt_o, x_o, _ = capi.propagate(common.oracle_tables(cfg), steps, x, a, threads=8)
    assert cnt == cnt_o and common.sort_photons(ph).tobytes() == common.sort_photons(ph_o).tobytes()
    assert np.array_equal(conv.GetRNGState(n), x_o)


def test_flasher_front_end():
    """I3CLSimLightSourceToStepConverterFlasher Initialize / EnqueueLightSource (Flasher.cxx:120-159, 214-265): photons after
    bias = photons x correction factor (bias at the peak for a delta peak, ratio of integrals for a tabulated spectrum),
    Poisson below 1e6 and Gaussian above, empty pulses skipped."""
    bias = CV.GetIceCubeDOMAcceptance()
    ob = B.icecube_dom_acceptance()

    def acceptance(w):
        q = (w - ob["start"]) / ob["step"]
        i = int(np.clip(math.floor(q), 0, len(ob["values"]) - 2))
        f = min(max(q - i, 0.0), 1.0)
        return ob["values"][i] * (1 - f) + ob["values"][i + 1] * f

    c405 = CV.FlasherPhotonNumberCorrectionFactor(bias, peakWavelength=405e-9)
    assert c405 == pytest.approx(acceptance(405e-9), rel=1e-12)
    # a tabulated LED-like spectrum: Gaussian around 400 nm, sigma 15 nm, on a 2 nm grid
    grid = 300e-9 + 2e-9 * np.arange(151)
    spec = np.exp(-0.5 * ((grid - 400e-9) / 15e-9) ** 2)
    table = CV.I3CLSimFunctionFromTable(grid[0], 2e-9, spec)
    got = CV.FlasherPhotonNumberCorrectionFactor(bias, spectrumNoBias=table, fromWlen=300e-9, toWlen=600e-9)

    def sp(w):
        q = (w - grid[0]) / 2e-9
        i = int(np.clip(math.floor(q), 0, len(spec) - 2))
        f = min(max(q - i, 0.0), 1.0)
        return spec[i] * (1 - f) + spec[i + 1] * f

    edges = np.linspace(300e-9, 600e-9, 151)
    num = sum(integrate.quad(lambda w: sp(w) * acceptance(w), a, b, epsrel=1e-10)[0] for a, b in zip(edges[:-1], edges[1:]))
    den = sum(integrate.quad(sp, a, b, epsrel=1e-10)[0] for a, b in zip(edges[:-1], edges[1:]))
    assert got == pytest.approx(num / den, rel=1e-6)

    n = 6000
    pulses = np.zeros(n, dtype=CV.FLASHER_PULSE_DTYPE)
    pulses["dz"] = 1.0
    pulses["identifier"] = np.arange(n)
    pulses["source_type"] = 1
    pulses["pulse_width"] = 70.0
    pulses["num_photons_no_bias"] = 2.0e5
    pulses["num_photons_no_bias"][:10] = [0.0, -5.0, 1e-9, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]      # skipped (the third: Poisson gives 0)
    req = CV.EnqueueFlasherPulses(pulses, c405, seed=3)
    assert len(req) == n - 10 and np.array_equal(req["identifier"], np.arange(10, n))
    assert np.all(req["pulse_width"] == 70.0) and np.all(req["source_type"] == 1)
    mean = 2.0e5 * c405
    z = (req["num_photons_with_bias"].astype(np.float64) - mean) / math.sqrt(mean)
    assert abs(z.mean()) < 4 / math.sqrt(n) and abs(z.var() - 1.0) < 0.1
    # above 1e6 photons after bias: Gaussian
    pulses["num_photons_no_bias"] = 5.0e9
    req = CV.EnqueueFlasherPulses(pulses[:2000], c405, seed=4)
    mean = 5.0e9 * c405
    z = (req["num_photons_with_bias"].astype(np.float64) - mean) / math.sqrt(mean)
    assert abs(z.mean()) < 0.1 and abs(z.std() - 1.0) < 0.06


@pytest.mark.gpu
def test_cascade_extension_can_be_disabled_like_the_reference_test():
    """resources/tests/testCascadeExtension.py: a 1 GeV electron at the origin along +z -- with the cascade extension (default)
    every step lies at z > 0, without it every step is at the origin."""
    cfg = common.config("mie")
    converter = CV.I3CLSimLightSourceToStepConverterPPC()
    converter.SetWlenBias(CV.GetIceCubeDOMAcceptance())
    converter.SetMediumProperties(cfg["med_p"])
    converter.SetRandomSeed(0)
    converter.Initialize()
    p = np.zeros(1, dtype=CV.PARTICLE_DTYPE)
    p["type"], p["energy"], p["dz"], p["length"], p["identifier"] = PT.EMinus, 1.0, 1.0, np.nan, 1
    steps = CV.GenerateSteps(converter.EnqueueLightSources(p), seed=1)
    steps = steps[steps["num"] > 0]
    assert len(steps) > 50 and np.all(steps["z"] > 0), "Steps are spread along the cascade by default"
    converter.SetUseCascadeExtension(False)
    p["identifier"] = 2
    steps = CV.GenerateSteps(converter.EnqueueLightSources(p), seed=2)
    steps = steps[steps["num"] > 0]
    assert len(steps) > 50 and np.all(steps["z"] == 0), "Steps are all at the origin"
