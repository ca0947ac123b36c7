"""Wavelength generators for emission spectra other than Cherenkov light: I3CLSimModuleHelper::makeWavelengthGenerator
(I3CLSimModuleHelper.cxx:73-171), I3CLSimFunctionFromTable with its own wavelengths (FromTable.cxx:57-70, :123-145),
I3CLSimRandomValueInterpolatedDistribution(x, y) (InterpolatedDistribution.cxx:40-55, :148-154, :292-297) and the flasher LEDs'
measured spectra (python/GetIceCubeFlasherSpectrum.py; the data files of resources/flasher_data/).  Host side: C ABI against
the oracle's builders and against plain numpy; the sampling itself is pinned on the reference's generated code by
tests/test_verbatim_cl.py (flasher_led405) and checked here as a distribution."""
import numpy as np
import pytest

from clsim_amd import converter as CV
from oracle import builders as B
from oracle import capi
from tests import common

LEDS = ["LED340nm", "LED370nm", "LED405nm", "LED450nm", "LED505nm"]


@pytest.mark.parametrize("kind", LEDS + ["SC1", "SC2"])
def test_make_wavelength_generator_equals_the_oracles(kind):
    cfg = common.config("lea")
    g_p = CV.makeWavelengthGenerator(CV.GetIceCubeFlasherSpectrum(kind), CV.GetIceCubeDOMAcceptance(), cfg["med_p"])
    g_o = B.make_wavelength_generator(B.flasher_spectrum(kind, CV.FLASHER_DATA), B.icecube_dom_acceptance(), cfg["med_o"])
    if kind.startswith("SC"):
        assert isinstance(g_p, CV.I3CLSimRandomValueConstant) and g_p.value == g_o["value"] == 337e-9
        return
    assert g_o["kind"] == "interp_x" and g_p.x is not None
    assert np.array_equal(g_p.x, g_o["x"]) and np.array_equal(g_p.y, g_o["y"])
    # an independent statement: the table's own wavelengths, every value times the linearly interpolated acceptance
    w, v = CV.GetIceCubeFlasherSpectrumData(kind)
    acc = CV.GetIceCubeDOMAcceptance()
    grid = acc.startWlen + acc.wlenStep * np.arange(len(acc.values))
    assert np.allclose(g_p.y, v * np.interp(w, grid, acc.values), rtol=1e-12, atol=0)
    assert np.all(np.diff(w) > 0) and 3.0e-7 < w[0] < w[-1] < 6.1e-7


def test_an_equally_spaced_spectrum_keeps_its_binning():
    cfg = common.config("mie")
    spectrum = CV.I3CLSimFunctionFromTable(300e-9, 5e-9, np.linspace(1.0, 3.0, 41))
    g = CV.makeWavelengthGenerator(spectrum, CV.GetIceCubeDOMAcceptance(), cfg["med_p"])
    assert g.x is None and g.first == 300e-9 and g.spacing == 5e-9 and len(g.y) == 41
    g_o = B.make_wavelength_generator(dict(kind="table", start=300e-9, step=5e-9, values=np.linspace(1.0, 3.0, 41)), B.icecube_dom_acceptance(), cfg["med_o"])
    assert np.array_equal(g.y, g_o["y"])


def test_refusals():
    cfg = common.config("mie")
    acc = CV.GetIceCubeDOMAcceptance()
    # the bias has to cover the spectrum (ModuleHelper.cxx:111-114)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="larger or equal to the spectrum wavelength range"):
        CV.makeWavelengthGenerator(CV.I3CLSimFunctionFromTable(np.array([200e-9, 300e-9, 400e-9]), np.ones(3)), acc, cfg["med_p"])
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="same size"):
        CV.I3CLSimFunctionFromTable(np.array([300e-9, 400e-9]), np.ones(3))
    # a table with its own wavelengths has no device code (FromTable.cxx:169-170): not a wavelength bias
    conv = CV.I3CLSimStepToPhotonConverterHIP(0)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="equal spacing"):
        conv.SetWlenBias(CV.I3CLSimFunctionFromTable(np.array([300e-9, 400e-9, 500e-9]), np.ones(3)))
    # abscissae must ascend
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="ascending"):
        conv.SetWlenGenerators([CV.I3CLSimRandomValueInterpolatedDistribution(np.array([300e-9, 290e-9, 400e-9]), np.ones(3))])


@pytest.mark.parametrize("kind", LEDS)
def test_oracle_samples_follow_the_biased_spectrum(kind):
    """inverse-CDF sampling of the piecewise linear density (InterpolatedDistribution.cxx:236-336): 200 000 draws of the oracle's
    generator against the exact CDF (Kolmogorov distance)"""
    cfg = common.config("lea")
    bias = B.icecube_dom_acceptance()
    gen = B.make_wavelength_generator(B.flasher_spectrum(kind, CV.FLASHER_DATA), bias, cfg["med_o"])
    g = cfg["geom"]
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    T = capi.make_tables(cfg["med_o"], geo, [B.cherenkov_wlen_generator(bias, cfg["med_o"]), gen], bias)
    w = np.sort(capi.generate_wavelengths(T, 1, 200000, seed=11).astype(np.float64))
    x, y = np.asarray(gen["x"]), np.asarray(gen["y"])
    cdf_nodes = np.concatenate([[0.0], np.cumsum(np.diff(x) * (y[1:] + y[:-1]) / 2.0)])
    cdf_nodes /= cdf_nodes[-1]
    k = np.clip(np.searchsorted(x, w, side="right") - 1, 0, len(x) - 2)
    dx = w - x[k]
    slope = (y[k + 1] - y[k]) / (x[k + 1] - x[k])
    total = np.sum(np.diff(x) * (y[1:] + y[:-1]) / 2.0)
    cdf = cdf_nodes[k] + (y[k] * dx + 0.5 * slope * dx * dx) / total
    emp = (np.arange(len(w)) + 0.5) / len(w)
    assert np.abs(cdf - emp).max() < 1.63 / np.sqrt(len(w)) + 2e-6          # 1 % Kolmogorov bound + single-precision table rounding
    assert x[0] <= w[0] and w[-1] <= x[-1] * (1 + 1e-6)
