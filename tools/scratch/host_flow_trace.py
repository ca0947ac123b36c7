"""timeline of bench.py --workload benchmark-host: when each bunch leaves the feeder, enters the propagator, returns"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from clsim_amd import converter as CV, step_store as SS, synthetic as S
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_lea"))
bias = CV.GetIceCubeDOMAcceptance(efficiency=0.95)
gens = [CV.makeCherenkovWavelengthGenerator(bias, medium)]
geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
bunch = 1 << 20
events = 8
ppc = CV.I3CLSimLightSourceToStepConverterPPC()
ppc.SetWlenBias(bias); ppc.SetMediumProperties(medium); ppc.SetRandomSeed(12345); ppc.Initialize()
f = SS.I3CLSimLightSourceToStepConverterAsync()
f.SetMaxBunchSize(bunch); f.SetBunchSizeGranularity(512); f.SetLightSourceParameterization(ppc, seed=12345, device=0); f.Initialize()
conv = CV.initializeHIP(0, geom, medium, bias, gens, pancakeFactor=5.0, enableDoubleBuffering=True, approximateNumberOfWorkItems=bunch, seed=12345)
warm = S.cascade_steps(bunch, seed=1, photons_per_step=200)
conv.EnqueueSteps(warm, 0); conv.GetConversionResult()
ev = np.zeros(events, dtype=CV.PARTICLE_DTYPE)
ev["type"], ev["energy"], ev["dz"], ev["length"] = CV.ParticleType.EMinus, 40.0e3, -1.0, np.nan
ev["identifier"] = np.arange(events, dtype=np.uint32)
log = []
state = {"n": 0, "done": False}
t0 = time.perf_counter()
def feed():
    for i in range(len(ev)):
        f.EnqueueLightSource(ev[i])
    f.EnqueueBarrier()
    log.append(("fed", time.perf_counter() - t0))
def forward():
    while True:
        r = f.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=600000)
        steps, _, last = r
        ta = time.perf_counter() - t0
        if len(steps):
            state["n"] += 1
            conv.EnqueueSteps(steps, state["n"])
            log.append(("bunch %d: from feeder at %.1f ms, in the propagator's queue at %.1f ms (%d steps)" % (state["n"], 1e3 * ta, 1e3 * (time.perf_counter() - t0), len(steps)), ta))
        if last:
            state["done"] = True
            return
th1 = threading.Thread(target=feed); th2 = threading.Thread(target=forward)
before = conv.GetStatistics()
th1.start(); th2.start()
got = 0
while not state["done"] or got < state["n"]:
    if got < state["n"]:
        ident, ph, rel = conv.GetConversionResultInPlace(); n = len(ph); rel()
        got += 1
        log.append(("result %d at %.1f ms (%d photons)" % (ident, 1e3 * (time.perf_counter() - t0), n), time.perf_counter() - t0))
    else:
        time.sleep(0.0005)
th1.join(); th2.join()
wall = time.perf_counter() - t0
st = conv.GetStatistics()
dev = (st["TotalDeviceTime"] - before["TotalDeviceTime"]) * 1e-9
for l in sorted(log, key=lambda x: x[1]):
    print(l[0])
print("wall %.1f ms, device %.1f ms, utilisation %.3f, kernel calls %d" % (1e3 * wall, 1e3 * dev, dev / wall, st["NumKernelCalls"] - before["NumKernelCalls"]))
