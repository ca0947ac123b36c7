import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda", 0)
dbg = torch.zeros(4 * 8192, dtype=torch.int64, device=dev)
os.environ["CLSIMHIP_DBG_PTR"] = str(dbg.data_ptr())
from clsim_amd import converter as CV, synthetic as S, _lib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
bias = CV.GetIceCubeDOMAcceptance()
conv = CV.initializeHIP(0, CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry()), medium, bias, [CV.makeCherenkovWavelengthGenerator(bias, medium)], pancakeFactor=5.0, approximateNumberOfWorkItems=n)
steps = S.cascade_steps(n, seed=1000)
d_steps = torch.from_numpy(steps.view(np.uint8).reshape(n, 48)).to(dev)
d_ph = torch.empty((1 << 22, 80), dtype=torch.uint8, device=dev); d_c = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(2):
    dbg.zero_()
    conv.PropagateDevice(d_steps.data_ptr(), n, d_ph.data_ptr(), 1 << 22, d_c.data_ptr())
    torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(-1, 4)
d = d[d[:, 1] > 0]
t0, t1, trips, active = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
tick = 1e-8  # wall_clock64: 100 MHz
T = (t1.max() - t0.min()) * tick * 1e3
print("n", n, "waves", len(d), "kernel span %.1f ms" % T)
end = (t1 - t0.min()) * tick * 1e3
start = (t0 - t0.min()) * tick * 1e3
print("wave start ms: min %.2f max %.2f" % (start.min(), start.max()))
print("wave end ms percentiles 1/10/50/90/99/100:", np.percentile(end, [1, 10, 50, 90, 99, 100]).round(1))
print("trips per wave: mean %.0f min %d max %d ; active lanes per trip %.1f" % (trips.mean(), trips.min(), trips.max(), active.sum() / trips.sum()))
print("us per trip: mean %.2f" % (((t1 - t0) * tick * 1e6) / trips).mean())
