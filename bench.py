#!/usr/bin/env python3
"""Headline benchmark: propagated photons per second (BASELINE.json).

One "step" = one pass of the hot path (the HIP propagation kernel behind the C
ABI) over one bunch of synthetic I3CLSimSteps that is already resident in HBM.
N=1 workload = BASELINE.json configs[1]: 1M cascade-like steps x 200 photons,
SPICE-Mie layered ice with tilt, synthetic 86-string detector, DOM oversize 5.
N>1: one process per GPU; a pass is the rank's whole shard of the configuration
BASELINE.json names -- C4 = configs[3]: 100M steps / 8 = 12 500 000 steps per
GPU, run as 3 bunches of 4 166 912 (a converter holds at most 6 139 850 RNG
streams, OpenCL.cxx:250; the last bunch padded with no-op steps), weak scaling:
the per-GPU shard is the same at N = 2, 4, 8; `--workload c5`: 10^9 photons / N,
strong scaling.  Steps are independent units, no data-path collective; the
detected photons of every bunch are gathered on rank 0 inside the timed region
through the C ABI's RCCL gather (clsimhip_gather_hits: counts all-gather + one
point-to-point transfer per peer), between two launches on the launch stream
(`--gather-overlap`: on a second stream while the next kernel runs -- measured
slower, the persistent propagation grid leaves a concurrent kernel only slivers
of the chip).  torch.distributed only
carries the RCCL unique id, the barriers and the max-over-ranks time.  If the C
ABI's communicator cannot be used the torch.distributed twin runs instead, the
line says so and the exit code is 3.

Prints ONE JSON line on rank 0.  At N=1 the line also carries `host_path`: the
reference's own calling pattern (a producer thread calls EnqueueSteps, the
consumer GetConversionResult, double buffering on; benchmark.py:300-360) over 8
bunches, PCIe transfers and index->ID conversion included -- measured after
and outside the timed region of `value`; `host_path_copy`: the same with the
caller's own copy of every result, which is what the reference's
GetConversionResult() means (I3CLSimStepToPhotonConverter.h:178-189); and
`table_maker`: two passes of `--workload tab` (BASELINE configs[4]'s table
maker half, 262 144 steps x 200 photons) with its roofline -- memory-side
atomic requests against the rate the memory side delivers -- and an oracle
baseline.  `roofline` prices the propagation kernel's
algorithmic HBM bytes against the 8 TB/s peak (the kernel is VALU-bound, so the
fraction is tiny by construction -- see DESIGN.md); `cpu_baseline` times the CPU
restatement of the reference kernel (oracle/, all host cores) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bunch", type=int, default=1 << 20, help="I3CLSimSteps per pass and GPU")
    ap.add_argument("--photons-per-step", type=int, default=200)
    ap.add_argument("--ice", default="spice_mie", choices=["spice_mie", "spice_lea"])
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c5", "tab", "tab5", "benchmark", "benchmark-host"],
                    help="BASELINE.json configs: c2 (default, the headline) 1M cascade steps SPICE-Mie; c3 10M steps "
                         "SPICE-Lea; c5 flasher: 405 nm point source at a DOM, 400 photons per step, SPICE-Lea; tab: the "
                         "table-maker half of configs[4] (point cascade, default spherical axes, SPICE-Mie) -- prints its own line")
    ap.add_argument("--events-per-pass", type=int, default=2, help="--workload benchmark: 40 TeV electrons per pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-path", action="store_true",
                    help="time EnqueueSteps -> GetConversionResult instead (host buffers, PCIe transfers and the index->ID "
                         "conversion included, double buffering on); NOT the headline value, see DESIGN.md 6")
    ap.add_argument("--no-host-path", action="store_true",
                    help="skip the host_path object (profile passes: counters then cover the timed launches only)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-table-maker", action="store_true", help="skip the table_maker object of the default line")
    ap.add_argument("--tab-generic-sampler", action="store_true",
                    help="--workload tab: clsimhip_tabulator_set_tuning(\"standard_sampler\", 0) -- the generic path sampler instead of the one "
                         "specialised for the reference's default table shape (an A/B measurement)")
    ap.add_argument("--gather-overlap", action="store_true",
                    help="N>1: gather the photons of launch k on a second stream while launch k+1 runs (two photon buffers). Default: "
                         "the gather runs between two launches on the launch stream -- the propagation kernel is a persistent grid "
                         "that fills every CU, a copy or RCCL kernel beside it gets slivers of the chip and slows it down (DESIGN.md 7)")
    ap.add_argument("--copy-results", dest="consume_in_place", action="store_false",
                    help="benchmark-host: the consumer copies every result (80 B per detected photon) into a buffer of its own before it counts "
                         "them, like the C++ adapter's GetConversionResult() fills an I3CLSimPhotonSeries.  Default: the records are read in the "
                         "library's page-locked result buffer, which is then released (GetConversionResultInPlace in both adapters) -- the "
                         "reference's benchmark.py only counts the photons of a result")
    ap.add_argument("--consume-in-place", dest="consume_in_place", action="store_true", default=True, help="(the default)")
    ap.add_argument("--shard-steps", type=int, default=0,
                    help="I3CLSimSteps per GPU and pass, cut into equal bunches of at most 6 139 850 (the converter's stream limit, "
                         "OpenCL.cxx:250).  Default: one --bunch at N=1; at N>1 the per-GPU shard of the configuration BASELINE names: "
                         "c2 -> C4 = 100M steps / 8 = 12 500 000 per GPU (weak scaling), c5 -> 10^9 photons / N (strong scaling)")
    ap.add_argument("--keep-detected", action="store_true",
                    help="SetStopDetectedPhotons(false): the instantiations without STOP_PHOTONS_ON_DETECTION (every DOM on a photon's way "
                         "records it; pooled kernel for large bunches like the default mode); an extra measurement, never the headline")
    ap.set_defaults(c3_as_written=False)
    ap.add_argument("--verify-gather", dest="verify_gather", action="store_true", default=None,
                    help="after the timed region, one more launch whose gathered photons on rank 0 are compared with every rank's own "
                         "buffer (record counts and a 64-bit sum); config.gather_verified.  DEFAULT at N>1; at N=1 only with "
                         "CLSIMHIP_BENCH_GATHER=1 and this flag")
    ap.add_argument("--no-verify-gather", dest="verify_gather", action="store_false", help="N>1: skip the gather verification launch")
    return ap.parse_args()


def cpu_baseline(args, steps_np, seconds):
    """Times oracle/ (CPU restatement of the reference kernel, the stand-in for
    the reference's UseCPUs=True path) on a bounded sample of the same steps."""
    from oracle import builders as B
    from oracle import capi
    from clsim_amd import synthetic as S
    cores = os.cpu_count() or 1
    g = S.ic86_geometry()
    geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
    med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", args.ice))
    bias = B.icecube_dom_acceptance()
    gens = [B.cherenkov_wlen_generator(bias, med)]
    if args.workload == "c5":
        gens.append(dict(kind="const", value=405e-9))
    T = capi.make_tables(med, geo, gens, bias, pancake=5.0)
    from clsim_amd import converter as CV
    probe = 64 * cores
    # stream set-up through the product's generators (tests/test_golden_reference.py pins both
    # implementations on the reference's multiplier file); the propagation below is the oracle's
    a = CV.mwc_multipliers(min(len(steps_np), 1 << 19))
    x = CV.seed_streams(a)
    t0 = time.time()
    capi.propagate(T, steps_np[:probe], x, a, threads=cores)
    rate = steps_np["num"][:probe].sum() / max(time.time() - t0, 1e-6)
    n = int(min(len(a), max(probe, seconds * rate / args.photons_per_step)))
    n = (n // cores) * cores
    t0 = time.time()
    _, hits, _, _ = capi.propagate(T, steps_np[:n], x, a, threads=cores)
    dt = time.time() - t0
    photons = int(steps_np["num"][:n].sum())
    return {"value": photons / dt, "unit": "photons/s", "cores": cores, "kind": "port",
            "sample": "%d steps x %d photons of the same bunch (%d photons, %d hits) in %.1f s, %d threads" %
                      (n, args.photons_per_step, photons, hits, dt, cores)}


def tabulator_measure(args, device, workload, passes, warmup, cpu_seconds, photons_per_step=200, bunch=262144):
    """Table maker (python/tablemaker/tabulator.py defaults): cascade-like steps at the origin,
    spherical table 200 x 36 x 100 x 105 bins (+ under/overflow), SPICE-Mie, 42 absorption lengths per photon, one
    path sample per metre.  Not the headline metric: photons/s and path samples/s of the TABULATE kernel and the
    oracle's rate on the host cores (the reference runs this kernel as a single CPU work item).  Returns the record."""
    import math
    from clsim_amd import converter as CV
    from clsim_amd import synthetic as S
    from clsim_amd import tabulator as TB
    n = (bunch // 256) * 256                                     # 262 144: >= 4 workgroups per CU
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
    axis_list = [TB.PowerAxis(0, 580, 200, 2), TB.LinearAxis(0, 180, 36), TB.LinearAxis(-1, 1, 100), TB.PowerAxis(0, 7e3, 105, 2)]
    if workload == "tab5":             # fifth axis = cosine of the impact angle (TABULATE_IMPACT_ANGLE), coarser in azimuth
        axis_list = [TB.PowerAxis(0, 580, 200, 2), TB.LinearAxis(0, 180, 12), TB.LinearAxis(-1, 1, 50), TB.PowerAxis(0, 7e3, 105, 2),
                     TB.LinearAxis(-1, 1, 10)]
    axes = TB.SphericalAxes(axis_list)
    ang = [0.32813, 0.63899, 0.20049, -1.2250, -0.14470, 4.1695, 0.76898, -5.8690, -2.0939, 2.3834, 1.0435]
    a = CV.mwc_multipliers(n)
    x = CV.seed_streams(a)
    tab = TB.I3CLSimStepToTableConverterHIP(device, axes, False, medium, math.pi * 0.16510 ** 2, CV.GetIceCubeDOMAcceptance(),
                                            TB.I3CLSimFunctionPolynomial(ang), (x, a))
    if getattr(args, "tab_generic_sampler", False):
        tab.SetTuning("standard_sampler", 0)
    steps = S.cascade_steps(n, seed=1000, vertex=(0.0, 0.0, 0.0), photons_per_step=photons_per_step)
    ref = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)
    for _ in range(warmup):
        tab.EnqueueSteps(steps, ref)
    tab.Finish()
    before = float(tab.GetBinSums().sum()) if warmup else 0.0
    k0 = tab.GetStatistics()["KernelTimeMs"]
    t0 = time.perf_counter()
    for _ in range(passes):
        tab.EnqueueSteps(steps, ref)
    tab.Finish()
    elapsed = time.perf_counter() - t0
    st = tab.GetStatistics()
    kernel_ms = (st["KernelTimeMs"] - k0) / passes
    photons = int(steps["num"].sum())
    sums = tab.GetBinSums()
    occupied = int((sums > 0).sum())
    total = float(sums.sum())
    del sums, tab
    out = {"metric": "tabulated photons/sec (TABULATE kernel, 1 GPU)", "value": photons * passes / elapsed, "unit": "photons/s",
           "n_gpus": 1, "steps": passes, "warmup": warmup, "ms_per_step": 1e3 * elapsed / passes,
           "kernel_ms_per_pass": kernel_ms, "photons_per_pass": photons, "table_bins": int(st["NumBins"]), "occupied_bins": occupied,
           "sum_of_weights_per_pass": (total - before) / passes,
           "config": {"workload": "%d steps x %d photons at the origin, spherical axes %s, spice_mie, 42 absorption lengths, "
                                  "1 m sampling; BASELINE.json configs[4] (tablemaker half)"
                                  % (n, photons_per_step, "x".join(str(ax.n_bins) for ax in axis_list)),
                      "sampler": "generic (standard_sampler = 0)" if getattr(args, "tab_generic_sampler", False) else
                                 ("specialised for the default table shape" if workload == "tab" else "generic")}}
    # The table sums are fp64 atomic adds that execute at the memory side: the L2 of an XCD hands every one of them on (TCC_EA0_ATOMIC = TCC_ATOMIC), one
    # request per wave instruction and 64-byte sector.  The memory side delivers 2.06e10 such requests per second whatever the lanes per sector and the sectors
    # per instruction (tools/micro/atomic_rate.hip on this chip; MI355X_MICROARCH.md, Global float atomics: 1.3 TB/s of added bytes for contiguous 256-byte wave
    # instructions = 2e10 requests).  Requests per pass come from a stored rocprofv3 --pmc pass of this command (profiles/r06/tab_pmc.json: TCC_EA0_ATOMIC); the
    # kernel time is live.  ROUND 6 CORRECTION: rounds 4-5 derived the count as WRITE_SIZE / 64 -- WRITE_SIZE tallies 32 bytes per atomic request, the count was
    # 1.9 x too low and the kernel was said to run at 0.53 of the rate.  It runs AT the rate (1.0-1.1 of the micro-benchmark's): this is the table maker's bound.
    ppath = next((q for q in (os.path.join(ROOT, "profiles", r, "tab_pmc.json") for r in ("r06", "r05")) if os.path.exists(q)), None)
    if ppath and workload == "tab" and photons_per_step == 200 and n == 262144:
        with open(ppath) as f:
            prof = json.load(f)
        requests = prof.get("tcc_ea0_atomic_per_launch") or (prof["write_bytes_per_launch"] / 32.0)
        seconds = kernel_ms * 1e-3
        peak = 2.06e10
        out["roofline"] = {"bound": "hbm",
                           "what": "memory-side fp64 atomic requests (one per wave instruction and 64-byte sector of the table; every one leaves the L2: TCC_EA0_ATOMIC), "
                                   "priced against the request rate the memory side delivers (tools/micro/atomic_rate.hip); `frac` >= 1 means the kernel sits on that limit -- "
                                   "with the atomics compiled out a pass takes 0.82 of the time (profiles/r06/ab_tab_std_sampler_noatomic.txt)",
                           "achieved": requests / seconds / 1e9, "peak": peak / 1e9, "unit": "1e9 atomic requests/s", "frac": requests / seconds / peak,
                           "atomic_requests_per_s": requests / seconds, "scattered_atomic_peak_per_s": peak,
                           "frac_of_scattered_rate": requests / seconds / peak,
                           "peaks": "tools/micro/atomic_rate.hip on this chip: 2.06e10 fp64 sector requests/s for 8 ... 64 lanes into 8 ... 64 random sectors of a 670 MB table per "
                                    "instruction (profiles/r05/atomic_rate_microbench.txt; 1.95e10 by TCC_ATOMIC in profiles/r06/tab_atomics_counters.json); "
                                    "MI355X_MICROARCH.md, Global float atomics: 1.3 TB/s of added bytes for contiguous 256-byte wave instructions = 2e10 requests/s",
                           "traffic": prof.get("fabric_bytes_per_launch"), "atomic_requests_per_launch": requests,
                           "traffic_source": {"file": os.path.relpath(ppath, ROOT), "git_revision": prof.get("git_revision"), "kernel": prof.get("kernel"),
                                              "profiled_kernel_ms": prof.get("kernel_ms")},
                           "avg_kernel_ms": kernel_ms}
    if cpu_seconds > 0:
        # the oracle with the reference's host loop around the kernel (oracle_tabulate_accumulate: every step in calls of 16 photons, entries
        # added up and dropped) on steps of the SAME bunch at their real 200 photons, all host cores, for about `cpu_seconds`
        from oracle import builders as B
        from oracle import capi
        cores = os.cpu_count() or 1
        med = B.load_ppc_ice(os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_mie"))
        o_axes = [B.power_axis(0, 580, 200, 2), B.linear_axis(0, 180, 36), B.linear_axis(-1, 1, 100), B.power_axis(0, 7e3, 105, 2)]
        if workload == "tab5":
            o_axes = [B.power_axis(0, 580, 200, 2), B.linear_axis(0, 180, 12), B.linear_axis(-1, 1, 50), B.power_axis(0, 7e3, 105, 2), B.linear_axis(-1, 1, 10)]
        tb = B.tabulator_config("spherical", o_axes, med, ang, entries_per_stream=65536)
        bias = B.icecube_dom_acceptance()
        g = S.single_string_geometry()
        geo = B.build_geometry(g["string_ids"], g["dom_ids"], g["x"], g["y"], g["z"], g["subdetectors"], g["om_radius"])
        T = capi.make_tables(med, geo, [B.cherenkov_wlen_generator(bias, med)], bias, pancake=1.0, tabulator=tb)
        ref_o = B.reference_particle(ref[:3], ref[3], ref[4:])
        probe = min(n, 2 * cores)
        t1 = time.time()
        capi.tabulate_accumulate(T, steps[:probe], x[:probe], a[:probe], ref_o, photons_per_call=16, threads=cores)
        pace = max(time.time() - t1, 1e-3) / probe
        m = int(min(n, max(probe, (cpu_seconds / pace) // cores * cores)))
        t1 = time.time()
        _, counts, _ = capi.tabulate_accumulate(T, steps[:m], x[:m], a[:m], ref_o, photons_per_call=16, threads=cores)
        dt = time.time() - t1
        done, samples = int(steps["num"][:m].sum()), int(counts.sum())
        out["samples_per_photon"] = float(samples) / max(done, 1)
        out["path_samples_per_sec"] = out["samples_per_photon"] * out["value"]
        out["cpu_baseline"] = {"value": done / dt, "unit": "photons/s", "cores": cores, "kind": "port",
                               "sample": "the first %d steps x %d photons of the same bunch (%d photons, %d path samples) in %.1f s, %d threads; the "
                                         "reference runs this kernel as ONE work item (StepToTableConverter.cxx:259)"
                                         % (m, photons_per_step, done, samples, dt, cores)}
    return out


def tabulator_bench(args, torch, device):
    """`--workload tab | tab5`: the table maker's own line."""
    bunch = 262144 if args.bunch == (1 << 20) else min(args.bunch, 1 << 19)
    emit(json.dumps(tabulator_measure(args, device, args.workload, args.steps, args.warmup, 0.0 if args.no_cpu_baseline else 10.0,
                                      photons_per_step=args.photons_per_step, bunch=bunch)))


def host_path_run(CV, args, steps_np, conv, bunches, in_place=None):
    """The reference's own calling pattern (benchmark.py:300-360): a producer thread enqueues bunches, the consumer takes
    results; wall clock from the first enqueue to the last result -- host buffers, PCIe transfers, index->ID conversion."""
    import threading
    from clsim_amd.synthetic import PHOTON_DTYPE
    in_place = args.consume_in_place if in_place is None else in_place
    conv.EnqueueSteps(steps_np, 0)          # warm-up bunch
    conv.GetConversionResult()
    before = conv.GetStatistics()
    recycled = None if in_place else np.zeros(conv.GetMaxNumWorkitems() * 10, dtype=PHOTON_DTYPE)     # (the copying consumer's buffer, reused per bunch)
    t0 = time.perf_counter()
    producer = threading.Thread(target=lambda: [conv.EnqueueSteps(steps_np, i) for i in range(bunches)])
    producer.start()
    hits = 0
    for _ in range(bunches):
        if in_place:               # the records where the library left them (benchmark.py only counts a result's photons)
            _, ph, release = conv.GetConversionResultInPlace()
            hits += len(ph)
            release()
        else:
            _, ph = conv.GetConversionResult(out=recycled)
            hits += len(ph)
    producer.join()
    elapsed = time.perf_counter() - t0
    st = conv.GetStatistics()
    photons = int(steps_np["num"].sum()) * bunches
    device_ns = st["TotalDeviceTime"] - before["TotalDeviceTime"]
    return {"value": photons / elapsed, "unit": "photons/s", "bunches": bunches, "steps_per_bunch": len(steps_np), "hits": hits,
            "seconds": elapsed, "device_utilization": device_ns * 1e-9 / elapsed,
            "device_ns_per_photon": device_ns / photons, "double_buffering": True,
            "consumer": "reads the records in the library's page-locked result buffer, then releases it" if in_place else
                        "copies every result into a buffer of its own (--copy-results)",
            "definition": "sum(numPhotons) / wall clock from the first EnqueueSteps to the last GetConversionResult "
                          "(reference benchmark.py:335-340), outside the timed region of `value`"}


def benchmark_workload(args, torch, device):
    """The reference's own benchmark (resources/scripts/benchmark.py:142-147, 300-316): 40 TeV electrons at the origin pointing
    down, SPICE-Lea, DOM oversize 5, UnshadowedFraction 0.95.  Particle -> step requests on the host (PPC front end,
    csrc/lightsource.cpp), steps born in HBM (steps_kernel.hip) and propagated there; nothing visits the host.  One pass =
    `--events-per-pass` events (2 x 2.56M steps = 1.02e9 photons); the steps of every pass are generated inside the timed region."""
    from clsim_amd import converter as CV
    from clsim_amd import synthetic as S
    dev = torch.device("cuda", device)
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_lea"))
    bias = CV.GetIceCubeDOMAcceptance(efficiency=0.95)
    gens = [CV.makeCherenkovWavelengthGenerator(bias, medium)]
    geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
    ppc = CV.I3CLSimLightSourceToStepConverterPPC()
    ppc.SetWlenBias(bias); ppc.SetMediumProperties(medium); ppc.SetRandomSeed(12345); ppc.Initialize()
    passes = args.warmup + args.steps
    per_pass = max(1, args.events_per_pass)
    requests, counts, photons = [], [], []
    for k in range(passes):
        ev = np.zeros(per_pass, dtype=CV.PARTICLE_DTYPE)
        ev["type"] = CV.ParticleType.EMinus
        ev["energy"] = 40.0e3
        ev["dz"] = -1.0
        ev["length"] = np.nan
        ev["identifier"] = np.arange(k * per_pass, (k + 1) * per_pass, dtype=np.uint32)
        req = ppc.EnqueueLightSources(ev)
        requests.append(req)
        counts.append(CV.CountGeneratedSteps(req, granularity=512)[1])
        photons.append(int((req["num_steps"] * req["photons_per_step"] + req["num_photons_in_last_step"]).sum()))
    n_max = max(counts)
    conv = CV.initializeHIP(device, geom, medium, bias, gens, pancakeFactor=5.0, approximateNumberOfWorkItems=n_max, seed=12345)
    if conv.GetMaxNumWorkitems() < n_max:
        raise SystemExit("events-per-pass too large: %d steps for %d RNG streams" % (n_max, conv.GetMaxNumWorkitems()))
    capacity = 32 * 1024 * 1024 * per_pass        # a 40 TeV cascade next to the central string leaves ~2e7 detected photons
    d_steps = torch.empty((n_max, 48), dtype=torch.uint8, device=dev)
    d_photons = torch.empty((capacity, 80), dtype=torch.uint8, device=dev)
    d_count = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one_pass(k):
        n = CV.GenerateStepsDevice(requests[k], 777 + k, d_steps.data_ptr(), n_max, granularity=512, device=device, stream=stream)
        conv.PropagateDevice(d_steps.data_ptr(), n, d_photons.data_ptr(), capacity, d_count.data_ptr(), stream=stream)

    for k in range(args.warmup):
        one_pass(k)
    torch.cuda.synchronize()
    conv.KernelTimeMs(reset=True)
    t0 = time.perf_counter()
    for k in range(args.warmup, passes):
        one_pass(k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = conv.KernelTimeMs(reset=True)
    hits = int(d_count.cpu().item())
    total = sum(photons[args.warmup:])
    n_last = counts[-1]
    avg_ms = kernel_ms / max(launches, 1)
    alg_bytes = n_last * 72.0 + min(hits, capacity) * 80.0
    emit(json.dumps({
        "metric": "propagated photons/sec (whole node)", "value": total / elapsed, "unit": "photons/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "reference benchmark.py: %d x 40 TeV e- at the origin pointing down per pass, spice_lea + tilt, 86 strings, oversize 5" % per_pass,
                   "kind": "particles -> step requests (host) -> steps born in HBM -> propagated; step generation inside the timed region",
                   "photons_per_meter_of_track": ppc.MeanPhotonsPerMeter(0), "steps_last_pass": n_last, "photons_last_pass": photons[-1],
                   "hits_last_pass": min(hits, capacity), "hit_counter_last_pass": hits, "photon_buffer_records": capacity,
                   "shower_parameters": "restated from the published parameterisation (parity unpinned)"},
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "prop_pool_kernel" if conv.KernelForBunch(n_last) == "pool" else "prop_kernel", "avg_kernel_ms": avg_ms,
                     "launches": int(launches), "algorithmic_bytes_per_launch": alg_bytes}}))


def benchmark_host_workload(args, torch, device):
    """The reference's benchmark as its host code runs it (resources/scripts/benchmark.py:284-340; I3CLSimModule): 40 TeV
    electrons at the origin -> I3CLSimLightSourceToStepConverterAsync (the feeder: PPC front end + GPU step producer + step
    store, clsimhip_feeder_*) -> HOST step bunches -> EnqueueSteps -> GetConversionResult with double buffering.  Three threads
    like the module's: one feeds particles, one moves step bunches from the feeder to the propagator, the caller collects
    photons.  Reports the reference's two figures (benchmark.py:326-340: device time per photon, wall time per photon with
    the device utilisation), the feeder's own rate (the same events with nothing downstream) and which stage limits the chain."""
    import threading
    from clsim_amd import converter as CV
    from clsim_amd import step_store as SS
    from clsim_amd import synthetic as S
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", "spice_lea"))
    bias = CV.GetIceCubeDOMAcceptance(efficiency=0.95)
    gens = [CV.makeCherenkovWavelengthGenerator(bias, medium)]
    geom = CV.I3CLSimSimpleGeometry.from_dict(S.ic86_geometry())
    bunch = (args.bunch // 512) * 512                    # steps per bunch handed to the propagator (default 1 048 576)
    events = max(1, args.events_per_pass) * max(1, args.steps)

    def make_feeder(seed):
        ppc = CV.I3CLSimLightSourceToStepConverterPPC()
        ppc.SetWlenBias(bias); ppc.SetMediumProperties(medium); ppc.SetRandomSeed(seed); ppc.Initialize()
        f = SS.I3CLSimLightSourceToStepConverterAsync()
        f.SetMaxBunchSize(bunch); f.SetBunchSizeGranularity(512); f.SetLightSourceParameterization(ppc, seed=seed, device=device); f.Initialize()
        return f, ppc

    def particles(first, n):
        ev = np.zeros(n, dtype=CV.PARTICLE_DTYPE)
        ev["type"], ev["energy"], ev["dz"], ev["length"] = CV.ParticleType.EMinus, 40.0e3, -1.0, np.nan
        ev["identifier"] = np.arange(first, first + n, dtype=np.uint32)
        return ev

    def feed(feeder, ev):
        for i in range(len(ev)):
            feeder.EnqueueLightSource(ev[i])
        feeder.EnqueueBarrier()

    # ---- the feeder alone: particles -> host step bunches, nothing downstream ----
    feeder, ppc = make_feeder(12345)
    ev = particles(0, events)
    t0 = time.perf_counter()
    th = threading.Thread(target=feed, args=(feeder, ev))
    th.start()
    steps_alone = photons_alone = 0
    while True:
        r = feeder.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=600000)
        assert r is not None
        steps_alone += len(r[0])
        photons_alone += int(r[0]["num"].sum())
        if r[2]:
            break
    th.join()
    feeder_seconds = time.perf_counter() - t0
    del feeder

    # ---- the whole chain ----
    conv = CV.initializeHIP(device, geom, medium, bias, gens, pancakeFactor=5.0, enableDoubleBuffering=True,
                            approximateNumberOfWorkItems=bunch, seed=12345)
    warm = S.cascade_steps(bunch, seed=1, photons_per_step=200)
    conv.EnqueueSteps(warm, 0)
    conv.GetConversionResult()
    feeder, ppc = make_feeder(12345)
    before = conv.GetStatistics()
    state = {"bunches": 0, "steps": 0, "photons": 0, "wait_feeder": 0.0, "wait_enqueue": 0.0, "wait_result": 0.0}

    def forward():
        while True:
            t_a = time.perf_counter()
            r = feeder.GetConversionResultWithBarrierInfoAndMarkers(timeout_ms=600000)
            t_b = time.perf_counter()
            state["wait_feeder"] += t_b - t_a           # the propagator's input queue is not being filled meanwhile
            assert r is not None
            steps, _, last = r
            if len(steps):
                state["bunches"] += 1
                state["steps"] += len(steps)
                state["photons"] += int(steps["num"].sum())
                t_c = time.perf_counter()
                conv.EnqueueSteps(steps, state["bunches"])
                state["wait_enqueue"] += time.perf_counter() - t_c     # (blocks while the input queue is full: the propagator is busy)
            if last:
                return
    t0 = time.perf_counter()
    th_feed = threading.Thread(target=feed, args=(feeder, particles(0, events)))
    th_fwd = threading.Thread(target=forward)
    th_feed.start(); th_fwd.start()
    hits = got = 0
    recycled = np.zeros(conv.GetMaxNumWorkitems() * 10, dtype=S.PHOTON_DTYPE)      # the consumer's photon buffer, reused per bunch
    while th_fwd.is_alive() or got < state["bunches"]:
        if got < state["bunches"]:
            t_r = time.perf_counter()
            if args.consume_in_place:
                _, ph, release = conv.GetConversionResultInPlace()
                hits += len(ph)
                release()
            else:
                _, ph = conv.GetConversionResult(out=recycled)
                hits += len(ph)
            state["wait_result"] += time.perf_counter() - t_r
            got += 1
        else:
            time.sleep(0.0005)
    th_feed.join(); th_fwd.join()
    elapsed = time.perf_counter() - t0
    st = conv.GetStatistics()
    device_ns = st["TotalDeviceTime"] - before["TotalDeviceTime"]
    photons = state["photons"]
    assert photons == photons_alone and state["steps"] == steps_alone
    value = photons / elapsed
    feeder_rate = photons_alone / feeder_seconds
    device_rate = photons / (device_ns * 1e-9)
    stages = {"feeder (particles -> host step bunches)": feeder_rate, "propagator device time": device_rate}
    emit(json.dumps({
        "metric": "propagated photons/sec through the reference's host flow (light sources -> feeder -> EnqueueSteps -> GetConversionResult)",
        "value": value, "unit": "photons/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "reference benchmark.py through host buffers: %d x 40 TeV e- at the origin pointing down, spice_lea + tilt, "
                               "86 strings, oversize 5, step bunches of %d, double buffering" % (events, bunch),
                   "events": events, "steps": state["steps"], "bunches": state["bunches"], "photons": photons, "hits": hits,
                   "consumer": "reads the records in the library's result buffer, then releases it" if args.consume_in_place else
                               "copies every result (80 B per detected photon) into a buffer of its own, like the C++ adapter fills an I3CLSimPhotonSeries",
                   "shower_parameters": "restated from the published parameterisation (parity unpinned)"},
        "reference_figures": {"AverageDeviceTimePerPhoton_ns": device_ns / photons, "AverageHostTimePerPhoton_ns": 1e9 * elapsed / photons,
                              "DeviceUtilization": device_ns * 1e-9 / elapsed,
                              "definition": "resources/scripts/benchmark.py:326-340"},
        "feeder": {"photons_per_s": feeder_rate, "steps_per_s": steps_alone / feeder_seconds, "seconds": feeder_seconds,
                   "note": "the same events with nothing downstream: PPC front end, GPU step producer, download, step store, bunching"},
        "threads": {"forwarding thread: seconds waiting for the feeder": state["wait_feeder"],
                    "forwarding thread: seconds inside EnqueueSteps (copy + waiting for room in the input queue)": state["wait_enqueue"],
                    "consumer: seconds inside GetConversionResult (waiting + taking the photons)": state["wait_result"],
                    "wall clock": elapsed},
        "limiting_stage": min(stages, key=stages.get), "stage_rates_photons_per_s": stages}))


VALU_PEAK_LANE_OPS = 256 * 4 * 2.4e9 * 32       # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, a wave64 vector operation holds a SIMD for 2 cycles
                                                # at 2.4 GHz: 7.86e13 single-precision lane operations per second (157 TFLOP/s counting an fma as two)


def valu_roofline(workload, photons_per_s, pmc=None, photons_per_launch=None, kernel_ms=None):
    """The operative roofline (the kernel is vector-issue bound, not HBM bound).  USEFUL work: the reference's arithmetic per photon,
    counted by the instrumented oracle on a sample of this very bunch and priced in gfx950 vector instructions
    (tools/count_reference_ops.py -> profiles/r06/reference_ops.json: `as_written` = every operation of the reference's
    expressions at the device's generic sequences, search arithmetic on every trip included; `transformed` = the same photon
    histories after the bit-preserving transformations of DESIGN.md section 2, every division and root at the cheapest form proven
    exact for its site, no search arithmetic: the floor) x the measured photons per second, against the chip's vector peak.
    ISSUED work (only where a rocprofv3 --pmc pass of this workload is stored): SQ_INSTS_VALU per launch -> issue slots and
    lanes, whose ratio to the floor is the overhead (filter, searches, scheduling, creation bookkeeping, idle lanes)."""
    path = os.path.join(ROOT, "profiles", "r06", "reference_ops.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        stored = json.load(f)
    w = stored["workloads"].get(workload)
    if w is None:
        return None
    floor = w["valu_per_photon"]["transformed"]
    slots = w["issue_slots_per_photon"]["transformed"]      # the floor's instructions, a quarter-rate one (v_rcp_f32, v_sqrt_f32, 32-bit integer multiply) as four
    out = {"reference_ops_per_photon": {"as_written": w["valu_per_photon"]["as_written"],
                                        "as_written_without_search": w["valu_per_photon"]["as_written_without_search"],
                                        "transformed": floor, "unit": "gfx950 vector instructions per lane",
                                        "transformed_issue_slots": slots,
                                        "source": "profiles/r06/reference_ops.json (oracle %s, %s)" % (stored["oracle_sha16"], w["sample"])},
           "trips_per_photon": w["events_per_photon"]["trips"],
           "useful_lane_slots_per_s": slots * photons_per_s, "peak_lane_slots_per_s": VALU_PEAK_LANE_OPS,
           "useful_frac": slots * photons_per_s / VALU_PEAK_LANE_OPS,
           "useful_frac_counting_instructions": floor * photons_per_s / VALU_PEAK_LANE_OPS}
    if pmc and photons_per_launch and kernel_ms:
        insts, lanes = pmc["sq_insts_valu_per_launch"], pmc.get("valu_lane_utilisation")
        out.update({"insts_per_launch": insts, "lane_utilisation": lanes,
                    "issue_slot_frac": insts * 2.0 / (1024 * 2.4e9 * kernel_ms * 1e-3),
                    "issued_lane_slots_per_photon": insts * 64.0 / photons_per_launch,
                    "issued_live_lane_ops_per_photon": (insts * 64.0 * lanes / photons_per_launch) if lanes else None,
                    "overhead_ratio": (insts * 64.0 * lanes / photons_per_launch / floor) if lanes else None,
                    "pmc_source": "rocprofv3 --pmc pass of this command (profiles/latest_traffic.json) x the live kernel time"})
        # `issue_slot_frac` prices every vector instruction at the two cycles the guide's fp32 peak assumes.  What instructions cost THIS kernel was
        # measured by adding 48 of a kind to its loop (profiles/r06/issue_cost_by_kind.txt): double-rate opcodes 2.1-2.6 cycles, the same with a
        # scalar-register source 2.8, single-rate ones (compares, conversions, min / max, selects, shift-adds) 3.0-3.2, v_rcp_f32 / v_sqrt_f32 10.2.
        # With the shipped C2 kernel's mix (profiles/r06/c2_valu_prices.json) a SIMD is busy issuing vector instructions for:
        prices = os.path.join(ROOT, "profiles", "r06", "c2_valu_prices.json")
        if workload == "c2" and os.path.exists(prices):
            with open(prices) as f:
                pr = json.load(f)
            out["issue_busy_at_measured_prices"] = insts * pr["mean_cycles_per_instruction_in_kernel"] / (1024 * pr["shader_clock_ghz"] * 1e9 * kernel_ms * 1e-3)
            out["issue_busy_note"] = ("SQ_INSTS_VALU x %.2f cycles (the kernel's instruction mix at marginal prices measured in this kernel) / (1024 SIMDs x %.2f GHz x kernel time); "
                                      ">= 1: the vector pipe is full, marginal prices overstate averages by what exceeds 1 (%s)"
                                      % (pr["mean_cycles_per_instruction_in_kernel"], pr["shader_clock_ghz"], "profiles/r06/issue_cost_by_kind.txt"))
    return out


WORKLOAD_NAMES = {"c2": "C2 = BASELINE configs[1]", "c3": "C3 (the ice and detector of BASELINE configs[2])", "c5": "C5 = BASELINE configs[4] (flasher half)"}


def check_world(everyone, world, library_communicator):
    """Did `world` ranks on `world` DIFFERENT devices take part?  `everyone`: one record per rank as the ranks reported themselves
    (pci_bus_id: hipDeviceGetPCIBusId of the device the rank's communicator lives on; rccl_ranks / rccl_rank: ncclCommCount /
    ncclCommUserRank of that communicator).  Returns (ok, reason).  Pure: tests/test_bench_multi_rank_rehearsal.py calls it on the CPU."""
    if len(everyone) != world:
        return False, "%d ranks reported, %d expected" % (len(everyone), world)
    buses = [e.get("pci_bus_id") for e in everyone]
    if any(not b for b in buses):
        return False, "a rank could not name its device: %s" % buses
    if len(set(buses)) != world:
        return False, "the ranks sit on %d distinct devices %s" % (len(set(buses)), buses)
    if sorted(e.get("rank") for e in everyone) != list(range(world)):
        return False, "the ranks are not 0..%d: %s" % (world - 1, [e.get("rank") for e in everyone])
    if library_communicator:
        counted = [e.get("rccl_ranks") for e in everyone]
        if any(c != world for c in counted):
            return False, "the RCCL communicators count %s ranks" % counted
        if sorted(e.get("rccl_rank") for e in everyone) != list(range(world)):
            return False, "the RCCL communicators' ranks are %s" % [e.get("rccl_rank") for e in everyone]
        if any(e.get("rccl_rank") != e.get("rank") for e in everyone):
            return False, "a communicator's rank differs from its process's"
    return True, ""


def emit(line):
    """The one JSON line goes to the process's ORIGINAL stdout (see main)."""
    os.write(JSON_FD, (line + "\n").encode())


JSON_FD = 1
EXIT_PORT_TAKEN = 98            # a self-launched rank found MASTER_PORT in use (launch_ranks starts the world again)
EXIT_NOT_N_DEVICES = 4          # an N>1 run whose ranks do not sit on N distinct devices (or whose communicator disagrees)


def launch_ranks(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK in the environment): this process becomes the launcher.
    It starts N fresh children of the same command line, one rank per GPU, with the rendezvous variables torchrun would set
    (127.0.0.1, a free port), waits for all of them and exits with the largest return code.  The parent never imports torch
    and never touches the GPU; nothing is re-exec'ed.  Rank 0's JSON line goes to the stdout the children inherit.  If a rank
    dies the others (which would wait for it in a collective) are ended by their own process IDs."""
    import socket
    import subprocess
    sys.stdout.flush()
    worst = 0
    for attempt in range(4):
        # a free port now is not a free port when rank 0 binds it: a child that finds it taken exits with EXIT_PORT_TAKEN
        # and the whole world is started again on another port
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        children = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CLSIMHIP_BENCH_SELF_LAUNCHED="1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
            children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        failed_at = None
        while any(c.poll() is None for c in children):
            for c in children:
                rc = c.poll()
                if rc is not None and rc not in (0, 3) and failed_at is None:
                    failed_at = time.time()             # (3 = the line was printed for the torch.distributed fallback)
                    if rc == EXIT_PORT_TAKEN:
                        failed_at -= 18.0               # nothing to wait for: the others are waiting for a store that is not ours
            if failed_at is not None and time.time() - failed_at > 20.0:
                for c in children:
                    if c.poll() is None:
                        c.terminate()
                for c in children:
                    try:
                        c.wait(timeout=10)
                    except subprocess.TimeoutExpired:
                        c.kill()
                break
            time.sleep(0.2)
        codes = []
        for c in children:
            rc = c.wait()
            codes.append(rc if rc >= 0 else 128 - rc)
        if EXIT_PORT_TAKEN in codes and attempt < 3:
            sys.stderr.write("bench.py: port %d was taken before rank 0 could bind it; starting the ranks again on another port\n" % port)
            continue
        for r, rc in enumerate(codes):
            if rc:
                sys.stderr.write("bench.py: rank %d exited with %d\n" % (r, rc))
        worst = max(codes)
        break
    sys.exit(worst)


def main():
    global JSON_FD
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    env_world = os.environ.get("WORLD_SIZE")
    if "RANK" not in os.environ:
        if env_world is not None and int(env_world) != args.gpus:
            sys.stderr.write("bench.py: WORLD_SIZE=%s but --gpus %d (and no RANK): refusing to guess\n" % (env_world, args.gpus))
            sys.exit(2)
        if args.gpus > 1:
            return launch_ranks(args)                   # does not return
    elif int(env_world or "1") != args.gpus:
        # a launcher started W ranks of a command that asks for N GPUs: a line printed from here would carry the wrong n_gpus
        sys.stderr.write("bench.py: started as rank %s of WORLD_SIZE=%s but --gpus %d: the two must agree\n"
                         % (os.environ["RANK"], env_world, args.gpus))
        sys.exit(2)
    # Native libraries print banners to the C stdout (RCCL: version / library path at communicator creation, flushed at
    # exit): file descriptor 1 becomes stderr for everything but the result line, which is written to a duplicate of
    # the original stdout.
    sys.stdout.flush()
    JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if not args.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # the checker's library is built (a `make` child process) BEFORE anything initialises the GPU: a GPU-initialised
        # process must not spawn build tools under a profiler's preload
        from oracle import capi
        capi.build()
    if args.workload == "c3":
        # BASELINE configs[2] as written: 10 000 000 steps -- two bunches of 5 000 192 per pass, because a converter holds at most
        # 6 139 850 streams (all 32-bit safeprime multipliers, OpenCL.cxx:250); `--bunch N`: one bunch of N instead
        args.ice = "spice_lea"
        if args.bunch == (1 << 20) and args.shard_steps == 0:
            args.shard_steps, args.c3_as_written = 10000000, True
    elif args.workload == "c5":
        args.ice, args.photons_per_step = "spice_lea", (400 if args.photons_per_step == 200 else args.photons_per_step)
    import torch
    import torch.distributed as dist
    from clsim_amd import converter as CV
    from clsim_amd import synthetic as S
    from clsim_amd.distributed import HitGatherer, gather_hits as dist_gather_hits

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CLSIMHIP_BENCH_ECHO_RANK") == "1":
        sys.stderr.write("bench.py: rank %d of %d (local rank %d)\n" % (rank, world, local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the propagator has no CPU path")
    # CLSIMHIP_BENCH_REHEARSAL=1: a FUNCTIONAL rehearsal of the --gpus N path on a box with one GPU -- every rank uses
    # device 0, torch.distributed runs over gloo (control messages on the CPU) and CLSIMHIP_RCCL_LIBRARY is expected to
    # name tests/libfake_rccl.so in its process mode (FAKE_RCCL_DIR).  The line is flagged; its rate measures nothing.
    rehearsal = os.environ.get("CLSIMHIP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    # A launcher that gives every rank a device mask of its own (HIP_VISIBLE_DEVICES=r: one visible device, ordinal 0) is as good as
    # one that shows every rank all of them: the rank takes what it sees.  Fewer visible devices than ranks any other way maps several
    # ranks onto one GPU -- which check_world() below then refuses to report as an N-GPU run.
    visible = torch.cuda.device_count()
    if visible >= 1 and local_rank >= visible:
        local_rank = 0 if visible == 1 else local_rank % visible
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctl = torch.device("cpu") if rehearsal else dev     # where the control tensors of torch.distributed live
    if world > 1:
        try:
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        except Exception as exc:
            if os.environ.get("CLSIMHIP_BENCH_SELF_LAUNCHED") == "1" and ("EADDRINUSE" in str(exc) or "address already in use" in str(exc).lower()):
                sys.stderr.write("bench.py: rank %d: %s\n" % (rank, str(exc)[:200]))
                sys.exit(EXIT_PORT_TAKEN)
            raise

    if args.workload in ("tab", "tab5"):
        return tabulator_bench(args, torch, local_rank)
    if args.workload == "benchmark":
        return benchmark_workload(args, torch, local_rank)
    if args.workload == "benchmark-host":
        return benchmark_host_workload(args, torch, local_rank)
    # ---- the rank's share of one pass: `shard` real steps in `n_bunches` equal bunches of n steps (a multiple of 512; the
    # last bunch is padded with no-op steps like the reference pads a bunch before a barrier, Async.cxx:240-257) ----
    STREAM_LIMIT = 6139850                              # 32-bit safeprime multipliers available (OpenCL.cxx:250)
    scaling = "weak"
    shard = args.shard_steps
    if shard <= 0 and world > 1 and args.workload == "c2":
        shard = 100000000 // 8                          # C4 = BASELINE configs[3]: 100M steps over 8 GPUs
    if shard <= 0 and world > 1 and args.workload == "c5":
        shard = -(-(10 ** 9 // args.photons_per_step) // world)     # C5 = BASELINE configs[4]: 10^9 photons over the GPUs
        scaling = "strong"
    if shard > 0:
        n_bunches = -(-shard // ((STREAM_LIMIT // 512) * 512))
        n = -(-(-(-shard // n_bunches)) // 512) * 512
    else:
        n_bunches, n = 1, (args.bunch // 512) * 512
        shard = n
    # ---- configuration (same sequence as I3CLSimModuleHelper::initializeOpenCL) ----
    medium = CV.MakeIceCubeMediumProperties(iceDataDirectory=os.path.join(ROOT, "clsim_amd", "data", "ice", args.ice))
    bias = CV.GetIceCubeDOMAcceptance()
    gens = [CV.makeCherenkovWavelengthGenerator(bias, medium)]
    g86 = S.ic86_geometry()
    geom = CV.I3CLSimSimpleGeometry.from_dict(g86)
    if args.workload == "c5":
        gens.append(CV.I3CLSimRandomValueConstant(405e-9))          # delta-peak spectrum (ModuleHelper.cxx:81-88)
    conv = CV.initializeHIP(local_rank, geom, medium, bias, gens, pancakeFactor=5.0, enableDoubleBuffering=args.host_path,
                            stopDetectedPhotons=not args.keep_detected, approximateNumberOfWorkItems=n, seed=12345 + rank)
    def make_bunch(b):
        real = min(n, shard - b * n)                    # the last bunch of a shard carries the padding
        seed = 1000 + rank + 7919 * b
        if args.workload == "c5":
            k = int(np.argmin(np.abs(g86["x"]) + np.abs(g86["y"]) + np.abs(g86["z"] + 100.0)))   # a DOM near the detector centre
            return S.flasher_steps(real, seed=seed, photons_per_step=args.photons_per_step, pad_to=n,
                                   position=(float(g86["x"][k]), float(g86["y"][k]), float(g86["z"][k])))
        return S.cascade_steps(real, seed=seed, photons_per_step=args.photons_per_step, pad_to=n)
    steps_np = make_bunch(0)
    assert len(steps_np) == n

    if args.host_path:
        hp = host_path_run(CV, args, steps_np, conv, max(args.steps, 1))
        emit(json.dumps({"metric": "propagated photons/sec through EnqueueSteps/GetConversionResult (host buffers)",
                          "value": hp["value"], "unit": "photons/s", "n_gpus": 1, "steps": hp["bunches"], "warmup": 1,
                          "ms_per_step": 1e3 * hp["seconds"] / hp["bunches"], "host_path": hp,
                          "config": {"workload": WORKLOAD_NAMES[args.workload], "steps_per_bunch": n}}))
        return

    d_steps = [torch.from_numpy(steps_np.view(np.uint8).reshape(n, 48)).to(dev)]
    photons_per_pass = int(steps_np["num"].sum())
    for b in range(1, n_bunches):
        more = make_bunch(b)
        photons_per_pass += int(more["num"].sum())
        d_steps.append(torch.from_numpy(more.view(np.uint8).reshape(n, 48)).to(dev))
        del more
    capacity = (4 if args.workload == "c2" else 48) * 1024 * 1024
    # two photon buffers: with several GPUs the gather of pass k (its own stream) runs while the kernel of pass k+1 does
    # (CLSIMHIP_BENCH_GATHER=1 runs the gather path in a world of one rank: the multi-GPU code path on a single GPU)
    use_gather = (world > 1) or (os.environ.get("CLSIMHIP_BENCH_GATHER") == "1")
    overlap = use_gather and args.gather_overlap
    n_buffers = 2 if overlap else 1
    d_photons = [torch.empty((capacity, 80), dtype=torch.uint8, device=dev) for _ in range(n_buffers)]
    d_count = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(n_buffers)]
    compute = torch.cuda.current_stream()
    stream = compute.cuda_stream
    gatherer = gathered = comm_stream = None
    if use_gather:
        # detected photons of all ranks on rank 0 through the C ABI (RCCL called from the library; torch.distributed
        # only distributes the unique id): counts all-gather, then one point-to-point transfer per peer
        gather_note = None
        try:
            HitGatherer.unique_id()                     # loads RCCL: a local call, before anything collective
        except Exception as exc:
            gather_note = "%s: %s" % (type(exc).__name__, str(exc)[:120])
        if world > 1:
            ok = torch.tensor([0 if gather_note else 1], dtype=torch.int32, device=ctl)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                gather_note = gather_note or "RCCL could not be loaded on another rank"
        if gather_note is None:
            try:
                gatherer = HitGatherer.from_process_group(local_rank) if world > 1 else HitGatherer(local_rank, 0, 1, HitGatherer.unique_id())
            except Exception as exc:                    # (communicator refused ...)
                gatherer, gather_note = None, "%s: %s" % (type(exc).__name__, str(exc)[:120])
        if world > 1:
            # every rank takes the same path: if the library's communicator failed anywhere, all fall back to the
            # same gather through torch.distributed (clsim_amd/distributed.py: gather_hits), and the line says so
            ok = torch.tensor([1 if gatherer is not None else 0], dtype=torch.int32, device=ctl)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if gatherer is not None:
                    gatherer.close()
                gatherer = None
                gather_note = gather_note or "the communicator failed on another rank"
        elif gatherer is None:
            raise RuntimeError(gather_note)
        # room on the root for every rank's photons of one bunch: a rank stores at most `capacity` records, and fills a
        # few per cent of that (C2: 0.2 hits per step) -- sized for four times the expected number, at least one
        # rank's capacity; clsimhip_gather_hits reports on every rank if it ever were too small
        expected = {"c2": 0.2, "c3": 0.2, "c5": 7.0}[args.workload] * n
        gathered_rows = int(min(world * capacity, max(capacity, 4 * world * expected)))
        gathered = torch.empty(((gathered_rows if rank == 0 else 1), 80), dtype=torch.uint8, device=dev)
        comm_stream = torch.cuda.Stream(device=dev) if overlap else compute
    kernel_done = [torch.cuda.Event() for _ in range(n_buffers)]
    gather_done = [torch.cuda.Event() for _ in range(n_buffers)]
    state = {"pending": None, "hits": 0, "pass": 0, "overflow": 0, "per_rank_hits": np.zeros(world, dtype=np.int64)}

    def do_gather(b):
        comm_stream.wait_event(kernel_done[b])
        if gatherer is not None:
            counts = gatherer.gather(d_photons[b].data_ptr(), d_count[b].data_ptr(), capacity, 0, gathered.data_ptr(), gathered.shape[0],
                                     comm_stream.cuda_stream)
        else:                                           # fallback: torch.distributed (synchronous on the comm stream)
            with torch.cuda.stream(comm_stream):
                raw = int(d_count[b].item())
                _, c = dist_gather_hits(d_photons[b], raw, dst=0, out=(gathered if rank == 0 else None))
                counts = c.numpy().astype(np.uint64)
                if raw > capacity:
                    counts[rank] = raw
        gather_done[b].record(comm_stream)
        state["hits"] += int(np.minimum(counts, capacity).sum())
        state["per_rank_hits"] += np.minimum(counts, capacity).astype(np.int64)
        state["overflow"] += int((counts > capacity).sum())

    def one_launch(bunch):
        b = state["pass"] % n_buffers
        state["pass"] += 1
        if use_gather:
            compute.wait_event(gather_done[b])          # buffer b is free once its previous gather has left it
        conv.PropagateDevice(d_steps[bunch].data_ptr(), n, d_photons[b].data_ptr(), capacity, d_count[b].data_ptr(), stream=stream)
        if use_gather:
            kernel_done[b].record(compute)
            if not overlap:
                do_gather(b)                            # in stream order, before the next launch
            else:
                if state["pending"] is not None:
                    do_gather(state["pending"])         # the previous launch's photons travel while this kernel runs
                state["pending"] = b

    def one_pass():                                     # the rank's whole shard: one launch per bunch
        for bunch in range(n_bunches):
            one_launch(bunch)

    def flush():
        if use_gather and state["pending"] is not None:
            do_gather(state["pending"])
            state["pending"] = None

    def barrier():
        # the library's communicator and torch's are two RCCL communicators on one device: torch's barrier starts only
        # once everything queued on this device (the last gather's transfers included) has finished
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    flush()
    barrier()
    conv.KernelTimeMs(reset=True)
    if gatherer is not None:
        gatherer.statistics(reset=True)
    state["hits"] = 0
    state["per_rank_hits"][:] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    flush()
    barrier()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    kernel_ms, launches = conv.KernelTimeMs(reset=True)
    gather_stats = gatherer.statistics() if gatherer is not None else None
    counted_last = int(d_count[(state["pass"] - 1) % n_buffers].cpu().item())
    hits_last = min(counted_last, capacity)             # the counter keeps counting past the buffer; `capacity` records are stored

    t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- who took part: every rank reports the device it ran on (PCI bus id as the LIBRARY's communicator
    # sees it), what its RCCL communicator says about the world (ncclCommCount / ncclCommUserRank), its own kernel and
    # gather times and its hits; all-gathered over the control group.  Outside a rehearsal an N>1 run whose ranks do not
    # sit on N distinct devices, or whose communicator does not count N ranks, prints no line and exits non-zero.
    identity = None
    if use_gather:
        info = gatherer.info() if gatherer is not None else {"rccl_ranks": None, "rccl_rank": None, "device": local_rank, "pci_bus_id": None}
        if not info.get("pci_bus_id"):
            props = torch.cuda.get_device_properties(dev)
            if hasattr(props, "pci_bus_id"):
                info["pci_bus_id"] = "%04x:%02x:%02x.0" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
        mine = dict(info, rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0")), pid=os.getpid(),
                    visible_devices=os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")),
                    devices_visible_to_rank=torch.cuda.device_count(), device_name=torch.cuda.get_device_name(dev),
                    kernel_ms=kernel_ms, launches=int(launches), seconds=own_elapsed,
                    gather_ms=(gather_stats["gather_ms"] if gather_stats else None),
                    gathers=(gather_stats["gathers"] if gather_stats else None),
                    records_sent=(gather_stats["records_sent"] if gather_stats else None),
                    records_received=(gather_stats["records_received"] if gather_stats else None))
        everyone = [None] * world
        if world > 1:
            dist.all_gather_object(everyone, mine)
        else:
            everyone = [mine]
        buses = [e["pci_bus_id"] for e in everyone]
        counted = [e["rccl_ranks"] for e in everyone]
        ok_world, why = check_world(everyone, world, gatherer is not None)
        if world > 1 and not rehearsal and not ok_world:
            if rank == 0:
                sys.stderr.write("bench.py: --gpus %d: %s: not an N-GPU run, no line printed\n" % (world, why))
            if gatherer is not None:
                gatherer.close()
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(EXIT_NOT_N_DEVICES)

        def spread(key):
            v = [e[key] for e in everyone]
            if any(x is None for x in v):
                return None
            return {"min": min(v), "max": max(v), "argmax_rank": int(np.argmax(v)), "mean": float(np.mean(v))}
        identity = {"rccl_ranks": (counted[0] if gatherer is not None else None),
                    "rccl_ranks_source": "ncclCommCount of the library's communicator on every rank (all equal)" if gatherer is not None else
                                         "none: the torch.distributed fallback ran",
                    "control_group": {"backend": (dist.get_backend() if world > 1 else None), "world_size": (dist.get_world_size() if world > 1 else 1)},
                    "rank_devices": buses, "distinct_devices": len(set(buses)),
                    "n_ranks_on_n_devices": ok_world, "n_ranks_on_n_devices_why_not": (why or None),
                    "rank_device_names": sorted(set(e["device_name"] for e in everyone)),
                    "per_rank": {"kernel_ms": spread("kernel_ms"), "gather_ms": spread("gather_ms"), "seconds": spread("seconds"),
                                 "launches": spread("launches"), "gathers": spread("gathers"),
                                 "hits_stored_timed_region": [int(v) for v in state["per_rank_hits"]],
                                 "records_sent": [e["records_sent"] for e in everyone],
                                 "records_received_by_root": everyone[0]["records_received"],
                                 "pids": [e["pid"] for e in everyone], "visible_devices": [e["visible_devices"] for e in everyone],
                                 "devices_visible_to_rank": [e["devices_visible_to_rank"] for e in everyone]}}

    verified = None
    verify = use_gather and (args.verify_gather if args.verify_gather is not None else world > 1)
    if verify:
        # outside the timed region: one more launch of the last bunch, gathered, and rank 0's buffer compared with what
        # each rank holds -- a 64-bit sum over every rank's records and the exact record counts
        hits_timed, state["hits"] = state["hits"], 0
        one_launch(n_bunches - 1)
        flush()
        barrier()
        b = (state["pass"] - 1) % n_buffers
        mine = min(int(d_count[b].cpu().item()), capacity)
        sums = torch.zeros(world, 2, dtype=torch.int64)
        sums[rank, 0] = mine
        sums[rank, 1] = int(d_photons[b][:mine].view(torch.int64).sum().item()) if mine else 0
        if world > 1:
            box = sums.to(ctl)
            dist.all_reduce(box, op=dist.ReduceOp.SUM)
            sums = box.cpu()
        if rank == 0:
            verified, at = True, 0
            for r in range(world):
                k = int(sums[r, 0])
                seen = int(gathered[at:at + k].view(torch.int64).sum().item()) if k else 0
                verified = verified and (seen == int(sums[r, 1]))
                at += k
            verified = bool(verified and at == state["hits"])
        state["hits"] = hits_timed

    if args.workload == "c2" and shard == 100000000 // 8:
        workload_name = "C4 = BASELINE configs[3]: 100M steps / 8 GPUs = 12 500 000 steps per GPU%s, hits gathered on rank 0" % (
            "" if world == 8 else " (this run: %d GPU%s, the same per-GPU shard)" % (world, "" if world == 1 else "s"))
    elif args.workload == "c5" and world > 1:
        workload_name = "C5 = BASELINE configs[4] (flasher half): 10^9 photons / %d GPUs, hits gathered on rank 0" % world
    elif args.workload == "c3" and shard == 10000000 and world == 1:
        workload_name = "C3 = BASELINE configs[2]: 10 000 000 steps"
    else:
        workload_name = WORKLOAD_NAMES[args.workload]
    if rank == 0:
        value = photons_per_pass * args.steps * world / elapsed
        avg_ms = kernel_ms / max(launches, 1)
        pooled = conv.KernelForBunch(n) == "pool"
        kernel_name = "prop_pool_kernel" if pooled else "prop_kernel"
        if args.keep_detected:
            workload_name += " -- WITHOUT STOP_PHOTONS_ON_DETECTION (SetStopDetectedPhotons(false))"
        # algorithmic HBM bytes per launch: step record + RNG state read and
        # write per step (48 + 12 + 12 B) plus one 80-byte record per detected photon
        alg_bytes = n * 72.0 + hits_last * 80.0
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = None
        traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.workload == "c2" and n == (1 << 20) and n_bunches == 1 and args.photons_per_step == 200:
            # HBM-side bytes per launch from separate rocprofv3 --pmc passes of this same command
            # (FETCH_SIZE, WRITE_SIZE; see the file for how they were taken).  The file names the kernel and the git
            # revision it was profiled at: numbers of another kernel are not quoted.
            with open(tpath) as f:
                prof = json.load(f)
            traffic_source = {"file": "profiles/latest_traffic.json", "kernel": prof.get("kernel"), "git_revision": prof.get("git_revision"),
                              "profiled_kernel_ms": prof.get("kernel_ms")}
            if kernel_name + "<" in str(prof.get("kernel")):
                traffic = prof.get("bytes_per_launch")
                if prof.get("sq_insts_valu_per_launch"):
                    pmc = prof
            else:
                traffic_source["stale"] = "profiled kernel differs from the one this run launched (%s)" % kernel_name
        kernel_rate = photons_per_pass / n_bunches / (avg_ms * 1e-3)
        valu = None
        if not args.keep_detected and args.photons_per_step == {"c2": 200, "c3": 200, "c5": 400}[args.workload]:
            valu = valu_roofline(args.workload, kernel_rate, pmc, photons_per_pass / n_bunches, avg_ms)
        out = {
            "metric": "propagated photons/sec (whole node)", "value": value, "unit": "photons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d steps x %d photons per GPU and pass in %d bunch%s of %d, %s + tilt, 86 strings, oversize 5" %
                                   (workload_name, shard, args.photons_per_step, n_bunches, "" if n_bunches == 1 else "es", n, args.ice),
                       "kind": {"c2": "cascade steps", "c3": "cascade steps", "c5": "flasher steps (405 nm point source at a DOM)"}[args.workload],
                       "ice_layers": 171, "doms": 5160,
                       "steps_per_gpu": shard, "bunches_per_pass": n_bunches, "steps_per_bunch": n, "photons_per_step": args.photons_per_step,
                       "photons_per_pass_all_gpus": photons_per_pass * world,
                       "hit_gather": ("none" if not use_gather else
                                      ("clsimhip_gather_hits (RCCL: counts all-gather + p2p to rank 0), " + ("overlapped with the next kernel" if overlap else "between two launches")) if gatherer is not None else
                                      "FALLBACK torch.distributed gather_hits (%s)" % gather_note),
                       "hits_last_pass_rank0": hits_last, "hit_counter_last_pass_rank0": counted_last,
                       "hits_gathered_per_pass": (state["hits"] / args.steps) if use_gather else None,
                       "overflowed_buffers": state["overflow"] + (1 if counted_last > capacity else 0),
                       "gather_verified": verified},
            "multi_gpu_evidence": ("none: no run on more than one GPU exists yet; this line is one GPU" if world == 1 else
                                   "this run: see config.rccl_ranks / rank_devices / per_rank; the scaling curve is the driver's to compute"),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "avg_kernel_ms": avg_ms, "launches": int(launches),
                         "algorithmic_bytes_per_launch": alg_bytes, "valu": valu,
                         "note": "VALU/divergence-bound kernel (`valu`: useful_frac = the reference's arithmetic per photon x photons/s over the "
                                 "chip's vector peak); kernel-only rate %.4g photons/s" % kernel_rate},
        }
        # the scalars of the operative (vector-issue) roofline directly in `roofline`: a parser that keeps scalars only still sees them
        rf = out["roofline"]
        rf["traffic_ratio"] = (traffic / alg_bytes) if traffic else None
        rf["pmc_is_stored"] = bool(pmc)         # PMC-derived fields = stored rocprofv3 --pmc counts of this command x the LIVE kernel time
        rf["valu_useful_frac"] = valu.get("useful_frac") if valu else None
        rf["valu_useful_frac_counting_instructions"] = valu.get("useful_frac_counting_instructions") if valu else None
        rf["valu_issue_slot_frac"] = valu.get("issue_slot_frac") if valu else None
        rf["valu_issue_busy_at_measured_prices"] = valu.get("issue_busy_at_measured_prices") if valu else None
        rf["valu_lane_utilisation"] = valu.get("lane_utilisation") if valu else None
        rf["valu_overhead_ratio"] = valu.get("overhead_ratio") if valu else None
        rf["valu_floor_issue_slots_per_photon"] = valu["reference_ops_per_photon"]["transformed_issue_slots"] if valu else None
        rf["kernel_photons_per_s"] = kernel_rate
        if identity is not None:
            out["config"].update(identity)
        if world > 1 or (args.shard_steps > 0 and not args.c3_as_written):
            # the like-for-like N = 1 point of a scaling curve: ONE GPU running this very per-GPU shard (same bunches, gather
            # path on), measured by the builder with `python bench.py --gpus 1 --shard-steps <shard> [--workload ...]` under
            # CLSIMHIP_BENCH_GATHER=1 and stored with its git revision.  The default N = 1 line is C2 (one bunch of 1M steps,
            # the configuration the metric is quoted on), which is NOT the shard the N > 1 lines run.
            spath = os.path.join(ROOT, "profiles", "single_gpu_shard_rates.json")
            same = None
            if os.path.exists(spath):
                with open(spath) as f:
                    same = json.load(f).get("%s:%d" % (args.workload, shard))
            out["config"]["single_gpu_rate_same_shard"] = same if same is not None else {
                "value": None, "note": "no stored one-GPU measurement of a %d-step shard of %s; take it with --gpus 1 --shard-steps %d"
                                       % (shard, args.workload, shard)}
            if same is not None and same.get("value"):
                out["config"]["predicted_value_if_ranks_scale_ideally"] = same["value"] * world
            if args.workload == "c5":
                out["config"]["stream_count_note"] = (
                    "one RNG stream per step (the reference's definition) makes a step sequential work: %d steps per GPU against "
                    "393 216 resident unit slots%s" % (shard, "; fewer steps than lanes, so this GPU runs at its small-bunch rate "
                                                       "and N GPUs cannot reach N x the whole-bunch rate" if shard < 393216 else ""))
        if rehearsal:
            out["rehearsal"] = ("FUNCTIONAL REHEARSAL, NOT A MEASUREMENT: %d ranks time-share one GPU, torch.distributed over gloo, "
                                "RCCL library = %s" % (world, os.environ.get("CLSIMHIP_RCCL_LIBRARY", "(default)")))
            out["value"] = None
        if world == 1 and not args.no_host_path:
            # the reference's "actual" metric, outside the timed region of `value`: a second converter with two buffer sets
            del d_photons, d_count
            torch.cuda.empty_cache()
            conv2 = CV.initializeHIP(local_rank, geom, medium, bias, gens, pancakeFactor=5.0, enableDoubleBuffering=True,
                                     approximateNumberOfWorkItems=n, seed=12345)
            out["host_path"] = host_path_run(CV, args, steps_np, conv2, 8, in_place=True)
            # the reference's own meaning of GetConversionResult() (I3CLSimStepToPhotonConverter.h:178-189): the caller receives a photon
            # series of its own -- every record is copied out of the library's buffer (what the C++ adapter's GetConversionResult() does)
            out["host_path_copy"] = host_path_run(CV, args, steps_np, conv2, 8, in_place=False)
            del conv2
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, steps_np, args.cpu_seconds)
        if world == 1 and args.workload == "c2" and not args.no_table_maker and not args.keep_detected and args.shard_steps == 0:
            # BASELINE configs[4]'s other half, bounded: one warm-up and two timed passes of the table maker's own
            # bench bunch (262 144 steps x 200 photons, ~2 s a pass) after and outside the timed region of `value`, with its roofline
            # and an oracle baseline of a few seconds -- the same record `--workload tab` prints, so that a driver-run line holds it
            del d_steps
            torch.cuda.empty_cache()
            tm = tabulator_measure(args, local_rank, "tab", passes=2, warmup=1, cpu_seconds=(0.0 if args.no_cpu_baseline else 6.0))
            tm["definition"] = ("tabulated photons / wall clock of 2 passes (EnqueueSteps ... Finish) on a converter of its own, after and outside "
                                "the timed region of `value`; the at-size check (1.25e8 photons) is tests/test_production_size_gpu.py")
            out["table_maker"] = tm
        emit(json.dumps(out))
    fell_back = use_gather and gatherer is None
    if gatherer is not None:
        gatherer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if fell_back:
        # the line above measured the torch.distributed twin of the gather, not the C ABI's: say so to whoever reads the
        # exit code as well
        sys.stderr.write("bench.py: the C ABI's RCCL gather could not be used (%s); the fallback was measured\n" % gather_note)
        sys.exit(3)


if __name__ == "__main__":
    main()
