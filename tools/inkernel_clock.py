#!/usr/bin/env python3
"""Which clock does pair_forces really run at?  (round 6)

Three readings of one run on one box, after >= 2 s of back-to-back headline steps (262 144 bodies fp32, pairwise):

    in_kernel    the diagnostic build of nbody_pair.hip (-DNB_PAIR_STAMPS): every wave of the LAST pair_forces launch stamps the
                 shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) at its start and at its exit;
                 clock = 100 MHz x cycles / ticks per wave, median over the waves (MI355X_MICROARCH.md, DVFS give-back, item 6)
    clocked      pair_forces_clocked (nb_set_pair_clock_words: what bench.py puts in its line): the shipping library's own variant of the
                 kernel -- four scalar instructions more -- whose workgroups note their lifetime on the same two counters
    probe        nb_clock_probe_launch (lab library): eight single-wave workgroups that stamp the same two counters around a 20 us spin,
                 launched right behind a step on the same stream -- the experiment that showed a kernel BEHIND the load cannot tell
    hwmon        the power management's own figure (sysfs freq1_input of the card), sampled while the steps ran

and the duration of pair_forces by HIP events, so that  cycles per launch = ms x clock  can be compared across boxes.

    tools/build_pair_variant.sh stamps "-DNB_PAIR_STAMPS"
    NBODY_HIP_LIB=expv/libnbody_hip_stamps.so python3 tools/inkernel_clock.py [--bodies 262144] [--seconds 2.0]     (in_kernel + clocked)
    python3 tools/inkernel_clock.py                                                                                  (clocked + probe: the lab library)

Prints one JSON line."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def probe_clock(pkg, lib, words_dev, stream, groups=8, micros=20):
    """MHz per workgroup of one nb_clock_probe_launch (blocking: reads the words back)"""
    pkg.check(lib.nb_clock_probe_launch(words_dev.ptr, groups, micros, stream), "nb_clock_probe_launch")
    host = np.zeros(groups * 4, np.uint64)
    pkg.check(lib.nb_d2h(host.ctypes.data_as(ctypes.c_void_p), words_dev.ptr, host.nbytes, stream), "nb_d2h")
    w = host.reshape(groups, 4)
    return [100.0 * float(c) / float(t) for c, t in zip(w[:, 0], w[:, 1]) if t], [int(x) for x in w[:, 2]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=262144)
    ap.add_argument("--seconds", type=float, default=2.0)
    args = ap.parse_args()
    from bench_support import ChipWatch, make_bodies

    pkg = entry.load_package()
    if "NBODY_HIP_LIB" not in os.environ:
        pkg.use_lab()  # (the probe kernel is the lab library's)
    lib = pkg.lib()
    pkg.check(lib.nb_set_device(0))
    n = args.bodies
    pos0, vel0 = make_bodies(n, np.float32)
    s = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), np.float32, pos0, vel0, mode=pkg.NB_MODE_FAST, workspace=True)
    pl = pkg.pair_plan(n, np.float32)
    dt = np.float32(0.016)
    stamps = None
    try:
        stamps = ctypes.CDLL(pkg.LAB_LIB_PATH if pkg.is_lab() else pkg.LIB_PATH).nb_debug_read_pair_stamps
        stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    except AttributeError:
        pass  # (the shipping library: no stamps, the probe and hwmon only)
    pci = None
    try:  # the PCI address of device 0, from the HIP runtime this process already runs on (ChipWatch reads that card's hwmon)
        hip = ctypes.CDLL("libamdhip64.so.7")
        text = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(text, 64, 0) == 0:
            pci = text.value.decode().lower()
    except OSError:
        pass
    words = pkg.DeviceBuffer(8 * 32)
    watch = ChipWatch(pci)
    for _ in range(4):
        s.update(dt)
    s.synchronize()
    watch.start()
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(16):
            s.update(dt)
        s.synchronize()
        steps += 16
    have_probe = pkg.is_lab()
    probes, xccs = probe_clock(pkg, lib, words, None) if have_probe else ([], [])  # right behind the last step
    a, b, c = pkg.Event(), pkg.Event(), pkg.Event()
    pkg.check(lib.nb_set_pair_probe_event(b.h))
    forces_ms = []
    more = []
    for _ in range(10):
        a.record(None)
        s.update(dt)
        c.record(None)
        if have_probe:
            got, _ = probe_clock(pkg, lib, words, None)
            more += got
        c.synchronize()
        forces_ms.append(a.elapsed_ms(b))
    pkg.check(lib.nb_set_pair_probe_event(None))
    # the shipping library's own reading: pair_forces_clocked for ten more steps, back to back
    from bench_support import delivered_clock

    clock_words = pkg.DeviceBuffer(pl.grid_blocks * 16)
    pkg.check(lib.nb_memset(clock_words.ptr, 0, pl.grid_blocks * 16, None))
    pkg.check(lib.nb_set_pair_clock_words(clock_words.ptr, pl.grid_blocks * 16), "nb_set_pair_clock_words")
    d0, d1, d2 = pkg.Event(), pkg.Event(), pkg.Event()
    pkg.check(lib.nb_set_pair_probe_event(d1.h))
    clocked_ms = []
    for _ in range(10):
        d0.record(None)
        s.update(dt)
        d2.record(None)
        d2.synchronize()
        clocked_ms.append(d0.elapsed_ms(d1))
    pkg.check(lib.nb_set_pair_probe_event(None))
    pkg.check(lib.nb_set_pair_clock_words(None, 0), "nb_set_pair_clock_words")
    host_words = np.zeros(pl.grid_blocks * 2, np.uint64)
    pkg.check(lib.nb_d2h(host_words.ctypes.data_as(ctypes.c_void_p), clock_words.ptr, host_words.nbytes, None))
    clocked = delivered_clock(host_words)
    watch.stop()
    lib_name = os.path.basename(pkg.LAB_LIB_PATH if pkg.is_lab() else pkg.LIB_PATH)
    out = {"bodies": n, "steps_before_the_reading": steps, "seconds_of_load": round(time.perf_counter() - t0, 2), "library": lib_name,
           "pair_forces_ms": round(float(np.median(forces_ms)), 4), "pair_forces_ms_min": round(min(forces_ms), 4),
           "pair_forces_clocked_ms": round(float(np.median(clocked_ms)), 4), "clocked": clocked, "hwmon": watch.summary()}
    if clocked:
        out["clocked_mcycles_per_pair_forces"] = round(out["pair_forces_ms"] * clocked["mhz"] * 1e-3, 3)
    if probes + more:
        out.update({"probe_mhz_median": round(float(np.median(probes + more)), 1), "probe_mhz_min": round(min(probes + more), 1), "probe_mhz_max": round(max(probes + more), 1),
                    "probe_xcc_ids": sorted(set(xccs))})
    if stamps is not None:
        waves = min(pl.grid_blocks * pl.waves_per_block, 8192)
        raw = np.zeros(8192 * 8, np.uint64)
        assert stamps(raw.ctypes.data_as(ctypes.c_void_p), raw.nbytes) == 0
        st = raw.reshape(8192, 8)[:waves]
        cycles = (st[:, 2] - st[:, 0]).astype(np.float64)
        ticks = (st[:, 7] - st[:, 6]).astype(np.float64)
        ok = ticks > 0
        mhz = 100.0 * cycles[ok] / ticks[ok]
        out["in_kernel_mhz_median"] = round(float(np.median(mhz)), 1)
        out["in_kernel_mhz_p10_p90"] = [round(float(np.percentile(mhz, 10)), 1), round(float(np.percentile(mhz, 90)), 1)]
        out["in_kernel_wave_cycles_median"] = int(np.median(cycles[ok]))
        out["in_kernel_wave_ms_median"] = round(float(np.median(ticks[ok])) / 1e5, 4)
        out["waves_stamped"] = int(ok.sum())
    for key in ("probe_mhz_median", "in_kernel_mhz_median"):
        if key in out:
            out[key.replace("_mhz_median", "_mcycles_per_pair_forces")] = round(out["pair_forces_ms"] * out[key] * 1e-3, 3)
    if out["hwmon"] and out["hwmon"].get("sclk_mhz"):
        out["hwmon_mcycles_per_pair_forces"] = round(out["pair_forces_ms"] * out["hwmon"]["sclk_mhz"] * 1e-3, 3)
    print(json.dumps(out), flush=True)
    s.free()
    words.free()
    clock_words.free()


if __name__ == "__main__":
    main()
