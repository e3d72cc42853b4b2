"""bench_support.py -- what bench.py measures AROUND its timed region (never inside it): the workload generator, the other
BASELINE configs, the one-rank projection, the per-kernel split of the pairwise step, the CPU baseline and the N > 1 diagnostics.
Benchmark support, not product: everything here goes through the C-ABI via cuda-nbody_amd/__init__.py; the CPU baseline is the
only place that loads oracle/ (test infrastructure).  Reference citations are relative to /root/reference/."""
from __future__ import annotations

import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md:41
FP64_VECTOR_PEAK_TFLOPS = 78.6   # public spec
HBM_PEAK_GBS = 8000.0


def make_bodies(n: int, dtype):
    """The bodies a fresh `nbody --numbodies=N [--fp64]` process starts from, drawn by the PRODUCT's randomise_bodies
    (libnbody_host.so, pinned bit-for-bit to the reference's own code in tests/test_host_cpp.py): SHELL configuration,
    third segment of the unseeded rand() stream (fp32 reset, fp64 reset with demo_params[0] scales, then the active
    precision with the N-scaled params).  "synthetic random bodies" of BASELINE.json."""
    host = ctypes.CDLL(os.path.join(ROOT, "cuda-nbody_amd", "libnbody_host.so"))
    f32p, f64p = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    host.nbh_srand.argtypes = [ctypes.c_uint]
    host.nbh_randomise_f32.argtypes = [ctypes.c_int, f32p, f32p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    host.nbh_randomise_f64.argtypes = [ctypes.c_int, f64p, f64p, ctypes.c_size_t, ctypes.c_float, ctypes.c_float]
    host.nbh_scale_params_for.argtypes = [ctypes.c_size_t, f32p, f32p]
    shell = 1  # NBodyConfig::NBODY_CONFIG_SHELL

    def draw(T, cluster, velocity):
        pos, vel = np.zeros(4 * n, T), np.zeros(4 * n, T)
        if T == np.float32:
            host.nbh_randomise_f32(shell, pos.ctypes.data_as(f32p), vel.ctypes.data_as(f32p), n, cluster, velocity)
        else:
            host.nbh_randomise_f64(shell, pos.ctypes.data_as(f64p), vel.ctypes.data_as(f64p), n, cluster, velocity)
        return pos, vel

    host.nbh_srand(1)
    draw(np.float32, 1.54, 8.0)
    draw(np.float64, 1.54, 8.0)
    c, v = ctypes.c_float(1.54), ctypes.c_float(8.0)
    host.nbh_scale_params_for(n, ctypes.byref(c), ctypes.byref(v))
    return draw(dtype, c.value, v.value)


# What the instruction mix of the production loop can do at best on this chip: the unit-mass inner loop run in isolation
# (tools/loop_microbench_gen.py, profiles/round2_loop_microbench.txt) sustains one packed interaction pair per 61.5 SIMD cycles
# with 3-4 runnable waves (11 v_pk_* at ~4 cycles + 2 v_rsq_f32 at ~8.3); at the nominal 2.4 GHz that is
# 1024 SIMDs x 128 interactions / 61.5 cycles = 5.115e12 interactions/s = 65.0 % of the 157.3 TFLOP/s "20 flop" roofline.
FP32_ISSUE_CEILING_INTERACTIONS_PER_S = 1024 * 128 * 2.4e9 / 61.5
# fp64 (profiles/round2_fp64_issue_probes.txt): 14 add/mul/fma at 4.41 cycles + one v_rsq_f64 at 16.3 = 78 cycles per interaction and wave
FP64_ISSUE_CEILING_INTERACTIONS_PER_S = 1024 * 64 * 2.4e9 / 78.0


_CHIP_SAMPLER = r"""
import os, sys, time
hwmon, out, parent = sys.argv[1], sys.argv[2], int(sys.argv[3])
def read(name):
    try:
        with open(os.path.join(hwmon, name)) as fh:
            return fh.read().strip()
    except OSError:
        return ""
end = time.time() + 900
with open(out, "a", buffering=1) as fh:
    while os.getppid() == parent and time.time() < end:
        fh.write("%.6f %s %s\n" % (time.time(), read("freq1_input") or "-", read("power1_input") or read("power1_average") or "-"))
        time.sleep(0.025)
"""


class ChipWatch:
    """The chip's shader clock and socket power WHILE the timed region runs, read from the amdgpu driver's sysfs files of the card
    with this device's PCI address (hwmon freq1_input / power1_input, or power1_average where the driver names it so; an ordinary user may read them).  Why it is in the line: the
    pairwise kernel keeps the vector ALUs ~97 % busy and the socket sits at its power cap (measured: 1 337-1 354 W of 1 400 W), so
    the clock the power management grants -- 2.04 ... 2.29 GHz by box, against the 2.4 GHz the peak is computed at -- decides
    `roofline.frac` between 0.85 and 0.93 for the same binary.
    The sampler is a CHILD PROCESS (two small files every 25 ms, time-stamped lines into a temporary file) started before the
    warm-up; start() / stop() only note the wall-clock window whose samples count.  It never opens the GPU and shares no
    interpreter lock with the timing loop (a sampler THREAD cost a 26 ms timed region 1.6 ms: the loop's ctypes calls give up the
    interpreter lock at every launch and wait to get it back).  Everything is None where the files are missing or unreadable."""

    PEAK_CLOCK_MHZ = 2400.0  # what FP32_VECTOR_PEAK_TFLOPS / FP64_VECTOR_PEAK_TFLOPS assume

    def __init__(self, pci_address, sysfs="/sys/class/drm"):
        import glob
        import subprocess
        import tempfile

        self.hwmon, self.child, self.path, self.window, self.samples = None, None, None, [None, None], []
        for dev in glob.glob(os.path.join(sysfs, "card*", "device")) if pci_address else ():
            try:
                if os.path.basename(os.path.realpath(dev)).lower() != str(pci_address).lower():
                    continue
                found = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
                if found and os.access(os.path.join(found[0], "freq1_input"), os.R_OK):
                    self.hwmon = found[0]
                    break
            except OSError:
                continue
        if self.hwmon is None:
            return
        try:
            fd, self.path = tempfile.mkstemp(prefix="nbody_chip_watch_", suffix=".txt")
            os.close(fd)
            # (an environment of its own: under rocprofv3 the parent's LD_PRELOAD / tool variables would make the child a second
            # profiled process that opens the GPU; this one must never touch it)
            self.child = subprocess.Popen([sys.executable, "-S", "-E", "-c", _CHIP_SAMPLER, self.hwmon, self.path, str(os.getpid())],
                                          stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env={"PATH": "/usr/bin:/bin"})
        except OSError:
            self.child = None
        import atexit

        atexit.register(self._collect)  # (a run that ends before summary(): the child and its file do not outlive it)

    def start(self):
        self.window[0] = time.time()

    def stop(self):
        """Only notes the end of the window: ending the child and reading its file waits for summary(), so that nothing idles the
        GPU between the timed region and what is timed right after it (30 ms of idling cost the next kernels 3 % of clock)."""
        self.window[1] = time.time()

    def _collect(self):
        if self.child is None:
            if self.path and os.path.exists(self.path):
                os.unlink(self.path)
            return
        if self.window[1] is None:
            self.window = [0.0, 0.0]  # (never stopped: nothing counts)
        wait = self.window[1] + 0.03 - time.time()  # (one more sample period: the last line of the window is on disk)
        if wait > 0:
            time.sleep(wait)
        self.child.terminate()
        try:
            self.child.wait(timeout=2)
        except Exception:  # noqa: BLE001
            self.child.kill()
        self.child = None
        try:
            with open(self.path) as fh:
                for text in fh:
                    part = text.split()
                    if len(part) == 3 and part[1] != "-" and self.window[0] <= float(part[0]) <= self.window[1]:
                        self.samples.append((float(part[1]) * 1e-6, None if part[2] == "-" else float(part[2]) * 1e-6))
            os.unlink(self.path)
        except (OSError, ValueError):
            pass

    def _read(self, name):
        try:
            with open(os.path.join(self.hwmon, name)) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    def summary(self):
        self._collect()
        if not self.samples:
            return None
        clocks = sorted(c for c, _ in self.samples)
        powers = sorted(p for _, p in self.samples if p is not None)
        cap = self._read("power1_cap")
        mid = lambda v: v[len(v) // 2] if v else None  # noqa: E731
        return {"what": "shader clock and socket power during the timed region (amdgpu hwmon freq1_input / power1_input, a child process sampling every 25 ms)",
                "sclk_mhz": mid(clocks), "sclk_mhz_min": clocks[0], "sclk_mhz_max": clocks[-1], "socket_power_w": mid(powers),
                "power_cap_w": None if cap is None else cap * 1e-6, "samples": len(clocks), "peak_is_computed_at_mhz": self.PEAK_CLOCK_MHZ}


def plan_dict(pkg, n, dtype, layout, world=1):
    """The launch plan of a step as the line (and profiles/*_pmc_summary.json, tools/summarize_prof.py) spell it."""
    if layout == "pairwise":
        pair = pkg.pair_plan(n, dtype)
        return {"layout": "pairwise", "bodies_per_lane": pair.bodies_per_lane, "waves_per_block": pair.waves_per_block, "workgroups_per_block": pair.splits,
                "blocks": pair.blocks, "block_bodies": pair.block_bodies, "reaction_slots": pair.reaction_slots, "grid": pair.grid_blocks,
                "lds_bytes": pair.lds_bytes, "workspace_bytes": pair.workspace_bytes}
    plan = pkg.plan(n // world, n, dtype)
    return {"bodies_per_lane": plan.bodies_per_lane, "lane_groups": plan.lanes_per_body, "lds_tile_bodies": plan.tile_bodies, "grid": plan.grid_blocks, "lds_bytes": plan.lds_bytes}


def pmc_summary(n, fp64, mode, layout, plan_now=None, kernel=""):
    """What the committed rocprofv3 --pmc passes say about this configuration's dominant kernel (tools/profile.sh ->
    tools/summarize_prof.py -> profiles/round*_n{N}_{f32|f64}[_pairwise|_strict][_finish]_pmc_summary.json; counters cannot be read
    from inside the process being timed): {"valu_busy", "hbm_bytes_per_launch", "source"} of the NEWEST such file, or None.  A file
    taken with another launch plan says nothing about this run: then only {"source": "... another launch plan"}."""
    import glob
    import json

    tag = f"n{n}_{'f64' if fp64 else 'f32'}" + ("_strict" if mode == "strict" else ("_pairwise" if layout == "pairwise" else "")) + kernel
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"round*_{tag}_pmc_summary.json")), key=lambda f: int(os.path.basename(f).split("_")[0][5:]))
    if not found:
        return None
    with open(found[-1]) as fh:
        summary = json.load(fh)
    source = os.path.relpath(found[-1], ROOT)
    if plan_now is not None and mode != "strict" and summary.get("kernel_plan") != plan_now:
        return {"source": f"{source} was taken with another launch plan: re-run tools/profile.sh"}
    d = summary.get("derived", {})
    return {"valu_busy": None if d.get("valu_busy_fraction") is None else round(d["valu_busy_fraction"], 4), "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"), "source": source}


def pair_evaluations(pair) -> float:
    """pair evaluations per step of the pairwise layout: NB x (NB/2 + 1) block pairs of (64 I)^2"""
    return float(pair.blocks) * (pair.blocks // 2 + 1) * pair.block_bodies * pair.block_bodies


def fractions(n, fp64, layout, ms, pair=None):
    """(frac, executed_frac) of the vector-FMA peak.  `frac`: ALGORITHMIC flop of the reference convention -- 20 (30) per directed
    interaction, N^2 interactions (compute.cpp:16-18) -- over the time; it can pass 1 for the pairwise layout, which
    evaluates each pair once.  `executed_frac`: the flop the kernels really issue -- 24 (36) per pair evaluation for the pairwise
    layout, the algorithmic count for the one-sided and STRICT kernels (they do evaluate every directed interaction) -- i.e. how
    busy the FMA pipes are; never above 1."""
    flops, peak = (30, FP64_VECTOR_PEAK_TFLOPS) if fp64 else (20, FP32_VECTOR_PEAK_TFLOPS)
    frac = flops * float(n) * n / (ms * 1e-3) / (peak * 1e12)
    if layout != "pairwise" or pair is None:
        return round(frac, 4), round(frac, 4)
    return round(frac, 4), round((36 if fp64 else 24) * pair_evaluations(pair) / (ms * 1e-3) / (peak * 1e12), 4)


# what every fp64 entry of the line says about its peak (round-5 review: 78.6 is a public figure, not a measurement)
FP64_PEAK_NOTE = {"fp64_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                  "fp64_peak_source": "public MI355X spec (half the packed-fp32 rate at 2.4 GHz); NOT confirmed on the box",
                  "fp64_issue_peak_tflops": round(1024 * 64 * 2 * 2.4e9 / 4.41 / 1e12, 1),
                  "fp64_issue_peak_source": "profiles/round2_fp64_issue_probes.txt: v_fma_f64 issues in 4.41 cycles per wave64 on a SIMD (not 4) -> 1024 SIMDs x 64 lanes x 2 flop x 2.4 GHz / 4.41"}

CONFIG_WARMUP_S = 0.25  # other_configs: steps run (untimed) for at least this long before a config is timed
GRAPH_SIZES = (16384, 65536)  # ... and these pairwise entries are also timed as hipGraph replays


def other_configs(pkg, lib, headline):
    """BASELINE.json configs besides the headline one, plus STRICT (the parity-exact mode) and, for FAST, both layouts
    (pairwise = nb_integrate_ws_* with a workspace, one-sided = nb_integrate_*), each as
    {workload, bodies, dtype, mode, layout, steps, warmup_steps, ms_per_step, frac_algorithmic, executed_frac, valu_busy}: untimed steps for at least
    CONFIG_WARMUP_S (steady clocks -- these entries are steady-state figures; the headline keeps the driver's W warm-up steps), then
    K steps between two HIP events on the launch stream (the reference's GPU protocol, compute_cuda.cpp:183-195); the fractions:
    see fractions()."""
    cases = [
        ("configs[1]", 65536, False, "fast", 200),
        ("configs[2]", 262144, False, "fast", 20),
        ("configs[4]", 262144, True, "fast", 5),
        ("configs[3]'s system on ONE GPU", 1048576, False, "fast", 3),
        ("STRICT = the CPU path's bits", 262144, False, "strict", 5),
        ("STRICT", 262144, True, "strict", 3),
        ("configs[0]'s system on the GPU", 1024, False, "fast", 2000),
        ("configs[0]'s system on the GPU", 1024, False, "strict", 500),
        ("small system", 16384, False, "fast", 1000),
        # between the powers of two: the launch geometry (R, and any number of workgroups per block) comes from plan_pair's cost model
        ("between the powers of two", 50000, False, "fast", 300),
        ("between the powers of two", 100000, False, "fast", 100),
        ("between the powers of two", 300000, False, "fast", 10),
        # beyond BASELINE's sizes: one tournament would want 206 GB of reaction slots; the tournament cut into slices inside 16 GB
        ("4 Mi bodies, workspace capped at 16 GB", 4194304, False, "fast", 2),
    ]
    out = []
    for what, n, fp64, mode_name, steps in cases:
        dtype = np.float64 if fp64 else np.float32
        mode = pkg.NB_MODE_FAST if mode_name == "fast" else pkg.NB_MODE_STRICT
        layouts = ["one-sided"] if mode_name == "fast" else ["strict"]
        cap = (16 << 30) if n > 1048576 else None
        if mode_name == "fast" and pkg.workspace_bytes(n, dtype, mode, cap):
            layouts.insert(0, "pairwise")
        if cap is not None:
            layouts = layouts[:1]  # (the one-sided kernel at this size: 3.6 s per step, nothing new)
        pos0 = vel0 = None
        for layout in layouts:
            if (n, fp64, mode_name, layout) == headline:
                continue
            if pos0 is None:
                pos0, vel0 = make_bodies(n, dtype)
            system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=mode, workspace=(layout == "pairwise"), workspace_cap=cap)
            dt = dtype(np.float32(0.016))
            system_bytes = system._workspace_bytes
            # warm-up by TIME, not by count: the chip leaves its sleep clock while the bodies are made on the host, and the power
            # management takes ~0.2 s under load to settle (a 13 ms run of 16 384 bodies right after a cold start is timed at
            # 2.1-2.2 GHz, the same run after 0.25 s of steps at 2.39 GHz: kernel fraction 0.58 against 0.60)
            warm_until, warmed = time.perf_counter() + CONFIG_WARMUP_S, 0
            while warmed == 0 or time.perf_counter() < warm_until:
                for _ in range(max(1, steps // 10)):
                    system.update(dt)
                system.synchronize()
                warmed += max(1, steps // 10)
            e0, e1 = pkg.Event(), pkg.Event()
            e0.record(None)
            for _ in range(steps):
                system.update(dt)
            e1.record(None)
            e1.synchronize()
            ms = e0.elapsed_ms(e1) / steps
            graph_ms = None
            if layout == "pairwise" and n in GRAPH_SIZES:
                # the same steps as ONE hipGraph launch per 100 (nb_graph_create_ws_*: the two kernels of a step captured back to back):
                # what a graph saves is the second launch and the gap of each step -- 8 + 3 us of a 56 us step at 16 384 bodies
                # (profiles/round4_*); timed right after the eager steps, same clocks
                per_replay = 100
                system.update_many(dt, per_replay)
                system.synchronize()
                replays = max(2, steps // per_replay)
                g0, g1 = pkg.Event(), pkg.Event()
                g0.record(None)
                for _ in range(replays):
                    system.update_many(dt, per_replay)
                g1.record(None)
                g1.synchronize()
                graph_ms = g0.elapsed_ms(g1) / (replays * per_replay)
            system.free()
            frac, executed = fractions(n, fp64, layout, ms, pkg.pair_plan(n, dtype) if layout == "pairwise" else None)
            # (kept short: the whole line should stay well under what a log tail holds; interactions/s = bodies^2 / ms_per_step)
            # frac_algorithmic: SURVEY 8(d)'s count (20 / 30 flop x N^2) over the peak -- NOT a utilisation figure for the pairwise layout
            # (each pair is evaluated once: it can pass 1); executed_frac: the flop really issued over the same peak; valu_busy: the
            # hardware's own answer, from the committed PMC passes of this configuration where its plan matches (None: no such pass)
            pmc = pmc_summary(n, fp64, mode_name, layout, plan_dict(pkg, n, dtype, layout) if cap is None else None)
            out.append({"workload": what, "bodies": n, "dtype": "f64" if fp64 else "f32", "mode": mode_name, "layout": layout, "steps": steps, "warmup_steps": warmed,
                        "ms_per_step": float(f"{ms:.5g}"), "frac_algorithmic": frac, "executed_frac": executed, "valu_busy": None if not pmc else pmc.get("valu_busy")})
            if layout == "pairwise":
                # (said per entry, so that nobody reads a fraction above 1 as utilisation: the count is the reference's, the work is half of it)
                out[-1]["frac_algorithmic_counts"] = "N^2 directed interactions (reference convention); the kernel evaluates each PAIR once = half of them: executed_frac is the utilisation figure"
            if fp64:
                out[-1].update(FP64_PEAK_NOTE)
            if what == "configs[1]":
                # BASELINE configs[1] says "LDS tile = 256 bodies": no shipping FAST kernel stages a tile of bodies j in LDS
                out[-1]["lds_tile"] = None
                out[-1]["why"] = ("no staged tile: the pairwise kernel rotates bodies j through the wavefront (DPP), the one-sided kernel streams them through the scalar unit; "
                                  "a staged 256-body LDS tile measured slower -- profiles/round2_scalar_stream_ab.txt, profiles/round4_lds_tile_experiment.txt; --blockSize stays a hint")
            if graph_ms is not None:
                out[-1]["hipgraph_ms_per_step"] = float(f"{graph_ms:.5g}")
                out[-1]["hipgraph_gain"] = round(ms / graph_ms - 1.0, 4)
            if cap is not None:
                out[-1]["workspace_bytes"] = system_bytes
    return out


def rank_projection(pkg, lib, n, dtype, dt, damping, single_ms):
    """ms of ONE rank's kernels of a 2 / 4 / 8-rank pairwise step on this GPU (rank G/2; diagonal + G/2 rectangles + folds + finish),
    no exchange -- tools/pair_rank_probe.py has the longer version."""
    f32 = np.dtype(dtype) == np.float32
    emulate = lib.nb_emulate_pair_rank_f32 if f32 else lib.nb_emulate_pair_rank_f64
    pos0, vel0 = make_bodies(n, dtype)
    bufs = [pkg.DeviceBuffer(pos0.nbytes) for _ in range(3)]
    bufs[0].upload(pos0), bufs[2].upload(vel0)
    out = {"what": "PROJECTION, not a multi-GPU measurement: kernel ms of ONE rank of a G-rank pairwise step, alone on one GPU, no exchange",
           "single_gpu_ms_per_step": float(f"{single_ms:.5g}"), "ranks": {}}
    for G in (2, 4, 8):
        need = ctypes.c_size_t(0)
        if emulate(None, None, None, None, ctypes.byref(need), n, G, 0, dt, damping, None) != 0:
            continue
        work = pkg.DeviceBuffer(need.value)

        def one_step():
            pkg.check(emulate(bufs[1].ptr, bufs[0].ptr, bufs[2].ptr, work.ptr, ctypes.byref(need), n, G, G // 2, dt, damping, None), "nb_emulate_pair_rank")

        one_step()
        pkg.check(lib.nb_device_synchronize())
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(None)
        for _ in range(10):
            one_step()
        e1.record(None)
        e1.synchronize()
        ms = e0.elapsed_ms(e1) / 10
        out["ranks"][str(G)] = {"kernel_ms": float(f"{ms:.5g}"), "speedup_excl_exchange": round(single_ms / ms, 2)}
        work.free()
    for b in bufs:
        b.free()
    # ... and the same rank's whole STEP with the real RCCL on the chip (round 5): a LOOPBACK rank (nb_comm_loopback_open) launches the
    # kernels, RCCL groups, events and waits of rank G/2 of a G-rank step, every send going to the rank itself -- a step minus what the
    # xGMI links add.  A child process (tools/exchange_contention.py) under a timeout: it brings up RCCL communicators, and nothing
    # a third-party library does may cost this process its headline line.
    if np.dtype(dtype) == np.float32:
        out["loopback"] = loopback_projection(n, single_ms)
        out["host_enqueue"] = host_enqueue_projection(n, out["loopback"])
    return out


def host_enqueue_projection(n, loopback, timeout_s=60):
    """What the HOST needs to enqueue one step of an 8-rank job (round 6), measured on this one GPU with the real RCCL, in child processes:
    one rank's whole step (the loopback rank: what each process of `bench.py --gpus 8`, and each thread of the crew that steps the devices of
    `nbody --numdevices 8`, enqueues -- in parallel with the others), and the whole world's step in ONE process (the lab's in-process world:
    eight ranks that share one RCCL communicator, so its RCCL groups are issued by one thread -- the crew only takes the kernels and
    events), with the crew and with the calling thread alone.  Labelled projections, never `value`."""
    import json
    import subprocess

    got = {"what": "host wall clock (ms) to enqueue ONE step of 8 ranks at this body count, real RCCL, one GPU; a step whose enqueue takes longer than its kernels is host-bound.  "
                   "one_rank_*: a loopback rank, in a torch process (torch's own HIP runtime and RCCL: what a rank of `bench.py --gpus 8` is) and in a plain process (ROCm's: "
                   "what a thread of the crew behind `nbody --numdevices 8` is); read over the first 8 steps of a stretch (40 steps in a torch process run into a full device queue)",
           "one_rank_of_8_torch_process_ms": None, "one_rank_of_8_plain_process_ms": None, "in_process_world_8_ranks": {}}
    try:
        got["one_rank_of_8_torch_process_ms"] = loopback["ranks"]["8"].get("host_enqueue_ms")
    except (KeyError, TypeError):
        pass
    tool = os.path.join(ROOT, "tools", "graph_capture_probe.py")
    env = dict(os.environ)
    for name in ("NBODY_RCCL_LIB", "FAKE_RCCL_IPC", "NCCL_DEBUG", "NBODY_HIP_LIB"):
        env.pop(name, None)
    try:
        done = subprocess.run([sys.executable, tool, "--worlds", "loopback", "--what", "none", "--bodies", str(n), "--world", "8", "--steps", "30"], capture_output=True, text=True, timeout=timeout_s, env=env)
        got["one_rank_of_8_plain_process_ms"] = next(json.loads(text) for text in done.stdout.splitlines() if text.startswith("{"))["eager_host_enqueue_ms_per_step"]
    except (subprocess.TimeoutExpired, StopIteration, KeyError, ValueError) as exc:
        got["error"] = f"loopback rank in a plain process: {exc!r}"
    for key, threads in (("crew_ms", "1"), ("calling_thread_alone_ms", "0")):
        try:
            done = subprocess.run([sys.executable, tool, "--worlds", "inprocess", "--what", "none", "--bodies", str(n), "--world", "8", "--steps", "30"],
                                  capture_output=True, text=True, timeout=timeout_s, env={**env, "NBODY_STEP_THREADS": threads})
            row = next(json.loads(text) for text in done.stdout.splitlines() if text.startswith("{"))
            got["in_process_world_8_ranks"][key] = row["eager_host_enqueue_ms_per_step"]
            got["in_process_world_8_ranks"]["stream_ms_per_step_all_8_ranks_on_one_gpu"] = row["eager_stream_ms_per_step"]
        except (subprocess.TimeoutExpired, StopIteration, KeyError, ValueError) as exc:
            got["error"] = f"in-process world ({key}): {exc!r}"
            break
    return got


def loopback_projection(n, single_ms, worlds=(2, 4, 8), timeout_s=180):
    import json
    import subprocess

    tool = os.path.join(ROOT, "tools", "exchange_contention.py")
    got = {"what": "ONE GPU, real RCCL: ms per step of a loopback rank (rank G/2 of a nominal G-rank communicator; its kernels, RCCL groups, events and waits -- "
                   "the transfers go to the rank itself), next to the same kernels with no communicator; a multi-GPU step minus the xGMI transfer time, NOT a "
                   "multi-GPU measurement", "ranks": {}}
    env = dict(os.environ)
    for name in ("NBODY_RCCL_LIB", "FAKE_RCCL_IPC", "NCCL_DEBUG"):
        env.pop(name, None)
    # one child per world size: a process that has brought up and destroyed one communicator after another has been seen to time the
    # LATER worlds' kernels-alone figure at twice its value (the two streams of nb_emulate_pair_rank_* no longer overlapping --
    # the runtime maps streams onto a few hardware queues as they come and go), gpurun call r5g
    for world in worlds:
        try:
            done = subprocess.run([sys.executable, tool, "--torch", "--bodies", str(n), "--world", str(world), "--steps", "40", "--rounds", "3",
                                   "--phases", "step_pairwise_late1_group_per_round,kernels_alone_pairwise_late1"], capture_output=True, text=True, timeout=timeout_s / len(worlds), env=env)
        except subprocess.TimeoutExpired:
            got["error"] = f"the child for {world} ranks did not finish within {timeout_s / len(worlds):.0f} s"
            break
        rows = [json.loads(text) for text in done.stdout.splitlines() if text.startswith("{")]
        for row in rows:
            step, alone = row.get("step_pairwise_late1_group_per_round"), row.get("kernels_alone_pairwise_late1")
            if step:
                got["ranks"][str(row["nominal_world"])] = {"step_ms": step, "kernels_alone_ms": alone, "exposed_exchange_ms": None if alone is None else round(step - alone, 4),
                                                           "speedup_excl_link_time": round(single_ms / step, 2),
                                                           "host_enqueue_ms": row.get("host_enqueue_ms_pairwise_late1_group_per_round")}
                got["rccl_version"], got["rccl_library"] = row.get("rccl_version"), row.get("rccl_library")
        if done.returncode != 0 or not rows:
            got["error"] = f"{world} ranks: exit status {done.returncode}" + (": " + done.stderr.strip().splitlines()[-1][:300] if done.stderr.strip() else "")
            break
    return got


def delivered_clock(words):
    """{"mhz": median, ...} from the words pair_forces_clocked left (two u64 per workgroup: its lifetime in shader cycles and in ticks of the
    constant 100 MHz counter), None when no workgroup wrote any"""
    w = np.asarray(words, dtype=np.uint64).reshape(-1, 2)
    mhz = sorted(100.0 * float(c) / float(t) for c, t in zip(w[:, 0], w[:, 1]) if t and c)
    if not mhz:
        return None
    cycles = sorted(float(c) for c, t in zip(w[:, 0], w[:, 1]) if t and c)
    return {"what": "the clock pair_forces REALLY ran at, read inside the kernel after the timed region: every workgroup of pair_forces_clocked (the same kernel + four "
                    "scalar instructions) notes its lifetime on the shader-cycle counter (s_memtime) and on the constant 100 MHz counter (s_memrealtime); "
                    "MHz = 100 x cycles / ticks, median over the workgroups of the last launch (nb_set_pair_clock_words).  The power management's own figure "
                    "(hwmon sclk) reads higher and lags; a probe launched BEHIND the kernel reads the unloaded clock (profiles/round6_delivered_clock.txt)",
            "mhz": round(mhz[len(mhz) // 2], 1), "mhz_p10": round(mhz[len(mhz) // 10], 1), "mhz_p90": round(mhz[(len(mhz) * 9) // 10], 1), "workgroups": len(mhz),
            "workgroup_mcycles_median": round(cycles[len(cycles) // 2] * 1e-6, 4)}


def pair_kernel_split(pkg, lib, step, stream, reps=10, grid_blocks=0):
    """Average duration of the two kernels of the pairwise step, each on its own: pair_forces (the dominant kernel) and pair_finish,
    from HIP events on the launch stream -- one before the step, one the library records BETWEEN the two launches
    (nb_set_pair_probe_event, tuning header), one after.  Taken after the timed region.  Round 6: then `reps` more steps with
    pair_forces_clocked in pair_forces' place (nb_set_pair_clock_words; grid_blocks = the launch's workgroups, 0 = skip): the clock
    the kernel really ran at.  Returns (forces ms, finish ms, delivered clock dict or None)."""
    # The reps are queued back to back (an event triple per rep, the probe event switched on the host before each call) and read
    # after ONE synchronisation: a synchronisation per rep idled the chip between steps, and the forces kernel then read 0.5 % slower
    # than the timed steps it belongs to.
    events = [(pkg.Event(), pkg.Event(), pkg.Event()) for _ in range(reps)]
    words = pkg.DeviceBuffer(grid_blocks * 16) if grid_blocks else None
    clock = None
    try:
        for before, between, after in events:
            pkg.check(lib.nb_set_pair_probe_event(between.h), "nb_set_pair_probe_event")
            before.record(stream)
            step()
            after.record(stream)
        pkg.check(lib.nb_set_pair_probe_event(None), "nb_set_pair_probe_event")
        if words is not None:  # (queued right behind: the chip stays under the same load)
            pkg.check(lib.nb_memset(words.ptr, 0, grid_blocks * 16, stream), "nb_memset")
            pkg.check(lib.nb_set_pair_clock_words(words.ptr, grid_blocks * 16), "nb_set_pair_clock_words")
            for _ in range(reps):
                step()
            pkg.check(lib.nb_set_pair_clock_words(None, 0), "nb_set_pair_clock_words")
            host = np.zeros(grid_blocks * 2, np.uint64)
            pkg.check(lib.nb_d2h(host.ctypes.data_as(ctypes.c_void_p), words.ptr, host.nbytes, stream), "nb_d2h")  # (blocking on the stream: everything above is done)
            clock = delivered_clock(host)
        events[-1][2].synchronize()
    finally:
        pkg.check(lib.nb_set_pair_probe_event(None), "nb_set_pair_probe_event")
        pkg.check(lib.nb_set_pair_clock_words(None, 0), "nb_set_pair_clock_words")
        if words is not None:
            words.free()
    forces = sum(before.elapsed_ms(between) for before, between, _ in events)
    finish = sum(between.elapsed_ms(after) for _, between, after in events)
    return forces / reps, finish / reps, clock


def cpu_baseline(n, dtype, pos0, vel0, sample_bodies):
    """The CPU path (oracle/: a port of BodySystemCPU<T>::update; test infrastructure, loaded here only) timed on this host:
    a bounded sample of the headline workload, and BASELINE configs[0] exactly as stated -- 1 024 bodies, fp32, 100 steps, no
    warm-up, steady clock around the loop (compute_cpu.cpp:72-88)."""
    O = entry.load_oracle()
    orc1 = O.Oracle()
    pos_h, vel_h = orc1.startup_state(n, dtype)
    # the workload above came from the product's randomise_bodies; the checker's must be the same bytes
    assert pos_h.tobytes() == pos0.tobytes() and vel_h.tobytes() == vel0.tobytes(), "product and oracle start-up bodies differ"
    sample = sample_bodies or max(8, min(n, int(2.0e10 // n) // 8 * 8))
    base = {}
    # OpenMP leg: the reference's fp32 loop forks INSIDE the j loop (bodysystemcpu.cpp:156-168), i.e. one fork/join per body j --
    # it is slow by construction, so it gets a smaller sample and at most 16 threads (`host_cores` in the line: what the box offers).
    for key, omp, smp in (("one_thread", False, sample), ("openmp", True, max(8, sample // 16 // 8 * 8))):
        orc = O.Oracle(openmp=omp)
        if omp:
            # SURVEY 8(d) says "all host cores"; this loop forks once per body j (262 144 fork/joins per pass), so MORE threads are SLOWER:
            # 64 threads on a 256-core box took 17 s for 1/16 of the sample (round 6).  16 threads -- a one-GPU share of this pool -- and
            # `host_cores` in the line says what the box had.
            orc.set_num_threads(min(16, os.cpu_count() or 1))
        ms = orc.benchmark_partial(pos_h, smp)
        base[key] = {"value": smp * float(n) / (ms * 1e-3), "cores": orc.num_threads() if omp else 1, "ms": ms, "sample_bodies_i": smp}
    p0, v0 = orc1.startup_state(1024, np.float32)
    ms0 = orc1.benchmark(p0, v0, np.float32(0.016), 100)
    return {
        "value": base["one_thread"]["value"],
        "unit": "interactions/s",
        "cores": 1,
        "kind": "port",
        "host_cores": os.cpu_count(),  # what the box offers this process (the OpenMP leg uses at most 16 of them: see above; `cores` = threads of the headline figure)
        "sample": f"force pass of BodySystemCPU::update (oracle/ port) for the first {sample} bodies i against all {n} bodies j = {sample * n:.3g} "
                  f"interactions; 1 thread is how the reference ships",
        "openmp": base["openmp"],
        "config0": {"what": "BASELINE configs[0]: 1024 bodies, fp32, 100 steps of the CPU path, no warm-up (compute_cpu.cpp:72-88), 1 thread",
                    "ms_total": float(f"{ms0:.5g}"), "interactions_per_s": 1024.0 * 1024.0 * 100 / (ms0 * 1e-3), "gflops": 20 * 1024.0 * 1024.0 * 100 / (ms0 * 1e-3) * 1e-9},
    }


def single_gpu_reference(pkg, n, dtype, mode, pos0, vel0, steps=10, warm_s=0.25):
    """The single-GPU step of the same system on THIS rank's GPU (nb_integrate_ws_* with its own workspace, the null stream), timed with
    events after `warm_s` of untimed steps: the N = 1 figure an N > 1 line carries along."""
    system = pkg.BodySystemHIP(n, 256, pkg.NBodyParams(), dtype, pos0, vel0, mode=mode, workspace=True)
    try:
        dt = np.dtype(dtype).type(np.float32(0.016))
        until = time.perf_counter() + warm_s
        while time.perf_counter() < until:
            for _ in range(4):
                system.update(dt)
            system.synchronize()
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(None)
        for _ in range(steps):
            system.update(dt)
        e1.record(None)
        e1.synchronize()
        ms = e0.elapsed_ms(e1) / steps
    finally:
        system.free()
    return {"what": "rank 0 alone on its GPU, the whole system, nb_integrate_ws_* (events, after the timed region; the other ranks wait)", "steps": steps,
            "ms_per_step": float(f"{ms:.5g}"), "value": float(n) * n / (ms * 1e-3)}


CONFIG3_BODIES = 1048576  # BASELINE.json configs[3]: 1 048 576 bodies over the GPUs of one node


def multi_gpu_diagnostics(pkg, lib, dist, torch, args, capi_rank, system, sharded, launch, fence, step, finish, lend, stream_ptr, rank, world, n, dtype, mode, dt, damping,
                          bufs, vel_t, acc_t, work_t, work_bytes, big, dev):
    """The same job timed other ways, by every rank together (max over ranks, ms per step): the other exchange grouping, the
    one-sided tile schedule, the exchange legs alone, the kernels alone; and BASELINE configs[3] through the same entry points.
    Returns a dict for the JSON line."""
    def timed(fn, reps):
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        finish()
        fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(f"{float(t.item()) / reps * 1e3:.5g}")

    by_rank = {}

    def stream_timed(fn, reps, label=None):
        """this rank's stream time (HIP events), max over ranks: for work that involves no other rank (`label`: every rank's own
        figure goes into out["by_rank"][label], so that "exchange cost = step - kernels" can be read per rank)"""
        fence()
        e0, e1 = pkg.Event(), pkg.Event()
        e0.record(stream_ptr)
        for _ in range(reps):
            fn()
        e1.record(stream_ptr)
        e1.synchronize()
        own = e0.elapsed_ms(e1) / reps
        t = torch.tensor([own], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if label is not None:
            every = [None] * world
            dist.all_gather_object(every, float(f"{own:.5g}"))
            by_rank[label] = every
        fence()
        return float(f"{float(t.item()):.5g}")

    reps = max(2, min(args.steps, 10))
    out = {"what": "ms per step, max over ranks, taken after the timed region; not part of `value`", "reps": reps}
    if capi_rank is None:
        out["exchange_alone_ms"] = timed(lambda: system.exchange_once(system.positions()), reps)
        i0, ni, schedule = system.i0, system.ni, system.schedule

        def kernels():
            for k, (j0, nj, _) in enumerate(schedule):
                launch(system.pos[1 - system.read], system.pos[system.read], system.vel, system.acc, i0, ni, j0, nj, (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(schedule) - 1 else 0))

        out["one_sided_kernels_alone_ms"] = stream_timed(kernels, reps)
        return out

    was_one_group, was_pairwise = capi_rank.exchange_grouping(), capi_rank.pairwise()
    label = lambda pw, og: ("pairwise" if pw else "one_sided") + ("_one_group" if og else "_group_per_round")  # noqa: E731
    steps = {}
    # (1) the step as timed in the headline, then with the other grouping of the position rounds
    for og in (was_one_group, not was_one_group):
        capi_rank.set_exchange_grouping(og)
        steps[label(was_pairwise, og)] = timed(step, reps)
    capi_rank.set_exchange_grouping(was_one_group)
    # (2) the exchange legs on their own
    out["position_exchange_alone_ms"] = {}
    for og in (True, False):
        capi_rank.set_exchange_grouping(og)
        out["position_exchange_alone_ms"]["one_group" if og else "group_per_round"] = timed(lambda: capi_rank.exchange_once(), reps)
    capi_rank.set_exchange_grouping(was_one_group)
    if was_pairwise:
        out["reaction_exchange_alone_ms"] = timed(capi_rank.reaction_exchange_once, reps)
        # (3) this rank's kernels alone: exactly what it launches in a pairwise step, no exchange, no waits
        emulate = lib.nb_emulate_pair_rank_f32 if np.dtype(dtype) == np.float32 else lib.nb_emulate_pair_rank_f64
        need = ctypes.c_size_t(work_bytes)
        r = capi_rank.read
        out["pairwise_kernels_alone_ms"] = stream_timed(lambda: pkg.check(emulate(bufs[1 - r].data_ptr(), bufs[r].data_ptr(), vel_t.data_ptr(), work_t.data_ptr(), ctypes.byref(need), n, world, rank, dt, damping,
                                                                                 stream_ptr), "nb_emulate_pair_rank"), reps, "pairwise_kernels_alone_ms")
    i0, ni = sharded.slice_of(rank, world, n)
    schedule = sharded.tile_schedule(rank, world, n, mode == pkg.NB_MODE_STRICT)

    def tile_kernels():
        r = capi_rank.read
        for k, (j0, nj, _) in enumerate(schedule):
            launch(bufs[1 - r], bufs[r], vel_t, acc_t, i0, ni, j0, nj, (pkg.NB_SHARD_ACC_IN if k else 0) | (pkg.NB_SHARD_FINALIZE if k == len(schedule) - 1 else 0))

    out["one_sided_kernels_alone_ms"] = stream_timed(tile_kernels, reps, "one_sided_kernels_alone_ms")
    out["by_rank"] = by_rank
    # (3b) the order of round 4 -- the diagonal as ONE launch, first -- against this round's (second half of it last, under which the
    # last reaction sums travel): a process-global plan setting, so every rank flips it and the communicator re-agrees its layout
    if was_pairwise:
        try:
            pkg.check(lib.nb_set_late_diagonal(0), "nb_set_late_diagonal")
            capi_rank.set_workspace(work_t.data_ptr(), work_bytes)
            if capi_rank.pairwise():  # (the one-launch form wants FEWER planes: what was lent suffices)
                steps["pairwise_diagonal_first_" + ("one_group" if was_one_group else "group_per_round")] = timed(step, reps)
        finally:
            pkg.check(lib.nb_set_late_diagonal(1), "nb_set_late_diagonal")
            capi_rank.set_workspace(work_t.data_ptr(), work_bytes)
        assert capi_rank.pairwise()
        # (3c) round 6: BOTH compute streams ending on local work (nb_set_late_diagonal(2): the second stream's last rectangle cut, half of the
        # late diagonal as its last kernel) -- slower on one GPU (profiles/round6_cut_rectangle_ab.txt), timed here on real links.  It wants a
        # few planes more than the shipping order: every rank lends a second, larger workspace for the duration (the call is collective)
        bigger = None
        try:
            pkg.check(lib.nb_set_late_diagonal(2), "nb_set_late_diagonal")
            need = capi_rank.workspace_bytes()
            bigger = lend(need) if need > work_bytes else work_t
            capi_rank.set_workspace(bigger.data_ptr() if bigger is not None else None, need if bigger is not None else 0)
            if capi_rank.pairwise():
                steps["pairwise_both_streams_end_on_local_work_" + ("one_group" if was_one_group else "group_per_round")] = timed(step, reps)
        finally:
            pkg.check(lib.nb_set_late_diagonal(1), "nb_set_late_diagonal")
            capi_rank.set_workspace(work_t.data_ptr(), work_bytes)
            torch.cuda.synchronize()
            del bigger
        assert capi_rank.pairwise()
    # (4) the other layout: every rank takes its workspace back (the call is collective) -> the one-sided tile schedule
    if was_pairwise:
        capi_rank.set_workspace(None, 0)
        assert not capi_rank.pairwise()
        for og in (True, False):
            capi_rank.set_exchange_grouping(og)
            steps[label(False, og)] = timed(step, reps)
        capi_rank.set_exchange_grouping(was_one_group)
        capi_rank.set_workspace(work_t.data_ptr(), work_bytes)
        assert capi_rank.pairwise()
    out["step_ms"] = steps
    out["headline_was"] = label(was_pairwise, was_one_group)
    # (5) BASELINE configs[3]: 1 048 576 bodies over the ranks, same communicator, same entry points
    if big is not None:
        pos_b, vel_b = big
        nb = pos_b.size // 4
        p0 = torch.from_numpy(pos_b.reshape(nb, 4)).to(dev)
        b_bufs, b_vel, b_acc = [p0, p0.clone()], torch.from_numpy(vel_b.reshape(nb, 4)).to(dev), torch.zeros_like(p0)
        job = pkg.ShardedRank(None, world, rank, [b.data_ptr() for b in b_bufs], b_vel.data_ptr(), b_acc.data_ptr(), nb, np.float32, mode, 256, stream_ptr, comm=capi_rank.comm)
        b_work = lend(job.workspace_bytes()) if was_pairwise else None
        job.set_workspace(b_work.data_ptr() if b_work is not None else None, b_work.numel() if b_work is not None else 0)
        job.exchange_once(0)  # (as for the headline system: every buffer the transport will see goes through it once before anything is timed)
        job.exchange_once(1)
        if job.pairwise():
            job.reaction_exchange_once()

        def big_fence():
            job.finish()
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

        big_step = lambda: job.update(np.float32(dt), np.float32(damping))  # noqa: E731
        big_step()
        big_fence()
        k = 3
        t0 = time.perf_counter()
        for _ in range(k):
            big_step()
        big_fence()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item()) / k * 1e3
        out["configs"] = [{"workload": "configs[3]" if nb == CONFIG3_BODIES else f"configs[3]'s shape at {nb} bodies (rehearsal)", "bodies": nb, "n_gpus": world, "dtype": "f32", "mode": "fast",
                           "layout": "pairwise across ranks" if job.pairwise() else "one-sided tiles", "steps": k, "ms_per_step": float(f"{ms:.5g}"),
                           "interactions_per_s": float(nb) * nb / (ms * 1e-3), "frac_algorithmic": round(20 * float(nb) * nb / (ms * 1e-3) / world / (FP32_VECTOR_PEAK_TFLOPS * 1e12), 4),
                           "workspace_bytes_per_rank": b_work.numel() if b_work is not None else 0}]
        # hand the communicator back to the headline system
        capi_rank.set_workspace(work_t.data_ptr() if work_t is not None else None, work_bytes)
        job.destroy()
    return out


