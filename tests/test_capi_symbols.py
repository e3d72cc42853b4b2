"""CPU test: libnbody_hip.so loads and exports EXACTLY the symbols include/nbody_hip.h (the drop-in boundary) and
include/nbody_hip_tuning.h (process-global tuning / introspection hooks) declare; libnbody_hip_lab.so -- the same object files plus
csrc/nbody_comm_lab.hip -- exports those and include/nbody_hip_lab.h (the lab bench) on top (no compute calls)."""
import os
import re
import subprocess
import sys

from conftest import ROOT


def declared_symbols(header="nbody_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"NB_API\s+[\w\s\*]+?\b(nb_\w+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_symbols()
    for must in ("nb_integrate_f32", "nb_integrate_f64", "nb_set_softening_sq_f32", "nb_set_softening_sq_f64",
                 "nb_integrate_shard_f32", "nb_alloc", "nb_h2d", "nb_d2h", "nb_event_elapsed_ms", "nb_device_info"):
        assert must in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()  # raises if the .so is missing: there is no fallback
    names = declared_symbols()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the ctypes signature table covers the whole header, nothing more
    assert sorted(pkg.SIGNATURES) == names


def exported_symbols(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if " T " in line)


def test_tuning_header_is_separate_and_exported(pkg):
    """Plan overrides, the one-rank projection hook, the probe event, the memory budget, the LDS opt-in counter and what a communicator
    says about its last step live in their own header: the boundary header declares none of it, the library exports all of it, and
    the binding's second table covers it exactly."""
    lib = pkg.lib()
    tuning = declared_symbols("nbody_hip_tuning.h")
    boundary = declared_symbols()
    assert tuning and not set(tuning) & set(boundary)
    for hook in ("nb_set_plan_override", "nb_set_pair_plan_override", "nb_comm_set_pair_min_slice", "nb_emulate_pair_rank_f32", "nb_lds_optin_count", "nb_comm_last_enqueue_ms"):
        assert hook in tuning and hook not in boundary
    assert not [n for n in tuning if not hasattr(lib, n)]
    assert sorted(pkg.TUNING_SIGNATURES) == tuning
    text = open(os.path.join(ROOT, "include", "nbody_hip_tuning.h")).read()
    assert "PROCESS-GLOBAL" in text and "NOT THREAD-SAFE" in text


def test_product_library_exports_the_two_headers_and_nothing_else(pkg):
    """Round 6: the lab bench (real-RCCL self-test, loopback rank, in-process world, stream-placement A/B hook, allocation-failure hook)
    is NOT in libnbody_hip.so any more -- a host links no test scaffolding.  The product's dynamic symbol table is exactly
    nbody_hip.h + nbody_hip_tuning.h; the lab library's is that plus nbody_hip_lab.h; the lab header repeats nothing."""
    boundary, tuning, lab = declared_symbols(), declared_symbols("nbody_hip_tuning.h"), declared_symbols("nbody_hip_lab.h")
    assert lab and not set(lab) & (set(boundary) | set(tuning))
    for hook in ("nb_comm_selftest_open", "nb_comm_selftest_f32", "nb_comm_self_transfer_f32", "nb_comm_loopback_open", "nb_comm_inprocess_open_all",
                 "nb_comm_replace_side_stream", "nb_set_alloc_limit"):
        assert hook in lab
    assert sorted(pkg.LAB_SIGNATURES) == lab
    assert exported_symbols(pkg.LIB_PATH) == sorted(boundary + tuning)
    assert exported_symbols(pkg.LAB_LIB_PATH) == sorted(boundary + tuning + lab)
    assert len(exported_symbols(pkg.LIB_PATH)) < 98  # (what round 5's library exported)
    # the product's comm translation unit does not even mention the lab's worlds
    comm = open(os.path.join(ROOT, "cuda-nbody_amd", "csrc", "nbody_comm.hip")).read()
    for word in ("inprocess", "nb_comm_selftest", "nb_comm_loopback_open", "nb_set_alloc_limit"):
        assert word not in comm, word
    # a process works with one of the two: after the product library is in, the binding refuses to switch
    pkg.lib()
    if not pkg.is_lab():
        try:
            pkg.use_lab()
        except RuntimeError:
            pass
        else:
            raise AssertionError("use_lab() after lib() must refuse")


def test_host_side_argument_errors_need_no_gpu(pkg):
    lib = pkg.lib()
    assert lib.nb_error_string(0) == b"success"
    assert lib.nb_error_string(10001) == b"NB_ERR_INVALID_ARGUMENT"
    assert lib.nb_set_plan_override(3, 0, 0) == 10001
    assert lib.nb_set_plan_override(0, 0, 0) == 0
    # null / zero-size arguments are rejected on the host before any HIP call
    assert lib.nb_integrate_f32(None, None, None, 0.016, 1.0, 0, 256, 1, None) == 10001
    assert lib.nb_integrate_f32(None, None, None, 0.016, 1.0, 1024, 256, 1, None) == 10001
    assert lib.nb_device_count(None) == 10001
    assert b"gfx950" in lib.nb_version()


def test_round5_hooks_reject_bad_arguments_on_the_host(pkg):
    """What a communicator says about its last step (tuning header) checks its arguments before it touches HIP or RCCL: no GPU
    needed, nothing loaded."""
    import ctypes

    lib = pkg.lib()
    word = ctypes.c_int(0)
    evals = ctypes.c_ulonglong(0)
    ms = ctypes.c_double(0)
    text = ctypes.create_string_buffer(64)
    assert lib.nb_comm_transport_info(None, ctypes.byref(word), text, len(text)) == 10001
    assert lib.nb_comm_pair_work_f32(None, 262144, ctypes.byref(evals), ctypes.byref(word)) == 10001
    assert lib.nb_comm_last_step_trace(None, text, len(text)) == 10001
    assert lib.nb_comm_last_enqueue_ms(None, ctypes.byref(ms)) == 10001
    assert lib.nb_comm_side_stream_collisions(None, ctypes.byref(word)) == 10001 and lib.nb_comm_settle_side_stream(None, None) == 10001
    assert lib.nb_set_late_diagonal(-1) == 10001
    # round 6: the in-kernel clock words -- a pointer without a size (or the reverse) and a misaligned pointer are argument errors; NULL, 0 takes them back
    assert lib.nb_set_pair_clock_words(None, 16) == 10001 and lib.nb_set_pair_clock_words(ctypes.c_void_p(0x1000), 0) == 10001
    assert lib.nb_set_pair_clock_words(ctypes.c_void_p(0x1004), 64) == 10001 and lib.nb_set_pair_clock_words(None, 0) == 0


LAB_ARGUMENT_CHECKS = """
import ctypes, sys
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as entry
pkg = entry.load_package()
pkg.use_lab()
lib = pkg.lib()
assert pkg.is_lab() and lib._name.endswith("libnbody_hip_lab.so")
comm, report = ctypes.c_void_p(), pkg.CommSelftest()
text = ctypes.create_string_buffer(64)
assert lib.nb_comm_selftest_open(None, None) == 10001 and lib.nb_comm_selftest_open(ctypes.byref(comm), None) == 10001
assert lib.nb_comm_loopback_open(ctypes.byref(comm), text, 1, 0) == 10001   # a nominal world of one is no world
assert lib.nb_comm_loopback_open(ctypes.byref(comm), None, 8, 4) == 10001   # no id
assert lib.nb_comm_loopback_open(ctypes.byref(comm), text, 8, 8) == 10001   # no such rank
assert comm.value is None
assert lib.nb_comm_inprocess_open_all(None, 8, text) == 10001 and lib.nb_comm_inprocess_open_all(ctypes.byref(comm), 1, text) == 10001
assert lib.nb_comm_selftest_f32(None, 1024, None, ctypes.byref(report)) == 10001
assert lib.nb_comm_self_transfer_f32(None, None, None, 0, 0, 1, None, None, None) == 10001
assert lib.nb_comm_replace_side_stream(None) == 10001
assert lib.nb_set_alloc_limit(0) == 0
assert lib.nb_error_string(10001) == b"NB_ERR_INVALID_ARGUMENT" and lib.nb_set_plan_override(0, 0, 0) == 0   # (the product's own exports are all there too)
print("lab ok")
"""


def test_lab_hooks_reject_bad_arguments_on_the_host():
    """The lab header's entry points (libnbody_hip_lab.so), in a process of their own: a process loads one of the two libraries."""
    env = {k: v for k, v in os.environ.items() if k != "NBODY_HIP_LAB"}
    done = subprocess.run([sys.executable, "-c", LAB_ARGUMENT_CHECKS, ROOT], capture_output=True, text=True, timeout=120, env=env)
    assert done.returncode == 0 and "lab ok" in done.stdout, done.stderr[-2000:]


def test_step_crew_under_thread_sanitizer(tmp_path):
    """The crew of threads that enqueues the local ranks of a multi-GPU step (csrc/step_crew.h: plain C++, so that this can run on a host
    without a GPU) under ThreadSanitizer: 20 000 jobs of varying size on eight threads -- every k of a job exactly once, no data race on
    the job's description or on memory handed through it, the first failure reported, sleeping workers woken, destruction joins.  The
    test has teeth: with a worker that answers only the tickets of jobs it takes part in (the crew's first form) it reports data races
    and a function run twice."""
    exe = tmp_path / "step_crew_tsan"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", os.path.join(ROOT, "tests", "step_crew_tsan.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe), "20000"], capture_output=True, text=True, timeout=300, env={**os.environ, "TSAN_OPTIONS": "halt_on_error=1"})
    assert run.returncode == 0 and "step crew ok" in run.stdout and "ThreadSanitizer" not in run.stderr, (run.stdout[-500:], run.stderr[-3000:])
    comm = open(os.path.join(ROOT, "cuda-nbody_amd", "csrc", "nbody_comm.hip")).read()
    assert '#include "step_crew.h"' in comm and "class StepCrew" not in comm  # (the product uses THIS class, not a copy of it)


def test_strict_translation_unit_has_no_fused_multiply_add():
    """The strict kernels must keep separate mul/add (bit-parity with the CPU path): the only v_fma in that
    object are inside the IEEE divide/sqrt expansions, never a contracted a*b+c of ours.  Checked structurally:
    the Makefile passes -ffp-contract=off to exactly that TU."""
    mk = open(os.path.join(ROOT, "cuda-nbody_amd", "csrc", "Makefile")).read()
    rule = re.search(r"nbody_strict\.o:.*?\n\t(.*)\n", mk).group(1)
    assert "-ffp-contract=off" in rule
    fast_rule = re.search(r"^%\.o: %\.hip.*?\n\t(.*)\n", mk, re.M).group(1)  # (every other object: the pattern rule)
    assert "-ffp-contract=off" not in fast_rule and "-ffp-contract" not in re.search(r"^COMMON\s*:=(.*)$", mk, re.M).group(1)


def test_launch_plan_heuristics_without_gpu(pkg):
    """nb_plan_* is pure host logic (256 CUs assumed when no device is visible): the geometry each BASELINE size and
    each multi-GPU shard of 262 144 bodies gets."""
    import numpy as np

    def plan(i, j, dtype=np.float32):
        p = pkg.plan(i, j, dtype)
        return p.bodies_per_lane, p.lanes_per_body, p.tile_bodies, p.block_threads, p.grid_blocks

    # full-size fp32 systems: 8 waves per 512-thread workgroup (two workgroups per CU), 4 bodies (2 packed pairs) per lane,
    # 256 bodies j per wave and chunk (round 4: 1-3 % over 128 at every size of the sweep)
    assert plan(262144, 262144) == (4, 8, 2048, 512, 1024)
    assert plan(65536, 65536) == (4, 16, 2048, 1024, 256)  # one workgroup per CU: 1024 threads, so that every SIMD still holds four waves
    assert plan(1048576, 1048576) == (4, 8, 2048, 512, 4096)
    # strong-scaling shards of 262 144 bodies on 2 / 4 / 8 GPUs keep whole rounds of 256 workgroups
    assert plan(131072, 262144)[:2] + plan(131072, 262144)[4:] == (4, 16, 512)
    assert plan(65536, 262144)[:2] + plan(65536, 262144)[4:] == (4, 16, 256)
    assert plan(32768, 262144)[:2] + plan(32768, 262144)[4:] == (2, 16, 256)
    # small or awkward sizes: wave-split layout (lanes_per_body == 64).  Its launch costs max over CUs of the waves on it (round 4):
    # whole rounds of 16-wave workgroups where the bodies allow (8 192 bodies = 256 workgroups of 32), smaller workgroups where one
    # more 16-wave workgroup would put 32 waves on a CU (8 193 bodies: 1 025 workgroups of 4 waves, 30 us instead of 41)
    for n, block in ((1, 256), (1024, 256), (2048, 256), (4096, 512), (8192, 1024), (8193, 256), (12000, 512), (16384, 1024), (40960, 512)):
        assert plan(n, n)[1] == 64 and plan(n, n)[3] == block, (n, plan(n, n))
        assert plan(n, n)[4] == -(-n // (plan(n, n)[0] * block // 64))
    # between the powers of two both layouts are held to the same estimate (round 4: the rule before lost up to 45 %): 18 000 bodies
    # are 2 250 wave-split workgroups of 8 bodies, 50 000 bodies 196 tile-layout workgroups of 256 (782 wave-split ones of 64
    # would be four rounds for the work of 3.05)
    assert plan(18000, 18000) == (2, 64, 1024, 256, 2250) and plan(50000, 50000)[:2] == (4, 16) and plan(50000, 50000)[4] == 196
    # fp64: one body per vector, up to 4 per lane
    assert plan(262144, 262144, np.float64) == (4, 8, 2048, 512, 1024)
    assert plan(1024, 1024, np.float64)[1] == 64
    # overrides are validated and reversible
    pkg.set_plan_override(2, 4, 512)
    try:
        assert plan(262144, 262144)[:4] == (2, 4, 512, 256)
    finally:
        pkg.set_plan_override(0, 0, 0)
    assert plan(262144, 262144)[1] == 8


def test_inline_asm_never_consumes_a_transcendental_result_directly():
    """gfx950 needs a wait state between a transcendental op (v_rsq_f32 ...) and a VALU op that reads its result; hipcc's
    hazard pass inserts it but does not look inside inline asm.  The FAST kernels carry ONE inline instruction (the packed
    multiply that broadcasts the mass from the high half of {z, m}); its operands must therefore never be the direct output
    of a v_rsq/v_rcp (it multiplies inv^3, the output of an ordinary v_pk_mul).  Checked on the compiled ISA of every
    instantiation: no inline v_pk_mul may have a transcendental writing one of its source registers among the two
    instructions before it."""
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, capture_output=True)
    lines = open(os.path.join(csrc, "nbody_fast.s")).read().split("\n")
    inline = 0
    for i, line in enumerate(lines):
        if "v_pk_mul_f32" in line and "op_sel:[1,0] op_sel_hi:[1,1]" in line:
            inline += 1
            m = re.search(r"v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]", line)
            sources = {int(m.group(k)) for k in (3, 4, 5, 6)}
            before = [x.strip() for x in lines[max(0, i - 8):i] if x.strip() and not x.strip().startswith(";")][-2:]
            for prev in before:
                t = re.match(r"v_(rsq|rcp|sqrt|exp|log|sin|cos)_f32(_e32|_e64)? v(\d+),", prev)
                assert not (t and int(t.group(3)) in sources), (i, prev, line)
    assert inline > 0  # the instruction is there (otherwise this test checks nothing)


def test_production_inner_loops_keep_their_instruction_mix():
    """A guard against a silent regression of the code generator: the headline kernel (fp32, 4 bodies i per lane, 8 waves per
    workgroup) must keep streaming loops of exactly 11 (unit / one-species chunks) and 12 (mixed masses) packed ops + 2 v_rsq per
    interaction pair, with the bodies j in scalar registers (s_load, no ds_* and no v_mov in the loop, no hazard s_nop), <= 128
    VGPRs and no scratch; and no FAST kernel at all may use scratch."""
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, capture_output=True)
    lines = open(os.path.join(csrc, "nbody_fast.s")).read().split("\n")
    kernel = "_ZN2nb12_GLOBAL__N_121integrate_bodies_fastIfLi2ELi8ELi2EEEvNS_5ShardIT_EE"
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    mixes = []
    for i in range(start, end):
        if "Inner Loop Header" not in lines[i]:
            continue
        label = lines[i - 1].split(":")[0].strip()
        stop = next((k for k in range(i, end) if ("s_cbranch" in lines[k] or "s_branch" in lines[k]) and label in lines[k]), None)
        if stop is None:
            continue
        body = [l.strip() for l in lines[i + 1:stop]]
        count = lambda prefix: sum(1 for l in body if l.startswith(prefix))
        if count("v_rsq_f32") == 32:  # a streaming loop: two groups of 4 bodies j x 2 packed pairs of bodies i
            mixes.append((count("v_pk_"), count("ds_"), count("v_mov"), count("s_nop"), count("s_load"),
                          sum(1 for l in body if l.startswith("v_") and not l.startswith(("v_pk_", "v_rsq_f32")))))
    assert sorted(m[0] for m in mixes) == [176, 176, 192], mixes  # unit, one-species, mixed
    for packed, lds, moves, nops, loads, other_valu in mixes:
        assert lds == 0 and moves == 0 and nops == 0 and other_valu == 0 and loads >= 2, mixes
    tail = "\n".join(lines[end:end + 60])
    assert int(re.search(r"; NumVgprs: (\d+)", tail).group(1)) <= 128
    assert int(re.search(r"; ScratchSize: (\d+)", tail).group(1)) == 0
    # no FAST kernel spills to scratch
    sizes = [int(m) for m in re.findall(r"; ScratchSize: (\d+)", "\n".join(lines))]
    assert sizes and max(sizes) == 0, max(sizes)


def test_strict_fast_form_loops_keep_their_instruction_mix():
    """The same guard for the bit-reproducing kernel: per four packed pairs of bodies j the fp32 fast form issues 80 packed ops
    (unit masses) or 92 (any masses) + 16 transcendentals + 24 adds -- the exhaustively verified sqrt / reciprocal / divide
    sequences and nothing else -- without scratch."""
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, capture_output=True)
    lines = open(os.path.join(csrc, "nbody_strict.s")).read().split("\n")
    kernel = "_ZN2nb12_GLOBAL__N_123integrate_bodies_strictIfEEvNS_5ShardIT_EE"
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    packed = []
    for i in range(start, end):
        if "Inner Loop Header" not in lines[i]:
            continue
        label = lines[i - 1].split(":")[0].strip()
        stop = next((k for k in range(i, end) if ("s_cbranch" in lines[k] or "s_branch" in lines[k]) and label in lines[k]), None)
        if stop is None:
            continue
        body = [l.strip() for l in lines[i + 1:stop]]
        count = lambda prefix: sum(1 for l in body if l.startswith(prefix))
        if count("v_rsq_f32") == 8 and count("v_rcp_f32") == 8:  # the U = 4 pair loops
            assert count("v_add_f32") == 24 and count("v_mov") == 0 and count("scratch_") == 0, body
            packed.append(count("v_pk_"))
    assert sorted(packed) == [80, 92], packed
    assert int(re.search(r"; ScratchSize: (\d+)", "\n".join(lines[end:end + 60])).group(1)) == 0


def test_pair_plan_heuristics_without_gpu(pkg):
    """nb_pair_plan_* / nb_workspace_bytes_* are pure host logic: where the pairwise layout applies, the geometry the
    BASELINE sizes get (measured: profiles/round3_pair_crossover_*.jsonl), and the workspace formula
    (workgroups per block + reaction slots) x 3 x padded bodies x sizeof(T)."""
    import ctypes

    import numpy as np

    def plan(n, dtype=np.float32):
        p = pkg.pair_plan(n, dtype)
        assert p.block_bodies == 64 * p.bodies_per_lane and p.blocks == -(-n // p.block_bodies) and p.grid_blocks == p.blocks * p.splits
        assert p.reaction_slots == (0 if p.blocks < 2 else (p.blocks // 2 if p.blocks % 2 else p.blocks // 2 - 1))
        assert p.workspace_bytes == (p.splits + p.reaction_slots) * 3 * p.blocks * p.block_bodies * np.dtype(dtype).itemsize
        return p.applies, p.bodies_per_lane, p.waves_per_block, p.splits, p.blocks

    assert plan(262144) == (1, 16, 8, 1, 256)      # the headline (round 4: sixteen bodies i per lane): 256 workgroups of 8 waves, one per CU, 127 reaction slots
    assert plan(1048576) == (1, 16, 8, 1, 1024)
    assert plan(65536) == (1, 16, 8, 4, 64)        # four workgroups share a block of bodies i and split its 528 units: 16.5 per wave, interleaved so that every SIMD gets 33
    assert plan(131072) == (1, 16, 8, 2, 128) and plan(32768) == (1, 8, 8, 4, 64)
    assert plan(16384) == (1, 4, 8, 4, 64)         # small systems: half the bodies per lane, twice the blocks
    assert plan(8192)[0] == 0 and plan(10240)[0] == 1
    assert plan(262144, np.float64) == (1, 8, 8, 1, 512)
    assert plan(4096, np.float64)[0] == 0 and plan(6144, np.float64)[0] == 0 and plan(6145, np.float64)[0] == 1
    assert plan(600, np.float32)[4] == 5 and plan(64, np.float32)[4] == 1  # odd block counts (blocks of 128 bodies: R = 1), a single block
    assert plan(9000) == (1, 2, 8, 3, 71)          # 8 200-10 500 bodies: two bodies i per lane, three workgroups per block of 128: 213 workgroups, one round
    need = ctypes.c_size_t(7)
    lib = pkg.lib()
    assert lib.nb_workspace_bytes_f32(262144, pkg.NB_MODE_FAST, ctypes.byref(need)) == 0 and need.value == 128 * 3 * 262144 * 4
    assert lib.nb_workspace_bytes_f32(262144, pkg.NB_MODE_STRICT, ctypes.byref(need)) == 0 and need.value == 0
    assert lib.nb_workspace_bytes_f32(4096, pkg.NB_MODE_FAST, ctypes.byref(need)) == 0 and need.value == 0
    assert lib.nb_workspace_bytes_f32(262144, pkg.NB_MODE_FAST, None) == 10001
    assert lib.nb_set_pair_plan_override(3, 8, 1, 0) == 10001 and lib.nb_set_pair_plan_override(4, 8, 65, 0) == 10001
    pkg.set_pair_plan_override(2, 16, 3, 1)
    try:
        assert plan(20000) == (1, 4, 16, 3, 79)
    finally:
        pkg.set_pair_plan_override(0, 0, 0, 0)


def test_pairwise_inner_loops_keep_their_instruction_mix():
    """The rotation loops of the pairwise headline kernel (fp32, 4 packed pairs of bodies i per lane), four steps per trip:
    per step 4 x (14 v_pk_* + 2 v_rsq_f32) + 9 v_mov_b32_dpp wave_ror:1 (one species on both sides; up to 16 + 2 and 10 moves with masses), nothing
    else on the vector unit but a few moves, no scratch access and no LDS inside the loops, <= 128 VGPRs."""
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, capture_output=True)
    lines = open(os.path.join(csrc, "nbody_pair.s")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"_ZN2nb\S*pair_forcesIfLi4ELi8EE\S*:", l))  # pair_forces<float, 4, 8>
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    mixes = []
    for i in range(start, end):
        if "Inner Loop Header: Depth=2" not in lines[i]:
            continue
        label = lines[i - 1].split(":")[0].strip()
        stop = next(k for k in range(i, end) if "s_cbranch" in lines[k] and label in lines[k])
        body = [l.strip() for l in lines[i + 1:stop] if l.strip() and not l.strip().startswith(";")]
        count = lambda what: sum(1 for l in body if what in l.split()[0])  # noqa: E731
        rotations = sum(1 for l in body if l.startswith("v_mov_b32_dpp") and "wave_ror:1" in l)
        other = sum(1 for l in body if l.startswith("v_") and not l.startswith(("v_pk_", "v_rsq_f32", "v_mov_b32")))
        mixes.append((count("v_pk_"), count("v_rsq_f32"), rotations, other, count("scratch_"), count("ds_"), count("s_nop")))
    # four loops: no mass multiply (one species on both sides), the bodies i differ in mass (+1 v_pk_mul per pair), the bodies j
    # differ (+1, and their relative mass rotates too), both
    assert sorted(m[:6] for m in mixes) == [(224, 32, 36, 0, 0, 0), (240, 32, 36, 0, 0, 0), (240, 32, 40, 0, 0, 0), (256, 32, 40, 0, 0, 0)], mixes
    assert all(m[6] <= 2 for m in mixes), mixes  # at most a stray hazard s_nop per four steps (296 instructions)
    tail = "\n".join(lines[end:end + 60])
    assert int(re.search(r"; NumVgprs: (\d+)", tail).group(1)) <= 128
    # the headline kernel since round 4: eight packed pairs of bodies i per lane -- per step 8 x (14 v_pk_* + 2 v_rsq_f32) + the same 9
    # rotation moves: 137 vector instructions for 32 directed interactions (4.3 each, against 4.6 with four pairs)
    start = next(i for i, l in enumerate(lines) if re.match(r"_ZN2nb\S*pair_forcesIfLi8ELi8EE\S*:", l))  # pair_forces<float, 8, 8>
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    mixes = []
    for i in range(start, end):
        if "Inner Loop Header: Depth=2" not in lines[i]:
            continue
        label = lines[i - 1].split(":")[0].strip()
        stop = next(k for k in range(i, end) if "s_cbranch" in lines[k] and label in lines[k])
        body = [l.strip() for l in lines[i + 1:stop] if l.strip() and not l.strip().startswith(";")]
        count = lambda what: sum(1 for l in body if what in l.split()[0])  # noqa: E731
        rotations = sum(1 for l in body if l.startswith("v_mov_b32_dpp") and "wave_ror:1" in l)
        other = sum(1 for l in body if l.startswith("v_") and not l.startswith(("v_pk_", "v_rsq_f32", "v_mov_b32")))
        mixes.append((count("v_pk_"), count("v_rsq_f32"), rotations, other, count("scratch_"), count("ds_")))
    assert sorted(mixes) == [(448, 64, 36, 0, 0, 0), (480, 64, 36, 0, 0, 0), (480, 64, 40, 0, 0, 0), (512, 64, 40, 0, 0, 0)], mixes


def test_pairwise_headline_kernels_register_budget():
    """The register budget of the kernels the headline numbers come from (round 4: R = 8 vectors per lane) -- pair_forces<float, 8, 8>
    (65 536 bodies and more, fp32: 256 VGPRs, two waves per SIMD, one 8-wave workgroup per CU), pair_forces<float, 8, 12> (reachable
    through the plan override: 168 VGPRs under launch_bounds(768), three waves per SIMD), pair_forces<double, 8, 8> (fp64: 254 VGPRs, four registers spilled outside the loops) -- and of
    pair_forces<float, 4, 8> (32 768 .. 65 535 bodies, slices and shards under 131 072: 128 VGPRs, four waves per SIMD): the
    occupancy asked for with amdgpu_waves_per_eu is what the compiler delivers, no AGPRs, and the VGPR spills that
    the per-tile prologue and epilogue carry neither grow nor reach the rotation loops: not one scratch_* instruction between a
    depth-2 loop header and its back-edge, in any of the four loops of any of these kernels."""
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, capture_output=True)
    text = open(os.path.join(csrc, "nbody_pair.s")).read()
    lines = text.split("\n")
    for template, vgprs, occupancy, spill_limit, scratch_limit, ops_limit in (("IfLi8ELi8E", 256, 2, 52, 104, 56), ("IfLi8ELi12E", 168, 3, 192, 464, 240), ("IdLi8ELi8E", 254, 2, 6, 16, 6),
                                                                                  ("IfLi4ELi8E", 128, 4, 58, 168, 64)):
        start = next(i for i, l in enumerate(lines) if re.match(r"_ZN2nb\S*pair_forces%sE\S*:" % template, l))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        name = lines[start].split(":")[0]
        loops = 0
        for i in range(start, end):
            if "Inner Loop Header: Depth=2" not in lines[i]:
                continue
            label = lines[i - 1].split(":")[0].strip()
            stop = next(k for k in range(i, end) if "s_cbranch" in lines[k] and label in lines[k])
            body = [l.strip() for l in lines[i + 1:stop] if l.strip() and not l.strip().startswith(";")]
            assert not any(l.startswith(("scratch_", "buffer_load", "buffer_store")) for l in body), (template, label)
            assert sum(1 for l in body if l.startswith("v_mov_b32_dpp") and "wave_ror:1" in l) > 0, (template, label)  # (it IS a rotation loop)
            loops += 1
        assert loops == 4, (template, loops)
        # the tile prologue / epilogue may touch scratch; count what the whole kernel holds so that growth shows
        scratch_ops = sum(1 for l in lines[start:end] if l.strip().startswith("scratch_"))
        assert scratch_ops <= ops_limit, (template, scratch_ops)
        tail = "\n".join(lines[end:end + 60])
        assert int(re.search(r"; NumVgprs: (\d+)", tail).group(1)) == vgprs, template
        assert int(re.search(r"; NumAgprs: (\d+)", tail).group(1)) == 0, template
        assert int(re.search(r"; Occupancy: (\d+)", tail).group(1)) == occupancy, template
        assert int(re.search(r"; ScratchSize: (\d+)", tail).group(1)) <= scratch_limit, template
        meta = re.search(r"  - \.agpr_count:(?:(?!  - \.agpr_count:).)*?\.name: +%s\n.*?\.wavefront_size: +\d+" % re.escape(name), text, re.S).group(0)
        field = lambda key: int(re.search(r"\.%s: +(\d+)" % key, meta).group(1))  # noqa: E731
        assert field("vgpr_count") == vgprs and field("sgpr_spill_count") <= 28, template  # (scalars parked in VGPR lanes outside the loops: v_writelane, no memory; more of them since the work items of a wave are whole units AND quarters)
        assert field("vgpr_spill_count") <= spill_limit, (template, field("vgpr_spill_count"))
        assert field("private_segment_fixed_size") <= scratch_limit, template


def test_pair_shard_plan_without_gpu(pkg):
    """(Slices of 32 768 bodies and more take sixteen bodies i per lane.)  The multi-GPU pairwise plan is host logic too: nb_emulate_pair_rank_* with no workspace only answers how many bytes a
    rank of a G-rank step needs -- (self sets + diagonal slots + two rectangle regions + send + receive planes) x 3 x the padded
    slice -- or says that the pairwise step does not apply (slices under 2 048 bodies, a world of one, bodies that do not shard)."""
    import ctypes

    import numpy as np

    lib = pkg.lib()

    def need(n, world, rank=0, fn=lib.nb_emulate_pair_rank_f32, dt=np.float32(0.016)):
        bytes_ = ctypes.c_size_t(0)
        rc = fn(None, None, None, None, ctypes.byref(bytes_), n, world, rank, dt, dt, None)
        return rc, bytes_.value

    rc, b8 = need(262144, 8)
    # 32 768 bodies per rank: R = 8 -> 32 blocks of 1 024, C = 4 for the diagonal and the rectangles (128 workgroups per launch: two
    # rectangles run at once) -- 8 for the split rectangle at distance 4 as its higher partner runs it (half of its blocks of
    # bodies i) --, H = 4 partners, 15 diagonal slots.  Round 5: the diagonal is two launches (offsets q = 0 .. 8 first, 9 .. 16 last),
    # each with its own C = 4 planes of i-side sums; nb_set_late_diagonal(0) is the single launch of before
    assert rc == 0 and b8 == ((4 + 4 + 3 * 4 + 8) + 15 + 2 * 32 + 4 + 4) * 3 * 32768 * 4
    six = need(262144 // 8 * 6, 6)
    assert lib.nb_set_late_diagonal(3) == 10001 and lib.nb_set_late_diagonal(0) == 0
    try:
        assert need(262144, 8) == (0, ((4 + 3 * 4 + 8) + 15 + 2 * 32 + 4 + 4) * 3 * 32768 * 4)
        # round 6, nb_set_late_diagonal(2): the late offsets 9 .. 16 dealt to BOTH streams (9 .. 12 / 13 .. 16) and rectangle 3 -- the second
        # stream's last -- cut at 4 x 16 = 64 tiles of bodies j: C = 4 planes more for the second stream's diagonal piece, 4 for the cut-off part
        assert lib.nb_set_late_diagonal(2) == 0
        assert need(262144, 8) == (0, ((4 + 4 + 3 * 4 + 8 + 4 + 4) + 15 + 2 * 32 + 4 + 4) * 3 * 32768 * 4)
        assert six[0] == 0 and need(262144 // 8 * 6, 6) == six  # (6 ranks: the split rectangle is the second stream's last -- the shipping deal)
    finally:
        assert lib.nb_set_late_diagonal(1) == 0
    rc, b2 = need(262144, 2)
    assert rc == 0 and b2 > b8
    assert need(262144, 8, fn=lib.nb_emulate_pair_rank_f64, dt=0.016)[0] == 0
    assert need(262144, 1)[0] == 10001          # a world of one is not a sharded step
    assert need(262144, 8, rank=8)[0] == 10001
    assert need(8192, 8)[0] == 10002            # 1 024 bodies per rank: too small a slice
    assert need(262145, 8)[0] == 10002          # does not shard evenly
    assert lib.nb_emulate_pair_rank_f32(None, None, None, None, None, 262144, 8, 0, np.float32(0.016), np.float32(1), None) == 10001


def test_workspace_memory_guard_without_gpu(pkg):
    """No workspace beyond a third of the device's memory is ever asked for, on one GPU and per rank of a multi-GPU step
    (the multi-GPU plan had no such guard: ~16 GB per rank at 1 Mi bodies over 2 ranks).  nb_set_memory_budget
    (tuning header) stands in for the device's memory figure: pure host logic."""
    import ctypes

    import numpy as np

    lib = pkg.lib()

    def single(n):
        need = ctypes.c_size_t(1)
        assert lib.nb_workspace_bytes_f32(n, pkg.NB_MODE_FAST, ctypes.byref(need)) == 0
        return need.value

    def rank(n, world):
        need = ctypes.c_size_t(0)
        rc = lib.nb_emulate_pair_rank_f32(None, None, None, None, ctypes.byref(need), n, world, 0, np.float32(0.016), np.float32(1), None)
        return rc, need.value

    try:
        assert lib.nb_set_memory_budget(0) == 0
        free_single, (rc, free_rank) = single(262144), rank(262144, 8)
        assert free_single == 128 * 3 * 262144 * 4 and rc == 0 and free_rank > 0
        assert lib.nb_set_memory_budget(3 * free_single) == 0          # exactly a third: still fine
        assert single(262144) == free_single
        assert lib.nb_set_memory_budget(3 * free_single - 1) == 0      # one byte less: the tournament is cut into two slices (bounded workspace)
        assert 0 < single(262144) < free_single and pkg.pair_plan(262144).applies == 1 and pkg.pair_plan(262144).slices == 2
        pkg.set_pair_slices_override(1)                                  # ... unless slicing is switched off: then the one-sided kernel, no workspace asked for
        assert single(262144) == 0 and pkg.pair_plan(262144).applies == 0
        pkg.set_pair_slices_override(0)
        assert rank(262144, 8) == (0, free_rank)                        # (a rank of eight needs far less: 54 MiB)
        assert lib.nb_set_memory_budget(3 * free_rank - 1) == 0
        assert rank(262144, 8)[0] == 10002                              # NB_ERR_UNSUPPORTED: the step would be the one-sided tile schedule
        # the case ADVICE raised: 1 Mi bodies over 2 ranks (7.5 GB per rank with sixteen bodies i per lane; ~16 GB in round 3) --
        # refused on a 16 GB device, accepted on a 288 GB one
        assert lib.nb_set_memory_budget(16 << 30) == 0 and rank(1048576, 2)[0] == 10002
        assert lib.nb_set_memory_budget(288 << 30) == 0 and rank(1048576, 2)[0] == 0
    finally:
        lib.nb_set_memory_budget(0)
        pkg.set_pair_slices_override(0)


def test_sliced_pairwise_plan_without_gpu(pkg):
    """The pairwise layout with a bounded workspace (round 4): one tournament over N bodies wants N^2 / (128 I) * 12 B of reaction
    slots -- 206 GB at 4 Mi bodies with eight bodies i per lane.  Cut into K slices that share one region of reaction planes it needs a fraction; the library
    picks the fewest slices that fit a third of the device's memory, a caller's cap, or the workspace a step is handed.  Host logic.
    (Round 4, later: with sixteen bodies i per lane one tournament wants half of that -- 103 GB at 4 Mi bodies.)"""
    import ctypes

    import numpy as np

    lib = pkg.lib()
    n = 4 * 1048576

    def capped(cap, n=n, dtype=np.float32):
        return pkg.workspace_bytes(n, dtype, pkg.NB_MODE_FAST, cap)

    try:
        assert lib.nb_set_memory_budget(192 << 30) == 0
        one = pkg.pair_plan(1048576)
        assert (one.applies, one.slices, one.workspace_bytes) == (1, 1, (1 + 511) * 3 * 1048576 * 4)  # 6.4 GB: affordable, one tournament
        big = pkg.pair_plan(n)
        assert big.applies == 1 and big.slices == 2 and big.workspace_bytes < (64 << 30)  # 103 GB in one piece; two slices fit a third of 192 GB
        assert pkg.workspace_bytes(n) == big.workspace_bytes
        # a caller's cap; more memory never means more slices
        sizes = [capped(c << 30) for c in (4, 8, 16, 32, 64, 128)]
        assert all(a <= b for a, b in zip(sizes, sizes[1:])) and all(0 < b <= (c << 30) for b, c in zip(sizes, (4, 8, 16, 32, 64, 128)))
        assert capped(16 << 30) <= (16 << 30) and capped(16 << 30) > (8 << 30)
        assert capped(1 << 20) == 0  # nothing fits one megabyte: the one-sided kernel
        assert capped(1 << 62) == big.workspace_bytes  # the device's third still binds
        # the formula of the sliced form, checked at a forced K on a size where everything is round: 262 144 bodies in 4 slices of
        # 64 blocks of 1 024: region max(31 diagonal slots, 64 rectangle planes) + per slice the self sets (C = 4 for the diagonal and the
        # full rectangle, 8 for the split one as its higher partner runs it) + (1 + 2) received arrays
        pkg.set_pair_slices_override(4)
        p = pkg.pair_plan(262144)
        assert p.slices == 4 and p.workspace_bytes == (64 + 4 * ((4 + 4 + 8) + 3)) * 3 * 65536 * 4
        assert capped(p.workspace_bytes - 1, 262144) == 0 and capped(p.workspace_bytes, 262144) == p.workspace_bytes
        pkg.set_pair_slices_override(0)
        assert lib.nb_set_pair_slices_override(16) == 10001 and lib.nb_set_pair_slices_override(-1) == 10001
        need = ctypes.c_size_t(1)
        assert lib.nb_workspace_bytes_capped_f64(n, pkg.NB_MODE_STRICT, 1 << 40, ctypes.byref(need)) == 0 and need.value == 0
        assert lib.nb_workspace_bytes_capped_f32(n, pkg.NB_MODE_FAST, 1 << 40, None) == 10001
    finally:
        lib.nb_set_memory_budget(0)
        pkg.set_pair_slices_override(0)


def test_makefile_rebuilds_an_object_when_any_header_it_includes_changes():
    """PairArgs / FinishArgs / Shard cross translation units BY VALUE: an object built against an older nbody_kernels.h launches
    kernels with a shifted argument block (round 4 met that as a GPU memory fault after one new field).  Since round 5 the
    COMPILER writes the dependencies (-MMD -MP, one .d file per object, included by csrc/Makefile): after a build, every header a
    source includes -- directly or through another header of ours -- must be in that object's .d file, and touching a header
    must make `make -q` say the library is out of date."""
    import re
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    with open(os.path.join(csrc, "Makefile")) as fh:
        mk = fh.read()
    assert "-MMD -MP" in mk and re.search(r"^-include \$\(OBJS:\.o=\.d\)$", mk, re.M)
    assert not re.search(r"^\w+\.o:.*\.h\b", mk, re.M), "an object rule lists headers by hand again"
    subprocess.run(["make", "-s", "-j", "8", "-C", csrc], check=True, capture_output=True)

    def includes(path, seen):
        with open(path) as fh:
            for name in re.findall(r'^\s*#include "([^"]+)"', fh.read(), re.M):
                full = os.path.normpath(os.path.join(os.path.dirname(path), name))
                if full not in seen:
                    seen.add(full)
                    includes(full, seen)
        return seen

    sources = sorted(f for f in os.listdir(csrc) if f.endswith(".hip") and f.startswith("nbody_"))
    objects = re.search(r"^OBJS\s*:=\s*(.*)$", mk, re.M).group(1).split() + re.search(r"^LAB_OBJS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    assert re.search(r"^-include \$\(LAB_OBJS:\.o=\.d\)$", mk, re.M)
    assert {s[:-4] + ".o" for s in sources} == set(objects), (sources, objects)
    for src in sources:
        with open(os.path.join(csrc, src[:-4] + ".d")) as fh:
            words = fh.read().replace("\\\n", " ").split()
        listed = {os.path.normpath(os.path.join(csrc, w.rstrip(":"))) for w in words}
        missing = includes(os.path.join(csrc, src), set()) - listed
        assert not missing, f"{src[:-4]}.d does not list {sorted(os.path.relpath(m, csrc) for m in missing)}"
    # and make acts on it: an up-to-date tree, then one header touched
    assert subprocess.run(["make", "-q", "-C", csrc], capture_output=True).returncode == 0
    header = os.path.join(csrc, "nbody_kernels.h")
    stat = os.stat(header)
    try:
        os.utime(header, None)
        assert subprocess.run(["make", "-q", "-C", csrc], capture_output=True).returncode == 1
        would = subprocess.run(["make", "-n", "-C", csrc], capture_output=True, text=True).stdout
        for obj in ("nbody_comm.o", "nbody_comm_lab.o", "nbody_pair.o", "nbody_capi.o", "nbody_fast.o", "nbody_strict.o"):
            assert f"-o {obj}" in would, obj
    finally:
        os.utime(header, ns=(stat.st_atime_ns, stat.st_mtime_ns))
    assert subprocess.run(["make", "-q", "-C", csrc], capture_output=True).returncode == 0


def test_rccl_entry_points_match_the_rccl_header_at_compile_time(tmp_path):
    """csrc/rccl_api.h spells the function-pointer types nbody_comm.hip calls RCCL through (the library binds RCCL with dlsym and
    includes no RCCL header).  `make check-rccl-abi` compiles a host-only unit that includes the image's rccl.h and static_asserts
    that each type IS decltype(&nccl...), that the enums the product passes as int are 4-byte enums with the values it uses, and
    that ncclUniqueId is the 128-byte struct it passes by value.  The check has teeth: with one argument of ncclSend's type
    swapped it no longer compiles."""
    import shutil
    import subprocess

    csrc = os.path.join(ROOT, "cuda-nbody_amd", "csrc")
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        import pytest

        pytest.skip("no RCCL header in this image")
    done = subprocess.run(["make", "-s", "-C", csrc, "check-rccl-abi"], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    with open(os.path.join(csrc, "nbody_comm_internal.h")) as fh:  # (the binding's struct lives in the header the two comm units share)
        product = fh.read()
    assert '#include "rccl_api.h"' in product and "nb_rccl::SendFn" in product and "(*Send)(" not in product  # (no second spelling of the types)
    for name in ("rccl_api.h", "rccl_abi_check.cpp"):
        shutil.copy(os.path.join(csrc, name), tmp_path / name)
    text = (tmp_path / "rccl_api.h").read_text()
    good = "using SendFn           = Result (*)(const void*, size_t, DataType, int, Comm, hipStream_t);"
    assert good in text
    (tmp_path / "rccl_api.h").write_text(text.replace(good, "using SendFn           = Result (*)(const void*, size_t, int, DataType, Comm, hipStream_t);"))
    broken = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "rccl_abi_check.cpp"], cwd=tmp_path, capture_output=True, text=True)
    assert broken.returncode != 0 and "static assertion failed" in broken.stderr


def test_pair_plan_fills_whole_rounds_at_any_body_count(pkg):
    """The geometry search of plan_pair (csrc/nbody_pair.hip): a launch of eight-wave workgroups costs ceil(grid / 256) rounds, so
    at ANY body count the plan must not leave much of its last round empty (the fixed table it replaces put 284 workgroups on
    256 CUs at 36 000 bodies: two rounds for the work of 1.1 -- the sample here never fills its last round under 0.88), must give every wave of a shared block at least two units, and must
    keep the choices measured at the powers of two.  Pure host logic."""
    import numpy as np

    worst = {}
    for dtype in (np.float32, np.float64):
        W = 2 if dtype is np.float32 else 1
        sizes = sorted(set([8193 + 977 * k for k in range(0, 120)] + [100003 + 20011 * k for k in range(0, 100)] + [1 << k for k in range(14, 23)]))
        for n in sizes:
            p = pkg.pair_plan(n, dtype)
            if not p.applies:
                continue
            assert p.waves_per_block == 8 and 1 <= p.splits <= 16 and p.bodies_per_lane // W in ((1, 2, 4, 8) if W == 2 else (2, 4, 8)), (n, p.bodies_per_lane, p.splits)
            units = (p.blocks // 2 + 1) * p.bodies_per_lane
            assert p.splits == 1 or units >= 2 * p.splits * 8, (n, units, p.splits)
            rounds = p.grid_blocks / 256
            fill = rounds / -(-p.grid_blocks // 256)
            if p.grid_blocks > 256:
                assert fill >= 0.88, (n, dtype.__name__, p.bodies_per_lane, p.splits, p.blocks, fill)
                worst[dtype.__name__] = min(worst.get(dtype.__name__, 1.0), fill)
            if n >= 200000:
                assert p.bodies_per_lane // W == 8, (n, p.bodies_per_lane)  # large systems: sixteen (eight) bodies i per lane
    assert worst["float32"] >= 0.88 and worst["float64"] >= 0.88


def test_python_mirror_constants_are_the_headers(pkg):
    """Every `#define NB_<NAME> <integer>` of include/nbody_hip.h that the Python mirror names has the header's value."""
    text = open(os.path.join(ROOT, "include", "nbody_hip.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(NB_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))u?\b", text, re.M)}
    defines.update({m.group(1): int(m.group(2), 0) for m in re.finditer(r"^\s*(NB_[A-Z0-9_]+)\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+))u?\s*,?\s*(?:/\*.*)?$", text, re.M)})  # enumerators
    shared = {name: value for name, value in defines.items() if hasattr(pkg, name)}
    assert {"NB_ERR_INVALID_ARGUMENT", "NB_ERR_UNSUPPORTED", "NB_ERR_RCCL_BASE", "NB_ERR_OUT_OF_MEMORY", "NB_MODE_FAST", "NB_MODE_STRICT"} <= set(shared), sorted(shared)
    for name, value in shared.items():
        assert getattr(pkg, name) == value, (name, getattr(pkg, name), value)
