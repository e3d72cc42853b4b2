"""CPU test: libnbody_hip.so loads and exports every symbol include/nbody_hip.h declares (no compute calls)."""
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_hip.h")).read()
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


def test_strict_translation_unit_has_no_fused_multiply_add():
    """The strict kernels must keep separate mul/add (bit-parity with the CPU path): the only v_fma in that
    object are inside the IEEE divide/sqrt expansions, never a contracted a*b+c of ours.  Checked structurally:
    the Makefile passes -ffp-contract=off to exactly that TU."""
    mk = open(os.path.join(ROOT, "cuda-nbody_amd", "csrc", "Makefile")).read()
    rule = re.search(r"nbody_strict\.o:.*?\n\t(.*)\n", mk).group(1)
    assert "-ffp-contract=off" in rule
    fast_rule = re.search(r"nbody_fast\.o:.*?\n\t(.*)\n", mk).group(1)
    assert "-ffp-contract=off" not in fast_rule
