"""The C-ABI library loads and exports every symbol include/sanm_hip.h (the interface) and include/sanm_hip_test.h
(this repository's test hooks) declare.
No compute calls: this runs in the GPU-less container."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols(headers=("sanm_hip.h", "sanm_hip_test.h")):
    out = set()
    for h in headers:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out |= set(re.findall(r"\b(sanm_[A-Za-z0-9_]+)\s*\(", txt))
    return sorted(out)


def test_the_interface_header_declares_no_test_hook():
    iface = _declared_symbols(("sanm_hip.h",))
    assert "sanm_anm_debug_inject" not in iface and "sanm_rtc_compile_check" not in iface
    assert set(_declared_symbols(("sanm_hip_test.h",))) == {"sanm_anm_debug_inject", "sanm_rtc_compile_check",
                                                            "sanm_rtc_cache_stats", "sanm_rtc_cache_probe",
                                                            "sanm_rtc_cache_drop_memory", "sanm_direct_solver_dist_plan",
                                                            "sanm_fea_spec_source", "sanm_rtc_source_key",
                                                            "sanm_rtc_compile_to_file", "sanm_rtc_embedded_hits", "sanm_test_set_p2p"}


def test_header_symbols_are_exported():
    import sanm_amd
    if not os.path.exists(sanm_amd.LIB_PATH):
        from sanm_amd.build import build
        build()
    lib = sanm_amd.load_library()
    syms = _declared_symbols()
    assert len(syms) > 50
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in sanm_hip.h but not exported"


def test_python_binding_lists_the_same_symbols():
    from sanm_amd.api import SYMBOLS
    assert sorted(SYMBOLS) == _declared_symbols()


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device init must fail."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sanm_amd
    from sanm_amd.api import Api, SanmError
    a = Api(sanm_amd.load_library())
    with pytest.raises(SanmError):
        a.init(0)


def test_package_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sanm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


@pytest.mark.gpu
def test_pytorch_initialised_after_the_product_library_still_finds_the_gpu():
    """the torch wheel bundles its own HIP runtime: a process in which libsanm_hip.so pulled in the system's copy first
    made a later torch.cuda initialisation report "No HIP GPUs are available" (seen on the MI355X boxes).  The package
    loads torch's runtime first (sanm_amd.load_library); a fresh process checks the order a user may well choose."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import sanm_amd; sanm_amd.get_api(0); import torch; "
            "torch.cuda.synchronize(); assert torch.cuda.is_available(); print('ok')" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
