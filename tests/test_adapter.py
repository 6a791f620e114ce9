"""The reference-side adapter (adapter/anm_hip.h: export_graph / export_desc and the three driver classes) compiles
against the reference's own headers: `g++ -std=c++20 -fsyntax-only -I/root/reference -Iinclude -Iadapter`.  The
reference's anm.h / symbolic.h / oprs/*.h are Eigen-free and parse in this container.  Nothing from
/root/reference is copied or shipped; where the reference is absent (the GPU box) the test is skipped."""
import os
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "libsanm")), reason="the reference tree is not on this box")
def test_adapter_compiles_against_the_reference_headers():
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", REF,
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "adapter"),
           os.path.join(ROOT, "adapter", "fea_callsites.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-4000:]


def test_adapter_uses_only_declared_entry_points():
    """every sanm_* function the adapter calls is declared in include/sanm_hip.h"""
    import re
    hdr = open(os.path.join(ROOT, "include", "sanm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(sanm_[A-Za-z0-9_]+)\s*\(", hdr))
    txt = open(os.path.join(ROOT, "adapter", "anm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    used = set(re.findall(r"\b(sanm_(?:hip|graph|sparse_desc|anm|hyper)_[A-Za-z0-9_]+)\s*\(", txt))
    assert used and used <= declared, sorted(used - declared)
