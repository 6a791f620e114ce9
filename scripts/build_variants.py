"""A/B builds of libsanm_hip.so that differ in the flags of backend_hip.hip only (the kernels): sanm_amd/libsanm_hip_<name>.so,
selected at run time with SANM_HIP_LIBRARY -- and a copy sanm_amd/variant_<name>.so, which travels to the GPU box
(.gpurunignore keeps libsanm_hip_*.so at home; both are git-ignored).   python scripts/build_variants.py name="flags" [name="flags" ...]
e.g. python scripts/build_variants.py novf="-mllvm -amdgpu-mfma-vgpr-form=0" old="-DSANM_MF_OLD_STAGING" """
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from sanm_amd import build as B  # noqa: E402

B.build()
objdir = os.path.join(B.HERE, "build")
for arg in sys.argv[1:]:
    name, flags = arg.split("=", 1)
    src = os.path.join(B.CSRC, "backend_hip.hip")
    obj = os.path.join(objdir, f"backend_hip.{name}.o")
    base = [f for f in B.FLAGS]
    if "NOVF" in flags:
        base = [f for i, f in enumerate(base) if f != "-amdgpu-mfma-vgpr-form" and not (f == "-mllvm" and i + 1 < len(base) and base[i + 1] == "-amdgpu-mfma-vgpr-form")]
        flags = flags.replace("NOVF", "")
    cmd = [B.HIPCC] + base + flags.split() + ["-c", src, "-o", obj]
    subprocess.run(cmd, check=True)
    objs = [os.path.join(objdir, s + ".o") for s in B.SOURCES if s != "backend_hip.hip"] + [obj]
    out = os.path.join(B.HERE, f"libsanm_hip_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic", "-o", out] + objs + ["-ldl"], check=True)
    import shutil
    shutil.copy(out, os.path.join(B.HERE, f"variant_{name}.so"))
    print(out)
