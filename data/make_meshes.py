"""Convert the reference's TetGen input meshes (config/model/*.{node,ele,face})
into compressed .npz data files (vertices, tets, surface vertex ids).

Run in the authoring container only (needs /root/reference); the .npz files
are input DATA for tests and bench.py on the GPU box, where the reference tree
does not exist.  Config JSONs of the BASELINE runs are copied as data too.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle.fea import read_tetgen  # noqa: E402

REF = "/root/reference/config"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "meshes")
CONFIGS = {
    "armadillo_small": ["armadillo_small.json"],
    "bob": ["bob.json"],
    "human_arap16": ["human.json", "override_arap.json", "override_order16.json"],
}

os.makedirs(OUT, exist_ok=True)
for name, files in CONFIGS.items():
    cfg = {}
    for f in files:
        cfg.update(json.load(open(os.path.join(REF, f))))
    mesh = read_tetgen(os.path.join(REF, cfg["mesh"]))
    base = os.path.basename(cfg["mesh"])
    np.savez_compressed(os.path.join(OUT, base + ".npz"), vertices=mesh.V,
                        tets=mesh.tets.astype(np.int32), surface_vtx=mesh.surface_vtx.astype(np.int32))
    cfg["mesh_npz"] = base + ".npz"
    json.dump(cfg, open(os.path.join(OUT, name + ".json"), "w"), indent=1)
    print(name, mesh.nr_vertices, mesh.nr_tet)
