"""Compile the reference's URDFs into the flat model JSONs shipped in assets/.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):
    python tools/compile_models.py [/root/reference]
The JSON holds numbers derived from the URDF (merged inertias, joint frames, collision spheres),
not the URDF text.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hcr_genesis_lr_cl_amd.model_compiler import compile_urdf, ASSET_DIR  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

GO2_DOFS = [f"{leg}_{j}_joint" for leg in ("FR", "FL", "RR", "RL") for j in ("hip", "thigh", "calf")]
TRON1_DOFS = [f"{j}_{s}_Joint" for s in ("L", "R") for j in ("abad", "hip", "knee")]

go2 = compile_urdf(os.path.join(REF, "resources/robots/go2/urdf/go2.urdf"), GO2_DOFS,
                   ["FL_foot", "FR_foot", "RL_foot", "RR_foot"], "foot", "base", name="go2")
go2.to_json(os.path.join(ASSET_DIR, "go2.json"))
tron = compile_urdf(os.path.join(REF, "resources/robots/PF_TRON1A/urdf/robot.urdf"), TRON1_DOFS,
                    ["foot_L_Link", "foot_R_Link"], "foot", "base_Link", name="tron1_pf")
tron.to_json(os.path.join(ASSET_DIR, "tron1_pf.json"))
SF_DOFS = [f"{j}_{s}_Joint" for s in ("L", "R") for j in ("abad", "hip", "knee", "ankle")]
sf = compile_urdf(os.path.join(REF, "resources/robots/SF_TRON1A/urdf/robot.urdf"), SF_DOFS, [], "ankle", "base_Link", name="tron1_sf")
sf.to_json(os.path.join(ASSET_DIR, "tron1_sf.json"))
for m in (go2, tron, sf):
    a = m.arrays
    print(m.name, "mass", m.total_mass, "links", m.n_links, m.link_names)
    print("  spheres", a["n_spheres"], "per body", list(a["body_sph_start"]))
    print("  feet", m.foot_names, list(a["foot_link"]), list(a["foot_sphere"]))
