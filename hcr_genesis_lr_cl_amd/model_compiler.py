"""URDF -> flat articulated-body model for the HIP engine.

Mirrors what the reference asks its backend to do when it loads a robot:
``gs.morphs.URDF(file=..., merge_fixed_links=True, links_to_keep=feet)``
(reference legged_gym/simulator/genesis_simulator.py:303-313).  Fixed joints are
folded into their parent body (composite mass / centre of mass / inertia), the
links named in ``links_to_keep`` survive as *reported* links (own name, own
contact-force row, own frame for feet_pos/feet_vel) although they move rigidly
with their parent body.

The output is a plain dict of numpy arrays (`RobotModel.arrays`) that is
(1) serialised to ``assets/<robot>.json`` so the GPU box needs no URDF, and
(2) packed into the C-ABI ``LgModelDesc`` struct (include/lgsim.h).

Engine restriction (checked here, and again by ``lg_create``): a floating base
carrying ``n_legs`` serial chains of exactly three revolute joints whose dofs are
listed leg-major in ``dof_names`` -- the Go2 (4x3) and TRON1 point-foot (2x3)
trees of the reference (resources/robots/go2/urdf/go2.urdf,
resources/robots/PF_TRON1A/urdf/robot.urdf).
"""
from __future__ import annotations

import json
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

JOINTS_PER_LEG = 3          # go2, TRON1 point foot; the TRON1 sole-foot biped has 4 (two legs)
MAX_LEGS = 4
MAX_BODIES = 1 + MAX_LEGS * JOINTS_PER_LEG
MAX_LINKS = 24
MAX_SPHERES = 64


def _floats(s, n=None, default=None):
    if s is None:
        return np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.size != n:
        raise ValueError(f"expected {n} floats, got {s!r}")
    return v


def rpy_to_mat(rpy):
    """URDF fixed-axis roll/pitch/yaw -> rotation matrix (child -> parent)."""
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return rz @ ry @ rx


def _origin(elem):
    o = elem.find("origin") if elem is not None else None
    if o is None:
        return np.zeros(3), np.eye(3)
    xyz = _floats(o.get("xyz"), 3, [0, 0, 0])
    rpy = _floats(o.get("rpy"), 3, [0, 0, 0])
    return xyz, rpy_to_mat(rpy)


@dataclass
class _Link:
    name: str
    mass: float = 0.0
    com: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inertia: np.ndarray = field(default_factory=lambda: np.zeros((3, 3)))  # about com, link axes
    collisions: list = field(default_factory=list)  # (kind, params, xyz, R)


@dataclass
class _Joint:
    name: str
    jtype: str
    parent: str
    child: str
    xyz: np.ndarray
    rot: np.ndarray
    axis: np.ndarray
    lower: float = 0.0
    upper: float = 0.0
    effort: float = 0.0
    velocity: float = 0.0
    damping: float = 0.0
    friction: float = 0.0


def _parse_urdf(path):
    root = ET.parse(path).getroot()
    links, joints = {}, []
    order = []
    for le in root.findall("link"):
        lk = _Link(le.get("name"))
        ine = le.find("inertial")
        if ine is not None:
            lk.mass = float(ine.find("mass").get("value"))
            xyz, R = _origin(ine)
            it = ine.find("inertia")
            I = np.zeros((3, 3))
            if it is not None:
                g = lambda k: float(it.get(k, 0.0))
                I = np.array([[g("ixx"), g("ixy"), g("ixz")],
                              [g("ixy"), g("iyy"), g("iyz")],
                              [g("ixz"), g("iyz"), g("izz")]])
            lk.com = xyz
            lk.inertia = R @ I @ R.T
        for ce in le.findall("collision"):
            xyz, R = _origin(ce)
            geo = ce.find("geometry")
            if geo is None:
                continue
            if geo.find("sphere") is not None:
                lk.collisions.append(("sphere", [float(geo.find("sphere").get("radius"))], xyz, R))
            elif geo.find("box") is not None:
                lk.collisions.append(("box", list(_floats(geo.find("box").get("size"), 3)), xyz, R))
            elif geo.find("cylinder") is not None:
                c = geo.find("cylinder")
                lk.collisions.append(("cylinder", [float(c.get("radius")), float(c.get("length"))], xyz, R))
            # meshes are not collision primitives this engine handles
        links[lk.name] = lk
        order.append(lk.name)
    for je in root.findall("joint"):
        if je.get("type") is None or je.find("parent") is None:
            continue  # <transmission> style stubs
        xyz, R = _origin(je)
        ax = je.find("axis")
        axis = _floats(ax.get("xyz"), 3) if ax is not None else np.array([1.0, 0, 0])
        j = _Joint(je.get("name"), je.get("type"), je.find("parent").get("link"),
                   je.find("child").get("link"), xyz, R, axis)
        lim = je.find("limit")
        if lim is not None:
            j.lower = float(lim.get("lower", 0.0))
            j.upper = float(lim.get("upper", 0.0))
            j.effort = float(lim.get("effort", 0.0))
            j.velocity = float(lim.get("velocity", 0.0))
        dyn = je.find("dynamics")
        if dyn is not None:
            j.damping = float(dyn.get("damping", 0.0))
            j.friction = float(dyn.get("friction", 0.0))
        joints.append(j)
    return links, joints, order


def _spheres_for(kind, params, xyz, R):
    """Sphere decomposition of a URDF collision primitive (local pos, radius).

    sphere  -> itself
    cylinder (axis = local z) -> capsule-like: spheres of the cylinder radius at
        +-(L/2 - R) (one at the centre when L/2 <= R, a third in the middle when long)
    box -> slender (two smallest dims < 5 cm): capsule along the long axis;
        otherwise its eight vertices as zero-radius points (exact for box/plane).
    """
    out = []
    if kind == "sphere":
        out.append((xyz.copy(), params[0]))
    elif kind == "cylinder":
        rad, length = params
        h = max(0.5 * length - rad, 0.0)
        if h == 0.0:
            out.append((xyz.copy(), rad))
        else:
            zs = [-h, h] if h / rad <= 4.0 else [-h, 0.0, h]
            for z in zs:
                out.append((xyz + R @ np.array([0, 0, z]), rad))
    elif kind == "box":
        s = np.array(params)
        idx = np.argsort(s)
        if s[idx[1]] < 0.05:
            rad = 0.5 * s[idx[0]]
            h = 0.5 * s[idx[2]] - rad
            e = np.zeros(3)
            e[idx[2]] = 1.0
            for z in (-h, h):
                out.append((xyz + R @ (e * z), rad))
        else:
            for sx in (-0.5, 0.5):
                for sy in (-0.5, 0.5):
                    for sz in (-0.5, 0.5):
                        out.append((xyz + R @ (s * np.array([sx, sy, sz])), 0.0))
    return out


def _sole_spheres(size, xyz, R):
    """A box that IS a foot body (TRON1 sole foot, SF_TRON1A/urdf/robot.urdf ankle_*_Link: 0.2 x 0.06 x 0.03): the inscribed sphere
    at the box centre -- the leg's foot sphere, solved against the exact operational-space inertia like a point foot -- followed by
    the four corners of the face turned away from the joint (the sole) as zero-radius points, which hold the foot flat."""
    s = np.array(size, dtype=float)
    k = int(np.argmin(s))                       # thickness axis
    c_joint = R.T @ (-xyz)                      # joint origin seen from the box centre, box axes
    sgn = -1.0 if c_joint[k] > 0 else 1.0       # the sole is the face on the far side of the joint
    out = [(xyz.copy(), 0.5 * s[k])]
    a, b = [i for i in range(3) if i != k]
    for sa in (-0.5, 0.5):
        for sb in (-0.5, 0.5):
            v = np.zeros(3)
            v[a], v[b], v[k] = sa * s[a], sb * s[b], sgn * 0.5 * s[k]
            out.append((xyz + R @ v, 0.0))
    return out


class RobotModel:
    """Flat model; see module docstring.  ``arrays`` holds everything numeric."""

    def __init__(self, name, arrays, link_names, dof_names, foot_names):
        self.name = name
        self.arrays = arrays
        self.link_names = list(link_names)
        self.dof_names = list(dof_names)
        self.foot_names = list(foot_names)

    # -- convenience -------------------------------------------------------
    @property
    def n_legs(self):
        return int(self.arrays["n_legs"])

    @property
    def n_dof(self):
        return int(self.arrays["n_bodies"]) - 1

    @property
    def joints_per_leg(self):
        return self.n_dof // self.n_legs

    @property
    def n_links(self):
        return len(self.link_names)

    @property
    def total_mass(self):
        return float(np.sum(self.arrays["mass"]))

    def find_link_indices(self, names):
        """Substring match in link order (reference genesis_simulator.py:333-342)."""
        return [i for i, ln in enumerate(self.link_names) if any(n in ln for n in names)]

    # -- (de)serialisation ---------------------------------------------------
    def to_json(self, path):
        d = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in self.arrays.items()}
        blob = dict(name=self.name, link_names=self.link_names, dof_names=self.dof_names,
                    foot_names=self.foot_names, arrays=d)
        with open(path, "w") as f:
            json.dump(blob, f, indent=1)

    @staticmethod
    def from_json(path):
        with open(path) as f:
            blob = json.load(f)
        arrays = {}
        for k, v in blob["arrays"].items():
            if isinstance(v, list):
                a = np.array(v)
                arrays[k] = a.astype(np.int32) if k in _INT_KEYS else a.astype(np.float64)
            else:
                arrays[k] = v
        return RobotModel(blob["name"], arrays, blob["link_names"], blob["dof_names"], blob["foot_names"])


_INT_KEYS = {"parent", "link_body", "sph_body", "sph_link", "foot_link", "foot_sphere", "body_sph_start"}


def compile_urdf(path, dof_names, links_to_keep, foot_name, base_link_name=None, name=None):
    """Build a :class:`RobotModel` from a URDF file.

    dof_names     policy order of actuated joints (reference cfg.asset.dof_names)
    links_to_keep fixed-joint children that stay reported links (cfg.asset.links_to_keep)
    foot_name     substring identifying feet (cfg.asset.foot_name)
    """
    links, joints, order = _parse_urdf(path)
    children = {j.child for j in joints}
    roots = [n for n in order if n not in children]
    if base_link_name and base_link_name in links:
        base = base_link_name
    else:
        base = roots[0]
        # skip a dummy world link welded to the real base
        while links[base].mass == 0.0 and not links[base].collisions:
            nxt = [j for j in joints if j.parent == base]
            if len(nxt) != 1 or nxt[0].jtype != "fixed":
                break
            base = nxt[0].child
    jmap_child = {j.child: j for j in joints}
    by_parent = {}
    for j in joints:
        by_parent.setdefault(j.parent, []).append(j)

    n_dof = len(dof_names)
    jbyname = {j.name: j for j in joints}
    # legs = actuated joints hanging (through fixed joints) off the base; every leg is a serial chain of jpl joints
    def _body_above(link):
        while link != base and jmap_child[link].jtype == "fixed":
            link = jmap_child[link].parent
        return link
    n_legs = sum(1 for dn in dof_names if dn in jbyname and _body_above(jbyname[dn].parent) == base)
    if n_legs == 0 or n_dof % n_legs:
        raise ValueError("dof_names must list the same number of joints for every leg")
    jpl = n_dof // n_legs
    if n_legs > MAX_LEGS or (jpl, n_legs) not in ((3, 2), (3, 4), (4, 2)):
        raise ValueError(f"unsupported topology: {n_legs} legs of {jpl} joints (supported: 2 or 4 legs of 3, 2 legs of 4)")
    for dn in dof_names:
        if dn not in jbyname or jbyname[dn].jtype not in ("revolute", "continuous"):
            raise ValueError(f"dof {dn} is not a revolute joint of {path}")

    nb = 1 + n_dof
    body_of_link = {base: 0}
    for d, dn in enumerate(dof_names):
        body_of_link[jbyname[dn].child] = 1 + d

    parent = np.full(nb, -1, np.int32)
    jpos = np.zeros((nb, 3))
    jrot = np.tile(np.eye(3).reshape(-1), (nb, 1))
    axis = np.zeros((nb, 3))
    for d, dn in enumerate(dof_names):
        j = jbyname[dn]
        # walk up through fixed joints until a body is reached
        xyz, R = j.xyz.copy(), j.rot.copy()
        p = j.parent
        while p not in body_of_link:
            pj = jmap_child[p]
            if pj.jtype != "fixed":
                raise ValueError(f"unsupported joint {pj.name} ({pj.jtype}) above {dn}")
            xyz = pj.xyz + pj.rot @ xyz
            R = pj.rot @ R
            p = pj.parent
        parent[1 + d] = body_of_link[p]
        jpos[1 + d] = xyz
        jrot[1 + d] = R.reshape(-1)
        a = j.axis / np.linalg.norm(j.axis)
        axis[1 + d] = a
        leg, k = divmod(d, jpl)
        expect = 0 if k == 0 else d  # body index of expected parent
        if parent[1 + d] != expect:
            raise ValueError(f"{dn}: tree is not base + {n_legs} serial {jpl}-joint legs in dof order")

    # fold every remaining link (fixed children) into its body
    mass = np.zeros(nb)
    mcom = np.zeros((nb, 3))
    parts = [[] for _ in range(nb)]  # (m, com, I) in body frame
    link_names, link_body, link_pos = [], [], []
    spheres = []  # (body, link_idx, pos, r)

    def visit(lname, body, xyz, R, rep_link):
        lk = links[lname]
        if lk.mass > 0.0:
            parts[body].append((lk.mass, xyz + R @ lk.com, R @ lk.inertia @ R.T))
        for kind, params, cxyz, cR in lk.collisions:
            sole = kind == "box" and lname in body_of_link and foot_name in lname
            for (p, r) in (_sole_spheres(params, cxyz, cR) if sole else _spheres_for(kind, params, cxyz, cR)):
                spheres.append((body, rep_link, xyz + R @ p, r))
        for j in by_parent.get(lname, []):
            if j.child in body_of_link:
                continue
            if j.jtype != "fixed":
                raise ValueError(f"joint {j.name} is not actuated and not fixed")
            cx, cRm = xyz + R @ j.xyz, R @ j.rot
            if j.child in links_to_keep:
                link_names.append(j.child)
                link_body.append(body)
                link_pos.append(cx)
                visit(j.child, body, cx, cRm, len(link_names) - 1)
            else:
                visit(j.child, body, cx, cRm, rep_link)

    # reported links: base first, then bodies in URDF link order, kept links right after their body
    body_links = sorted(body_of_link.items(), key=lambda kv: order.index(kv[0]))
    for lname, body in body_links:
        link_names.append(lname)
        link_body.append(body)
        link_pos.append(np.zeros(3))
        visit(lname, body, np.zeros(3), np.eye(3), len(link_names) - 1)

    com = np.zeros((nb, 3))
    inertia = np.zeros((nb, 6))
    for b in range(nb):
        m = sum(p[0] for p in parts[b])
        if m <= 0.0:
            raise ValueError(f"body {b} has no mass")
        c = sum(p[0] * p[1] for p in parts[b]) / m
        I = np.zeros((3, 3))
        for (pm, pc, pI) in parts[b]:
            d = pc - c
            I += pI + pm * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        mass[b], com[b] = m, c
        inertia[b] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]

    # spheres sorted by body so the kernel can walk [body_sph_start[b], body_sph_start[b+1])
    spheres.sort(key=lambda s: s[0])
    ns = len(spheres)
    if ns > MAX_SPHERES:
        raise ValueError("too many collision spheres")
    sph_body = np.array([s[0] for s in spheres], np.int32)
    sph_link = np.array([s[1] for s in spheres], np.int32)
    sph_pos = np.array([s[2] for s in spheres]).reshape(ns, 3)
    sph_r = np.array([s[3] for s in spheres])
    # conservative point inverse mass: the carrying body treated as a free rigid body
    sph_w = np.zeros(ns)
    for i in range(ns):
        b = sph_body[i]
        I = _sym(inertia[b])
        lam = np.linalg.eigvalsh(I)[0]
        r = sph_pos[i] - com[b]
        sph_w[i] = 1.0 / mass[b] + np.dot(r, r) / max(lam, 1e-9)
    body_sph_start = np.zeros(nb + 1, np.int32)
    for b in range(nb):
        body_sph_start[b + 1] = body_sph_start[b] + int(np.sum(sph_body == b))

    foot_links = [i for i, ln in enumerate(link_names) if foot_name in ln]
    foot_link = np.full(MAX_LEGS, -1, np.int32)
    foot_sphere = np.full(MAX_LEGS, -1, np.int32)
    for li in foot_links:
        b = link_body[li]
        leg = (b - 1) // jpl
        if b == 0 or (b - 1) % jpl != jpl - 1:
            raise ValueError(f"foot {link_names[li]} is not on the last body of a leg")
        cand = [i for i in range(ns) if sph_link[i] == li]
        if link_names[li] in body_of_link:
            # the foot is the chain's last body itself (sole foot): its first sphere is the sole centre (_sole_spheres)
            if not cand:
                raise ValueError(f"foot body {link_names[li]} has no collision geometry")
            # sole corners: the kernel adds the ankle joint's own compliance at the contact point (lg_kernel.h sphere_contact);
            # what is stored here covers every other way the corner can give: shank + foot as a point mass
            for i in cand[1:]:
                sph_w[i] = 1.0 / (mass[b] + mass[b - 1])
            cand = cand[:1]
        elif len(cand) != 1:
            raise ValueError(f"foot {link_names[li]} must carry exactly one collision sphere")
        foot_link[leg] = li
        foot_sphere[leg] = cand[0]
    if np.any(foot_link[:n_legs] < 0):
        raise ValueError("every leg needs a foot link (links_to_keep / foot_name)")

    q_lo = np.array([jbyname[d].lower for d in dof_names])
    q_hi = np.array([jbyname[d].upper for d in dof_names])
    arrays = dict(
        n_legs=n_legs, n_bodies=nb, n_links=len(link_names), n_spheres=ns,
        parent=parent, mass=mass, com=com, inertia=inertia, jpos=jpos, jrot=jrot, axis=axis,
        q_lo=q_lo, q_hi=q_hi,
        effort=np.array([jbyname[d].effort for d in dof_names]),
        vel_limit=np.array([jbyname[d].velocity for d in dof_names]),
        damping=np.array([jbyname[d].damping for d in dof_names]),
        frictionloss=np.array([jbyname[d].friction for d in dof_names]),
        armature=np.zeros(n_dof),
        link_body=np.array(link_body, np.int32), link_pos=np.array(link_pos).reshape(-1, 3),
        sph_body=sph_body, sph_link=sph_link, sph_pos=sph_pos, sph_r=sph_r, sph_w=sph_w,
        body_sph_start=body_sph_start, foot_link=foot_link, foot_sphere=foot_sphere,
    )
    if len(link_names) > MAX_LINKS:
        raise ValueError("too many links")
    feet_in_link_order = [link_names[i] for i in foot_links]
    return RobotModel(name or os.path.splitext(os.path.basename(path))[0], arrays,
                      link_names, dof_names, feet_in_link_order)


def _sym(v6):
    xx, yy, zz, xy, xz, yz = v6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_model(name):
    """Load a pre-compiled model shipped in ``assets/`` (go2, tron1_pf)."""
    return RobotModel.from_json(os.path.join(ASSET_DIR, f"{name}.json"))
