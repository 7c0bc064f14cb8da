"""Host-side model compiler: fruit-fly MJCF -> numeric model for the batched HIP environment.

This is the MI355X build's replacement for the two things the reference does with the MJCF
before physics ever runs:

1. the walker/task *edits* (`fruitfly/fruitfly.py:115-326` `FruitFly._build`;
   `tasks/base.py:264-328` `Flying.__init__`; `tasks/base.py:159-161` bound mass/inertia), and
2. MuJoCo's model *compilation* (third-party; the reference reaches it through
   `composer.Environment(...)`, `fly_envs.py:67-72`): defaults, frames, inertia from geoms,
   dof tree, spring-damper solve, inverse weights, fluid coefficients.

The result is a `CompiledModel` holding (a) the full, un-welded body list the way MuJoCo keeps
it (consumed by the float64 CPU oracle) and (b) a *welded link* view in which every joint-less
body is folded into the nearest jointed ancestor (consumed by the HIP kernels).  Both views are
serialised into one blob (`blob.py`).

All arithmetic here is float64.  Units are the model's CGS (cm, g, s).
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

from . import quat as Q
from .mesh import load_mesh, mass_properties
from .mjcf import Document, Element

MJMINVAL = 1e-15

# Empirical part masses (mg) the reference's model was tuned to: `build_fruitfly/make_fruitfly.py:24`.
MASS_TABLE_MG = {"head": 0.15, "thorax": 0.34, "abdomen": 0.38, "leg": 0.0162, "wing": 0.008}

JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
TRN_JOINT, TRN_TENDON, TRN_BODY = 0, 1, 2


# ---------------------------------------------------------------------------------------------
# Step 1: the reference walker's MJCF surgery
# ---------------------------------------------------------------------------------------------

_NAME_SUBSTR = {
    # `fruitfly.py:174-183`
    "adhesion": [],
    "head": ["head"],
    "mouth": ["rostrum", "haustellum", "labrum"],
    "antennae": ["antenna"],
    "wings": ["wing"],
    "abdomen": ["abdomen"],
    "legs": ["T1", "T2", "T3"],
    "user": [],
}
ACTION_CLASSES = ("adhesion", "head", "mouth", "antennae", "wings", "abdomen", "legs", "user")  # `fruitfly.py:25`


def _any_in(subs, s):
    return any(x in s for x in subs)


def _body_quat_from_springrefs(doc: Document, body: Element):
    """`fruitfly.py:61-80`: orientation of a leg segment with its joints parked at springref.

    Mirrors the reference's lookup exactly: springref is read from the joint, else from the
    joint's *own* class only; axis from the joint, its class, then that class's parent."""
    joints = [c for c in body.children if c.tag == "joint"]
    if not joints:
        return None
    quats = []
    for j in joints:
        cls = doc.classes.get(j.attrib.get("class", ""), None)
        theta = j.num("springref")
        if theta is None and cls is not None:
            v = cls.own_attr("joint", "springref")
            theta = None if v is None else np.array([float(v)])
        theta = 0.0 if theta is None else float(theta[0])
        axis = j.num("axis")
        if axis is None and cls is not None:
            v = cls.own_attr("joint", "axis")
            if v is None and cls.parent is not None:
                v = cls.parent.own_attr("joint", "axis")
            axis = np.array([float(x) for x in v.split()])
        quats.append(np.hstack((np.cos(theta / 2), np.sin(theta / 2) * axis)))
    quat = np.array([1.0, 0, 0, 0])
    for i in range(len(quats)):
        quat = Q.mul(quats[-1 - i], quat)
    bq = body.num("quat")
    if bq is not None:
        quat = Q.mul(bq, quat)
    return quat


def _change_body_frame(body: Element, frame_pos, frame_quat):
    """`fruitfly.py:83-106`: re-orient a body's frame, keeping its children where they were.
    Joints carry `pos` but no `quat`, so their axes are *not* counter-rotated - which is the
    whole point of the edit (the wing hinge axes turn with the new frame)."""
    body_pos = body.num("pos", np.zeros(3))
    frame_pos = body_pos if frame_pos is None else frame_pos
    dpos = body_pos - frame_pos
    body_quat = body.num("quat", np.array([1.0, 0, 0, 0]))
    dquat = Q.mul(Q.neg(frame_quat), body_quat)
    body.set("pos", frame_pos)
    body.set("quat", frame_quat)
    for child in body.children:
        if child.tag not in ("body", "geom", "site", "joint", "camera", "light", "inertial"):
            continue
        if child.tag in ("body", "geom", "site", "camera", "inertial"):
            cq = child.num("quat", np.array([1.0, 0, 0, 0]))
            child.set("quat", Q.mul(dquat, cq))
        cp = child.num("pos", np.zeros(3))
        pos_in_parent = Q.rot(cp, body_quat) + dpos
        child.set("pos", Q.rot(pos_in_parent, Q.neg(frame_quat)))


@dataclass
class WalkerOptions:
    """Keyword surface of `FruitFly._build` (`fruitfly.py:115-131`) that affects physics."""

    use_legs: bool = True
    use_wings: bool = False
    use_mouth: bool = False
    use_antennae: bool = False
    joint_filter: float = 0.01
    adhesion_filter: float = 0.007
    body_pitch_angle: float = 47.5
    stroke_plane_angle: float = 0.0
    num_user_actions: int = 0


def apply_walker_edits(doc: Document, opt: WalkerOptions) -> dict:
    """Replays `FruitFly._build` on the document.  Returns the action bookkeeping
    (`_ctrl_indices`, `_action_indices`, observable joints) the walker derives from it."""
    observable_joints = [j.name for j in doc.find_all("joint") if j.tag == "joint"]

    def drop_obs(name):
        if name in observable_joints:
            observable_joints.remove(name)

    if not opt.use_legs:  # `fruitfly.py:188-218`
        for b in doc.find_all("body"):
            if _any_in(_NAME_SUBSTR["legs"], b.name):
                b.set("quat", _body_quat_from_springrefs(doc, b))
        for t in doc.find_all("tendon"):
            if _any_in(_NAME_SUBSTR["legs"], t.name):
                a = doc.find("actuator", t.name)
                if a is not None:
                    a.remove()
                t.remove()
        for j in [j for j in doc.find_all("joint") if _any_in(_NAME_SUBSTR["legs"], j.name)]:
            a = doc.find("actuator", j.name)
            if a is not None:
                a.remove()
            drop_obs(j.name)
            j.remove()
        for a in doc.find_all("actuator"):
            if "adhere" in a.name and _any_in(_NAME_SUBSTR["legs"], a.name):
                a.remove()
        for s in doc.find_all("sensor"):
            if _any_in(_NAME_SUBSTR["legs"], s.name):
                s.remove()

    def strip_group(group, adhesion_too):
        for j in [j for j in doc.find_all("joint") if _any_in(_NAME_SUBSTR[group], j.name)]:
            doc.find("actuator", j.name).remove()
            drop_obs(j.name)
        if adhesion_too:
            for a in doc.find_all("actuator"):
                if "adhere" in a.name and _any_in(_NAME_SUBSTR[group], a.name):
                    a.remove()

    if not opt.use_wings:  # `fruitfly.py:221-229`
        strip_group("wings", False)
    if not opt.use_mouth:  # `fruitfly.py:232-240`
        strip_group("mouth", True)
    if not opt.use_antennae:  # `fruitfly.py:243-247`
        strip_group("antennae", False)

    if opt.use_wings:  # `fruitfly.py:250-269`
        site = doc.find("site", "hover_up_dir")
        up_dir = site.num("quat")
        up_dir_angle = 2 * np.arccos(up_dir[0])
        delta = np.deg2rad(opt.body_pitch_angle) - up_dir_angle
        dquat = np.array([np.cos(delta / 2), 0, np.sin(delta / 2), 0])
        up_dir = Q.mul(dquat, up_dir)
        site.set("quat", up_dir)
        spa = np.deg2rad(opt.stroke_plane_angle)
        stroke_plane_quat = np.array([np.cos(spa / 2), 0, np.sin(spa / 2), 0])
        for quat, wing in [(np.array([0.0, 0, 0, 1]), "wing_left"), (np.array([0.0, -1, 0, 0]), "wing_right")]:
            dq = Q.mul(Q.neg(stroke_plane_quat), quat)
            new_wing_quat = Q.mul(dq, Q.neg(up_dir))
            body = doc.find("body", wing)
            _change_body_frame(body, body.num("pos"), new_wing_quat)

    if opt.joint_filter > 0:  # `fruitfly.py:272-276`
        for a in doc.find_all("actuator"):
            if a.tag != "adhesion":
                a.set("dyntype", "filter")
                a.set("dynprm", np.array([opt.joint_filter]))
    if opt.adhesion_filter > 0:  # `fruitfly.py:277-281` (edits the parent default class)
        for a in doc.find_all("actuator"):
            if a.tag == "adhesion":
                parent = doc.classes[a.attrib["class"]].parent
                parent.own.setdefault("general", {})["dyntype"] = "filter"
                parent.own["general"]["dynprm"] = str(opt.adhesion_filter)

    names = [a.name for a in doc.find_all("actuator")]
    ctrl_indices = {}
    for cls in ACTION_CLASSES:  # `fruitfly.py:285-296`
        idx = [i for i, n in enumerate(names) if _any_in(_NAME_SUBSTR[cls], n) and "adhere" not in n]
        ctrl_indices[cls] = idx if idx else None
    idx = [i for i, n in enumerate(names) if "adhere" in n]
    ctrl_indices["adhesion"] = idx if idx else None
    num_actions = {c: 0 for c in ACTION_CLASSES}  # `fruitfly.py:299-307`
    num_actions["user"] = opt.num_user_actions
    for c in ACTION_CLASSES:
        if ctrl_indices[c] is not None:
            num_actions[c] = len(ctrl_indices[c])
    action_indices, counter = {}, 0  # `fruitfly.py:310-318`
    for c in ACTION_CLASSES:
        if num_actions[c]:
            action_indices[c] = list(range(counter, counter + num_actions[c]))
            counter += num_actions[c]
        else:
            action_indices[c] = []
    return {
        "ctrl_indices": ctrl_indices,
        "action_indices": action_indices,
        "num_actions": num_actions,
        "observable_joints": observable_joints,
        "actuator_names": names,
    }


def apply_flying_edits(doc: Document, wing_gainprm=(18.0, 18.0, 18.0), wing_stiffness=0.01,
                       wing_damping=0.007769230, fluidcoef=(1.0, 0.5, 1.5, 1.7, 1.0)) -> None:
    """`tasks/base.py:304-325` with the constants of `tasks/constants.py:29-37`."""
    for i, dclass in enumerate(["yaw", "roll", "pitch"]):
        doc.classes[dclass].own.setdefault("general", {})["gainprm"] = str(wing_gainprm[i])
    for g in doc.find_all("geom"):
        if "fluid" in g.name:
            g.set("fluidshape", "ellipsoid")
            g.set("fluidcoef", np.asarray(fluidcoef, dtype=np.float64))
    wj = doc.classes["wing"].own.setdefault("joint", {})
    wj["stiffness"] = str(wing_stiffness)
    wj["damping"] = str(wing_damping)


# ---------------------------------------------------------------------------------------------
# Step 2: compilation
# ---------------------------------------------------------------------------------------------


@dataclass
class CompiledModel:
    """MuJoCo-style flat model.  Field names follow `mjModel` where the meaning is the same."""

    # options
    timestep: float = 0.0
    gravity: np.ndarray = None
    density: float = 0.0
    viscosity: float = 0.0
    # bodies (index 0 = world)
    body_name: list = field(default_factory=list)
    body_parentid: np.ndarray = None
    body_pos: np.ndarray = None
    body_quat: np.ndarray = None
    body_mass: np.ndarray = None
    body_ipos: np.ndarray = None
    body_iquat: np.ndarray = None
    body_inertia: np.ndarray = None
    body_jntadr: np.ndarray = None
    body_jntnum: np.ndarray = None
    body_dofadr: np.ndarray = None
    body_dofnum: np.ndarray = None
    body_fluid_kind: np.ndarray = None  # 0 none, 1 inertia box, 2 ellipsoid geoms
    body_box: np.ndarray = None
    # joints / dofs
    jnt_name: list = field(default_factory=list)
    jnt_type: np.ndarray = None
    jnt_bodyid: np.ndarray = None
    jnt_qposadr: np.ndarray = None
    jnt_dofadr: np.ndarray = None
    jnt_pos: np.ndarray = None
    jnt_axis: np.ndarray = None
    jnt_limited: np.ndarray = None
    jnt_range: np.ndarray = None
    jnt_stiffness: np.ndarray = None
    jnt_margin: np.ndarray = None
    jnt_solref: np.ndarray = None
    jnt_solimp: np.ndarray = None
    qpos0: np.ndarray = None
    qpos_spring: np.ndarray = None
    dof_bodyid: np.ndarray = None
    dof_jntid: np.ndarray = None
    dof_parentid: np.ndarray = None
    dof_damping: np.ndarray = None
    dof_armature: np.ndarray = None
    dof_invweight0: np.ndarray = None
    dof_M0: np.ndarray = None
    # fluid ellipsoid geoms
    fl_bodyid: np.ndarray = None
    fl_pos: np.ndarray = None
    fl_quat: np.ndarray = None
    fl_size: np.ndarray = None
    fl_coef: np.ndarray = None  # (n, 12): interaction, blunt, slender, ang, kutta, magnus, vmass[3], vinertia[3]
    # fixed tendons
    ten_name: list = field(default_factory=list)
    ten_adr: np.ndarray = None
    ten_num: np.ndarray = None
    wrap_dof: np.ndarray = None
    wrap_coef: np.ndarray = None
    # actuators
    act_name: list = field(default_factory=list)
    act_trntype: np.ndarray = None
    act_trnid: np.ndarray = None
    act_gear: np.ndarray = None
    act_gainprm: np.ndarray = None
    act_biasprm: np.ndarray = None  # (nu, 3)
    act_ctrllimited: np.ndarray = None
    act_ctrlrange: np.ndarray = None
    act_forcelimited: np.ndarray = None
    act_forcerange: np.ndarray = None
    act_dyntype: np.ndarray = None  # 0 none, 1 filter
    act_dynprm: np.ndarray = None
    # sensor site (thorax)
    site_bodyid: int = 0
    site_pos: np.ndarray = None
    site_quat: np.ndarray = None
    # collision geoms (contype|conaffinity != 0 only), MuJoCo type codes
    geom_name: list = field(default_factory=list)
    geom_bodyid: np.ndarray = None
    geom_type: np.ndarray = None
    geom_size: np.ndarray = None
    geom_pos: np.ndarray = None
    geom_quat: np.ndarray = None
    geom_condim: np.ndarray = None
    geom_friction: np.ndarray = None
    geom_margin: np.ndarray = None
    geom_gap: np.ndarray = None
    geom_solref: np.ndarray = None
    geom_solimp: np.ndarray = None
    geom_solmix: np.ndarray = None
    geom_priority: np.ndarray = None
    geom_contype: np.ndarray = None
    geom_conaffinity: np.ndarray = None
    exclude_pairs: np.ndarray = None  # (n, 2) body ids from <contact><exclude>
    body_weldid: np.ndarray = None
    body_invweight0: np.ndarray = None  # (nbody, 2): translational, rotational
    # named sites (all of them) and the sensors that refer to them
    sites_name: list = field(default_factory=list)
    sites_bodyid: np.ndarray = None
    sites_pos: np.ndarray = None
    sites_quat: np.ndarray = None
    sites_type: np.ndarray = None
    sites_size: np.ndarray = None
    touch_site: np.ndarray = None
    force_site: np.ndarray = None
    # solver options
    cone_elliptic: int = 0
    noslip_iterations: int = 0
    impratio: float = 1.0
    # bookkeeping from the walker
    walker: dict = field(default_factory=dict)
    notes: dict = field(default_factory=dict)

    @property
    def nbody(self):
        return len(self.body_parentid)

    @property
    def njnt(self):
        return len(self.jnt_type)

    @property
    def nq(self):
        return len(self.qpos0)

    @property
    def nv(self):
        return len(self.dof_bodyid)

    @property
    def nu(self):
        return len(self.act_trntype)


def _prim_mass_props(gtype: str, size: np.ndarray):
    """Unit-density volume and principal inertia (about own centre, own axes) of a primitive."""
    if gtype == "sphere":
        r = size[0]
        v = 4 / 3 * np.pi * r**3
        return v, np.full(3, 0.4 * v * r * r)
    if gtype == "ellipsoid":
        a, b, c = size[:3]
        v = 4 / 3 * np.pi * a * b * c
        return v, v / 5 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == "box":
        a, b, c = size[:3]
        v = 8 * a * b * c
        return v, v / 3 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    if gtype == "cylinder":
        r, h = size[0], size[1]
        v = np.pi * r * r * 2 * h
        ixy = v * (3 * r * r + 4 * h * h) / 12
        return v, np.array([ixy, ixy, v * r * r / 2])
    if gtype == "capsule":
        r, h = size[0], size[1]
        vc = np.pi * r * r * 2 * h
        vs = 4 / 3 * np.pi * r**3
        v = vc + vs
        izz = vc * r * r / 2 + vs * 0.4 * r * r
        ixy = vc * (3 * r * r + 4 * h * h) / 12 + vs * (0.4 * r * r + h * h + 0.75 * r * h)
        return v, np.array([ixy, ixy, izz])
    raise ValueError(gtype)


def _added_mass_kappa(dx, dy, dz):
    """kappa_x = dx dy dz * integral_0^inf dl / ((dx^2+l)^(3/2) sqrt((dy^2+l)(dz^2+l)))  (Lamb 1932),
    the quantity MuJoCo tabulates for ellipsoid added mass."""
    from scipy.integrate import quad

    # integrate in log space (l = s e^x): the integrand decays exponentially on both sides
    s = max(dx, dy, dz) ** 2
    a2, b2, c2 = dx * dx / s, dy * dy / s, dz * dz / s

    def g(x):
        l = np.exp(x)
        return l / ((a2 + l) ** 1.5 * np.sqrt((b2 + l) * (c2 + l)))

    total = quad(g, -80.0, 0.0, epsabs=0, epsrel=1e-13, limit=400)[0]
    total += quad(g, 0.0, 80.0, epsabs=0, epsrel=1e-13, limit=400)[0]
    # dx dy dz * s^(-3/2) from the change of variable (dl = s e^x dx; denominators scale s^(5/2))
    return dx * dy * dz * total / s**1.5


def ellipsoid_fluid_coefs(size, fluidcoef):
    """Virtual mass / inertia of an ellipsoid moving in an ideal fluid (per unit fluid density)."""
    dx, dy, dz = size
    kx = _added_mass_kappa(dx, dy, dz)
    ky = _added_mass_kappa(dy, dz, dx)
    kz = _added_mass_kappa(dz, dx, dy)
    vol = 4 / 3 * np.pi * dx * dy * dz
    eps = MJMINVAL
    dx2, dy2, dz2 = dx * dx, dy * dy, dz * dz
    ixf = (dy2 - dz2) ** 2 * abs(kz - ky) / max(eps, abs(2 * (dy2 - dz2) + (dy2 + dz2) * (ky - kz)))
    iyf = (dz2 - dx2) ** 2 * abs(kx - kz) / max(eps, abs(2 * (dz2 - dx2) + (dz2 + dx2) * (kz - kx)))
    izf = (dx2 - dy2) ** 2 * abs(ky - kx) / max(eps, abs(2 * (dx2 - dy2) + (dx2 + dy2) * (kx - ky)))
    vmass = vol * np.array([kx / max(eps, 2 - kx), ky / max(eps, 2 - ky), kz / max(eps, 2 - kz)])
    vinertia = vol * np.array([ixf, iyf, izf]) / 5
    return np.hstack(([1.0], fluidcoef, vmass, vinertia))


class _MeshBank:
    """Loads mesh assets once; falls back to the legacy `.msh` twins shipped under
    `build_fruitfly/assets/` for the six `.obj` files absent from the reference snapshot
    (`.MISSING_LARGE_BLOBS:4-10`), and reconstructs `head_red` (no copy anywhere) from the
    lateral faces of `head_body`, calibrated to the reference's head-mass target."""

    def __init__(self, doc: Document, mesh_dir: str, fallback_dir: str, rule: str):
        self.rule = rule
        self.mesh_dir = mesh_dir
        self.fallback_dir = fallback_dir
        self.files, self.scale = {}, {}
        dflt_scale = doc.main.own.get("mesh", {}).get("scale", "1 1 1")
        for m in doc.section("asset").children:
            if m.tag != "mesh":
                continue
            name = m.attrib.get("name") or os.path.splitext(m.attrib["file"])[0]
            self.files[name] = m.attrib["file"]
            self.scale[name] = np.array([float(x) for x in m.attrib.get("scale", dflt_scale).split()])
        self.cache = {}
        self.substituted = []
        self.reconstructed = {}

    def raw(self, name):
        fn = self.files[name]
        p = os.path.join(self.mesh_dir, fn)
        if not os.path.exists(p):
            alt = os.path.join(self.fallback_dir, os.path.splitext(fn)[0] + ".msh")
            if not os.path.exists(alt):
                return None
            self.substituted.append(fn)
            p = alt
        return load_mesh(p)

    def props(self, name):
        """(volume, com, inertia about com) in *scaled* mesh coordinates, unit density."""
        if name in self.cache:
            return self.cache[name]
        if name in self.reconstructed:
            v, f = self.reconstructed[name]
        else:
            r = self.raw(name)
            if r is None:
                raise FileNotFoundError(name)
            v, f = r
        s = self.scale[name]
        vol, com, inertia = mass_properties(v * s, f, self.rule)
        self.cache[name] = (vol, com, inertia)
        return self.cache[name]

    def reconstruct_lateral_caps(self, name, donor, target_volume_scaled):
        """`name` := faces of `donor` with |y| above a threshold, threshold bisected so that the
        legacy-rule volume equals `target_volume_scaled`."""
        v, f = self.raw(donor)
        s = self.scale[donor]
        cen_y = np.abs(v[f].mean(1)[:, 1])
        lo, hi = 0.0, float(cen_y.max())
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            ff = f[cen_y > mid]
            vol = mass_properties(v * s, ff, self.rule)[0] if len(ff) > 3 else 0.0
            if vol > target_volume_scaled:
                lo = mid
            else:
                hi = mid
        thr = 0.5 * (lo + hi)
        self.reconstructed[name] = (v, f[cen_y > thr])
        self.scale[name] = s
        self.cache.pop(name, None)
        return thr


def _geom_frame(g: dict):
    """(pos, quat, size) of a geom from resolved attributes, handling `fromto` and `euler`."""
    size = g.get("size")
    size = np.zeros(3) if size is None else np.pad(_f(size), (0, 3))[:3]
    if "fromto" in g:
        ft = _f(g["fromto"])
        a, b = ft[:3], ft[3:]
        pos = 0.5 * (a + b)
        quat = Q.z_to_vec(b - a)
        size = np.array([size[0], 0.5 * np.linalg.norm(b - a), 0.0])
        return pos, quat, size
    pos = _f(g["pos"]) if "pos" in g else np.zeros(3)
    if "quat" in g:
        quat = Q.normalize(_f(g["quat"]))
    elif "euler" in g:
        quat = Q.from_euler_xyz(_f(g["euler"]))
    else:
        quat = np.array([1.0, 0, 0, 0])
    return pos, quat, size


def _f(v):
    if isinstance(v, str):
        return np.array([float(x) for x in v.split()], dtype=np.float64)
    return np.atleast_1d(np.asarray(v, dtype=np.float64))


def _eig_inertia(tensor):
    """Principal moments (descending, as `mju_eig3` orders them) and the frame quaternion."""
    w, vec = np.linalg.eigh(tensor)
    order = np.argsort(-w)
    w, vec = w[order], vec[:, order]
    if np.linalg.det(vec) < 0:
        vec[:, 2] *= -1
    return w, Q.from_mat(vec)


def compile_model(doc: Document, mesh_dir: str, fallback_mesh_dir: str, timestep: float,
                  mesh_rule: str = "legacy", free_root: bool = True,
                  spawn_pos=(0.0, 0.0, 0.1278)) -> CompiledModel:
    m = CompiledModel()
    opt = doc.section("option").attrib
    m.timestep = float(timestep)
    m.gravity = _f(opt.get("gravity", "0 0 -9.81"))
    m.density = float(opt.get("density", 0))
    m.viscosity = float(opt.get("viscosity", 0))

    bank = _MeshBank(doc, mesh_dir, fallback_mesh_dir, mesh_rule)

    # ---- body list, DFS document order -------------------------------------------------------
    bodies: list[Element] = []
    parent_of: dict[int, int] = {}

    def walk(e: Element, pid: int):
        for c in e.children:
            if c.tag == "body":
                bodies.append(c)
                bid = len(bodies)  # world is 0
                parent_of[bid] = pid
                walk(c, bid)

    walk(doc.worldbody, 0)
    nb = len(bodies) + 1
    m.body_name = ["world"] + [b.name for b in bodies]
    m.body_parentid = np.zeros(nb, dtype=np.int32)
    m.body_pos = np.zeros((nb, 3))
    m.body_quat = np.tile([1.0, 0, 0, 0], (nb, 1))
    for i, b in enumerate(bodies, start=1):
        m.body_parentid[i] = parent_of[i]
        m.body_pos[i] = b.num("pos", np.zeros(3))
        m.body_quat[i] = Q.normalize(b.num("quat", np.array([1.0, 0, 0, 0])))
    # dm_control attaches the walker under a frame body at the spawn site (`tasks/base.py:130-133`);
    # thorax sits at that frame's origin.  With a free joint on the frame, giving the thorax the free
    # joint with qpos0 = spawn pose is the same model; without one (`walk_on_ball.py:30`) the thorax is
    # simply fixed to the world at the spawn position.
    root_idx = m.body_name.index("thorax")
    m.body_pos[root_idx] = np.asarray(spawn_pos, dtype=np.float64)

    # ---- geoms -> body inertials ---------------------------------------------------------------
    geoms_of = {i: [c for c in b.children if c.tag == "geom"] for i, b in enumerate(bodies, start=1)}

    def geom_contrib(g: Element):
        a = doc.resolved(g)
        gtype = a.get("type")
        if gtype is None:
            gtype = "mesh" if "mesh" in a else "sphere"
        pos, quat, size = _geom_frame(a)
        if gtype == "mesh":
            vol, com, inertia = bank.props(a["mesh"])
        else:
            vol, ip = _prim_mass_props(gtype, size)
            com, inertia = np.zeros(3), np.diag(ip)
        if "mass" in a:
            mass = float(_f(a["mass"])[0])
            dens = mass / vol if vol > 0 else 0.0
        else:
            dens = float(_f(a.get("density", "1000"))[0])
            mass = dens * vol
        R = Q.to_mat(quat)
        return mass, pos + R @ com, R @ (dens * inertia) @ R.T

    # head_red has no mesh file anywhere in the snapshot: calibrate its volume so the head subtree
    # reaches the reference's 0.15 mg target (`make_fruitfly.py:24`), then rebuild it from
    # head_body's lateral faces.
    head_idx = m.body_name.index("head") if "head" in m.body_name else -1
    if head_idx > 0 and bank.raw("head_red") is None:
        subtree = [i for i in range(1, nb) if _is_descendant(m.body_parentid, i, head_idx)]
        mass_wo = 0.0
        dens_red = None
        for i in subtree:
            for g in geoms_of[i]:
                a = doc.resolved(g)
                if a.get("mesh") == "head_red":
                    dens_red = float(_f(a.get("density", "1000"))[0])
                    continue
                mass_wo += geom_contrib(g)[0]
        target_mass = MASS_TABLE_MG["head"] * 1e-3 - mass_wo
        thr = bank.reconstruct_lateral_caps("head_red", "head", target_mass / dens_red)
        m.notes["head_red_threshold_y"] = thr
        m.notes["head_red_mass"] = target_mass

    m.body_mass = np.zeros(nb)
    m.body_ipos = np.zeros((nb, 3))
    m.body_iquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    m.body_inertia = np.zeros((nb, 3))
    body_tensor = np.zeros((nb, 3, 3))
    for i in range(1, nb):
        parts = [geom_contrib(g) for g in geoms_of[i]]
        mass = sum(p[0] for p in parts)
        if mass <= 0:
            continue
        com = sum(p[0] * p[1] for p in parts) / mass
        tensor = np.zeros((3, 3))
        for pm, pc, pi in parts:
            d = pc - com
            tensor += pi + pm * (d @ d * np.eye(3) - np.outer(d, d))
        w, iq = _eig_inertia(tensor)
        m.body_mass[i], m.body_ipos[i], m.body_iquat[i], m.body_inertia[i] = mass, com, iq, w
        body_tensor[i] = tensor
    m.notes["mesh_substituted"] = sorted(set(bank.substituted))
    m.notes["mesh_rule"] = mesh_rule

    # ---- joints and dofs -----------------------------------------------------------------------
    jt, jb, jqa, jda, jpos, jax, jlim, jrng, jst, jmar, jsr, jsi = ([] for _ in range(12))
    jspringdamper = []
    qpos0, qspring = [], []
    dbody, djnt, dpar, ddamp, darm = [], [], [], [], []
    m.body_jntadr = np.full(nb, -1, dtype=np.int32)
    m.body_jntnum = np.zeros(nb, dtype=np.int32)
    m.body_dofadr = np.full(nb, -1, dtype=np.int32)
    m.body_dofnum = np.zeros(nb, dtype=np.int32)
    last_dof_of_body = np.full(nb, -1, dtype=np.int32)
    for i, b in enumerate(bodies, start=1):
        joints = [c for c in b.children if c.tag in ("joint", "freejoint")]
        if i == root_idx and free_root and not any(c.tag == "freejoint" or c.attrib.get("type") == "free" for c in joints):
            joints = [Element("freejoint", {"name": "free"}, b)] + joints
        pid = m.body_parentid[i]
        inherit = last_dof_of_body[pid] if pid >= 0 else -1
        if joints:
            m.body_jntadr[i] = len(jt)
            m.body_jntnum[i] = len(joints)
            m.body_dofadr[i] = len(dbody)
        prev = inherit
        for j in joints:
            a = {} if j.tag == "freejoint" else doc.resolved(j)
            jtype = "free" if j.tag == "freejoint" else a.get("type", "hinge")
            m.jnt_name.append(j.name)
            jb.append(i)
            jqa.append(len(qpos0))
            jda.append(len(dbody))
            jpos.append(_f(a["pos"]) if "pos" in a else np.zeros(3))
            if jtype == "free":
                jt.append(JNT_FREE)
                jax.append(np.array([0.0, 0, 1]))
                jlim.append(0)
                jrng.append(np.zeros(2))
                jst.append(0.0)
                jmar.append(0.0)
                jsr.append(np.array([0.02, 1.0]))
                jsi.append(np.array([0.9, 0.95, 0.001, 0.5, 2.0]))
                jspringdamper.append(np.zeros(2))
                qpos0.extend(list(m.body_pos[i]) + list(m.body_quat[i]))
                qspring.extend(list(m.body_pos[i]) + list(m.body_quat[i]))
                ndof = 6
                damp, arm = 0.0, 0.0
            elif jtype == "ball":
                jt.append(JNT_BALL)
                jax.append(np.array([0.0, 0, 1]))
                jlim.append(0)
                jrng.append(np.zeros(2))
                jst.append(float(_f(a.get("stiffness", "0"))[0]))
                jmar.append(0.0)
                jsr.append(np.array([0.02, 1.0]))
                jsi.append(np.array([0.9, 0.95, 0.001, 0.5, 2.0]))
                jspringdamper.append(np.zeros(2))
                qpos0.extend([1.0, 0, 0, 0])
                qspring.extend([1.0, 0, 0, 0])
                ndof = 3
                damp = float(_f(a.get("damping", "0"))[0])
                arm = float(_f(a.get("armature", "0"))[0])
            elif jtype == "hinge":
                jt.append(JNT_HINGE)
                ax = _f(a["axis"]) if "axis" in a else np.array([0.0, 0, 1])
                jax.append(ax / np.linalg.norm(ax))
                rng = _f(a["range"]) if "range" in a else np.zeros(2)
                lim = a.get("limited", "auto")
                limited = (lim == "true") or (lim == "auto" and "range" in a)
                jlim.append(int(limited))
                jrng.append(rng)
                jst.append(float(_f(a.get("stiffness", "0"))[0]))
                jmar.append(float(_f(a.get("margin", "0"))[0]))
                jsr.append(_f(a.get("solreflimit", "0.02 1")))
                si = _f(a.get("solimplimit", "0.9 0.95 0.001 0.5 2"))
                jsi.append(np.hstack((si, [0.9, 0.95, 0.001, 0.5, 2.0][len(si):])))
                jspringdamper.append(_f(a.get("springdamper", "0 0")))
                ref = float(_f(a.get("ref", "0"))[0])
                qpos0.append(ref)
                qspring.append(float(_f(a.get("springref", "0"))[0]))
                ndof = 1
                damp = float(_f(a.get("damping", "0"))[0])
                arm = float(_f(a.get("armature", "0"))[0])
            else:
                raise NotImplementedError(jtype)
            for _ in range(ndof):
                dbody.append(i)
                djnt.append(len(jt) - 1)
                dpar.append(prev)
                ddamp.append(damp)
                darm.append(arm)
                prev = len(dbody) - 1
        m.body_dofnum[i] = (len(dbody) - m.body_dofadr[i]) if joints else 0
        last_dof_of_body[i] = prev
    m.jnt_type = np.array(jt, dtype=np.int32)
    m.jnt_bodyid = np.array(jb, dtype=np.int32)
    m.jnt_qposadr = np.array(jqa, dtype=np.int32)
    m.jnt_dofadr = np.array(jda, dtype=np.int32)
    m.jnt_pos = np.array(jpos)
    m.jnt_axis = np.array(jax)
    m.jnt_limited = np.array(jlim, dtype=np.int32)
    m.jnt_range = np.array(jrng)
    m.jnt_stiffness = np.array(jst)
    m.jnt_margin = np.array(jmar)
    m.jnt_solref = np.array(jsr)
    m.jnt_solimp = np.array(jsi)
    m.qpos0 = np.array(qpos0)
    m.qpos_spring = np.array(qspring)
    m.dof_bodyid = np.array(dbody, dtype=np.int32)
    m.dof_jntid = np.array(djnt, dtype=np.int32)
    m.dof_parentid = np.array(dpar, dtype=np.int32)
    m.dof_damping = np.array(ddamp)
    m.dof_armature = np.array(darm)

    # ---- M at qpos0 -> spring-damper solve, inverse weights ----------------------------------------
    from .pyref import mass_matrix

    M0 = mass_matrix(m, m.qpos0)
    m.dof_M0 = np.diag(M0).copy()
    for j in range(m.njnt):
        tc, dr = jspringdamper[j]
        if tc > 0 and dr > 0:  # MJCF `springdamper`: set stiffness/damping from joint inertia at qpos0
            d = m.jnt_dofadr[j]
            inertia = m.dof_M0[d]
            m.jnt_stiffness[j] = inertia / max(MJMINVAL, tc * tc * dr * dr)
            m.dof_damping[d] = 2 * inertia / max(MJMINVAL, tc)
    Minv = np.linalg.inv(M0)
    m.dof_invweight0 = np.diag(Minv).copy()
    for j in range(m.njnt):
        if m.jnt_type[j] == JNT_FREE:
            d = m.jnt_dofadr[j]
            m.dof_invweight0[d : d + 3] = np.mean(np.diag(Minv)[d : d + 3])
            m.dof_invweight0[d + 3 : d + 6] = np.mean(np.diag(Minv)[d + 3 : d + 6])
        elif m.jnt_type[j] == JNT_BALL:
            d = m.jnt_dofadr[j]
            m.dof_invweight0[d : d + 3] = np.mean(np.diag(Minv)[d : d + 3])

    # ---- fluid ---------------------------------------------------------------------------------
    m.body_fluid_kind = np.zeros(nb, dtype=np.int32)
    m.body_box = np.zeros((nb, 3))
    fb, fp, fq, fs, fc = [], [], [], [], []
    for i in range(1, nb):
        if m.body_mass[i] < MJMINVAL:
            continue
        ell = []
        for g in geoms_of[i]:
            a = doc.resolved(g)
            if a.get("fluidshape") == "ellipsoid":
                ell.append(a)
        if ell:
            m.body_fluid_kind[i] = 2
            for a in ell:
                pos, quat, size = _geom_frame(a)
                fb.append(i)
                fp.append(pos)
                fq.append(quat)
                fs.append(size)
                fc.append(ellipsoid_fluid_coefs(size, _f(a.get("fluidcoef", "0.5 0.25 1.5 1.0 1.0"))))
        else:
            m.body_fluid_kind[i] = 1
            I, mass = m.body_inertia[i], m.body_mass[i]
            m.body_box[i] = [
                np.sqrt(max(MJMINVAL, I[1] + I[2] - I[0]) / mass * 6.0),
                np.sqrt(max(MJMINVAL, I[0] + I[2] - I[1]) / mass * 6.0),
                np.sqrt(max(MJMINVAL, I[0] + I[1] - I[2]) / mass * 6.0),
            ]
    m.fl_bodyid = np.array(fb, dtype=np.int32)
    m.fl_pos = np.array(fp).reshape(-1, 3)
    m.fl_quat = np.array(fq).reshape(-1, 4)
    m.fl_size = np.array(fs).reshape(-1, 3)
    m.fl_coef = np.array(fc).reshape(-1, 12)

    # ---- tendons ---------------------------------------------------------------------------------
    jname2id = {n: k for k, n in enumerate(m.jnt_name)}
    tadr, tnum, wd, wc = [], [], [], []
    for t in doc.find_all("tendon"):
        if t.tag != "fixed":
            raise NotImplementedError(t.tag)
        m.ten_name.append(t.name)
        tadr.append(len(wd))
        n = 0
        for w in t.children:
            if w.tag == "joint":
                wd.append(m.jnt_dofadr[jname2id[w.attrib["joint"]]])
                wc.append(float(w.attrib.get("coef", 1)))
                n += 1
        tnum.append(n)
    m.ten_adr = np.array(tadr, dtype=np.int32)
    m.ten_num = np.array(tnum, dtype=np.int32)
    m.wrap_dof = np.array(wd, dtype=np.int32)
    m.wrap_coef = np.array(wc)

    # ---- actuators --------------------------------------------------------------------------------
    tname2id = {n: k for k, n in enumerate(m.ten_name)}
    bname2id = {n: k for k, n in enumerate(m.body_name)}
    tt, ti, gear, gain, bias, cl, cr, fl, fr, dt, dp = ([] for _ in range(11))
    for aelem in doc.find_all("actuator"):
        a = doc.resolved(aelem)
        m.act_name.append(aelem.name)
        if "joint" in a:
            tt.append(TRN_JOINT)
            ti.append(jname2id[a["joint"]])
        elif "tendon" in a:
            tt.append(TRN_TENDON)
            ti.append(tname2id[a["tendon"]])
        elif "body" in a:
            tt.append(TRN_BODY)
            ti.append(bname2id[a["body"]])
        else:
            raise NotImplementedError(str(a))
        gear.append(float(_f(a.get("gear", "1"))[0]))
        if aelem.tag == "adhesion":
            gain.append(float(_f(a.get("gain", "1"))[0]))
            bias.append(np.zeros(3))
        else:
            gain.append(float(_f(a.get("gainprm", "1"))[0]))
            bp = _f(a.get("biasprm", "0 0 0")) if a.get("biastype", "none") == "affine" else np.zeros(3)
            bias.append(np.pad(bp, (0, 3))[:3])
        crange = _f(a["ctrlrange"]) if "ctrlrange" in a else np.zeros(2)
        climited = a.get("ctrllimited", "auto")
        cl.append(int(climited == "true" or (climited == "auto" and "ctrlrange" in a)))
        cr.append(crange)
        frange = _f(a["forcerange"]) if "forcerange" in a else np.zeros(2)
        flimited = a.get("forcelimited", "auto")
        fl.append(int(flimited == "true" or (flimited == "auto" and "forcerange" in a)))
        fr.append(frange)
        dyn = a.get("dyntype", "none")
        dt.append({"none": 0, "filter": 1}[dyn])
        dp.append(float(_f(a.get("dynprm", "1"))[0]))
    m.act_trntype = np.array(tt, dtype=np.int32)
    m.act_trnid = np.array(ti, dtype=np.int32)
    m.act_gear = np.array(gear)
    m.act_gainprm = np.array(gain)
    m.act_biasprm = np.array(bias).reshape(-1, 3)
    m.act_ctrllimited = np.array(cl, dtype=np.int32)
    m.act_ctrlrange = np.array(cr).reshape(-1, 2)
    m.act_forcelimited = np.array(fl, dtype=np.int32)
    m.act_forcerange = np.array(fr).reshape(-1, 2)
    m.act_dyntype = np.array(dt, dtype=np.int32)
    m.act_dynprm = np.array(dp)

    # ---- sensor site --------------------------------------------------------------------------------
    site = doc.find("site", "thorax")
    m.site_bodyid = m.body_name.index(site.parent.name)
    m.site_pos = site.num("pos", np.zeros(3))
    m.site_quat = Q.normalize(site.num("quat", np.array([1.0, 0, 0, 0])))

    # ---- solver options ---------------------------------------------------------------------------------
    m.cone_elliptic = int(opt.get("cone", "pyramidal") == "elliptic")
    m.noslip_iterations = int(opt.get("noslip_iterations", 0))
    m.impratio = float(opt.get("impratio", 1.0))

    # ---- collision geoms, contact excludes, weld ids ------------------------------------------------------
    GEOM_CODE = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6, "mesh": 7}
    gb, gt, gs, gp, gq, gcd, gfr, gma, gga, gsr, gsi, gmx, gpr, gct, gca = ([] for _ in range(15))
    world_geoms = [c for c in doc.worldbody.children if c.tag == "geom"]  # (arena geoms such as a floor plane; none in the fly's own file)
    for i in range(0, nb):
        for g in (world_geoms if i == 0 else geoms_of[i]):
            a = doc.resolved(g)
            ct, ca = int(a.get("contype", 1)), int(a.get("conaffinity", 1))
            if ct == 0 and ca == 0:
                continue
            gtype = a.get("type") or ("mesh" if "mesh" in a else "sphere")
            pos, quat, size = _geom_frame(a)
            m.geom_name.append(g.name)
            gb.append(i)
            gt.append(GEOM_CODE[gtype])
            gs.append(size)
            gp.append(pos)
            gq.append(quat)
            gcd.append(int(a.get("condim", 3)))
            fr = _f(a.get("friction", "1 0.005 0.0001"))
            gfr.append(np.hstack((fr, [1.0, 0.005, 0.0001][len(fr):])))
            gma.append(float(_f(a.get("margin", "0"))[0]))
            gga.append(float(_f(a.get("gap", "0"))[0]))
            gsr.append(_f(a.get("solref", "0.02 1")))
            si = _f(a.get("solimp", "0.9 0.95 0.001 0.5 2"))
            gsi.append(np.hstack((si, [0.9, 0.95, 0.001, 0.5, 2.0][len(si):])))
            gmx.append(float(_f(a.get("solmix", "1"))[0]))
            gpr.append(int(a.get("priority", 0)))
            gct.append(ct)
            gca.append(ca)
    m.geom_bodyid = np.array(gb, dtype=np.int32)
    m.geom_type = np.array(gt, dtype=np.int32)
    m.geom_size = np.array(gs).reshape(-1, 3)
    m.geom_pos = np.array(gp).reshape(-1, 3)
    m.geom_quat = np.array(gq).reshape(-1, 4)
    m.geom_condim = np.array(gcd, dtype=np.int32)
    m.geom_friction = np.array(gfr).reshape(-1, 3)
    m.geom_margin = np.array(gma)
    m.geom_gap = np.array(gga)
    m.geom_solref = np.array(gsr).reshape(-1, 2)
    m.geom_solimp = np.array(gsi).reshape(-1, 5)
    m.geom_solmix = np.array(gmx)
    m.geom_priority = np.array(gpr, dtype=np.int32)
    m.geom_contype = np.array(gct, dtype=np.int32)
    m.geom_conaffinity = np.array(gca, dtype=np.int32)
    ex = []
    csec = doc.section("contact")
    for e in csec.children if csec is not None else []:
        if e.tag == "exclude":
            ex.append([bname2id[e.attrib["body1"]], bname2id[e.attrib["body2"]]])
    m.exclude_pairs = np.array(ex, dtype=np.int32).reshape(-1, 2)
    m.body_weldid = np.zeros(nb, dtype=np.int32)
    for i in range(1, nb):
        m.body_weldid[i] = i if m.body_jntnum[i] > 0 else m.body_weldid[m.body_parentid[i]]

    # ---- body inverse weights at qpos0 (mj: setM0/`mj_setConst`: mean diagonal of J M^-1 J' per body) ----
    from .pyref import body_jacobians

    m.body_invweight0 = np.zeros((nb, 2))
    if len(m.geom_bodyid):
        jacp, jacr = body_jacobians(m, m.qpos0)
        for i in range(1, nb):
            if m.body_weldid[i] == 0:
                continue
            Ap = jacp[i] @ Minv @ jacp[i].T
            Ar = jacr[i] @ Minv @ jacr[i].T
            m.body_invweight0[i] = [np.trace(Ap) / 3, np.trace(Ar) / 3]

    # ---- all sites, touch / force sensors -------------------------------------------------------------------
    SITE_CODE = {"sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6}
    sb, sp, sq, st, ss = [], [], [], [], []
    for i, b in enumerate(bodies, start=1):
        for c in b.children:
            if c.tag != "site":
                continue
            a = doc.resolved(c)
            pos, quat, size = _geom_frame(a)
            m.sites_name.append(c.name)
            sb.append(i)
            sp.append(pos)
            sq.append(quat)
            st.append(SITE_CODE[a.get("type", "sphere")])
            ss.append(size)
    m.sites_bodyid = np.array(sb, dtype=np.int32)
    m.sites_pos = np.array(sp).reshape(-1, 3)
    m.sites_quat = np.array(sq).reshape(-1, 4)
    m.sites_type = np.array(st, dtype=np.int32)
    m.sites_size = np.array(ss).reshape(-1, 3)
    sname2id = {n: k for k, n in enumerate(m.sites_name)}
    m.touch_site = np.array([sname2id[e.attrib["site"]] for e in doc.find_all("sensor") if e.tag == "touch"], dtype=np.int32)
    m.force_site = np.array([sname2id[e.attrib["site"]] for e in doc.find_all("sensor") if e.tag == "force"], dtype=np.int32)
    return m


def _is_descendant(parentid, i, root):
    while i > 0:
        if i == root:
            return True
        i = parentid[i]
    return False


# ---------------------------------------------------------------------------------------------
# Step 3: welded link view for the GPU
# ---------------------------------------------------------------------------------------------


@dataclass
class LinkModel:
    """Joint-less bodies folded into their nearest jointed ancestor ("link").  Frames of the
    folded bodies are constant in the link frame, so their inertia adds once at compile time and
    their per-body fluid boxes become a fixed list of (frame-in-link, box) records."""

    link_body: np.ndarray = None  # body id of each link
    link_parent: np.ndarray = None  # parent link (-1 for the root link)
    link_pos: np.ndarray = None  # frame in parent link
    link_quat: np.ndarray = None
    link_mass: np.ndarray = None
    link_ipos: np.ndarray = None  # composite CoM in link frame
    link_iquat: np.ndarray = None  # composite principal frame
    link_inertia: np.ndarray = None  # composite principal moments
    link_dofadr: np.ndarray = None
    link_dofnum: np.ndarray = None
    link_depth: np.ndarray = None
    link_subtree: np.ndarray = None  # number of links in the subtree (DFS order => contiguous)
    body_link: np.ndarray = None  # link that carries each body
    # fluid records
    fbox_link: np.ndarray = None
    fbox_pos: np.ndarray = None  # body CoM in link frame
    fbox_mat: np.ndarray = None  # inertial frame axes in link frame (3x3 row-major, columns = axes)
    fbox_box: np.ndarray = None
    fell_link: np.ndarray = None
    fell_pos: np.ndarray = None
    fell_mat: np.ndarray = None
    fell_size: np.ndarray = None
    fell_coef: np.ndarray = None


def weld(m: CompiledModel) -> LinkModel:
    nb = m.nbody
    L = LinkModel()
    link_of_body = np.full(nb, -1, dtype=np.int32)
    rel_pos = np.zeros((nb, 3))
    rel_quat = np.tile([1.0, 0, 0, 0], (nb, 1))
    links = []
    for i in range(1, nb):
        if m.body_jntnum[i] > 0:
            link_of_body[i] = len(links)
            links.append(i)
        else:
            p = m.body_parentid[i]
            if p == 0:
                raise NotImplementedError("static bodies attached to the world")
            link_of_body[i] = link_of_body[p]
            rel_pos[i] = rel_pos[p] + Q.rot(m.body_pos[i], rel_quat[p])
            rel_quat[i] = Q.normalize(Q.mul(rel_quat[p], m.body_quat[i]))
    nl = len(links)
    L.link_body = np.array(links, dtype=np.int32)
    L.body_link = link_of_body
    L.link_parent = np.full(nl, -1, dtype=np.int32)
    L.link_pos = np.zeros((nl, 3))
    L.link_quat = np.tile([1.0, 0, 0, 0], (nl, 1))
    L.link_depth = np.zeros(nl, dtype=np.int32)
    for k, b in enumerate(links):
        p = m.body_parentid[b]
        if p == 0:
            L.link_pos[k], L.link_quat[k] = m.body_pos[b], m.body_quat[b]
            continue
        pl = link_of_body[p]
        L.link_parent[k] = pl
        L.link_depth[k] = L.link_depth[pl] + 1
        L.link_pos[k] = rel_pos[p] + Q.rot(m.body_pos[b], rel_quat[p])
        L.link_quat[k] = Q.normalize(Q.mul(rel_quat[p], m.body_quat[b]))
    L.link_subtree = np.ones(nl, dtype=np.int32)
    for k in range(nl - 1, 0, -1):
        L.link_subtree[L.link_parent[k]] += L.link_subtree[k]
    # composite inertials
    L.link_mass = np.zeros(nl)
    L.link_ipos = np.zeros((nl, 3))
    L.link_iquat = np.tile([1.0, 0, 0, 0], (nl, 1))
    L.link_inertia = np.zeros((nl, 3))
    first = np.zeros((nl, 3))
    for i in range(1, nb):
        k = link_of_body[i]
        c = rel_pos[i] + Q.rot(m.body_ipos[i], rel_quat[i])
        L.link_mass[k] += m.body_mass[i]
        first[k] += m.body_mass[i] * c
    for k in range(nl):
        L.link_ipos[k] = first[k] / L.link_mass[k]
    tens = np.zeros((nl, 3, 3))
    for i in range(1, nb):
        k = link_of_body[i]
        if m.body_mass[i] <= 0:
            continue
        R = Q.to_mat(Q.mul(rel_quat[i], m.body_iquat[i]))
        c = rel_pos[i] + Q.rot(m.body_ipos[i], rel_quat[i])
        d = c - L.link_ipos[k]
        tens[k] += R @ np.diag(m.body_inertia[i]) @ R.T + m.body_mass[i] * (d @ d * np.eye(3) - np.outer(d, d))
    for k in range(nl):
        L.link_inertia[k], L.link_iquat[k] = _eig_inertia(tens[k])
    L.link_dofadr = np.array([m.body_dofadr[b] for b in links], dtype=np.int32)
    L.link_dofnum = np.array([m.body_dofnum[b] for b in links], dtype=np.int32)
    # fluid records
    bl, bp, bm, bb = [], [], [], []
    for i in range(1, nb):
        if m.body_fluid_kind[i] != 1:
            continue
        bl.append(link_of_body[i])
        bp.append(rel_pos[i] + Q.rot(m.body_ipos[i], rel_quat[i]))
        bm.append(Q.to_mat(Q.mul(rel_quat[i], m.body_iquat[i])))
        bb.append(m.body_box[i])
    L.fbox_link = np.array(bl, dtype=np.int32)
    L.fbox_pos = np.array(bp).reshape(-1, 3)
    L.fbox_mat = np.array(bm).reshape(-1, 9)
    L.fbox_box = np.array(bb).reshape(-1, 3)
    el, ep, em = [], [], []
    for g in range(len(m.fl_bodyid)):
        i = m.fl_bodyid[g]
        el.append(link_of_body[i])
        ep.append(rel_pos[i] + Q.rot(m.fl_pos[g], rel_quat[i]))
        em.append(Q.to_mat(Q.mul(rel_quat[i], m.fl_quat[g])))
    L.fell_link = np.array(el, dtype=np.int32)
    L.fell_pos = np.array(ep).reshape(-1, 3)
    L.fell_mat = np.array(em).reshape(-1, 9)
    L.fell_size = m.fl_size.copy()
    L.fell_coef = m.fl_coef.copy()
    # collision geoms in link coordinates (the flight kernel's contact stage; `tasks/base.py:299-302` leaves fly - fly collisions on)
    ng = len(m.geom_bodyid)
    L.cgeom_link = np.array([link_of_body[m.geom_bodyid[g]] for g in range(ng)], dtype=np.int32)
    L.cgeom_pos = np.array([rel_pos[m.geom_bodyid[g]] + Q.rot(m.geom_pos[g], rel_quat[m.geom_bodyid[g]]) for g in range(ng)]).reshape(-1, 3)
    L.cgeom_quat = np.array([Q.normalize(Q.mul(rel_quat[m.geom_bodyid[g]], m.geom_quat[g])) for g in range(ng)]).reshape(-1, 4)
    return L


# ---------------------------------------------------------------------------------------------
# Front doors
# ---------------------------------------------------------------------------------------------

REFERENCE_ASSETS = "/root/reference/vnl_ray/fruitfly/assets"
REFERENCE_MSH = "/root/reference/vnl_ray/fruitfly/build_fruitfly/assets"


def build_flight_model(assets_dir: str = REFERENCE_ASSETS, msh_dir: str = REFERENCE_MSH,
                       mesh_rule: str = "legacy"):
    """The model `fly_envs.flight_imitation` compiles (`fly_envs.py:29-72`): legs retracted, wings
    on, mouth/antennae passive, `joint_filter=0`, one user action, 5e-5 s physics step."""
    doc = Document(os.path.join(assets_dir, "fruitfly.xml"))
    fj = doc.find("freejoint", "free")
    if fj is not None:
        fj.remove()  # `fruitfly.py:168`; re-created by the arena attachment (see compile_model)
    wopt = WalkerOptions(use_legs=False, use_wings=True, use_mouth=False, use_antennae=False,
                         joint_filter=0.0, adhesion_filter=0.007, body_pitch_angle=47.5,
                         num_user_actions=1)
    walker = apply_walker_edits(doc, wopt)
    apply_flying_edits(doc)
    m = compile_model(doc, assets_dir, msh_dir, timestep=5e-5, mesh_rule=mesh_rule)
    m.walker = walker
    return m, weld(m)


def apply_walking_edits(doc: Document) -> None:
    """`tasks/base.py:159-161` (mass bounds) are no-ops here; `Walking.__init__` only touches the arena's
    ground geoms (`tasks/base.py:353-357`), which `build_ball_model` writes directly."""


def apply_walk_on_ball_edits(doc: Document, claw_friction=1.0) -> None:
    """`tasks/walk_on_ball.py:30-45`: thorax-children contact excludes and the claw friction override."""
    csec = doc.section("contact")
    thorax = doc.find("body", "thorax")
    for child in thorax.children:
        if child.tag == "body":
            csec.add(Element("exclude", {"name": f"thorax_{child.name}", "body1": "thorax", "body2": child.name}))
    if claw_friction is not None:
        doc.classes["adhesion-collision"].own.setdefault("geom", {})["friction"] = str(claw_friction)


def build_ball_model(assets_dir: str = REFERENCE_ASSETS, msh_dir: str = REFERENCE_MSH, mesh_rule: str = "legacy",
                     ball_pos=(-0.05, 0.0, -0.419), ball_radius=0.454, ball_density=0.0025,
                     joint_filter=0.01, adhesion_filter=0.007, claw_friction=1.0) -> CompiledModel:
    """The model `fly_envs.walk_on_ball` compiles (`fly_envs.py:125-157`): legs on, wings / mouth / antennae
    passive, thorax fixed to the world, a free-spinning ball under the legs (`tasks/arenas/ball.py:61-69`),
    filtered position actuators, 2e-4 s physics step (`tasks/constants.py:16-17`)."""
    doc = Document(os.path.join(assets_dir, "fruitfly.xml"))
    fj = doc.find("freejoint", "free")
    if fj is not None:
        fj.remove()
    wopt = WalkerOptions(use_legs=True, use_wings=False, use_mouth=False, use_antennae=False,
                         joint_filter=joint_filter, adhesion_filter=adhesion_filter, num_user_actions=0)
    walker = apply_walker_edits(doc, wopt)
    apply_walk_on_ball_edits(doc, claw_friction)
    # the arena's ball precedes the attached walker in the world body (`tasks/arenas/ball.py:61`, then
    # `tasks/base.py:130`); ground-geom contact parameters from `tasks/base.py:353-357`
    ball = Element("body", {"name": "ball", "pos": np.asarray(ball_pos, dtype=np.float64)}, doc.worldbody)
    ball.add(Element("geom", {"name": "ball_geom", "type": "sphere", "size": np.array([ball_radius, 0.0, 0.0]),
                              "density": str(ball_density), "friction": "0.5 0.005 0.0001", "solref": "0.001 1",
                              "solimp": "0.95 0.99 0.01", "contype": "1", "conaffinity": "1", "condim": "3"}))
    ball.add(Element("joint", {"name": "ball", "type": "ball"}))
    doc.worldbody.children.insert(0, ball)
    m = compile_model(doc, assets_dir, msh_dir, timestep=2e-4, mesh_rule=mesh_rule, free_root=False)
    m.walker = walker
    return m


def build_walk_model(assets_dir: str = REFERENCE_ASSETS, msh_dir: str = REFERENCE_MSH, mesh_rule: str = "legacy",
                     joint_filter=0.01, adhesion_filter=0.007, claw_friction=1.0, floor_size=(8.0, 8.0, 0.25)) -> CompiledModel:
    """The model `fly_envs.walk_imitation` compiles (`fly_envs.py:75-122`): the `Walking` configuration of the fly
    (`tasks/base.py:331-364`: legs on, wings / mouth / antennae passive, filtered position actuators, 2e-4 s physics step)
    on a free joint over dm_control's `floors.Floor()` (one plane geom at z = 0; dm_control is third-party and absent,
    its default floor is restated: size 8 x 8, default contact dimension 3), with the ground-geom contact parameters of
    `tasks/base.py:353-357` and the claw friction override of `tasks/walk_imitation.py:67-69`.  The ghost walker has no
    contacts and only mirrors the reference pose (`walk_imitation.py:117-132`): it is not part of the compiled physics."""
    doc = Document(os.path.join(assets_dir, "fruitfly.xml"))
    fj = doc.find("freejoint", "free")
    if fj is not None:
        fj.remove()
    wopt = WalkerOptions(use_legs=True, use_wings=False, use_mouth=False, use_antennae=False,
                         joint_filter=joint_filter, adhesion_filter=adhesion_filter, num_user_actions=0)
    walker = apply_walker_edits(doc, wopt)
    if claw_friction is not None:
        doc.classes["adhesion-collision"].own.setdefault("geom", {})["friction"] = str(claw_friction)
    floor = Element("geom", {"name": "groundplane", "type": "plane", "size": np.asarray(floor_size, dtype=np.float64),
                             "friction": "0.5 0.005 0.0001", "solref": "0.001 1", "solimp": "0.95 0.99 0.01",
                             "contype": "1", "conaffinity": "1", "condim": "3"}, doc.worldbody)
    doc.worldbody.children.insert(0, floor)
    m = compile_model(doc, assets_dir, msh_dir, timestep=2e-4, mesh_rule=mesh_rule, free_root=True)
    m.walker = walker
    return m
