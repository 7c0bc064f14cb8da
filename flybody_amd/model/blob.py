"""Model blob: a flat, named-tensor archive shared by the Python host, the C oracle and the HIP
library (each has its own ~50-line reader).

Layout (little endian):
    char[4] "FFMB" | u32 version | u32 count
    repeat count: u16 name_len | name bytes | u8 dtype (0=f64, 1=i32) | u8 ndim | u32 dims[ndim]
                  | zero padding to an 8-byte boundary | raw data (C order)
"""

from __future__ import annotations

import struct

import numpy as np

MAGIC = b"FFMB"
VERSION = 1


def write_blob(path: str, tensors: dict) -> None:
    out = bytearray()
    out += MAGIC + struct.pack("<II", VERSION, len(tensors))
    for name, arr in tensors.items():
        a = np.asarray(arr)
        if a.dtype.kind == "f":
            a, code = np.ascontiguousarray(a, dtype="<f8"), 0
        elif a.dtype.kind in "iub":
            a, code = np.ascontiguousarray(a, dtype="<i4"), 1
        else:
            raise TypeError(f"{name}: {a.dtype}")
        nb = name.encode()
        out += struct.pack("<H", len(nb)) + nb + struct.pack("<BB", code, a.ndim)
        out += struct.pack(f"<{a.ndim}I", *a.shape)
        out += b"\0" * ((-len(out)) % 8)
        out += a.tobytes()
    with open(path, "wb") as f:
        f.write(bytes(out))


def read_blob(path: str) -> dict:
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != MAGIC:
        raise ValueError("not a model blob")
    version, count = struct.unpack_from("<II", data, 4)
    if version != VERSION:
        raise ValueError(f"blob version {version}")
    off, out = 12, {}
    for _ in range(count):
        (nl,) = struct.unpack_from("<H", data, off)
        off += 2
        name = data[off : off + nl].decode()
        off += nl
        code, ndim = struct.unpack_from("<BB", data, off)
        off += 2
        dims = struct.unpack_from(f"<{ndim}I", data, off)
        off += 4 * ndim
        off += (-off) % 8
        dt = "<f8" if code == 0 else "<i4"
        n = int(np.prod(dims)) if ndim else 1
        out[name] = np.frombuffer(data, dtype=dt, count=n, offset=off).reshape(dims).copy()
        off += n * (8 if code == 0 else 4)
    return out


def model_tensors(m, L=None, with_collision=False) -> dict:
    """Flatten a `CompiledModel` + `LinkModel` (+ the walker's action/observation bookkeeping)."""
    w = m.walker
    nu = m.nu
    act_action = np.full(nu, -1, dtype=np.int32)
    for cls, cidx in w["ctrl_indices"].items():
        aidx = w["action_indices"][cls]
        if cidx and aidx:
            for c, a in zip(cidx, aidx):
                act_action[c] = a
    naction = sum(len(v) for v in w["action_indices"].values())
    amin, amax = np.zeros(naction), np.zeros(naction)
    for c in range(nu):
        if act_action[c] >= 0:
            amin[act_action[c]], amax[act_action[c]] = m.act_ctrlrange[c]
    for a in w["action_indices"]["user"]:
        amin[a], amax[a] = -1.0, 1.0
    jid = {n: k for k, n in enumerate(m.jnt_name)}
    obs_j = [jid[n] for n in w["observable_joints"]]
    wing = [jid[f"wing_{ax}_{side}"] for side in ("left", "right") for ax in ("yaw", "roll", "pitch")
            if f"wing_{ax}_{side}" in jid]
    t = {
        "opt": np.array([m.timestep, m.density, m.viscosity, *m.gravity]),
        "body_parentid": m.body_parentid, "body_pos": m.body_pos, "body_quat": m.body_quat,
        "body_mass": m.body_mass, "body_ipos": m.body_ipos, "body_iquat": m.body_iquat,
        "body_inertia": m.body_inertia, "body_jntadr": m.body_jntadr, "body_jntnum": m.body_jntnum,
        "body_dofadr": m.body_dofadr, "body_dofnum": m.body_dofnum,
        "body_fluid_kind": m.body_fluid_kind, "body_box": m.body_box,
        "jnt_type": m.jnt_type, "jnt_bodyid": m.jnt_bodyid, "jnt_qposadr": m.jnt_qposadr,
        "jnt_dofadr": m.jnt_dofadr, "jnt_pos": m.jnt_pos, "jnt_axis": m.jnt_axis,
        "jnt_limited": m.jnt_limited, "jnt_range": m.jnt_range, "jnt_stiffness": m.jnt_stiffness,
        "jnt_margin": m.jnt_margin, "jnt_solref": m.jnt_solref, "jnt_solimp": m.jnt_solimp,
        "qpos0": m.qpos0, "qpos_spring": m.qpos_spring,
        "dof_bodyid": m.dof_bodyid, "dof_jntid": m.dof_jntid, "dof_parentid": m.dof_parentid,
        "dof_damping": m.dof_damping, "dof_armature": m.dof_armature,
        "dof_invweight0": m.dof_invweight0, "dof_M0": m.dof_M0,
        "fl_bodyid": m.fl_bodyid, "fl_pos": m.fl_pos, "fl_quat": m.fl_quat, "fl_size": m.fl_size,
        "fl_coef": m.fl_coef,
        "ten_adr": m.ten_adr, "ten_num": m.ten_num, "wrap_dof": m.wrap_dof, "wrap_coef": m.wrap_coef,
        "act_trntype": m.act_trntype, "act_trnid": m.act_trnid, "act_gear": m.act_gear,
        "act_gainprm": m.act_gainprm, "act_biasprm": m.act_biasprm,
        "act_ctrllimited": m.act_ctrllimited, "act_ctrlrange": m.act_ctrlrange,
        "act_forcelimited": m.act_forcelimited, "act_forcerange": m.act_forcerange,
        "act_dyntype": m.act_dyntype, "act_dynprm": m.act_dynprm,
        "site": np.array([m.site_bodyid, *m.site_pos, *m.site_quat], dtype=np.float64),
        # walker / task bookkeeping
        "act_action": act_action, "action_min": amin, "action_max": amax,
        "wing_action": np.array(w["action_indices"]["wings"], dtype=np.int32),
        "user_action": np.array(w["action_indices"]["user"], dtype=np.int32),
        "wing_jnt": np.array(wing, dtype=np.int32),
        "obs_jnt": np.array(obs_j, dtype=np.int32),
    }
    if with_collision:  # contact-capable models (walk_on_ball)
        sid = {n: k for k, n in enumerate(m.sites_name)}
        app = [sid[n] for n in ("claw_T1_left", "claw_T1_right", "claw_T2_left", "claw_T2_right", "claw_T3_left",
                                "claw_T3_right", "head") if n in sid]  # `fruitfly.py:421-447`
        t.update({
            "geom_bodyid": m.geom_bodyid, "geom_type": m.geom_type, "geom_size": m.geom_size, "geom_pos": m.geom_pos,
            "geom_quat": m.geom_quat, "geom_condim": m.geom_condim, "geom_friction": m.geom_friction,
            "geom_margin": m.geom_margin, "geom_gap": m.geom_gap, "geom_solref": m.geom_solref,
            "geom_solimp": m.geom_solimp, "geom_solmix": m.geom_solmix, "geom_priority": m.geom_priority,
            "geom_contype": m.geom_contype, "geom_conaffinity": m.geom_conaffinity,
            "exclude_pairs": m.exclude_pairs, "body_weldid": m.body_weldid, "body_invweight0": m.body_invweight0,
            "sites_bodyid": m.sites_bodyid, "sites_pos": m.sites_pos, "sites_quat": m.sites_quat,
            "sites_type": m.sites_type, "sites_size": m.sites_size,
            "touch_site": m.touch_site, "force_site": m.force_site, "appendage_site": np.array(app, dtype=np.int32),
            "solver_opt": np.array([m.cone_elliptic, m.noslip_iterations, m.impratio], dtype=np.float64),
        })
        # static candidate list of the device kernels' broad phase (flybody_amd/model/reach.py)
        from .reach import candidate_pairs
        cand, _ = candidate_pairs(m)
        t["cand_g1"] = np.array([c[0] for c in cand], dtype=np.int32)
        t["cand_g2"] = np.array([c[1] for c in cand], dtype=np.int32)
    if L is None:
        return t
    t.update({
        # welded links
        "link_body": L.link_body, "link_parent": L.link_parent, "link_pos": L.link_pos,
        "link_quat": L.link_quat, "link_mass": L.link_mass, "link_ipos": L.link_ipos,
        "link_iquat": L.link_iquat, "link_inertia": L.link_inertia, "link_dofadr": L.link_dofadr,
        "link_dofnum": L.link_dofnum, "link_depth": L.link_depth, "link_subtree": L.link_subtree,
        "body_link": L.body_link,
        "fbox_link": L.fbox_link, "fbox_pos": L.fbox_pos, "fbox_mat": L.fbox_mat, "fbox_box": L.fbox_box,
        "fell_link": L.fell_link, "fell_pos": L.fell_pos, "fell_mat": L.fell_mat,
        "fell_size": L.fell_size, "fell_coef": L.fell_coef,
    })
    if with_collision:
        t.update({"cgeom_link": L.cgeom_link, "cgeom_pos": L.cgeom_pos, "cgeom_quat": L.cgeom_quat})
    return t


def names_tensor(names: list) -> np.ndarray:
    return np.frombuffer("\n".join(names).encode(), dtype=np.uint8).astype(np.int32)
