"""Small float64 numpy evaluation of position-stage quantities on a `CompiledModel`.

The compiler needs the joint-space inertia at `qpos0` (for MJCF `springdamper` and the
constraint inverse weights), so it carries its own kinematics + composite-rigid-body pass.
It is also a third, independent statement of those two stages that the tests use to
cross-check the C oracle.
"""

from __future__ import annotations

import numpy as np

from . import quat as Q

JNT_FREE, JNT_BALL, JNT_HINGE = 0, 1, 3


def kinematics(m, qpos):
    nb = m.nbody
    xpos = np.zeros((nb, 3))
    xquat = np.tile([1.0, 0, 0, 0], (nb, 1))
    xanchor = np.zeros((m.njnt, 3))
    xaxis = np.zeros((m.njnt, 3))
    for i in range(1, nb):
        p = m.body_parentid[i]
        adr, num = m.body_jntadr[i], m.body_jntnum[i]
        if num == 1 and m.jnt_type[adr] == JNT_FREE:
            qa = m.jnt_qposadr[adr]
            xpos[i] = qpos[qa : qa + 3]
            xquat[i] = Q.normalize(qpos[qa + 3 : qa + 7])
            xanchor[adr] = xpos[i]
            xaxis[adr] = [0, 0, 1]
        else:
            pos = xpos[p] + Q.rot(m.body_pos[i], xquat[p])
            quat = Q.mul(xquat[p], m.body_quat[i])
            for j in range(adr, adr + num):
                if m.jnt_type[j] == JNT_FREE:
                    # free joint not alone on the body does not occur in this model
                    raise NotImplementedError
                xanchor[j] = Q.rot(m.jnt_pos[j], quat) + pos
                xaxis[j] = Q.rot(m.jnt_axis[j], quat)
                if m.jnt_type[j] == JNT_BALL:
                    qa = m.jnt_qposadr[j]
                    quat = Q.mul(quat, Q.normalize(qpos[qa : qa + 4]))
                else:
                    ang = qpos[m.jnt_qposadr[j]] - m.qpos0[m.jnt_qposadr[j]]
                    quat = Q.mul(quat, Q.axis_angle(m.jnt_axis[j], ang))
                pos = xanchor[j] - Q.rot(m.jnt_pos[j], quat)
            xpos[i], xquat[i] = pos, Q.normalize(quat)
    xipos = np.array([xpos[i] + Q.rot(m.body_ipos[i], xquat[i]) for i in range(nb)])
    ximat = np.array([Q.to_mat(Q.mul(xquat[i], m.body_iquat[i])) for i in range(nb)])
    return dict(xpos=xpos, xquat=xquat, xipos=xipos, ximat=ximat, xanchor=xanchor, xaxis=xaxis)


def subtree_com(m, k):
    nb = m.nbody
    mass = m.body_mass.copy()
    mom = m.body_mass[:, None] * k["xipos"]
    for i in range(nb - 1, 0, -1):
        p = m.body_parentid[i]
        mass[p] += mass[i]
        mom[p] += mom[i]
    com = np.where(mass[:, None] > 1e-15, mom / np.maximum(mass[:, None], 1e-300), k["xipos"])
    return com, mass


def mass_matrix(m, qpos):
    """Dense joint-space inertia via the composite-rigid-body construction in the
    world-aligned frame centred at the root subtree's CoM."""
    k = kinematics(m, qpos)
    com, _ = subtree_com(m, k)
    c0 = com[1]
    nb, nv = m.nbody, m.nv
    cdof = _cdof(m, k, c0)
    # 6x6 spatial inertia (angular first) of every body about c0, world axes
    I6 = np.zeros((nb, 6, 6))
    for i in range(1, nb):
        R = k["ximat"][i]
        Ic = R @ np.diag(m.body_inertia[i]) @ R.T
        d = k["xipos"][i] - c0
        dx = np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        mass = m.body_mass[i]
        I6[i, :3, :3] = Ic - mass * dx @ dx
        I6[i, :3, 3:] = mass * dx
        I6[i, 3:, :3] = -mass * dx
        I6[i, 3:, 3:] = mass * np.eye(3)
    for i in range(nb - 1, 0, -1):
        I6[m.body_parentid[i]] += I6[i]
    M = np.zeros((nv, nv))
    for i in range(nv):
        f = I6[m.dof_bodyid[i]] @ cdof[i]
        j = i
        while j >= 0:
            M[i, j] = M[j, i] = cdof[j] @ f
            j = m.dof_parentid[j]
        M[i, i] += m.dof_armature[i]
    return M


def _cdof(m, k, c0):
    """Motion axes of every dof (angular; linear at point c0), world frame."""
    cdof = np.zeros((m.nv, 6))
    for j in range(m.njnt):
        b = m.jnt_bodyid[j]
        d = m.jnt_dofadr[j]
        off = c0 - k["xanchor"][j]
        if m.jnt_type[j] == JNT_FREE:
            for a in range(3):
                cdof[d + a, 3 + a] = 1
            R = Q.to_mat(k["xquat"][b])
            for a in range(3):
                ax = R[:, a]
                cdof[d + 3 + a, :3] = ax
                cdof[d + 3 + a, 3:] = np.cross(ax, off)
        elif m.jnt_type[j] == JNT_BALL:
            R = Q.to_mat(k["xquat"][b])
            for a in range(3):
                ax = R[:, a]
                cdof[d + a, :3] = ax
                cdof[d + a, 3:] = np.cross(ax, off)
        else:
            ax = k["xaxis"][j]
            cdof[d, :3] = ax
            cdof[d, 3:] = np.cross(ax, off)
    return cdof


def body_jacobians(m, qpos):
    """Translational (at the body CoM) and rotational Jacobians of every body: (nbody, 3, nv) each."""
    k = kinematics(m, qpos)
    c0 = np.zeros(3)
    cdof = _cdof(m, k, c0)
    nb, nv = m.nbody, m.nv
    jacp = np.zeros((nb, 3, nv))
    jacr = np.zeros((nb, 3, nv))
    for i in range(1, nb):
        b = i
        while b > 0 and m.body_dofnum[b] == 0:
            b = m.body_parentid[b]
        if b <= 0:
            continue
        d = m.body_dofadr[b] + m.body_dofnum[b] - 1
        while d >= 0:
            w, v0 = cdof[d, :3], cdof[d, 3:]
            jacr[i][:, d] = w
            jacp[i][:, d] = v0 + np.cross(w, k["xipos"][i] - c0)
            d = m.dof_parentid[d]
    return jacp, jacr
