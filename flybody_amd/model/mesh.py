"""Triangle-mesh loading and solid-body mass properties.

Used by the model compiler to turn the fruit-fly's visual meshes (the only mass-bearing geoms
besides two boxes, reference `fruitfly/assets/fruitfly.xml:14,37,80,94` densities on
`type="mesh"` geoms) into volume / centre of mass / inertia tensor.

Two integration rules are provided, mirroring the two rules MuJoCo has shipped for mesh
inertia (the reference does not pin a MuJoCo version, `requirements.txt:3`):

* ``exact``  - signed tetrahedra (origin, v0, v1, v2); exact for any closed, consistently
  oriented surface.
* ``legacy`` - unsigned pyramids from the area-weighted face centroid to every face; exact only
  for shapes that are star-convex about that centroid.
"""

from __future__ import annotations

import struct

import numpy as np


def load_obj(path: str) -> tuple[np.ndarray, np.ndarray]:
    """Returns (vertices (V,3) f64, faces (F,3) int) of a Wavefront OBJ; polygons are fanned."""
    verts: list[list[float]] = []
    faces: list[list[int]] = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float64), np.asarray(faces, dtype=np.int64)


def load_msh(path: str) -> tuple[np.ndarray, np.ndarray]:
    """Legacy MuJoCo binary .msh: int32 header (nvert, nnormal, ntexcoord, nface), then float32
    vertices, normals, texcoords and int32 faces."""
    with open(path, "rb") as f:
        data = f.read()
    nvert, nnormal, ntex, nface = struct.unpack("<4i", data[:16])
    expect = 16 + 4 * (3 * nvert + 3 * nnormal + 2 * ntex + 3 * nface)
    if expect != len(data):
        raise ValueError(f"{path}: size {len(data)} does not match header ({expect})")
    off = 16
    verts = np.frombuffer(data, dtype="<f4", count=3 * nvert, offset=off).reshape(-1, 3)
    off += 4 * (3 * nvert + 3 * nnormal + 2 * ntex)
    faces = np.frombuffer(data, dtype="<i4", count=3 * nface, offset=off).reshape(-1, 3)
    return verts.astype(np.float64), faces.astype(np.int64)


def load_mesh(path: str) -> tuple[np.ndarray, np.ndarray]:
    if path.endswith(".msh"):
        return load_msh(path)
    return load_obj(path)


def _tet_second_moments(a: np.ndarray, b: np.ndarray, c: np.ndarray, vol: np.ndarray) -> np.ndarray:
    """Sum over tetrahedra (0, a, b, c) of the second-moment matrix  integral(x x^T) dV."""
    s = a + b + c
    # integral over tet with one vertex at the origin: V/20 * (a a^T + b b^T + c c^T + s s^T)
    m = (
        np.einsum("n,ni,nj->ij", vol, a, a)
        + np.einsum("n,ni,nj->ij", vol, b, b)
        + np.einsum("n,ni,nj->ij", vol, c, c)
        + np.einsum("n,ni,nj->ij", vol, s, s)
    ) / 20.0
    return m


def mass_properties(verts: np.ndarray, faces: np.ndarray, rule: str = "exact"):
    """Unit-density solid properties of a closed triangle surface.

    Returns (volume, com (3,), inertia (3,3) about the com, in mesh coordinates).
    """
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    if rule == "exact":
        ref = np.zeros(3)
    elif rule == "legacy":
        n = np.cross(v1 - v0, v2 - v0)
        area = 0.5 * np.linalg.norm(n, axis=1)
        ref = (area[:, None] * (v0 + v1 + v2) / 3.0).sum(0) / area.sum()
    else:
        raise ValueError(rule)
    a, b, c = v0 - ref, v1 - ref, v2 - ref
    vol = np.einsum("ni,ni->n", a, np.cross(b, c)) / 6.0
    if rule == "legacy":
        vol = np.abs(vol)
    volume = vol.sum()
    com = (vol[:, None] * (a + b + c) / 4.0).sum(0) / volume + ref
    if rule == "legacy":
        # second pass: pyramids from the centre of mass, unsigned
        a, b, c = v0 - com, v1 - com, v2 - com
        vol2 = np.abs(np.einsum("ni,ni->n", a, np.cross(b, c)) / 6.0)
        sm = _tet_second_moments(a, b, c, vol2)
    else:
        sm = _tet_second_moments(a, b, c, vol)
        d = com - ref
        sm = sm - volume * np.outer(d, d)
    inertia = np.trace(sm) * np.eye(3) - sm
    return float(volume), com, inertia
