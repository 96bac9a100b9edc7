"""ORACLE / TEST INFRASTRUCTURE ONLY.

numpy restatement of the closest-hit query `Raytracing::trace` (sources/etx/rt/rt.cxx:428-466) by brute force over
every triangle: skip Material::Class::Void (rt.cxx:441-444), keep the closest accepted hit in [tmin, tmax], report
(u, v, t, triangle) with Embree's barycentric convention (u, v weight vertices 1 and 2, rt.cxx:352-353 + math.hxx:764).
The stochastic alpha test (scene_bsdf.hxx:128-144) is the identity for opaque materials (opacity 1, no alpha texture);
for alpha-tested surfaces `exclude` gives the second outcome (see closest_hits). "parity unpinned": the reference holds no vectors for this boundary;
the restatement is cross-checked against the reference-based oracle binary through the rendered images instead.
"""
import numpy as np

ETX_MAT_VOID = 10


def closest_hits(snapshot, rays, exclude=()):
    """rays: (n, 8) float32 {ox,oy,oz,tmin,dx,dy,dz,tmax} -> (n, 4) float64 {u, v, t, triangle or -1}.
    `exclude`: triangle indices treated as absent - the two outcomes of the stochastic alpha test (scene_bsdf.hxx:128-144) for
    an alpha-tested surface are "hit it" (exclude nothing) and "pass through" (exclude its triangles); the test checks that
    every device hit is one of the two and that the pass-through frequency matches opacity x texture alpha."""
    vertices = snapshot.vertices()[:, 0:3].astype(np.float64)
    triangles = snapshot.triangles()
    material_class = snapshot.material_classes()
    rays = np.asarray(rays, dtype=np.float64).reshape(-1, 8)
    n = rays.shape[0]
    o, d = rays[:, 0:3], rays[:, 4:7]
    best_t = rays[:, 7].copy()
    out = np.full((n, 4), -1.0)
    out[:, 0:2] = 0.0
    for ti in range(triangles.shape[0]):
        i0, i1, i2, mat = (int(x) for x in triangles[ti, 0:4])
        if (material_class[mat] == ETX_MAT_VOID) or (ti in exclude):
            continue
        v0, e1, e2 = vertices[i0], vertices[i1] - vertices[i0], vertices[i2] - vertices[i0]
        p = np.cross(d, e2)
        det = p @ e1
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            s = o - v0
            u = np.einsum("ij,ij->i", s, p) * inv
            q = np.cross(s, e1)
            v = np.einsum("ij,ij->i", d, q) * inv
            t = (q @ e2) * inv
        ok = (det != 0.0) & (u >= 0.0) & (u <= 1.0) & (v >= 0.0) & (u + v <= 1.0) & (t >= rays[:, 3]) & (t <= best_t)
        best_t = np.where(ok, t, best_t)
        out[ok, 0], out[ok, 1], out[ok, 2], out[ok, 3] = u[ok], v[ok], t[ok], ti
    return out
