"""Scenes of 10^4..10^6 triangles without scene files of that size: the gems snapshot with scaled copies of its gem meshes scattered
through the box (real shape statistics: small closed facetted objects). Used by tests/test_gpu_scene_update.py, tests/test_host_lbvh.py,
tools/bvh_build_bench.py and bench.py --workload gems1m."""
import numpy as np


def replicate_gems(etx, snapshot_path, copies, seed=9):
    """-> SceneSnapshot of `snapshot_path` (a cornell_gems_* snapshot) with `copies` extra copies of its conductor / dielectric
    triangles: 2 892 + copies * 2 880 triangles. Emissive triangles keep their indices (emitter instances name them)."""
    snap = etx.SceneSnapshot(snapshot_path)
    vertices, triangles, to_emitter = snap.vertices().copy(), snap.triangles().copy(), snap.triangle_to_emitter().copy()
    classes = snap.material_classes()
    gem = np.nonzero(np.isin(classes[triangles[:, 3]], (3, 4)))[0]
    rng = np.random.default_rng(seed)
    new_vertices, new_triangles = [vertices], [triangles]
    base = vertices.shape[0]
    corner_rows = vertices[triangles[gem, 0:3].reshape(-1).astype(np.int64)]  # (3 * G, 14): unshared copies
    centre = corner_rows[:, 0:3].mean(axis=0)
    for _ in range(copies):
        scale = np.float32(rng.uniform(0.1, 0.3))
        offset = np.float32([rng.uniform(-0.8, 0.8), rng.uniform(0.15, 1.8), rng.uniform(-0.8, 0.8)])
        rows = corner_rows.copy()
        rows[:, 0:3] = (rows[:, 0:3] - centre) * scale + offset
        tri = triangles[gem].copy()
        tri[:, 0:3] = base + np.arange(3 * len(gem), dtype=np.uint32).reshape(-1, 3)
        new_vertices.append(rows)
        new_triangles.append(tri)
        base += rows.shape[0]
    all_triangles = np.concatenate(new_triangles)
    snap.replace_geometry(np.concatenate(new_vertices), all_triangles, np.concatenate([to_emitter, np.full(all_triangles.shape[0] - to_emitter.shape[0], 0xFFFFFFFF, dtype=np.uint32)]))
    return snap


def _icosphere(subdivisions):
    """-> (unit vertices float64 (V, 3), faces int64 (F, 3)) of a subdivided icosahedron: 20 * 4^subdivisions outward-wound faces."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], dtype=np.float64)
    verts /= np.linalg.norm(verts, axis=1, keepdims=True)
    faces = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                      (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)], dtype=np.int64)
    for _ in range(subdivisions):
        edges = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
        keys = np.sort(edges, axis=1)
        unique, inverse = np.unique(keys[:, 0] * (1 << 32) + keys[:, 1], return_inverse=True)
        mid = verts[unique >> 32] + verts[unique & 0xffffffff]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = verts.shape[0]
        verts = np.concatenate([verts, mid])
        f = faces.shape[0]
        ab, bc, ca = base + inverse[0:f], base + inverse[f:2 * f], base + inverse[2 * f:3 * f]
        a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
        faces = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    return verts, faces


def _blob(subdivisions, center, radius, displace):
    """Closed blob: the icosphere's vertices moved radially by `displace(unit vectors) -> factors`; -> (vertex rows float32 (V, 14) in
    the layout of etx::Vertex = pos, nrm, tan, btn, tex; faces (F, 3) local indices; geometric normals float32 (F, 3))."""
    unit, faces = _icosphere(subdivisions)
    pos = np.asarray(center, dtype=np.float64) + radius * unit * displace(unit)[:, None]
    e1, e2 = pos[faces[:, 1]] - pos[faces[:, 0]], pos[faces[:, 2]] - pos[faces[:, 0]]
    fn = np.cross(e1, e2)  # length = 2 x area: the vertex normal is the area-weighted mean of its faces
    nrm = np.zeros_like(pos)
    for k in range(3):
        np.add.at(nrm, faces[:, k], fn)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    helper = np.where(np.abs(nrm[:, 1:2]) < 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    tan = np.cross(helper, nrm)
    tan /= np.linalg.norm(tan, axis=1, keepdims=True)
    btn = np.cross(nrm, tan)
    rows = np.zeros((pos.shape[0], 14), dtype=np.float32)
    rows[:, 0:3], rows[:, 3:6], rows[:, 6:9], rows[:, 9:12] = pos, nrm, tan, btn
    return rows, faces, (fn / np.linalg.norm(fn, axis=1, keepdims=True)).astype(np.float32)


def sss_dragon(etx, snapshot_path, subdivisions=(6, 5)):
    """-> SceneSnapshot of BASELINE configs[3]'s shape: `snapshot_path` (the cornell_sss_* box: two subsurface materials on the two
    boxes) with the boxes swapped for closed, non-convex blob meshes (thick lobes, thin necks - the statistics of a figurine, the
    reference tree ships no mesh): 20 * 4^6 + 20 * 4^5 = 102 400 triangles by default. Emitter instances name triangles by index; they
    are re-pointed."""
    snap = etx.SceneSnapshot(snapshot_path)
    vertices, triangles, to_emitter = snap.vertices().copy(), snap.triangles().copy(), snap.triangle_to_emitter().copy()
    sss_materials = np.nonzero(snap.materials()[:, 26] != 0)[0]  # etx_abi_material::subsurface.cls
    if len(sss_materials) != 2:
        raise ValueError("sss_dragon: expected the two subsurface materials of the cornell_sss box, found %d" % len(sss_materials))

    def knobbly(v):
        x, y, z = v[:, 0], v[:, 1], v[:, 2]
        return 1.0 + 0.22 * np.sin(4.0 * x + 1.0) * np.sin(5.0 * y) * np.sin(3.0 * z + 2.0) + 0.10 * np.sin(9.0 * x) * np.sin(7.0 * y + 0.5) + 0.06 * np.sin(13.0 * z + 1.5)

    def lumpy(v):
        x, y, z = v[:, 0], v[:, 1], v[:, 2]
        return 1.0 + 0.15 * np.sin(3.0 * x + 0.3) * np.sin(4.0 * z + 1.0) + 0.12 * np.sin(6.0 * y + 2.0) * np.sin(5.0 * x)

    keep = ~np.isin(triangles[:, 3], sss_materials)
    new_index = np.cumsum(keep) - 1
    new_vertices, new_triangles = [vertices], [triangles[keep]]
    base = vertices.shape[0]
    for material, level, center, radius, displace in ((sss_materials[0], subdivisions[0], (0.35, 0.42, 0.40), 0.36, knobbly), (sss_materials[1], subdivisions[1], (-0.38, 0.58, -0.30), 0.46, lumpy)):
        rows, faces, geo_n = _blob(level, center, radius, displace)
        tri = np.zeros((faces.shape[0], 8), dtype=np.uint32)
        tri[:, 0:3] = (faces + base).astype(np.uint32)
        tri[:, 3] = material
        tri[:, 4:7] = geo_n.view(np.uint32)
        new_vertices.append(rows)
        new_triangles.append(tri)
        base += rows.shape[0]
    all_triangles = np.concatenate(new_triangles)
    kept_emitters = to_emitter[keep]
    snap.replace_geometry(np.concatenate(new_vertices), all_triangles, np.concatenate([kept_emitters, np.full(all_triangles.shape[0] - kept_emitters.shape[0], 0xFFFFFFFF, dtype=np.uint32)]))
    emitters = snap.emitter_instances()
    area = emitters[:, 2] != 0xFFFFFFFF  # etx_abi_emitter::triangle_index
    emitters[area, 2] = new_index[emitters[area, 2]].astype(np.uint32)
    return snap


def sss_sheets(etx, snapshot_path, slabs=6, fill=0.6):
    """-> SceneSnapshot of `snapshot_path` (a cornell_sss* box) whose first subsurface object is cut into `slabs` separate closed slabs stacked
    along y (each `fill` of its share of the height thick): a probe ray of the Christensen-Burley gather that runs along the stack meets
    2 * slabs surfaces of ONE material - more than the eight hits Raytracing::continuous_trace keeps (rt.cxx:412-421 keeps the first eight
    in TRAVERSAL order; tests/test_gpu_sssmesh.py::test_more_than_eight_hits_along_a_probe)."""
    snap = etx.SceneSnapshot(snapshot_path)
    vertices, triangles, to_emitter = snap.vertices().copy(), snap.triangles().copy(), snap.triangle_to_emitter().copy()
    sss_materials = np.nonzero(snap.materials()[:, 26] != 0)[0]  # etx_abi_material::subsurface.cls
    if len(sss_materials) == 0:
        raise ValueError("sss_sheets: no subsurface material in the scene")
    material = sss_materials[0]
    mine = triangles[:, 3] == material
    corners = vertices[triangles[mine, 0:3].reshape(-1).astype(np.int64), 0:3]
    lo, hi = corners.min(axis=0), corners.max(axis=0)
    keep = ~mine
    new_index = np.cumsum(keep) - 1
    # one axis-aligned box = 6 faces x 4 vertices with face normals (flat shading), 12 outward-wound triangles
    faces = (((0, -1), (1, 2)), ((0, +1), (2, 1)), ((1, -1), (2, 0)), ((1, +1), (0, 2)), ((2, -1), (0, 1)), ((2, +1), (1, 0)))
    rows, tris = [], []
    base = vertices.shape[0]
    for k in range(slabs):
        y0 = lo[1] + (hi[1] - lo[1]) * k / slabs
        y1 = y0 + (hi[1] - lo[1]) * fill / slabs
        blo, bhi = np.array([lo[0], y0, lo[2]]), np.array([hi[0], y1, hi[2]])
        for (axis, sign), (ua, va) in faces:
            n = np.zeros(3)
            n[axis] = sign
            t = np.zeros(3)
            t[ua] = 1.0
            b = np.cross(n, t)
            quad = []
            for (su, sv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = np.where(np.arange(3) == axis, bhi if sign > 0 else blo, 0.0)
                p[ua] = bhi[ua] if su else blo[ua]
                p[va] = bhi[va] if sv else blo[va]
                quad.append(p)
            # outward winding: (q1 - q0) x (q3 - q0) must point along n
            if np.dot(np.cross(quad[1] - quad[0], quad[3] - quad[0]), n) < 0.0:
                quad = [quad[0], quad[3], quad[2], quad[1]]
            for i, p in enumerate(quad):
                row = np.zeros(14, dtype=np.float32)
                row[0:3], row[3:6], row[6:9], row[9:12] = p, n, t, b
                row[12:14] = ((0, 0), (1, 0), (1, 1), (0, 1))[i]
                rows.append(row)
            for a, bq, c in ((0, 1, 2), (0, 2, 3)):
                tri = np.zeros(8, dtype=np.uint32)
                tri[0:3] = base + np.array([a, bq, c], dtype=np.uint32)
                tri[3] = material
                tri[4:7] = n.astype(np.float32).view(np.uint32)
                tris.append(tri)
            base += 4
    all_triangles = np.concatenate([triangles[keep], np.array(tris, dtype=np.uint32)])
    kept_emitters = to_emitter[keep]
    snap.replace_geometry(np.concatenate([vertices, np.array(rows, dtype=np.float32)]), all_triangles,
                          np.concatenate([kept_emitters, np.full(all_triangles.shape[0] - kept_emitters.shape[0], 0xFFFFFFFF, dtype=np.uint32)]))
    emitters = snap.emitter_instances()
    area = emitters[:, 2] != 0xFFFFFFFF
    emitters[area, 2] = new_index[emitters[area, 2]].astype(np.uint32)
    return snap

