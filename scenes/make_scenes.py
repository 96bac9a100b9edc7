#!/usr/bin/env python3
"""Authors the benchmark / parity scenes as OBJ + MTL + JSON files that the reference's own loader reads.

The reference snapshot ships no geometry (`bin/assets/cornellbox/cornellbox.obj` is a missing blob, see
/root/reference/.MISSING_LARGE_BLOBS); only the Cornell camera (`cornellbox.json:9-35`) and materials
(`cornellbox.mtl`) survive. This script rebuilds the Cornell box for that camera (box x in [-1,1], y in [0,2],
z in [-1,1], open towards +z, camera at (0,1,3.82) looking down -z, fov 39.5978 deg) with those material
definitions, in two flavours:

  cornell_classic : diffuse walls / short box, delta silver conductor tall box, blackbody area light
  cornell_full    : + `et::env`, `et::dir`, and the `fog` boundary box with the `fog__vol` scattering medium
                    (the full surviving cornellbox.mtl)

and one JSON per configuration of BASELINE.json (resolution / spp), plus small variants for tests.
Run: python3 scenes/make_scenes.py   (writes scenes/cornell/*)
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "cornell")


def cornell_to_scene(X, Y, Z):
    """Classic Cornell box data (x 0..556, y 0..548.8, z 0..559.2, camera at z=-800 looking +z) -> our frame."""
    return (-(X - 278.0) / 278.0, Y / 274.4, 1.0 - Z / 279.6)


class Mesh:
    def __init__(self):
        self.v = []
        self.vn = []
        self.vt = []      # texture coordinates, only for quads created with uv=True
        self.groups = []  # (material, [(vi, ni) * 3]) or (material, [(vi, ni, ti) * 3])

    def quad(self, mat, pts, normal_hint=None, uv=False):
        """pts: 4 points, counter-clockwise seen from the side the normal points to. uv: corners get (0,0) (1,0) (1,1) (0,1)."""
        base = len(self.v)
        self.v.extend(pts)
        ax = [pts[1][i] - pts[0][i] for i in range(3)]
        bx = [pts[3][i] - pts[0][i] for i in range(3)]
        n = [ax[1] * bx[2] - ax[2] * bx[1], ax[2] * bx[0] - ax[0] * bx[2], ax[0] * bx[1] - ax[1] * bx[0]]
        ln = sum(c * c for c in n) ** 0.5
        n = [c / ln for c in n]
        if normal_hint is not None and sum(n[i] * normal_hint[i] for i in range(3)) < 0:
            raise ValueError("winding does not match the requested normal for %s" % mat)
        self.vn.append(n)
        ni = len(self.vn)
        if uv:
            tb = len(self.vt)
            self.vt.extend([(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)])
            self.groups.append((mat, [(base + 1, ni, tb + 1), (base + 2, ni, tb + 2), (base + 3, ni, tb + 3)]))
            self.groups.append((mat, [(base + 1, ni, tb + 1), (base + 3, ni, tb + 3), (base + 4, ni, tb + 4)]))
            return
        self.groups.append((mat, [(base + 1, ni), (base + 2, ni), (base + 3, ni)]))
        self.groups.append((mat, [(base + 1, ni), (base + 3, ni), (base + 4, ni)]))

    def box(self, mat, top, height_y, uv=False):
        """top: 4 points of the top face (counter-clockwise seen from above); sides go down to y=0."""
        bottom = [(p[0], 0.0, p[2]) for p in top]
        top = [(p[0], height_y, p[2]) for p in top]
        self.quad(mat, top, (0, 1, 0), uv=uv)
        for i in range(4):
            j = (i + 1) % 4
            self.quad(mat, [top[j], top[i], bottom[i], bottom[j]], uv=uv)

    def aabb(self, mat, lo, hi):
        """closed axis aligned box with outward normals"""
        x0, y0, z0 = lo
        x1, y1, z1 = hi
        self.quad(mat, [(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)], (0, -1, 0))
        self.quad(mat, [(x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)], (0, 1, 0))
        self.quad(mat, [(x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)], (-1, 0, 0))
        self.quad(mat, [(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)], (1, 0, 0))
        self.quad(mat, [(x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)], (0, 0, -1))
        self.quad(mat, [(x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], (0, 0, 1))

    def icosphere(self, mat, center, radius, subdivisions, flat=False):
        """Triangle mesh of a sphere (20 * 4^subdivisions faces), smooth vertex normals unless `flat`."""
        t = (1.0 + 5.0 ** 0.5) / 2.0
        verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
        faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                 (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]

        def unit(v):
            ln = sum(c * c for c in v) ** 0.5
            return tuple(c / ln for c in v)

        verts = [unit(v) for v in verts]
        for _ in range(subdivisions):
            cache, new_faces = {}, []

            def mid(a, b):
                key = (min(a, b), max(a, b))
                if key not in cache:
                    verts.append(unit(tuple((verts[a][i] + verts[b][i]) * 0.5 for i in range(3))))
                    cache[key] = len(verts) - 1
                return cache[key]

            for a, b, c in faces:
                ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
                new_faces += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
            faces = new_faces
        vbase, nbase = len(self.v), len(self.vn)
        for v in verts:
            self.v.append(tuple(center[i] + radius * v[i] for i in range(3)))
            if not flat:
                self.vn.append(v)
        for a, b, c in faces:
            if flat:
                n = unit(tuple(verts[a][i] + verts[b][i] + verts[c][i] for i in range(3)))
                self.vn.append(n)
                ni = len(self.vn)
                self.groups.append((mat, [(vbase + a + 1, ni), (vbase + b + 1, ni), (vbase + c + 1, ni)]))
            else:
                self.groups.append((mat, [(vbase + a + 1, nbase + a + 1), (vbase + b + 1, nbase + b + 1), (vbase + c + 1, nbase + c + 1)]))

    def write(self, path, mtllib):
        with open(path, "w") as f:
            f.write("# Cornell box rebuilt for etx-tracer's surviving camera/materials (scenes/make_scenes.py)\n")
            f.write("mtllib %s\n" % mtllib)
            for p in self.v:
                f.write("v %.6f %.6f %.6f\n" % tuple(p))
            for n in self.vn:
                f.write("vn %.6f %.6f %.6f\n" % tuple(n))
            for t in self.vt:
                f.write("vt %.6f %.6f\n" % tuple(t))
            current = None
            for mat, idx in self.groups:
                if mat != current:
                    f.write("usemtl %s\n" % mat)
                    current = mat
                f.write("f " + " ".join(("%d/%d/%d" % (c[0], c[2], c[1])) if len(c) == 3 else ("%d//%d" % c) for c in idx) + "\n")


def build_mesh(with_fog, spheres=False, sss_meshes=False, box_uv=False):
    m = Mesh()
    # room, normals pointing inside
    m.quad("floor", [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)], (0, 1, 0))
    m.quad("ceiling", [(-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)], (0, -1, 0))
    m.quad("frontWall", [(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)], (0, 0, 1))
    m.quad("leftWall", [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)], (1, 0, 0))
    m.quad("rightWall", [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)], (-1, 0, 0))
    # light: classic 130 x 105 quad just below the ceiling, facing down
    lx0, _, lz0 = cornell_to_scene(343.0, 0, 227.0)
    lx1, _, lz1 = cornell_to_scene(213.0, 0, 332.0)
    ly = 1.98
    m.quad("light", [(lx0, ly, lz1), (lx1, ly, lz1), (lx1, ly, lz0), (lx0, ly, lz0)], (0, -1, 0))
    # boxes
    short_top = [cornell_to_scene(*p) for p in [(130.0, 165.0, 65.0), (82.0, 165.0, 225.0), (240.0, 165.0, 272.0), (290.0, 165.0, 114.0)]]
    tall_top = [cornell_to_scene(*p) for p in [(423.0, 330.0, 247.0), (265.0, 330.0, 296.0), (314.0, 330.0, 456.0), (472.0, 330.0, 406.0)]]

    def ccw_from_above(pts):
        # signed area in the xz plane; counter-clockwise seen from +y means the face normal is +y
        a = 0.0
        for i in range(4):
            j = (i + 1) % 4
            a += pts[i][2] * pts[j][0] - pts[j][2] * pts[i][0]
        return pts if a > 0 else list(reversed(pts))

    if sss_meshes:
        # subsurface objects as MESHES: 20 480 + 1 280 triangles -> the material-filtered traversal of the walks and the inline
        # traversal of the shade kernels run on the tree (the box scenes are swept linearly), at a depth whose stack bound exceeds
        # the per-lane LDS stack
        m.icosphere("shortBox", (0.35, 0.36, 0.40), 0.36, 5)
        m.icosphere("tallBox", (-0.38, 0.50, -0.30), 0.50, 3)
    elif spheres:
        # BVH-sized geometry (2 x 1280 + 320 triangles): a faceted "gem" and two smooth spheres instead of the boxes
        m.icosphere("shortBox", (0.35, 0.33, 0.45), 0.33, 3, flat=True)
        m.icosphere("tallBox", (-0.40, 0.45, -0.30), 0.45, 3)
        m.icosphere("gem", (-0.45, 0.22, 0.55), 0.22, 2, flat=True)
    else:
        m.box("shortBox", ccw_from_above(short_top), short_top[0][1], uv=box_uv)
        m.box("tallBox", ccw_from_above(tall_top), tall_top[0][1], uv=box_uv)
    if with_fog:
        m.aabb("fog", (-0.99, 0.01, -0.99), (0.99, 1.99, 0.99))
    return m


MTL_COMMON = """newmtl ceiling
material class diffuse
Kd 1.000 1.000 1.000
Pr 0.000
two_sided 1

newmtl floor
material class diffuse
Kd 1.000 1.000 1.000
Pr 0.000
two_sided 1

newmtl frontWall
material class diffuse
Kd 0.906 0.906 0.906
Pr 0.000
two_sided 1

newmtl leftWall
material class diffuse
Kd 1.000 0.000 0.000
Pr 0.000
two_sided 1

newmtl rightWall
material class diffuse
Kd 0.000 1.000 0.000
Pr 0.000
two_sided 1

newmtl shortBox
material class diffuse
Kd 0.906 0.906 0.906
Pr 0.000
two_sided 1

newmtl tallBox
material class conductor
int_ior silver
Ks 1.000 1.000 1.000
Pr 0.000
two_sided 1

"""

# Material coverage scenes (same geometry as the classic box): every BSDF class of scene_bsdf.hxx:56-107 appears once.
# "Pr r" sets alpha = r^2 (scene_representation.cxx:1731-1737).
MTL_MATERIALS_ROUGH = """newmtl ceiling
material class diffuse
diffuse 1
Kd 0.900 0.900 0.900
Pr 0.600
two_sided 1

newmtl floor
material class diffuse
diffuse 2
Kd 0.900 0.900 0.900
Pr 0.700
two_sided 1

newmtl frontWall
material class principled
Kd 0.800 0.800 0.700
Ks 1.000 1.000 1.000
metalness 0.500
transmission 0.000
Pr 0.600
two_sided 1

newmtl leftWall
material class velvet
Kd 0.800 0.050 0.050
Ks 0.600 0.600 0.600
Pr 0.700
two_sided 1

newmtl rightWall
material class plastic
Kd 0.050 0.800 0.050
Ks 1.000 1.000 1.000
int_ior 1.5
Pr 0.500
two_sided 1

newmtl shortBox
material class dielectric
Ks 1.000 1.000 1.000
Kt 0.950 0.950 1.000
int_ior 1.5
Pr 0.450
two_sided 1

newmtl tallBox
material class conductor
int_ior gold
Ks 1.000 1.000 1.000
Pr 0.500
thinfilm range 250 450 ior 1.33
two_sided 1

"""

MTL_MATERIALS_DELTA = """newmtl ceiling
material class diffuse
Kd 1.000 1.000 1.000
two_sided 1

newmtl floor
material class diffuse
Kd 1.000 1.000 1.000
two_sided 1

newmtl frontWall
material class diffuse
Kd 0.906 0.906 0.906
two_sided 1

newmtl leftWall
material class diffuse
Kd 1.000 0.000 0.000
two_sided 1

newmtl rightWall
material class diffuse
Kd 0.000 1.000 0.000
two_sided 1

newmtl shortBox
material class dielectric
Ks 1.000 1.000 1.000
Kt 1.000 1.000 1.000
int_ior 1.5
Pr 0.000
two_sided 1

newmtl tallBox
material class thinfilm
Ks 1.000 1.000 1.000
Kt 1.000 1.000 1.000
int_ior 1.5
thinfilm range 300 600 ior 1.33
two_sided 1

"""

# A scene without any non-area emitter gets a default atmosphere (sun + sky images) from the loader
# (scene_representation.cxx: "if (_private->data.emitter_profiles.empty())"); a zero-power directional emitter
# keeps the classic variant at "area light only" (weight 0 => never sampled, not in environment_emitters).
MTL_LIGHT_CLASSIC = """newmtl et::dir
direction 0.0 1.0 0.0
color 0.0 0.0 0.0
angular_diameter 0.0000

newmtl light
material class diffuse
Kd 0.000 0.000 0.000
emitter nblackbody 2700 scale 5.0000
two_sided 1

"""

MTL_LIGHT_FOG = """newmtl light
material class diffuse
Kd 0.000 0.000 0.000
ext_medium fog__vol
emitter nblackbody 2700 scale 5.0000
two_sided 1

"""

MTL_FULL_EXTRA = """newmtl et::env
color nblackbody 12000 scale 0.1000

newmtl et::dir
direction -0.0000 0.8660 0.5000
color nblackbody 5800 scale 1.0000
angular_diameter 0.0000

newmtl et::medium
id fog__vol
scattering 0.8000 0.8000 0.8000

newmtl fog
material class boundary
int_medium fog__vol

"""

CAMERA = {
    "class": "perspective",
    "origin": [0.0, 1.000000238418579, 3.819999933242798],
    "target": [0.0, 1.000000238418579, -6.179999351501465],
    "up": [0.0, 0.9999999403953552, -0.0],
    "fov": 39.597755335771296,
    "lens-radius": 0.0,
    "focal-distance": 0.0,
    "clip-near": 0.10000000149011612,
    "clip-far": 100.0,
}


def write_png(path, rgba):
    """8-bit RGBA PNG (rows top to bottom), written by hand: no image library needed where the scenes are regenerated."""
    import struct
    import zlib
    h, w = len(rgba), len(rgba[0])
    raw = b"".join(b"\x00" + bytes(c for px in row for c in px) for row in rgba)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def write_hdr(path, rgb):
    """Radiance .hdr (flat RGBE scanlines, rows top to bottom)."""
    import math
    h, w = len(rgb), len(rgb[0])
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        for row in rgb:
            for r, g, b in row:
                m = max(r, g, b)
                if m < 1e-32:
                    f.write(bytes([0, 0, 0, 0]))
                else:
                    mant, e = math.frexp(m)
                    scale = mant * 256.0 / m
                    f.write(bytes([min(255, int(r * scale)), min(255, int(g * scale)), min(255, int(b * scale)), e + 128]))


def build_extra_textures():
    """Images of the feature-coverage scenes (all procedural, a few KB each)."""
    import math
    tex = os.path.join(OUT, "textures")
    os.makedirs(tex, exist_ok=True)
    # leaf.png: colour stripes with an alpha cut-out pattern (discs) -> stochastic alpha test inside traversal
    n = 64
    rows = []
    for y in range(n):
        row = []
        for x in range(n):
            cx, cy = (x % 16) - 7.5, (y % 16) - 7.5
            hole = (cx * cx + cy * cy) < 20.0
            stripe = ((x // 8) % 2) == 0
            row.append((230 if stripe else 60, 200 if stripe else 120, 40 if stripe else 200, 0 if hole else 255))
        rows.append(row)
    write_png(os.path.join(tex, "leaf.png"), rows)
    # checker.png: opaque albedo texture for the back wall
    rows = [[(230, 230, 230, 255) if ((x // 8 + y // 8) % 2) == 0 else (90, 90, 200, 255) for x in range(64)] for y in range(64)]
    write_png(os.path.join(tex, "checker.png"), rows)
    # bumps.png: tangent-space normal map (sinusoidal ripples)
    rows = []
    for y in range(64):
        row = []
        for x in range(64):
            dx = 0.5 * math.cos(x / 64.0 * 8.0 * math.pi)
            dy = 0.5 * math.cos(y / 64.0 * 6.0 * math.pi)
            ln = math.sqrt(dx * dx + dy * dy + 1.0)
            row.append((int((dx / ln * 0.5 + 0.5) * 255), int((dy / ln * 0.5 + 0.5) * 255), int((1.0 / ln * 0.5 + 0.5) * 255), 255))
        rows.append(row)
    write_png(os.path.join(tex, "bumps.png"), rows)
    # sky.hdr: a dim gradient sky with one bright "sun" patch -> image environment map with 2-D sampling tables
    w, h = 64, 32
    rows = []
    for y in range(h):
        row = []
        for x in range(w):
            sun = math.exp(-(((x - 44) / 3.0) ** 2 + ((y - 7) / 2.5) ** 2))
            base = 0.25 + 0.35 * (1.0 - y / float(h))
            row.append((base * 0.8 + 60.0 * sun, base * 0.9 + 52.0 * sun, base * 1.2 + 40.0 * sun))
        rows.append(row)
    write_hdr(os.path.join(tex, "sky.hdr"), rows)
    # aperture.png: hexagonal aperture (lens image: sampling table, uniform)
    rows = []
    for y in range(32):
        row = []
        for x in range(32):
            px, py = (x + 0.5) / 16.0 - 1.0, (y + 0.5) / 16.0 - 1.0
            inside = max(abs(px), abs(px) * 0.5 + abs(py) * 0.8660254) < 0.9
            v = 255 if inside else 0
            row.append((v, v, v, 255))
        rows.append(row)
    write_png(os.path.join(tex, "aperture.png"), rows)


def write_json(name, geometry, materials, viewport, samples, max_path_length=1023, rr_start=6, spectral=False, camera=None):
    cam = dict(camera if camera is not None else CAMERA)
    cam["viewport"] = list(viewport)
    doc = {
        "geometry": geometry,
        "materials": materials,
        "samples": samples,
        "max-path-length": max_path_length,
        "random-termination-start": rr_start,
        "spectral": spectral,
        "force-tangents": False,
        "camera": cam,
    }
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(doc, f, indent=2)


def main():
    os.makedirs(OUT, exist_ok=True)
    build_mesh(False).write(os.path.join(OUT, "cornell_classic.obj"), "cornell_classic.mtl")
    build_mesh(True).write(os.path.join(OUT, "cornell_full.obj"), "cornell_full.mtl")
    with open(os.path.join(OUT, "cornell_classic.mtl"), "w") as f:
        f.write(MTL_COMMON + MTL_LIGHT_CLASSIC)
    with open(os.path.join(OUT, "cornell_full.mtl"), "w") as f:
        f.write(MTL_FULL_EXTRA + MTL_COMMON + MTL_LIGHT_FOG)

    with open(os.path.join(OUT, "cornell_rough.mtl"), "w") as f:
        f.write(MTL_MATERIALS_ROUGH + MTL_LIGHT_CLASSIC)
    with open(os.path.join(OUT, "cornell_glass.mtl"), "w") as f:
        f.write(MTL_MATERIALS_DELTA + MTL_LIGHT_CLASSIC)
    write_json("rough_test_128.json", "cornell_classic.obj", "cornell_rough.mtl", (128, 128), 64)
    write_json("glass_test_128.json", "cornell_classic.obj", "cornell_glass.mtl", (128, 128), 64)
    # spectral mode (one wavelength per path): the classic box, and the delta-glass box with a dispersive diamond
    # (BASELINE configs[2] family: dielectric-heavy SDS paths, spectral)
    with open(os.path.join(OUT, "cornell_diamond.mtl"), "w") as f:
        f.write(MTL_MATERIALS_DELTA.replace("newmtl shortBox\nmaterial class dielectric\nKs 1.000 1.000 1.000\nKt 1.000 1.000 1.000\nint_ior 1.5",
                                            "newmtl shortBox\nmaterial class dielectric\nKs 1.000 1.000 1.000\nKt 1.000 1.000 1.000\nint_ior diamond") + MTL_LIGHT_CLASSIC)
    # "gems": 2 900 triangles (BVH traversal instead of the flat sweep), dispersive diamond + glass + rough gold, spectral:
    # the shape of BASELINE configs[2] at test size
    build_mesh(False, spheres=True).write(os.path.join(OUT, "cornell_gems.obj"), "cornell_gems.mtl")
    with open(os.path.join(OUT, "cornell_gems.mtl"), "w") as f:
        f.write(MTL_COMMON.split("newmtl shortBox")[0] + """newmtl shortBox
material class dielectric
Ks 1.000 1.000 1.000
Kt 1.000 1.000 1.000
int_ior diamond
Pr 0.000
two_sided 1

newmtl tallBox
material class conductor
int_ior gold
Ks 1.000 1.000 1.000
Pr 0.350
two_sided 1

newmtl gem
material class dielectric
Ks 1.000 1.000 1.000
Kt 0.950 1.000 0.950
int_ior 1.5
Pr 0.000
two_sided 1

""" + MTL_LIGHT_CLASSIC)
    write_json("gems_test_128.json", "cornell_gems.obj", "cornell_gems.mtl", (128, 128), 64, spectral=True)
    write_json("gems_c3_1080p.json", "cornell_gems.obj", "cornell_gems.mtl", (1920, 1080), 64, spectral=True)
    # random-walk subsurface scattering (the material family of BASELINE configs[3]): a waxy short box, a tall box whose walk
    # starts along the refracted direction of a plastic coat
    with open(os.path.join(OUT, "cornell_sss.mtl"), "w") as f:
        f.write(MTL_COMMON.split("newmtl shortBox")[0] + """newmtl shortBox
material class diffuse
Kd 0.900 0.750 0.550
subsurface distances 0.30 0.15 0.08 scale 0.5
two_sided 1

newmtl tallBox
material class plastic
Kd 0.550 0.750 0.900
Ks 1.000 1.000 1.000
int_ior 1.5
Pr 0.300
subsurface path refracted distances 0.10 0.20 0.40 scale 0.5
two_sided 1

""" + MTL_LIGHT_CLASSIC)
    write_json("sss_test_128.json", "cornell_classic.obj", "cornell_sss.mtl", (128, 128), 64)
    # ... in spectral mode: the walk medium the bidirectional integrator derives from colour and distances depends on the wavelength
    write_json("sssspec_test_128.json", "cornell_classic.obj", "cornell_sss.mtl", (128, 128), 64, spectral=True)
    # the same two materials on sphere meshes (BVH traversal instead of the flat sweep)
    build_mesh(False, sss_meshes=True).write(os.path.join(OUT, "cornell_sssmesh.obj"), "cornell_sss.mtl")
    write_json("sssmesh_test_128.json", "cornell_sssmesh.obj", "cornell_sss.mtl", (128, 128), 64)
    # BASELINE configs[3] at its own size: this box at 1920 x 1080 is the BASE of bench.py --workload sssdragon_bdpt, which swaps the
    # two boxes for closed blob meshes of 102 400 triangles (tools/synthetic_scenes.py sss_dragon: the tree ships no mesh)
    write_json("sss_c4_1080p.json", "cornell_classic.obj", "cornell_sss.mtl", (1920, 1080), 64)
    # the same box with the Christensen-Burley class ("class approximate", scene_representation.cxx:1996-2001): three probe rays,
    # up to 24 exit points per vertex (subsurface::gather_cb + Raytracing::continuous_trace)
    with open(os.path.join(OUT, "cornell_sss.mtl")) as f:
        text = f.read()
    with open(os.path.join(OUT, "cornell_ssscb.mtl"), "w") as f:
        f.write(text.replace("subsurface distances 0.30 0.15 0.08 scale 0.5", "subsurface distances 0.30 0.15 0.08 scale 0.5 class approximate")
                    .replace("subsurface path refracted distances 0.10 0.20 0.40 scale 0.5", "subsurface path refracted distances 0.10 0.20 0.40 scale 0.5 class approximate"))
    write_json("ssscb_test_128.json", "cornell_classic.obj", "cornell_ssscb.mtl", (128, 128), 64)
    # the random-walk box with a TEXTURED scattering colour on the short box (checker albedo through map_Kd; every face of the boxes carries texture coordinates):
    # without an interior medium the walk's coefficients are derived from the colour at the ENTRY POINT (subsurface_step, bidirectional.cxx:757-765) - under the
    # bidirectional integrator every walk then runs through a medium of its own
    build_extra_textures()
    build_mesh(False, box_uv=True).write(os.path.join(OUT, "cornell_ssstex.obj"), "cornell_ssstex.mtl")
    with open(os.path.join(OUT, "cornell_ssstex.mtl"), "w") as f:
        f.write(text.replace("newmtl shortBox\nmaterial class diffuse\nKd 0.900 0.750 0.550", "newmtl shortBox\nmaterial class diffuse\nKd 1.000 0.900 0.700\nmap_Kd textures/checker.png"))
    write_json("ssstex_test_128.json", "cornell_ssstex.obj", "cornell_ssstex.mtl", (128, 128), 64)
    write_json("spectral_test_128.json", "cornell_classic.obj", "cornell_classic.mtl", (128, 128), 64, spectral=True)
    write_json("diamond_test_128.json", "cornell_classic.obj", "cornell_diamond.mtl", (128, 128), 64, spectral=True)

    # Feature-coverage scenes (branches no other scene reaches): textures + stochastic alpha + normal map; image environment
    # map as the only light; thin lens with an aperture image; equirectangular camera.
    build_extra_textures()
    m = build_mesh(False)
    # re-author two surfaces with texture coordinates: the back wall (checker albedo), and a free-standing cut-out card
    textured = Mesh()
    textured.quad("floor", [(-1, 0, 1), (1, 0, 1), (1, 0, -1), (-1, 0, -1)], (0, 1, 0), uv=True)
    textured.quad("ceiling", [(-1, 2, -1), (1, 2, -1), (1, 2, 1), (-1, 2, 1)], (0, -1, 0))
    textured.quad("frontWall", [(-1, 0, -1), (1, 0, -1), (1, 2, -1), (-1, 2, -1)], (0, 0, 1), uv=True)
    textured.quad("leftWall", [(-1, 0, 1), (-1, 0, -1), (-1, 2, -1), (-1, 2, 1)], (1, 0, 0))
    textured.quad("rightWall", [(1, 0, -1), (1, 0, 1), (1, 2, 1), (1, 2, -1)], (-1, 0, 0))
    lx0, _, lz0 = cornell_to_scene(343.0, 0, 227.0)
    lx1, _, lz1 = cornell_to_scene(213.0, 0, 332.0)
    textured.quad("light", [(lx0, 1.98, lz1), (lx1, 1.98, lz1), (lx1, 1.98, lz0), (lx0, 1.98, lz0)], (0, -1, 0))
    textured.quad("card", [(-0.6, 0.2, 0.1), (0.5, 0.2, 0.4), (0.5, 1.3, 0.4), (-0.6, 1.3, 0.1)], uv=True)
    textured.write(os.path.join(OUT, "cornell_textured.obj"), "cornell_textured.mtl")
    with open(os.path.join(OUT, "cornell_textured.mtl"), "w") as f:
        f.write(MTL_COMMON.split("newmtl shortBox")[0].replace("newmtl frontWall\nmaterial class diffuse\nKd 0.906 0.906 0.906", "newmtl frontWall\nmaterial class diffuse\nKd 1.000 1.000 1.000\nmap_Kd textures/checker.png")
                .replace("newmtl floor\nmaterial class diffuse\nKd 1.000 1.000 1.000", "newmtl floor\nmaterial class diffuse\nKd 1.000 1.000 1.000\nnormalmap image textures/bumps.png scale 1.0") + """newmtl card
material class diffuse
Kd 1.000 1.000 1.000
map_Kd textures/leaf.png
opacity 0.85
two_sided 1

""" + MTL_LIGHT_CLASSIC)
    write_json("textured_test_128.json", "cornell_textured.obj", "cornell_textured.mtl", (128, 128), 64)
    # the same box in spectral mode: RGB textures go through apply_rgb / rgb_response (scene.hxx:249-260)
    write_json("spectex_test_128.json", "cornell_textured.obj", "cornell_textured.mtl", (128, 128), 64, spectral=True)
    # image environment map only: the box without its area light, lit through the open front
    with open(os.path.join(OUT, "cornell_envmap.mtl"), "w") as f:
        f.write("newmtl et::env\nimage textures/sky.hdr\nrotation 30\n\n" + MTL_COMMON + "newmtl light\nmaterial class diffuse\nKd 0.780 0.780 0.780\ntwo_sided 1\n\n")
    write_json("envmap_test_128.json", "cornell_classic.obj", "cornell_envmap.mtl", (128, 128), 64)
    # thin lens + aperture image ("shape" is a key of the et::camera material form, scene_representation.cxx:1134-1138)
    with open(os.path.join(OUT, "cornell_lens.mtl"), "w") as f:
        f.write(MTL_COMMON + MTL_LIGHT_CLASSIC + """newmtl et::camera
class perspective
viewport 128 128
origin 0.0 1.0 3.82
target 0.0 1.0 -6.18
up 0.0 1.0 0.0
fov 39.597755
lens-radius 0.08
focal-distance 4.6
clip-near 0.1
clip-far 100.0
shape textures/aperture.png

""")
    write_json("lens_test_128.json", "cornell_classic.obj", "cornell_lens.mtl", (128, 128), 64, camera={})
    # equirectangular camera in the middle of the box (no light image: sample_film returns nothing for this class)
    eq = dict(CAMERA)
    eq.update({"class": "eq", "origin": [0.0, 1.0, 0.3], "target": [0.0, 1.0, -1.0]})
    write_json("equirect_test_128.json", "cornell_classic.obj", "cornell_classic.mtl", (128, 64), 64, camera=eq)

    for flavour in ("classic", "full"):
        obj, mtl = "cornell_%s.obj" % flavour, "cornell_%s.mtl" % flavour
        # BASELINE.json configs[0]: PT 512x512 16 spp ; configs[1]: VCM 1920x1080 64 spp
        write_json("%s_c1_512.json" % flavour, obj, mtl, (512, 512), 16)
        write_json("%s_c2_1080p.json" % flavour, obj, mtl, (1920, 1080), 64)
        if flavour == "full":
            # BASELINE configs[4]: 2048 x 2048; the fog medium becomes a 256^3 heterogeneous density grid at load time
            # (etx_oracle --inject-density 256 / SceneSnapshot.inject_density: the tree ships no .nvdb file)
            write_json("cloud_c5_2048.json", obj, mtl, (2048, 2048), 64)
        # test-sized variants (oracle finishes in seconds)
        write_json("%s_test_128.json" % flavour, obj, mtl, (128, 128), 64)
        write_json("%s_test_192x108.json" % flavour, obj, mtl, (192, 108), 64)
        # short paths: isolates direct lighting + first connections
        write_json("%s_test_128_len3.json" % flavour, obj, mtl, (128, 128), 64, max_path_length=3)


if __name__ == "__main__":
    main()
