"""Scene snapshots: a byte-exact copy of the reference's `etx::Scene` / `etx::Camera` (the backend's input ABI,
include/etx_scene_abi.h) plus every array they point to, written by oracle/_ref/etx_oracle --snapshot through the
reference's own scene loader. Loading = relocating the listed pointer fields into this process' address space.

File layout ("ETXSCENE1", see oracle/driver/etx_oracle.cxx write_snapshot):
  bytes 0..15  magic, then u64 scene_offset, camera_offset, fixup_count, fixup_offset, total_size, (sizeof(Scene)<<32|sizeof(Camera))
  fixups: u64 offsets of pointer fields; each field holds a payload offset (0 = null).
"""
import ctypes
import struct

import numpy as np

SCENE_SIZE = 528   # sizeof(etx::Scene)  == sizeof(etx_abi_scene)
CAMERA_SIZE = 176  # sizeof(etx::Camera) == sizeof(etx_abi_camera)

# field offsets inside etx_abi_scene / etx_abi_camera used on the host side (include/etx_scene_abi.h)
_SCENE_TRIANGLES = 16
_SCENE_VERTICES = 0
_SCENE_TRIANGLE_TO_EMITTER = 32
_SCENE_MATERIALS = 48
_SCENE_EMITTERS = 80
_SCENE_MEDIUMS = 112
_SCENE_SAMPLES = 464
_SCENE_MAX_PATH = 460
_SCENE_MIN_PATH = 456
_SCENE_RR_START = 468
_SCENE_FLAGS = 520
_SCENE_NOISE_THRESHOLD = _SCENE_RR_START + 4  # float noise_threshold follows random_path_termination (etx_scene_abi.h)
_SCENE_RADIUS = 444
_CAMERA_FILM_SIZE = 144


class SceneSnapshot:
    def __init__(self, path):
        with open(path, "rb") as f:
            data = f.read()
        if data[:9] != b"ETXSCENE1":
            raise ValueError("%s: not an ETXSCENE1 snapshot" % path)
        scene_off, camera_off, fix_count, fix_off, total, sizes = struct.unpack_from("<6Q", data, 16)
        if total != len(data):
            raise ValueError("%s: truncated snapshot" % path)
        if sizes != ((SCENE_SIZE << 32) | CAMERA_SIZE):
            raise ValueError("%s: snapshot was written for a different Scene/Camera layout" % path)
        # 16-byte aligned, writable storage that outlives the upload call
        self._raw = ctypes.create_string_buffer(len(data) + 16)
        base = ctypes.addressof(self._raw)
        aligned = (base + 15) & ~15
        ctypes.memmove(aligned, data, len(data))
        self.base = aligned
        self.size = len(data)
        fixups = struct.unpack_from("<%dQ" % fix_count, data, fix_off)
        for field in fixups:
            ptr = ctypes.c_uint64.from_address(aligned + field)
            if ptr.value != 0:
                ptr.value = ptr.value + aligned
        self.scene_address = aligned + scene_off
        self.camera_address = aligned + camera_off
        self.path = path
        self._fixups = fixups  # offsets of the pointer fields (save())
        self.version = 0  # bumped by every setter: the integrators re-upload when it changed since their last upload

    def _u32(self, address):
        return ctypes.c_uint32.from_address(address)

    def _array(self, offset):
        ptr = ctypes.c_uint64.from_address(self.scene_address + offset).value
        count = ctypes.c_uint64.from_address(self.scene_address + offset + 8).value
        return ptr, count

    @property
    def film_size(self):
        return (self._u32(self.camera_address + _CAMERA_FILM_SIZE).value, self._u32(self.camera_address + _CAMERA_FILM_SIZE + 4).value)

    @property
    def samples(self):
        return self._u32(self.scene_address + _SCENE_SAMPLES).value

    @samples.setter
    def samples(self, value):
        self._u32(self.scene_address + _SCENE_SAMPLES).value = int(value)
        self.version += 1

    @property
    def noise_threshold(self):
        """Scene::noise_threshold: > 0 switches the path tracer's adaptive sampling on (Film::estimate_noise_levels)."""
        return ctypes.c_float.from_address(self.scene_address + _SCENE_NOISE_THRESHOLD).value

    @noise_threshold.setter
    def noise_threshold(self, value):
        ctypes.c_float.from_address(self.scene_address + _SCENE_NOISE_THRESHOLD).value = float(value)
        self.version += 1

    @property
    def max_path_length(self):
        return self._u32(self.scene_address + _SCENE_MAX_PATH).value

    @max_path_length.setter
    def max_path_length(self, value):
        self._u32(self.scene_address + _SCENE_MAX_PATH).value = int(value)
        self.version += 1

    @property
    def bounding_sphere_radius(self):
        return ctypes.c_float.from_address(self.scene_address + _SCENE_RADIUS).value

    @property
    def triangle_count(self):
        return self._array(_SCENE_TRIANGLES)[1]

    def vertices(self):
        """float32 view (count, 14): pos, nrm, tan, btn, tex"""
        ptr, count = self._array(_SCENE_VERTICES)
        buf = (ctypes.c_float * (count * 14)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float32).reshape(count, 14)

    def triangles(self):
        """uint32 view (count, 8): i0, i1, i2, material, geo_n (float bits) x3, pad"""
        ptr, count = self._array(_SCENE_TRIANGLES)
        buf = (ctypes.c_uint32 * (count * 8)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(count, 8)

    def replace_geometry(self, vertices, triangles, triangle_to_emitter):
        """New vertex / triangle / triangle_to_emitter arrays for the scene (float32 (V, 14), uint32 (T, 8), uint32 (T,)): what a host
        with changed TOPOLOGY hands to etx_hip_upload_scene. The snapshot keeps the arrays alive; emitter instances name triangles
        by index, so the caller keeps emissive triangles where they were."""
        vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 14)
        triangles = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 8)
        triangle_to_emitter = np.ascontiguousarray(triangle_to_emitter, dtype=np.uint32).reshape(-1)
        if (triangle_to_emitter.shape[0] != triangles.shape[0]) or (int(triangles[:, 0:3].max()) >= vertices.shape[0]):
            raise ValueError("replace_geometry: one emitter entry per triangle, vertex indices inside the vertex array")
        self._geometry = (vertices, triangles, triangle_to_emitter)
        for offset, array in ((_SCENE_VERTICES, vertices), (_SCENE_TRIANGLES, triangles), (_SCENE_TRIANGLE_TO_EMITTER, triangle_to_emitter)):
            ctypes.c_uint64.from_address(self.scene_address + offset).value = array.ctypes.data
            ctypes.c_uint64.from_address(self.scene_address + offset + 8).value = array.shape[0]
        self.version += 1

    def triangle_to_emitter(self):
        ptr, count = self._array(_SCENE_TRIANGLE_TO_EMITTER)
        return np.frombuffer((ctypes.c_uint32 * count).from_address(ptr), dtype=np.uint32)

    def materials(self):
        """writable uint32 view (count, 50) of the etx::Material table (class at column 41): for hosts / tests that edit a material
        in place and then call Context.update_scene(snapshot, CHANGED_MATERIALS)"""
        ptr, count = self._array(_SCENE_MATERIALS)
        buf = (ctypes.c_uint32 * (count * 50)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(count, 50)

    def emitter_instances(self):
        """writable uint32 view (count, 8) of the etx::Emitter table (cls, profile, triangle_index, weights as float bits)"""
        ptr, count = self._array(_SCENE_EMITTERS)
        return np.frombuffer((ctypes.c_uint32 * (count * 8)).from_address(ptr), dtype=np.uint32).reshape(count, 8)

    def material_classes(self):
        ptr, count = self._array(_SCENE_MATERIALS)
        buf = (ctypes.c_uint32 * (count * 50)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(count, 50)[:, 41].copy()  # offset 164 / 4

    def inject_density(self, n):
        """Every medium of the scene becomes Heterogeneous with the procedural n^3 density grid of `etx_oracle --inject-density n`
        (oracle/driver/etx_oracle.cxx: two blobs + a product of sines, normalised to a maximum of 1 = the state MediumPool::add leaves
        behind, medium_pool.cxx:41-58). The reference's loader only reads .nvdb files and its tree ships none; a 256^3 grid (67 MB,
        BASELINE configs[4]) cannot be a committed fixture either, so both sides build it from the same formula."""
        n = int(n)
        c = (np.arange(n, dtype=np.float32) + np.float32(0.5)) / np.float32(n)
        fx, fy, fz = c[None, None, :], c[None, :, None], c[:, None, None]
        f32 = np.float32
        sq = lambda v: v * v
        d = np.exp(f32(-12.0) * (sq(fx - f32(0.35)) + sq(fy - f32(0.40)) + sq(fz - f32(0.55))))
        d = d + f32(0.8) * np.exp(f32(-20.0) * (sq(fx - f32(0.70)) + sq(fy - f32(0.65)) + sq(fz - f32(0.35))))
        d = d + f32(0.15) * (f32(1.0) + np.sin(f32(9.0) * fx) * np.sin(f32(7.0) * fy + f32(1.0)) * np.sin(f32(8.0) * fz + f32(2.0)))
        grid = np.ascontiguousarray((d / d.max()).astype(np.float32)).reshape(-1)  # index x + y * n + z * n * n
        ptr, count = self._array(_SCENE_MEDIUMS)
        self._density = grid
        for i in range(count):
            m = ptr + i * 80  # sizeof(etx_abi_medium)
            ctypes.c_uint64.from_address(m).value = grid.ctypes.data
            ctypes.c_uint64.from_address(m + 8).value = grid.shape[0]
            ctypes.c_uint16.from_address(m + 48).value = 1  # Medium::Class::Heterogeneous
            for k in range(3):
                ctypes.c_uint32.from_address(m + 68 + 4 * k).value = n
        self.version += 1
        return grid

    def save(self, path):
        """Writes the snapshot back in the ETXSCENE1 layout WITH what the setters changed: the in-memory image (scene scalars, materials,
        emitters ... are edited in place) with every listed pointer field turned back into a payload offset; arrays that
        replace_geometry / inject_density swapped in are appended as new 16-byte aligned chunks. etx_oracle --load-snapshot reads the
        result (CPU baseline and golden films of scenes that are assembled in memory, tools/synthetic_scenes.py)."""
        out = bytearray(ctypes.string_at(self.base, self.size))
        owned = [a for a in getattr(self, "_geometry", ())] + ([self._density] if getattr(self, "_density", None) is not None else [])
        appended = {}
        for field in self._fixups:
            live = ctypes.c_uint64.from_address(self.base + field).value
            if live == 0:
                offset = 0
            elif self.base <= live < self.base + self.size:
                offset = live - self.base
            elif live in appended:
                offset = appended[live]
            else:
                match = [a for a in owned if a.ctypes.data == live]
                if not match:
                    raise ValueError("save: the pointer field at offset %d points to memory the snapshot does not own" % field)
                out.extend(b"\0" * ((-len(out)) % 16))
                offset = appended[live] = len(out)
                out.extend(match[0].tobytes())
            struct.pack_into("<Q", out, field, offset)
        struct.pack_into("<Q", out, 16 + 4 * 8, len(out))  # total_size
        with open(path, "wb") as f:
            f.write(bytes(out))
