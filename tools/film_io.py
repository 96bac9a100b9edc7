"""Readers for the raw film / scene-snapshot files written by oracle/_ref/etx_oracle and by the HIP backend tools."""
import struct
import numpy as np

LAYER_NAMES = ("camera", "light", "result", "normal", "albedo")


def read_film(path):
    """Returns dict(width, height, spp, seconds, threads, camera, light, result) with HxWx4 float32 arrays."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"ETXFILM1":
        raise ValueError("%s: not an ETXFILM1 file" % path)
    w, h, layers, spp = struct.unpack_from("<4I", data, 8)
    (seconds,) = struct.unpack_from("<d", data, 24)
    threads, _ = struct.unpack_from("<2I", data, 32)
    off = 40
    out = {"width": w, "height": h, "spp": spp, "seconds": seconds, "threads": threads}
    for i in range(layers):
        arr = np.frombuffer(data, dtype=np.float32, count=w * h * 4, offset=off).reshape(h, w, 4).copy()
        out[LAYER_NAMES[i] if i < len(LAYER_NAMES) else "layer%d" % i] = arr
        off += w * h * 16
    return out


def write_film(path, camera, light, spp=0, seconds=0.0, threads=0):
    h, w = camera.shape[:2]
    result = np.maximum(camera + light, 0.0)
    result[..., 3] = 1.0
    with open(path, "wb") as f:
        f.write(b"ETXFILM1")
        f.write(struct.pack("<4I", w, h, 3, spp))
        f.write(struct.pack("<d", seconds))
        f.write(struct.pack("<2I", threads, 0))
        for arr in (camera, light, result):
            f.write(np.ascontiguousarray(arr, dtype=np.float32).tobytes())


def tonemap(rgb, exposure=1.0):
    """Same display transform as the reference's PNG export (sources/raytracer/app.cxx:268-283)."""
    tm = 1.0 - np.exp(-exposure * np.maximum(rgb, 0.0))
    g = np.where(tm <= 0.0031308, 12.92 * tm, 1.055 * np.power(np.maximum(tm, 1e-12), 1.0 / 2.4) - 0.055)
    return (np.clip(g, 0.0, 1.0) * 255.0).astype(np.uint8)


def save_png(path, rgb, exposure=1.0):
    from PIL import Image
    Image.fromarray(tonemap(rgb[..., :3], exposure)).save(path)


def rmse(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


if __name__ == "__main__":
    import sys
    film = read_film(sys.argv[1])
    print({k: v for k, v in film.items() if not hasattr(v, "shape")})
    for name in LAYER_NAMES:
        print(name, "mean rgb", film[name][..., :3].mean(axis=(0, 1)))
    if len(sys.argv) > 2:
        save_png(sys.argv[2], film["result"])
