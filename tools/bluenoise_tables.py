#!/usr/bin/env python3
"""Blue-noise sample tables for etx_hip_upload_bluenoise.

The C ABI takes what the host's sample_blue_noise (reference path_tracing.cxx:173-178 -> thirdparty/bluenoise BNSampler)
returns for one sample-count class, tabulated as bytes: value[py][px][sample][dimension] for a 128 x 128 pixel tile,
256 samples, 8 dimensions (float = (0.5 + value) / 256). That is 32 MiB - too big for a test fixture - but every
(pixel, dimension) column is the same 256-entry sequence with its index and value XOR-ed by two per-pixel bytes
(Heitz et al. 2019), so the fixture stores that factorisation (258 KiB) and the table is expanded at run time.

usage: bluenoise_tables.py factor table.raw out.npz     (the raw table comes from oracle/_ref/etx_oracle --dump-bluenoise)
"""
import sys

import numpy as np

TILE, SAMPLES, DIMS = 128, 256, 8


def factor(table):
    """table: uint8 [128,128,256,8] -> (base [8,256], index_xor [128,128,8], value_xor [128,128,8])"""
    table = np.asarray(table, dtype=np.uint8).reshape(TILE * TILE, SAMPLES, DIMS)
    base = np.ascontiguousarray(table[0].T)  # [8,256]: the sequence as pixel (0,0) sees it
    index_xor = np.zeros((TILE * TILE, DIMS), dtype=np.uint8)
    value_xor = np.zeros((TILE * TILE, DIMS), dtype=np.uint8)
    s = np.arange(SAMPLES)
    for d in range(DIMS):
        column = table[:, :, d]  # [pixels, 256]
        found = np.zeros(TILE * TILE, dtype=bool)
        for r in range(SAMPLES):
            diff = column ^ base[d][s ^ r][None, :]
            constant = (diff == diff[:, :1]).all(axis=1) & ~found
            index_xor[constant, d] = r
            value_xor[constant, d] = diff[constant, 0]
            found |= constant
            if found.all():
                break
        if not found.all():
            raise ValueError("dimension %d: %d columns are not an index/value XOR of the base sequence" % (d, (~found).sum()))
    return base, index_xor.reshape(TILE, TILE, DIMS), value_xor.reshape(TILE, TILE, DIMS)


def expand(base, index_xor, value_xor):
    """inverse of factor: uint8 [128,128,256,8]"""
    s = np.arange(SAMPLES)
    out = np.empty((TILE, TILE, SAMPLES, DIMS), dtype=np.uint8)
    for d in range(DIMS):
        idx = s[None, None, :] ^ index_xor[:, :, d, None].astype(np.int64)
        out[:, :, :, d] = base[d][idx] ^ value_xor[:, :, d, None]
    return out


def set_index(samples):
    """sample-count class of BNSampler (bluenoise.cxx:78): next power of two of clamp(samples, 1, 256), as log2"""
    samples = 1 if samples == 0 else min(samples, 256)
    p = 1
    while p < samples:
        p *= 2
    return p.bit_length() - 1


def load(path):
    z = np.load(path)
    return expand(z["base"], z["index_xor"], z["value_xor"])


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "factor":
        raw = np.fromfile(sys.argv[2], dtype=np.uint8)
        b, i, v = factor(raw)
        assert np.array_equal(expand(b, i, v).reshape(-1), raw)
        np.savez_compressed(sys.argv[3], base=b, index_xor=i, value_xor=v)
        print("wrote", sys.argv[3])
    else:
        print(__doc__)
