"""numpy restatements of small pieces of index math / RNG used by the tests (test infrastructure)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox4x32_10(ctr, key):
    """ctr [n,4] uint32, key (k0, k1) -> [n,4] uint32.  Salmon et al. SC'11."""
    c = [ctr[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return np.stack(c, 1).astype(np.uint32)


def u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def deinterleave(gathered, W, H, world, tile_w=8, tile_h=8):
    """[world][tprp][64][3] -> [H][W][3]; restates deinterleave_kernel's index math."""
    tiles_x = (W + tile_w - 1) // tile_w
    rows, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    gt = (rows // tile_h) * tiles_x + xs // tile_w
    k = (rows % tile_h) * tile_w + xs % tile_w
    return gathered[gt % world, gt // world, k]


def unit_vector_to_uniforms(v):
    """Invert sample_unit_vector: z = 1-2u1, phi = 2 pi u2."""
    u1 = (1.0 - v[:, 2].astype(np.float64)) / 2.0
    u2 = (np.arctan2(v[:, 1].astype(np.float64), v[:, 0].astype(np.float64)) / (2 * np.pi)) % 1.0
    return u1.astype(np.float32), u2.astype(np.float32)
