"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the per-slice keyed step noise of the ancestral sampler
(founddiff_amd/csrc/fd_sched.hip: Philox4x32-10 keyed by the slice seed, counter = (pixel // 4, t, tag, 0),
Box-Muller on 23-bit uniforms (k + 0.5) / 2^23).  The reference has no counterpart: it draws torch.randn_like(x) from the device
generator (/root/reference/src/DADiff.py:1228), which ties a slice's noise to its place in the batch; the keyed stream
is what makes a sharded BASELINE configs[3] volume independent of the world size.  Parity of the noise VALUES with the
reference is therefore neither possible nor claimed -- parity tests feed explicit noise tensors (`step_noise=`)."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
TAG = 0x46444E5A
MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & np.uint64(MASK), p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & np.uint64(MASK), lo1, (hi0 ^ c3 ^ k1) & np.uint64(MASK), lo0
        k0, k1 = (k0 + np.uint64(W0)) & np.uint64(MASK), (k1 + np.uint64(W1)) & np.uint64(MASK)
    return c0, c1, c2, c3


def keyed_normal(seed, t, npix):
    """(npix,) float32 standard normals of slice `seed` at step `t`."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    ng = (npix + 3) // 4
    g = np.arange(ng, dtype=np.uint64)
    z = np.zeros(ng, dtype=np.uint64)
    r = philox4x32_10(g, z + np.uint64(t & MASK), z + np.uint64(TAG), z, seed & MASK, seed >> 32)
    u = [((v >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0) for v in r]
    ra = np.sqrt(np.float32(-2.0) * np.log(u[0])).astype(np.float32)
    rb = np.sqrt(np.float32(-2.0) * np.log(u[2])).astype(np.float32)
    a0, a1 = np.float32(2.0 * np.pi) * u[1], np.float32(2.0 * np.pi) * u[3]
    out = np.stack([ra * np.cos(a0), ra * np.sin(a0), rb * np.cos(a1), rb * np.sin(a1)], 1).reshape(-1)
    return out[:npix].astype(np.float32)
