"""Philox4x32-10 (Salmon et al., SC'11; Random123 1.14) in numpy, and the normals tdx_randn* / tdx_p_sample_step_rng make of
it: counter = (offset + i as two 32-bit words, trajectory stream id as two words), key = the seed's two words; from output
words c0..c3: u = (c0 + 1) 2^-32, v = c1 2^-32 -> sqrt(-2 ln u) (cos 2 pi v, sin 2 pi v), and the same from (c2, c3).
Test infrastructure only (the third-party algorithm restated from its publication, checked against its known-answer vectors)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint64 arrays holding 32-bit words; k0, k1 python ints."""
    c = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def normals(n, seed, stream_id, offset=0):
    """The first n values tdx_randn(out, n, seed, stream_id, offset) writes."""
    n4 = (n + 3) // 4
    ctr = np.uint64(offset) + np.arange(n4, dtype=np.uint64)
    sid = np.uint64(stream_id)
    c = philox4x32_10(ctr & MASK, ctr >> np.uint64(32), np.full(n4, sid & MASK), np.full(n4, sid >> np.uint64(32)),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    out = np.empty((n4, 4), dtype=np.float64)
    for k in range(2):
        u1 = (c[2 * k].astype(np.float64) + 1.0) * 2.0**-32
        u2 = c[2 * k + 1].astype(np.float64) * 2.0**-32
        rad = np.sqrt(-2.0 * np.log(u1))
        out[:, 2 * k], out[:, 2 * k + 1] = rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)
    return out.reshape(-1)[:n]
