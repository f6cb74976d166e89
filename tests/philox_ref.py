"""Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) written
straight from the paper, as the independent check of the device generator (csrc/k_small.hip rng_normal_kernel),
plus the known-answer vectors of the Random123 distribution (kat_vectors, "philox4x32 10")."""
import math

import numpy

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF

# (counter[4], key[2]) -> output[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(ctr, key):
    c = list(ctr)
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return tuple(c)


def device_normals(n, seed, stream, counter):
    """The first n normals of launch ``counter`` of the device stream (seed, stream), restated from
    rng_normal_kernel: pair p uses counter words (p_lo, p_hi, counter_lo, counter_hi ^ (stream * W0)),
    key (seed_lo, seed_hi); u1 = (top 53 bits of c0:c1 + 1) 2^-53 in (0, 1], u2 = (top 53 bits of c2:c3 + 1/2)
    2^-53; (x0, x1) = sqrt(-2 ln u1) (cos, sin)(2 pi u2)."""
    out = numpy.empty(2 * ((n + 1) // 2))
    for p in range((n + 1) // 2):
        ctr = (p & MASK, (p >> 32) & MASK, counter & MASK, ((counter >> 32) ^ (stream * W0)) & MASK)
        c = philox4x32_10(ctr, (seed & MASK, (seed >> 32) & MASK))
        a = ((c[0] << 32) | c[1]) >> 11
        b = ((c[2] << 32) | c[3]) >> 11
        u1 = (a + 1.0) / 9007199254740992.0
        u2 = (b + 0.5) / 9007199254740992.0
        rad = math.sqrt(-2.0 * math.log(u1))
        out[2 * p] = rad * math.cos(2.0 * math.pi * u2)
        out[2 * p + 1] = rad * math.sin(2.0 * math.pi * u2)
    return out[:n]


def philox4x32_10_vec(c0, c1, c2, c3, k0, k1):
    """philox4x32_10 on numpy uint64 arrays holding 32-bit words (same arithmetic, vectorised)."""
    c0, c1, c2, c3 = [numpy.asarray(x, dtype=numpy.uint64) for x in numpy.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = numpy.uint64(k0), numpy.uint64(k1)
    m32, s32 = numpy.uint64(MASK), numpy.uint64(32)
    for _ in range(10):
        p0, p1 = numpy.uint64(M0) * c0, numpy.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> s32) ^ c1 ^ k0) & m32, p1 & m32, ((p0 >> s32) ^ c3 ^ k1) & m32, p0 & m32
        k0, k1 = (k0 + numpy.uint64(W0)) & m32, (k1 + numpy.uint64(W1)) & m32
    return c0, c1, c2, c3


def device_normals_fast(n, seed, stream, counter):
    """device_normals, vectorised (numpy's log / cos / sin instead of libm's through math: the same to the last bit or one
    unit in it)."""
    npair = (n + 1) // 2
    p = numpy.arange(npair, dtype=numpy.uint64)
    m32, s32 = numpy.uint64(MASK), numpy.uint64(32)
    c = philox4x32_10_vec(p & m32, (p >> s32) & m32, numpy.uint64(counter & MASK),
                          numpy.uint64(((counter >> 32) ^ (stream * W0)) & MASK), seed & MASK, (seed >> 32) & MASK)
    a = ((c[0] << s32) | c[1]) >> numpy.uint64(11)
    b = ((c[2] << s32) | c[3]) >> numpy.uint64(11)
    u1 = (a.astype(numpy.float64) + 1.0) / 9007199254740992.0
    u2 = (b.astype(numpy.float64) + 0.5) / 9007199254740992.0
    rad = numpy.sqrt(-2.0 * numpy.log(u1))
    out = numpy.empty(2 * npair)
    out[0::2] = rad * numpy.cos(2.0 * numpy.pi * u2)
    out[1::2] = rad * numpy.sin(2.0 * numpy.pi * u2)
    return out[:n]
