"""The Python restatement of Philox4x32-10 used to check the device generator reproduces the Random123
known-answer vectors (so a GPU mismatch in tests/test_gpu_rng.py is the device's)."""
from tests.philox_ref import KAT, philox4x32_10


def test_philox_reference_known_answers():
    for ctr, key, out in KAT:
        assert philox4x32_10(ctr, key) == out


def test_vectorised_restatement_equals_the_scalar_one():
    import numpy
    from tests.philox_ref import device_normals, device_normals_fast, philox4x32_10_vec
    for ctr, key, out in KAT:
        got = philox4x32_10_vec(*[numpy.array([x], dtype=numpy.uint64) for x in ctr], key[0], key[1])
        assert tuple(int(x[0]) for x in got) == out
    for seed, stream, counter in ((7, 0, 0), (7, 3, 41), (0x1234567890ABCDEF, 5, 2 ** 33 + 9)):
        a, b = device_normals(2001, seed, stream, counter), device_normals_fast(2001, seed, stream, counter)
        assert numpy.max(numpy.abs(a - b)) < 1e-14
