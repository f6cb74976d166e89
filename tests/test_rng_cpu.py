"""The Python restatement of Philox4x32-10 used to check the device generator reproduces the Random123
known-answer vectors (so a GPU mismatch in tests/test_gpu_rng.py is the device's)."""
from tests.philox_ref import KAT, philox4x32_10


def test_philox_reference_known_answers():
    for ctr, key, out in KAT:
        assert philox4x32_10(ctr, key) == out
