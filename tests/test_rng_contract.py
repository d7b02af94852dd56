"""The counter-based generator of include/ptk.h, through the oracle's exported functions."""
import numpy as np

from oracle import binding as ob

M = (1 << 64) - 1


def mix64(z):
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return z


G = 0x9E3779B97F4A7C15


def test_matches_the_written_contract():
    o = ob.oracle_lib()
    for seed, pixel, sample in [(0, 0, 0), (1, 2, 3), (2**63 + 5, 2**31 + 7, 999), (12345, 1920 * 1080 - 1, 1023)]:
        key = mix64(mix64((seed + G) & M) ^ ((pixel << 32) | sample))
        assert o.oracle_rng_key(seed, pixel, sample) == key
        for n in (0, 1, 17, 2**32 - 1):
            bits = mix64((key + (n + 1) * G) & M)
            assert o.oracle_rng_uniform(key, n) == (bits >> 11) * 2.0**-53
            assert o.oracle_rng_fork(key, n) == mix64(key ^ mix64((n + 0xD1B54A32D192ED03) & M))


def test_uniforms_are_uniform_and_streams_distinct():
    o = ob.oracle_lib()
    key = o.oracle_rng_key(7, 11, 13)
    u = np.array([o.oracle_rng_uniform(key, n) for n in range(20000)])
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    hist = np.histogram(u, bins=16, range=(0, 1))[0]
    assert hist.min() > 1000
    keys = {o.oracle_rng_key(s, p, k) for s in range(3) for p in range(40) for k in range(40)}
    assert len(keys) == 3 * 40 * 40
    v = np.array([o.oracle_rng_uniform(o.oracle_rng_key(0, p, 0), 0) for p in range(5000)])
    assert abs(np.corrcoef(v[:-1], v[1:])[0, 1]) < 0.05
