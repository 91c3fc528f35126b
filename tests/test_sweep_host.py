"""The GPU-free half of the one-sweep mode: band construction and the resolve step, the very functions the
runtime calls (papr_level_key, papr_sweep_bands, papr_sweep_resolve in libpaprhip.so), against a numpy model
of what the sweep kernel leaves behind.  Whatever the guess: either the resolved counts equal the brute-force
`power > level` counts, or the function says "out of band" and the runtime reads the shard again."""
import numpy as np
import pytest


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def model_sweep(power, edges):
    """what papr_sweep_kernel computes: bin = #{edges <= bits(power)}; even bins counted, odd bins stashed"""
    k = np.searchsorted(edges, bits(power), side="right")
    k[np.isnan(power)] = 0
    m = edges.size // 2
    hist = np.bincount(k, minlength=2 * m + 1).astype(np.uint64)
    above_band = np.array([hist[2 * j + 2::2].sum() for j in range(m)], dtype=np.uint64)
    return above_band, power[(k & 1) == 1]


def test_level_key_is_the_float_compare(pkg):
    L = pkg.lib()
    rng = np.random.default_rng(1)
    levels = np.concatenate([rng.standard_normal(200).astype(np.float32) * 10, np.float32([0.0, -0.0, np.inf, -np.inf, np.nan,
                             1e-45, 3.4e38, 1.17549435e-38])])
    probes = np.abs(np.concatenate([rng.standard_normal(500).astype(np.float32) * 10, levels[np.isfinite(levels)],
                                    np.float32([0.0, np.inf, 1e-45])])).astype(np.float32)
    for t in levels:
        key = L.papr_level_key(float(t))
        with np.errstate(invalid="ignore"):
            want = probes > t
        got = bits(probes) >= key if key != 0xFFFFFFFF else np.zeros(probes.size, bool)
        assert np.array_equal(got, want), t


def test_bands_shape_and_rejections(pkg):
    keys, edges = pkg.sweep_bands(np.float32([8.0, 2.0, 2.0, 4.0, np.nan, np.inf]), 14)
    assert np.array_equal(keys, bits(np.float32([2.0, 4.0, 8.0])) + 1)
    assert np.array_equal(edges[0::2], keys - (1 << 14)) and np.array_equal(edges[1::2], keys + (1 << 14))
    assert np.all(np.diff(edges.astype(np.int64)) > 0)
    for bad in ([], [np.nan], [np.inf], [0.0, 1.0], [-1.0, 2.0], [1e-42, 2.0], [3.4e38], [2.0, 2.0001]):
        assert pkg.sweep_bands(np.float32(bad), 14) is None, bad
    assert pkg.sweep_bands(np.float32([2.0, 2.0001]), 7) is not None      # narrower bands fit between close thresholds


@pytest.mark.parametrize("graph", [False, True], ids=["default", "graph"])
def test_resolve_equals_brute_force_whatever_the_guess(pkg, graph):
    rng = np.random.default_rng(7 + graph)
    resolved = missed = 0
    for case in range(120):
        n = int(rng.integers(1000, 40000))
        scale = 10.0 ** rng.uniform(-3, 3)
        iq = (rng.standard_normal((n, 2)) * scale).astype(np.float32)
        power = (iq[:, 0] * iq[:, 0] + iq[:, 1] * iq[:, 1]).astype(np.float32)
        if case % 7 == 0:
            power[rng.integers(0, n, 5)] = np.float32([np.inf, 0.0, 0.0, 3e38, 1e-40])
        st = pkg.Stats()
        st.sum, st.n, st.peak = float(power.astype(np.float64).sum()), n, float(power.max())
        mean, papr, table = pkg.levels(st, graph)
        est = pkg.Stats()
        err = rng.choice([0.0, 1e-4, -3e-4, 8e-4, -2e-3, 5e-2])       # some guesses good, some hopeless
        est.sum, est.n = st.sum * (1 + err), n
        band_log2 = int(rng.choice([12, 14, 15]))
        made = pkg.sweep_bands(pkg.guess_levels(est, graph), band_log2)
        if made is None:
            continue
        keys, edges = made
        above_band, stash = model_sweep(power, edges)
        with np.errstate(invalid="ignore"):
            stash_above = np.array([(stash > t).sum() for t in table], dtype=np.uint64)
            want = np.array([(power > t).sum() for t in table], dtype=np.uint64)
        got = pkg.sweep_resolve(keys, band_log2, above_band, table, stash_above)
        if got is None:
            missed += 1
            assert abs(err) * 2.0 ** 23 > (1 << band_log2) / 2.1 or not np.isfinite(st.sum)   # only bad guesses may miss
        else:
            resolved += 1
            assert np.array_equal(got, want), (case, err, band_log2)
        # a table from somewhere else entirely: never a wrong answer, at worst a refusal
        other = (table * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32) if table.size else table
        with np.errstate(invalid="ignore"):
            o_stash = np.array([(stash > t).sum() for t in other], dtype=np.uint64)
            o_want = np.array([(power > t).sum() for t in other], dtype=np.uint64)
        o_got = pkg.sweep_resolve(keys, band_log2, above_band, other, o_stash)
        assert o_got is None or np.array_equal(o_got, o_want)
    assert resolved >= 40 and missed >= 10


def test_the_kernels_walk_covers_every_tile_once(tmp_path):
    """SkewWalk (csrc/papr_skew_walk.h: which tiles a persistent workgroup folds — grid stride with the odd XCDs' workgroups
    sitting one round in R out) compiled for the host: every tile exactly once, a workgroup's tiles in increasing order (the
    trackers' first-occurrence rule needs that), the shares in the ratio (R - 1) / R; the bench shard's geometry (163 840 tiles of
    64 KiB, 256 workgroups) with every period the kernels use, odd sizes, no skew, a workgroup count the skew does not apply to."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "skew_harness")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "dtv-utils_amd", "csrc"),
                           os.path.join(root, "tests", "c", "skew_harness.cpp"), "-o", exe])
    for ntiles, blocks, R in [(163840, 256, 24), (163840, 256, 20), (163840, 256, 48), (163840, 256, 96), (163841, 256, 24),
                              (1000003, 256, 2), (12345, 64, 3), (255, 256, 24), (0, 256, 24), (163840, 256, 0), (9999, 250, 24),
                              (4097, 8, 5)]:
        for slow in (1, 0):   # (which parity of workgroups the queue put on the odd XCDs)
            p = subprocess.run([exe, str(ntiles), str(blocks), str(R), str(slow)], capture_output=True, text=True)
            assert p.returncode == 0 and p.stdout.startswith("ok"), (ntiles, blocks, R, slow, p.stdout)
            even, odd = (int(v) for v in p.stdout.split()[1:3])
            assert even + odd == ntiles
            less, more = (odd, even) if slow else (even, odd)
            if R >= 2 and blocks % 8 == 0 and ntiles >= 50 * R * blocks:   # (many whole periods: the shares are what they should be)
                assert abs(less / more - (R - 1) / R) < 2e-3, (ntiles, blocks, R, slow, even, odd)
            if R < 2 or blocks % 8:
                assert abs(even - odd) <= blocks   # plain grid stride
