"""The synthetic IQ generator (include/papr_synth.h) is the bench workload and
the full-size parity input, so its byte stream is pinned here."""
import hashlib
import subprocess

import numpy as np


def test_known_samples_and_digest(orc, tmp_path):
    path = str(tmp_path / "s.cfile")
    subprocess.check_call([orc.MKCFILE, path, "4096"])
    a = np.fromfile(path, dtype=np.float32)
    assert a.size == 8192
    # exact multiples of 2^-16 bounded by 4.0
    assert np.all(np.abs(a) <= 4.0) and np.all(a * 65536 == np.round(a * 65536))
    assert hashlib.sha256(a.tobytes()).hexdigest() == PINNED_SHA256_4096
    assert abs(float(np.mean(a.astype(np.float64) ** 2)) * 2 - 4.0 / 3.0) < 0.05


def test_index_addressable(orc, tmp_path):
    """A shard generated alone equals the same range of the full stream
    (extra floats continue the stream; seeds differ)."""
    p1, p2, p3 = (str(tmp_path / f"{k}.cfile") for k in "abc")
    subprocess.check_call([orc.MKCFILE, p1, "3000"])
    subprocess.check_call([orc.MKCFILE, p2, "1000", "--extra-floats", "4000"])
    subprocess.check_call([orc.MKCFILE, p3, "3000", "--seed", "12345"])
    a, b, c = (np.fromfile(p, dtype=np.float32) for p in (p1, p2, p3))
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_spike_spec_python_matches_c(pkg, orc, tmp_path):
    for n in (16, 1000, 20000, 131072, 1342177):
        sp = pkg.SynthSpec.spike(n)
        path = str(tmp_path / "sp.cfile")
        subprocess.check_call([orc.MKCFILE, path, str(n), "--spike"])
        a = np.fromfile(path, dtype=np.float32).reshape(-1, 2)
        hits = np.flatnonzero(a[:, 0] == np.float32(36.9375))
        assert sorted(hits.tolist()) == sorted({sp.ov[0].index, sp.ov[1].index}), n
        assert np.all(a[hits, 1] == 0)


PINNED_SHA256_4096 = "f6f948d9d86d54152867c5b9decf53fc9941f60dec2ab5a4f369ca85227b2474"
