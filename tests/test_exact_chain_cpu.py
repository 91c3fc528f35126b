"""papr_exact_chain is host-side product code (it replays a shard's "sum program" into the
reference's sequential double sum), so it is tested here without a GPU: the programs are built in
numpy by a model of what papr_exact.hip computes on the device — per-segment rounding pairs obtained
by running the additions from the two canonical entry states of a binade — and the replay must equal
the oracle's sequential sum bit for bit."""
import struct

import numpy as np
import pytest

TILE, SEG, GROUP = 2048, 1024, 128
AMBIG, ZERO = -(2 ** 31), -(2 ** 31) + 1
MAGIC, VERSION = 0x31535850, 3
RUN = 16


def powers(iq):
    re, im = iq[0::2], iq[1::2]
    return (re * re) + (im * im)          # float32, separate roundings


def seg_pair(pw, E):
    """(D0, D1) of one segment for binade E: run the adds from 2^E and 2^E + ulp."""
    m0 = np.float64(2.0) ** E
    m1 = m0 + np.float64(2.0) ** (E - 52)
    x0, x1 = m0, m1
    for v in pw.astype(np.float64):
        x0 = x0 + v
        x1 = x1 + v
    return float(x0 - m0), float(x1 - m1)


def classify(P, s, delta):
    if s == 0.0:
        return ZERO
    if not (s > 0 and np.isfinite(s) and P > 0 and np.isfinite(P)):
        return AMBIG
    E = int(np.floor(np.log2(P)))
    if 2.0 ** E > P:
        E -= 1
    m0 = 2.0 ** E
    if P * (1 - delta) >= m0 and (P + s) * (1 + delta) < 2 * m0 and s <= 0.25 * m0:
        return E
    return AMBIG


def run_records(pw_tile, P_tile, runs):
    """What the pack kernel appends to a raw tile: per 16-sample run the binade of an APPROXIMATE prefix and the pair for
    it (none where the canonical chains leave the binade).  runs: "model" as the device does it, "none" no pairs at all,
    "wrong" every pair's binade off by one (the host must notice and add those runs sample by sample)."""
    es, ds = [], []
    P = P_tile
    for r in range(TILE // RUN):
        pw = pw_tile[r * RUN:(r + 1) * RUN]
        s = float(np.sum(pw.astype(np.float64)))
        e, d = AMBIG, (0.0, 0.0)
        if s == 0.0:
            e = ZERO
        elif runs != "none" and s > 0 and np.isfinite(s) and P > 0 and np.isfinite(P):
            E = int(np.floor(np.log2(P)))
            if 2.0 ** E > P:
                E -= 1
            if E >= -960:
                if runs == "wrong":
                    E += 1
                m0 = np.float64(2.0) ** E
                d0, d1 = seg_pair(pw, E)
                if (m0 + np.float64(2.0) ** (E - 52)) + d1 < 2 * m0:
                    e, d = E, (d0, d1)
        es.append(e)
        ds.append(d)
        P += s
    return struct.pack(f"<{TILE // RUN}i", *es) + b"".join(struct.pack("<dd", *d) for d in ds)


def build_program(iq, before=0.0, force_mixed=True, runs="model"):
    """Every group is emitted as 'mixed' (per-tile classes + per-segment pairs), unsafe tiles raw."""
    pw = powers(iq)
    n = pw.size
    ntiles, tail = n // TILE, n % TILE
    ngroups = (ntiles + GROUP - 1) // GROUP
    delta = max(1e-6, 8 * n * 2.0 ** -53)
    tile_sums = [float(np.sum(pw[t * TILE:(t + 1) * TILE].astype(np.float64))) for t in range(ntiles)]
    P, cls, prefix = before, [], []
    for s in tile_sums:
        cls.append(classify(P, s, delta))
        prefix.append(P * (1 + 3e-7))    # (the device's prefix is only good to `delta`)
        P += s
    groups = b"".join(struct.pack("<iidd", AMBIG, 0, 0.0, 0.0) for _ in range(ngroups))
    mixed, raw = b"", b""
    nraw = 0
    for g in range(ngroups):
        te = [cls[g * GROUP + j] if g * GROUP + j < ntiles else ZERO for j in range(GROUP)]
        pairs = []
        for j in range(GROUP):
            t = g * GROUP + j
            for h in range(2):
                if t < ntiles and te[j] not in (AMBIG, ZERO):
                    pairs.append(seg_pair(pw[t * TILE + h * SEG:t * TILE + (h + 1) * SEG], te[j]))
                else:
                    pairs.append((0.0, 0.0))
            if t < ntiles and te[j] == AMBIG:
                raw += (struct.pack("<Q", t) + pw[t * TILE:(t + 1) * TILE].tobytes() +
                        run_records(pw[t * TILE:(t + 1) * TILE], prefix[t], runs))
                nraw += 1
        mixed += struct.pack("<Q", g) + struct.pack(f"<{GROUP}i", *te) + b"".join(struct.pack("<dd", *p) for p in pairs)
    header = struct.pack("<IIQQQIIII", MAGIC, VERSION, n, ntiles, ngroups, tail, ngroups, nraw, 0)
    return header + groups + mixed + raw + iq[2 * ntiles * TILE:].tobytes(), nraw


@pytest.mark.parametrize("case", ["gauss", "growing", "constant", "zeros_mixed", "tiny", "short"])
def test_chain_replays_the_sequential_sum(pkg, orc, case):
    rng = np.random.default_rng(12)
    n = 700 if case == "short" else 3 * GROUP * TILE // 2 + 777
    iq = rng.standard_normal(2 * n).astype(np.float32)
    if case == "growing":
        iq *= np.exp2(np.repeat(np.linspace(-15, 15, n), 2)).astype(np.float32)
    elif case == "constant":
        iq[0::2], iq[1::2] = 0.75, 0.25
    elif case == "zeros_mixed":
        iq[2 * 5000:2 * 200000] = 0
    elif case == "tiny":
        iq *= np.float32(3.3717e-8)
    prog, nraw = build_program(iq)
    want = orc.run_mem(iq, False)["sum"]
    assert pkg.exact_chain([prog]) == want
    # raw tiles without run pairs, and with pairs built for the wrong binade: the same sum, sample by sample
    assert pkg.exact_chain([build_program(iq, runs="none")[0]]) == want
    assert pkg.exact_chain([build_program(iq, runs="wrong")[0]]) == want
    if case not in ("short", "growing"):        # (a sum that doubles every few tiles is mostly binade crossings)
        assert nraw < (n // TILE) // 2          # most tiles really go through the pair path
    # two shards: the second program is built knowing only the accurate sum of the first
    cut = (n // 2) // 8192 * 8192
    if cut:
        a, b = iq[:2 * cut], iq[2 * cut:]
        before = float(np.sum(powers(a).astype(np.float64)))
        assert pkg.exact_chain([build_program(a)[0], build_program(b, before)[0]]) == want


def test_chain_continued_window_by_window(pkg, orc):
    """papr_exact_chain_continue (what papr_hip_stream_stats does for a stream that crosses one window of HBM many times):
    every window's program is built from the EXACT accumulator in front of it and replayed at once; the accumulator behind
    the last window is the oracle's sequential sum of the whole stream, whatever the windows' size."""
    rng = np.random.default_rng(5)
    n = 5 * GROUP * TILE // 4 + 333
    iq = (rng.standard_normal(2 * n) * np.exp2(np.repeat(np.linspace(-6, 4, n), 2))).astype(np.float32)
    want = orc.run_mem(iq, False)["sum"]
    for window in (8192, 8192 * 5, GROUP * TILE):
        S = 0.0
        for first in range(0, n, window):
            S = pkg.exact_chain_continue(S, [build_program(iq[2 * first:2 * min(n, first + window)], S)[0]])
        assert S == want, window
    assert pkg.exact_chain_continue(want, []) == want
    with pytest.raises(pkg.PaprError):
        pkg.exact_chain_continue(float("nan"), [])


def test_chain_rejects_malformed_programs(pkg):
    rng = np.random.default_rng(1)
    iq = rng.standard_normal(2 * 100000).astype(np.float32)   # 48 tiles: one group, the later tiles safe
    prog, _ = build_program(iq)
    assert pkg.exact_chain([]) == 0.0
    for bad in (prog[:40], b"\x00" * 64, prog[:-8], prog[:4] + b"\x09" + prog[5:]):
        with pytest.raises(pkg.PaprError) as e:
            pkg.exact_chain([bad])
        assert e.value.code == -3
    # a pair applied in the wrong binade is caught, not silently added
    h = list(struct.unpack_from("<IIQQQIIII", prog, 0))
    assert h[4] == 1                             # one group
    hacked = bytearray(prog)
    off_te = 48 + 24 + 8                         # header + group table + mixed.group
    te = list(struct.unpack_from(f"<{GROUP}i", hacked, off_te))
    k = next(i for i, e in enumerate(te) if e not in (AMBIG, ZERO))
    te[k] += 3
    struct.pack_into(f"<{GROUP}i", hacked, off_te, *te)
    with pytest.raises(pkg.PaprError) as e:
        pkg.exact_chain([bytes(hacked)])
    assert e.value.code == -8
