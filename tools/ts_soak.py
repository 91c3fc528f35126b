#!/usr/bin/env python3
"""A randomised differential soak of the packet scan ON THE GPU BOX: N random streams (150 000 ... 400 000 packets, 188- and 192-byte
units alternating) with 100 ... 1500 damaged spots of seven kinds at random places — bytes inserted (1 ... 400; 1 ... 7), bytes missing,
a sync byte overwritten, a run overwritten with false sync bytes in it, whole packets gone, an adaptation_field_length that runs
over its packet (what leaves a span un-clean on its boundary) — each scanned in all three forms with 256, 600 and 48 spans, twice
(the second scan starts in the form the first ended in), and compared with the oracle: report, packet count, the three tables,
both event lists.   gpurun -- 'python tools/ts_soak.py 400'   (7200 scans, ~80 s)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from dtv_utils_amd import ts
import ts_oracle

def ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return ts.TsHip(0)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v

def damaged(clean, unit, hdmv, rng, nplaces):
    npackets = clean.size // unit
    places = np.sort(rng.choice(npackets - 8, size=nplaces, replace=False) + 4)
    parts, at = [], 0
    for k in places:
        cut = int(k) * unit + int(rng.integers(0, unit))
        if cut < at:
            continue
        parts.append(clean[at:cut])
        kind = int(rng.integers(0, 7))
        if kind == 0:
            parts.append(rng.integers(0, 256, size=int(rng.integers(1, 401)), dtype=np.uint8)); at = cut
        elif kind == 1:
            at = cut + int(rng.integers(1, 401))
        elif kind == 2:
            nxt = (int(k) + 1) * unit + (4 if hdmv else 0)
            parts.append(clean[cut:nxt]); parts.append(np.array([int(rng.integers(0, 256))], dtype=np.uint8)); at = nxt + 1
        elif kind == 3:
            n = int(rng.integers(1, 600)); junk = rng.integers(0, 256, size=n, dtype=np.uint8)
            junk[rng.integers(0, n, size=max(1, n // 40))] = 0x47
            parts.append(junk); at = cut + n
        elif kind == 4:
            at = cut + unit * int(rng.integers(1, 4))
        elif kind == 5:
            parts.append(rng.integers(0, 256, size=int(rng.integers(1, 8)), dtype=np.uint8)); at = cut
        else:   # an adaptation field length that runs over the packet (what leaves a span un-clean on its boundary)
            p0 = int(k) * unit + (4 if hdmv else 0)
            if p0 + 5 <= cut:
                continue
            parts.append(clean[cut:p0 + 3]); parts.append(np.array([0x30 | int(rng.integers(0, 16)), int(rng.integers(184, 256))], dtype=np.uint8)); at = p0 + 5
    parts.append(clean[at:])
    return np.concatenate(parts).tobytes()

bad = 0; total = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    rng = np.random.default_rng(9000 + seed)
    hdmv = bool(seed & 1); unit = 192 if hdmv else 188
    npackets = int(rng.integers(150_000, 400_000))
    with ts.TsHip(0) as g:
        g.generate(npackets, seed=0x7500001 + seed, hdmv=hdmv)
        clean = np.frombuffer(g.download(0, npackets * unit), dtype=np.uint8)
    data = damaged(clean, unit, hdmv, rng, int(rng.integers(100, 1500)))
    ref = ts_oracle.scan_mem(data, hdmv); want = ts_oracle.report_lines(ref)
    for form in ("auto", "slots", "full"):
        for spans in ("256", "600", "48"):
            with ctx({"TS_SCAN_FORM": form, "TS_SCAN_SPANS": spans, "TS_SCAN_MIN_SPAN": "16384"}) as g:
                g.upload(data)
                for rep in range(2):
                    res = g.scan(hdmv); total += 1
                    cnt, first, last = res.tables()
                    ok = res.report() == want and res.packets == ref["packets"] and np.array_equal(cnt, ref["count"]) and \
                        np.array_equal(first, ref["first"]) and np.array_equal(last, ref["last"]) and \
                        res.sync_error_list() == ref["sync_errors"] and res.discontinuity_list() == ref["discontinuities"]
                    if not ok:
                        bad += 1
                        print("MISMATCH seed", seed, form, spans, "rep", rep, "launches", res.launches, "packets", res.packets, ref["packets"], flush=True)
print("scans", total, "mismatches", bad)
