#!/usr/bin/env python3
"""Regenerate tests/golden/: small .cfile fixtures + the REAL reference's stdout.

Run in the build container only (needs /root/reference): it compiles the
reference papr.c where it lies into oracle/_ref/papr (oracle/Makefile `ref`),
writes every fixture below, and records the reference's stdout/stderr/exit code
for `papr <f>` and `papr -g <f>` in manifest.json + <name>.default.txt /
<name>.graph.txt.  Fixtures are data only (inputs and expected outputs); no
reference source is stored.

    python tests/golden/make_golden.py            # small fixtures
    python tests/golden/make_golden.py --big FILE # also record the 10 GiB bench
                                                  # workload from a file made by
                                                  # `oracle/mkcfile FILE 1342177280 --spike`
"""
import argparse
import json
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ORACLE = os.path.join(ROOT, "oracle")
MKCFILE = os.path.join(ORACLE, "mkcfile")
REF = os.path.join(ORACLE, "_ref", "papr")


def mk(name, n, *extra):
    subprocess.check_call([MKCFILE, os.path.join(HERE, name + ".cfile"), str(n), *map(str, extra)])


def raw(name, floats=(), tail=b""):
    with open(os.path.join(HERE, name + ".cfile"), "wb") as f:
        f.write(np.asarray(floats, dtype=np.float32).tobytes())
        f.write(tail)


def build_fixtures():
    # hand-specified vectors from SURVEY.md section 8(a)
    raw("k8", [1, 0, 0, 1, 1, 1, 2, 0, 0, -2, -1, -1, 3, 4, 0.5, 0.5])
    tie = [0.1] * 20
    tie[6], tie[7], tie[14], tie[15] = 2, 1, 1, 2
    raw("tie", tie)
    raw("one", [3.0])
    raw("oneb", [3.0], b"\xa5")
    raw("empty")
    raw("stray3", [], b"\x01\x02\x03")              # not even one whole float
    raw("zeros", [0.0] * 2000)
    raw("negzero", [-0.0, 0.0, 0.0, -0.0] * 100)
    # BASELINE.json configs[0]: 1 MiB synthetic gr_complex
    mk("g1m", 131072)
    mk("spike20k", 20000, "--spike")
    # tail quirks (papr.c:102-103 with the stale static buffer)
    mk("odd", 20000, "--extra-floats", 1)
    mk("oddb", 20000, "--extra-floats", 1, "--extra-bytes", 3)
    mk("evenb", 5000, "--extra-bytes", 2)
    mk("chunk3odd", 24576, "--extra-floats", 1, "--extra-bytes", 2)
    mk("chunk1exact", 8192)
    mk("chunk1plus", 8192, "--extra-floats", 1)
    mk("tileplus", 4096 * 3 + 17)
    # non-finite samples
    mk("nan_i", 5000, "--set", 100, "nan", 1)
    mk("nan_q_neg", 5000, "--set", 100, 1, "-nan")
    mk("nan_both", 5000, "--set", 100, "nan", "-nan", "--set", 200, "-nan", "nan")
    mk("nan_order", 5000, "--set", 4500, "nan", 1, "--set", 4097, 1, "-nan")
    mk("inf", 5000, "--set", 100, "inf", 1)
    mk("inf_nan", 5000, "--set", 100, "inf", 1, "--set", 50, "-nan", 0)
    mk("overflow", 5000, "--scale", 1e14)
    # amplitude scales: GNU Radio tx scaling, denormal powers
    mk("tiny", 50000, "--scale", 3.3717e-8)
    mk("denorm", 5000, "--scale", 4e-26)
    # ties: equal peaks / equal extrema in different places, first must win
    mk("ties", 30000, "--set", 29000, 5, 5, "--set", 123, 5, 5, "--set", 7000, -5, -5, "--set", 6999, -5, -5)
    # a signalling NaN bit pattern in I and in Q
    snan = np.frombuffer(struct.pack("<I", 0x7FA00001), dtype=np.float32)[0]
    base = np.fromfile(os.path.join(HERE, "nan_i.cfile"), dtype=np.float32).copy()
    base[2 * 100] = 0.25
    raw_bits = base.view(np.uint32)
    raw_bits[2 * 300] = 0x7FA00001
    raw_bits[2 * 17 + 1] = 0xFFA00001
    raw("snan", base)
    del snan
    # producer-side formats: what the GNU Radio flowgraphs that feed papr write —
    # blocks.file_sink(gr.sizeof_gr_complex) after a constant scale (dvbt-blade.py:189,214 / dvbt2-blade.py:132,159 /
    # qam-blade.py:59,83).  Seeded numpy stand-ins with the same statistics and amplitude scales.
    rng = np.random.default_rng(20260928)
    n = 40000
    ofdm = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) / np.sqrt(2.0)
    raw("ofdm_dvbt", np.column_stack([ofdm.real, ofdm.imag]).ravel() * 0.0022097087)
    # DVB-T2 with PAPR reduction: the envelope is clipped ~8.6 dB above the mean, scale 0.2
    clip = np.sqrt(10 ** 0.86)
    mag = np.abs(ofdm)
    t2 = ofdm * np.minimum(1.0, clip / np.maximum(mag, 1e-30))
    raw("ofdm_dvbt2_clipped", np.column_stack([t2.real, t2.imag]).ravel() * 0.2)
    # J.83B 64-QAM through a root-raised-cosine interpolator (x2), roll-off 0.18
    sym = (rng.integers(0, 8, n // 2) * 2 - 7) + 1j * (rng.integers(0, 8, n // 2) * 2 - 7)
    up = np.zeros(n, dtype=np.complex128)
    up[::2] = sym
    k = np.arange(-44, 45) / 2.0
    beta = 0.18
    with np.errstate(divide="ignore", invalid="ignore"):
        rrc = (np.sin(np.pi * k * (1 - beta)) + 4 * beta * k * np.cos(np.pi * k * (1 + beta))) / \
              (np.pi * k * (1 - (4 * beta * k) ** 2))
    rrc[np.isnan(rrc) | np.isinf(rrc)] = 1.0 - beta + 4 * beta / np.pi
    qam = np.convolve(up, rrc / np.sqrt(np.sum(rrc ** 2)), mode="same") * 0.1
    raw("qam64_rrc", np.column_stack([qam.real, qam.imag]).ravel())
    # (added later; drawn after everything above so that the earlier fixtures keep their bytes)
    # DVB-S2 32-APSK (rings 4 + 12 + 16, radius ratios 2.84 / 5.27) through the same RRC, roll-off 0.2 filter shape
    ring = rng.integers(0, 32, n // 2)
    radius = np.where(ring < 4, 1.0, np.where(ring < 16, 2.84, 5.27)) / 5.27
    count = np.where(ring < 4, 4, np.where(ring < 16, 12, 16))
    index = np.where(ring < 4, ring, np.where(ring < 16, ring - 4, ring - 16))
    apsk = radius * np.exp(2j * np.pi * (index + 0.5) / count)
    up = np.zeros(n, dtype=np.complex128)
    up[::2] = apsk
    s2 = np.convolve(up, rrc / np.sqrt(np.sum(rrc ** 2)), mode="same") * 0.5
    raw("apsk32_dvbs2_rrc", np.column_stack([s2.real, s2.imag]).ravel())
    # ATSC 8-VSB: 8-level symbols + pilot, one sideband kept (analytic-signal approximation of the VSB filter)
    vsb_sym = (rng.integers(0, 8, n) * 2 - 7 + 1.25).astype(np.float64)
    spec = np.fft.fft(vsb_sym)
    spec[n // 2 + 1:] = 0.0          # drop the lower sideband
    spec[1:n // 2] *= 2.0
    vsb = np.fft.ifft(spec) * 0.02
    raw("vsb8_atsc", np.column_stack([vsb.real, vsb.imag]).ravel())
    # TDMA-like bursts: QPSK bursts of random length and level separated by exact silence (runs of +0 power)
    tdma = np.zeros(n, dtype=np.complex128)
    pos = 0
    while pos < n:
        gap = int(rng.integers(50, 900))
        length = int(rng.integers(200, 3000))
        level = 10.0 ** rng.uniform(-2.0, 0.0)
        a, b = min(n, pos + gap), min(n, pos + gap + length)
        tdma[a:b] = level * (rng.choice([-1.0, 1.0], b - a) + 1j * rng.choice([-1.0, 1.0], b - a)) / np.sqrt(2.0)
        pos = b
    raw("tdma_bursts", np.column_stack([tdma.real, tdma.imag]).ravel())


def record(path, graph):
    args = [REF] + (["-g"] if graph else []) + [path]
    p = subprocess.run(args, capture_output=True)
    return p.returncode, p.stdout, p.stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", help="path of the 10 GiB spike workload file (optional)")
    args = ap.parse_args()
    subprocess.check_call(["make", "-C", ORACLE, "mkcfile", "ref"])
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/papr missing: this script needs /root/reference")
    manifest_path = os.path.join(HERE, "manifest.json")
    manifest = {}
    if os.path.exists(manifest_path):
        manifest = json.load(open(manifest_path))
    build_fixtures()
    names = sorted(f[:-6] for f in os.listdir(HERE) if f.endswith(".cfile"))
    for name in names:
        path = os.path.join(HERE, name + ".cfile")
        entry = {"bytes": os.path.getsize(path)}
        for graph, tag in ((False, "default"), (True, "graph")):
            rc, out, err = record(path, graph)
            with open(os.path.join(HERE, f"{name}.{tag}.txt"), "wb") as f:
                f.write(out)
            entry[tag] = {"rc": rc, "stderr": err.decode(), "lines": out.count(b"\n")}
        manifest[name] = entry
    if args.big:
        # BASELINE.json configs[1] / configs[2]: the full-size bench workload.
        # Only the reference's stdout is committed; the GPU test regenerates
        # the same stream on the device from include/papr_synth.h.
        entry = {"bytes": os.path.getsize(args.big), "synthetic": "spike", "nsamples": os.path.getsize(args.big) // 8}
        for graph, tag in ((False, "default"), (True, "graph")):
            rc, out, err = record(args.big, graph)
            with open(os.path.join(HERE, f"big_spike10g.{tag}.txt"), "wb") as f:
                f.write(out)
            entry[tag] = {"rc": rc, "stderr": err.decode(), "lines": out.count(b"\n")}
        # the reference never prints its double accumulator in full; the oracle (pinned to the reference by
        # tests/test_oracle.py) restates papr.c:100-129, so its pass-1 sum is the sequential sum to reproduce
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        entry["oracle_sequential_sum_hex"] = ge.load_oracle().run_file(args.big, False)["sum"].hex()
        manifest["big_spike10g"] = entry
    json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)
    print(f"recorded {len(names)} fixtures")


if __name__ == "__main__":
    main()
