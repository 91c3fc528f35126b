"""Seeded synthetic MPEG-2 transport streams for the TS packet-scan tests.

Own generator (numpy, seeded): 188-byte packets (or 192-byte HDMV packets: 4 bytes of tp_extra_header in front),
a handful of PIDs with realistic proportions, null packets, a well-formed PAT on PID 0, adaptation fields of every
legal length (and, on request, illegal ones), transport_error_indicator packets, and damage: leading / inserted
garbage with or without 0x47 bytes, deleted bytes, a truncated tail.  `offset` shifts the whole stream so that packets
can be made to end exactly one byte past a 16384-byte read (the reference's chunk-boundary quirk, oracle/ts_oracle.c).
"""
from __future__ import annotations

import numpy as np

PAT_PROGRAMS = ((1, 0x0030), (2, 0x0040))   # program numbers the tests never ask for are used on the command line


def _crc32_mpeg(data: bytes) -> int:
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc


def pat_payload() -> bytes:
    body = bytearray([0x00, 0x01, 0xC1, 0x00, 0x00])            # ts_id 1, version 0, current, section 0 of 0
    for prog, pid in PAT_PROGRAMS:
        body += bytes([prog >> 8, prog & 0xFF, 0xE0 | (pid >> 8), pid & 0xFF])
    length = len(body) + 4
    sec = bytearray([0x00, 0xB0 | (length >> 8), length & 0xFF]) + body
    sec += _crc32_mpeg(bytes(sec)).to_bytes(4, "big")
    return bytes([0x00]) + bytes(sec)                            # pointer_field 0


def packet(rng, pid: int, cc: int, *, tei=0, pusi=0, af_len=None, payload: bytes = None, no_sync_in_payload=False,
           no_payload=False) -> bytes:
    afc = 1 if af_len is None else 3
    if no_payload:   # adaptation field only (adaptation_field_control = 2): its continuity counter is neither checked nor kept
        afc, af_len = 2, 183
    head = bytes([0x47, (tei << 7) | (pusi << 6) | ((pid >> 8) & 0x1F), pid & 0xFF, (afc << 4) | (cc & 0xF)])
    body = bytearray()
    if af_len is not None:
        body.append(af_len & 0xFF)
        if af_len > 0:
            flags = int(rng.integers(0, 256)) & ~0x10 if af_len < 7 else int(rng.integers(0, 256))
            body.append(flags)
            body += bytes(rng.integers(0, 256, max(af_len - 1, 0), dtype=np.uint8))
    room = 184 - len(body)
    if room > 0:
        pay = payload if payload is not None else bytes(rng.integers(0, 256, room, dtype=np.uint8))
        pay = (pay + b"\xff" * room)[:room]
        body += pay
    body = bytearray(body[:184])
    if no_sync_in_payload:
        for k in range(len(body)):
            if body[k] == 0x47:
                body[k] = 0x48
    return head + bytes(body)


def make_stream(seed: int, npackets: int, *, hdmv=False, pat=True, pids=(0x100, 0x101, 0x102, 0x1FFB, 0x1FFF, 0x31),
                af_rate=0.2, bad_af_rate=0.0, tei_rate=0.01, offset_garbage=0, garbage_has_sync=False, damage=(),
                truncate=0, last_byte_sync_at=None, plain_at=(), extra_header_at=None, cc_jump_at=None, drop_at=(),
                no_payload_at=(), repeat_at=()) -> bytes:
    """damage: list of (packet_index, kind, amount): kind 'insert' (amount garbage bytes before that packet),
    'delete' (drop `amount` bytes from the start of that packet).  last_byte_sync_at: packet index whose last
    payload byte is forced to 0x47 (a false sync byte for the chunk-boundary quirk).  plain_at: packet indices that are
    forced to be an ordinary payload-only packet on PID 0x100 (what the quirk needs).  extra_header_at: {packet index: the
    four tp_extra_header bytes of that HDMV unit}.  Continuity counters (xport.c:2872-2889): cc_jump_at {packet index: what
    is added to that packet's counter, and to its PID's from there on}; drop_at: packets that are left out of the stream (their
    PID's counter still advances: a lost packet); no_payload_at: packets that carry an adaptation field only; repeat_at:
    packets that are sent twice (the copy repeats the counter)."""
    rng = np.random.default_rng(seed)
    cc = {}
    out = bytearray()

    def garbage(n):
        g = rng.integers(0, 256, n, dtype=np.uint8)
        if not garbage_has_sync:
            g[g == 0x47] = 0x46
        return bytes(g)

    out += garbage(offset_garbage)
    dmg = {d[0]: d for d in damage}
    for k in range(npackets):
        if pat and k % 40 == 0:
            pid, payload, pusi = 0, pat_payload(), 1
        else:
            pid, payload, pusi = int(pids[int(rng.integers(0, len(pids)))]), None, int(rng.integers(0, 8) == 0)
            if pid == 0x1FFB:
                pusi = 0    # no PSIP section ever starts: the reference would parse it (and act on a Master Guide Table)
        if k in plain_at:   # (decided before the counter is taken: the packet counts for the PID it ends up on)
            pid = 0x100
        c = cc.get(pid, 0)
        if cc_jump_at and k in cc_jump_at:
            c = (c + cc_jump_at[k]) & 15
        cc[pid] = (c + 1) & 15
        af_len = None
        if pid != 0 and rng.random() < af_rate:
            af_len = int(rng.choice([0, 1, 7, 20, 100, 181, 182, 183]))
            if rng.random() < bad_af_rate:
                af_len = int(rng.choice([184, 200, 255]))
        tei = int(rng.random() < tei_rate)
        if k in plain_at:
            pid, payload, pusi, af_len, tei = 0x100, None, 0, None, 0
        p = bytearray(packet(rng, pid, c, tei=tei, pusi=pusi, af_len=af_len, payload=payload, no_payload=k in no_payload_at))
        if k in drop_at:
            continue
        if last_byte_sync_at is not None and k == last_byte_sync_at:
            p[187] = 0x47
        if hdmv:
            extra = bytearray(rng.integers(0, 256, 4, dtype=np.uint8).tobytes())
            if extra_header_at and k in extra_header_at:
                extra = bytearray(extra_header_at[k])
            p = extra + p
        if k in dmg:
            _, kind, amount = dmg[k]
            if kind == "insert":
                out += garbage(amount)
            elif kind == "delete":
                p = p[amount:]
        out += p
        if k in repeat_at:
            out += p
    if truncate:
        out = out[:len(out) - truncate]
    return bytes(out)


def quirk_offset(packet_index: int, packet_size: int = 188) -> int:
    """Leading garbage that makes packet `packet_index` start at 16197 (mod 16384): it then ends one byte past a read.
    (192-byte units: the packet starts four bytes into its unit.)"""
    return (16197 - (packet_size - 188) - packet_index * packet_size) % 16384


# name -> kwargs: the committed fixtures (tests/golden/make_golden_ts.py records the reference's lines for them)
FIXTURES = {
    "ts_clean": dict(seed=101, npackets=1500),
    "ts_no_pat": dict(seed=102, npackets=900, pat=False),
    "ts_af_heavy": dict(seed=103, npackets=1200, af_rate=0.8),
    "ts_tei": dict(seed=104, npackets=800, tei_rate=0.3),
    "ts_lead_garbage": dict(seed=105, npackets=700, offset_garbage=1234),
    "ts_lead_garbage_sync": dict(seed=106, npackets=700, offset_garbage=999, garbage_has_sync=True),
    "ts_insert": dict(seed=107, npackets=1000, damage=[(300, "insert", 77), (650, "insert", 1)]),
    "ts_insert_sync": dict(seed=108, npackets=1000, garbage_has_sync=True, damage=[(200, "insert", 500), (700, "insert", 4000)]),
    "ts_delete": dict(seed=109, npackets=1000, damage=[(400, "delete", 10), (800, "delete", 187)]),
    "ts_truncated": dict(seed=110, npackets=500, truncate=100),
    "ts_trunc_header": dict(seed=111, npackets=300, truncate=186),
    "ts_quirk": dict(seed=112, npackets=600, pat=False, offset_garbage=quirk_offset(90)),
    "ts_quirk_false_sync": dict(seed=113, npackets=600, pat=False, offset_garbage=quirk_offset(90), last_byte_sync_at=90),
    "ts_quirk_af": dict(seed=114, npackets=600, pat=False, af_rate=1.0, offset_garbage=quirk_offset(120)),
    "ts_bad_af": dict(seed=115, npackets=800, af_rate=0.5, bad_af_rate=0.3),
    "ts_hdmv": dict(seed=116, npackets=1200, hdmv=True),
    "ts_hdmv_damaged": dict(seed=117, npackets=900, hdmv=True, damage=[(250, "insert", 33), (600, "delete", 5)]),
    "ts_only_garbage": dict(seed=118, npackets=0, offset_garbage=3000),
    "ts_two_packets": dict(seed=119, npackets=2, pat=False),
    # the read-boundary quirk where `skipped 1 bytes` is NOT the whole story: the quirk packet ends the stream (no line
    # at all), damage follows it (one line with the sum), the byte the HDMV search tests first is a 0x47 (re-lock one
    # byte early)
    "ts_quirk_last": dict(seed=120, npackets=91, pat=False, offset_garbage=quirk_offset(90), plain_at=(90,)),
    "ts_quirk_then_garbage": dict(seed=121, npackets=400, pat=False, offset_garbage=quirk_offset(90), plain_at=(90,),
                                  damage=[(91, "insert", 50)]),
    "ts_quirk_then_delete": dict(seed=122, npackets=400, pat=False, offset_garbage=quirk_offset(90), plain_at=(90,),
                                 damage=[(91, "delete", 3)]),
    "ts_quirk_then_truncated": dict(seed=123, npackets=92, pat=False, offset_garbage=quirk_offset(90), plain_at=(90,),
                                    truncate=40),
    "ts_hdmv_quirk": dict(seed=124, npackets=500, hdmv=True, pat=False, offset_garbage=quirk_offset(85, 192), plain_at=(85,),
                          extra_header_at={86: b"\x11\x22\x33\x44"}),
    "ts_hdmv_quirk_false_sync": dict(seed=125, npackets=500, hdmv=True, pat=False, offset_garbage=quirk_offset(85, 192),
                                     plain_at=(85,), extra_header_at={86: b"\x11\x22\x33\x47"}),
    "ts_hdmv_quirk_last": dict(seed=126, npackets=86, hdmv=True, pat=False, offset_garbage=quirk_offset(85, 192), plain_at=(85,)),
    # continuity counters (xport.c:2872-2889): jumps on every kind of PID (the null PID and PID 0 stay silent), lost and
    # repeated packets, packets without a payload (neither checked nor remembered), error-indicator packets (checked all
    # the same), and discontinuities between sync errors
    "ts_cc_jumps": dict(seed=127, npackets=3000, tei_rate=0.05, cc_jump_at={k: 1 + k % 14 for k in range(50, 3000, 97)}),
    "ts_cc_lost_and_repeated": dict(seed=128, npackets=2500, drop_at=tuple(range(30, 2500, 171)), repeat_at=tuple(range(77, 2500, 233)),
                                    no_payload_at=tuple(range(5, 2500, 41))),
    "ts_cc_between_sync_errors": dict(seed=129, npackets=2000, garbage_has_sync=True, cc_jump_at={k: 3 for k in range(10, 2000, 53)},
                                      damage=[(100, "insert", 700), (101, "delete", 9), (102, "insert", 2), (900, "delete", 100),
                                              (1500, "insert", 5000)], drop_at=(103, 104, 901)),
    "ts_cc_hdmv": dict(seed=130, npackets=1500, hdmv=True, cc_jump_at={k: 7 for k in range(20, 1500, 61)}, drop_at=(700, 701),
                       damage=[(400, "insert", 33)]),
}


def fixture_bytes(name: str) -> bytes:
    return make_stream(**FIXTURES[name])


def is_hdmv(name: str) -> bool:
    return bool(FIXTURES[name].get("hdmv"))
