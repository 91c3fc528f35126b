#!/usr/bin/env python3
"""A randomised soak of bin/papr ON THE GPU BOX against the compiled reference (oracle/_ref/papr; the oracle port where it is
absent): N random cfiles (1000 ... 24 million samples, random seed and scale, the bench's spikes or none, 0-1 stray floats and 0-3
stray bytes at the end) through the CLI resident, streamed (a few MiB of HBM: the one-sweep ingest, its reader threads and
chunks), as three shards on one GPU over the in-process hub, as three streamed shards, and through a FIFO (windows of HBM of a
random size, against the reference through a FIFO of its own) — default and -g each: stdout must be the reference's, byte for byte.   gpurun -- 'python tools/cli_soak.py 40'"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(777)
    ref = os.path.join(ROOT, "oracle", "_ref", "papr")
    if not os.path.exists(ref):
        ref = os.path.join(ROOT, "oracle", "papr_oracle")
    cli = os.path.join(ROOT, "bin", "papr")
    path = "/dev/shm/cli_soak.cfile"
    bad = runs = 0
    for it in range(n_iter):
        n = int(10 ** rng.uniform(3.0, 7.38))
        args = [os.path.join(ROOT, "oracle", "mkcfile"), path, str(n), "--seed", str(int(rng.integers(1, 1 << 31))),
                "--extra-floats", str(int(rng.integers(0, 2))), "--extra-bytes", str(int(rng.integers(0, 4)))]
        if rng.integers(0, 3):
            args.append("--spike")
        if rng.integers(0, 3) == 0:
            args += ["--scale", str(float(rng.choice([0.01, 3.0, 100.0])))]
        subprocess.check_call(args, stdout=subprocess.DEVNULL)
        size_mb = max(1, os.path.getsize(path) >> 20)
        for mode in ([], ["-g"]):
            want = subprocess.run([ref] + mode + [path], capture_output=True).stdout
            small = str(max(2, size_mb // int(rng.integers(3, 9))))
            for name, env in (("resident", {}), ("streamed", {"PAPR_HBM_BUDGET_MB": small}),
                              ("three shards", {"PAPR_GPUS": "3", "PAPR_OVERSUBSCRIBE": "1", "PAPR_XCH": "threads"}),
                              ("three shards, streamed", {"PAPR_GPUS": "3", "PAPR_OVERSUBSCRIBE": "1", "PAPR_HBM_BUDGET_MB": small})):
                p = subprocess.run([cli] + mode + [path], capture_output=True, env=dict(os.environ, **env), timeout=300)
                runs += 1
                if p.returncode != 0 or p.stdout != want:
                    bad += 1
                    print("MISMATCH", it, n, " ".join(args[3:]), mode, name, env, "rc", p.returncode, p.stderr[-300:], flush=True)
            # ... and as a stream that cannot be rewound: through a FIFO, one window of HBM of a random size (round 6), against the
            # reference fed through a FIFO of its own (its pass 2 counts nothing there: papr.c:142-143)
            fifo = "/dev/shm/cli_soak.fifo"
            outs = []
            for prog, env in ((ref, {}), (cli, {"PAPR_STREAM_WINDOW_MB": str(int(rng.integers(2, 40)))})):
                if os.path.exists(fifo):
                    os.unlink(fifo)
                os.mkfifo(fifo)
                feeder = subprocess.Popen(f"exec cat '{path}' > '{fifo}'", shell=True)
                outs.append(subprocess.run([prog] + mode + [fifo], capture_output=True, env=dict(os.environ, **env), timeout=300))
                feeder.wait(timeout=60)
                os.unlink(fifo)
            runs += 1
            if outs[1].returncode != outs[0].returncode or outs[1].stdout != outs[0].stdout or outs[1].stderr != outs[0].stderr:
                bad += 1
                print("MISMATCH", it, n, " ".join(args[3:]), mode, "through a FIFO", "rc", outs[1].returncode, outs[1].stderr[-300:], flush=True)
        if it % 10 == 9:
            print("...", it + 1, "files,", runs, "runs,", bad, "mismatches", flush=True)
    os.unlink(path)
    print("files", n_iter, "runs", runs, "mismatches", bad, "reference:", os.path.relpath(ref, ROOT))


if __name__ == "__main__":
    main()
