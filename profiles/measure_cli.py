#!/usr/bin/env python3
"""End-to-end drop-in time of BASELINE cfg2 through bin/debruijn_contig (host parse -> GPU -> host KmerSet): writes the 10 M reads
as one-line FASTA to $TMPDIR, runs the command line with DBGK_TIMINGS=1 and prints wall clock + the phase lines.
    python profiles/measure_cli.py [n_reads] [extra env as K=V ...]"""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402  (the generator only: writing the input file)

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
extra = dict(kv.split("=", 1) for kv in sys.argv[2:])
tmp = os.environ.get("TMPDIR", "/tmp")
fa, libf = os.path.join(tmp, "cfg2_cli.fa"), os.path.join(tmp, "cfg2_cli.lib")
if not os.path.exists(fa) or os.path.getsize(fa) < n_reads * 150:
    P = O.synth_params(50_000_000, 150, cfg=2)
    t0 = time.time()
    O.lib().orc_synth_write_file(C.byref(P), 0, n_reads, os.fsencode(fa), 2, 0)
    print("wrote %s (%.2f GB) in %.0f s" % (fa, os.path.getsize(fa) / 1e9, time.time() - t0), flush=True)
open(libf, "w").write(fa + "\n")
cli = os.path.join(ROOT, "dbg_assembly_amd", "bin", "debruijn_contig")
env = dict(os.environ, DBGK_TIMINGS="1")
env.update(extra)
for rep in range(3):
    time.sleep(2.0)   # (the driver wipes the ~25 GB of device memory of the run before; a process started at once can wait > 1 s in its first hipMalloc)
    t0 = time.time()
    r = subprocess.run([cli, "-k", "31", "-f", "2", "-t", "16", "-i", "0.6", "-o", os.path.join(tmp, "cfg2_cli_out"), libf], env=env,
                       capture_output=True, text=True)
    wall = time.time() - t0
    lines = [l for l in r.stderr.splitlines() if l.startswith(("Host phases", "GPU phases", "count:", "Wall phases", "Reader", "dbgk_create", "dbgk export"))]
    print("run %d: rc %d wall %.3f s %s" % (rep, r.returncode, wall, " | ".join(lines)), flush=True)
