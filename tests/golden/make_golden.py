#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference): it (re)builds oracle/_ref/ref_dbg from
the reference sources in place (oracle/Makefile), writes small FASTA/FASTQ inputs, runs the
reference's own build_debruijn_graph() on them through oracle/ref_driver.cpp and stores

    tests/golden/<case>/case.json      parameters, input file names, the reference's summary
                                       (reads / kmers / count / size / max) and sha256 of the dump
    tests/golden/<case>/<inputs>       the input files (small cases only)
    tests/golden/<case>/dump.txt       canonical dump "kmer<TAB>l_link<TAB>r_link" sorted by kmer
                                       (kept for a few small cases only; elsewhere the sha256 of
                                       this exact text, recorded in case.json, is the fixture)
    tests/golden/kat.txt               known-answer values of seq2bit / rev-comp / hash_code / primes

Fixtures are DATA (inputs + the reference's outputs); no reference source text is stored.
Cases whose input comes from the counter-based generator (include/dbgk_synth.h) store only the
generator parameters and the sha256 of the dump ("synth_*" cases).

usage: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def sample_reads(rng, n, genome_len, read_len, sub=0.01, n_rate=0.002, var_len=False):
    g = "".join(rng.choice("ACGT") for _ in range(genome_len))
    out = []
    for _ in range(n):
        ln = read_len
        if var_len and rng.random() < 0.3:
            ln = rng.randint(1, read_len + 60)
        s = rng.randint(0, genome_len - ln)
        r = g[s:s + ln]
        if rng.random() < 0.5:
            r = revcomp(r)
        r = list(r)
        for j in range(len(r)):
            x = rng.random()
            if x < sub:
                r[j] = rng.choice("ACGT")
            elif x < sub + n_rate:
                r[j] = "N"
        out.append("".join(r).encode())
    return out, g


def write_fastq_tricky(path, seqs):
    """FASTQ whose quality lines start with '@' """
    with open(path, "wb") as fh:
        for i, s in enumerate(seqs):
            fh.write(b"@r%d\n" % i + s + b"\n+\n" + b"@" + b"I" * (len(s) - 1 if len(s) else 0) + b"\n")


def run_case(name, files, params, keep_inputs=True, keep_dump=False):
    """files: list of (filename, writer(path)) ; params: dict for ref_build"""
    cdir = os.path.join(HERE, name)
    shutil.rmtree(cdir, ignore_errors=True)
    os.makedirs(cdir)
    paths = []
    for fname, writer in files:
        p = os.path.join(cdir, fname)
        writer(p)
        paths.append(p)
    with tempfile.TemporaryDirectory() as tmp:
        libf = os.path.join(tmp, "reads.lib")
        with open(libf, "w") as fh:
            fh.write("\n".join(paths) + "\n")
        dump = os.path.join(cdir, "dump.txt")
        js = O.ref_build(libf, dump=dump, timeout=120, **params)
    body = open(dump, "rb").read()
    sha = hashlib.sha256(body).hexdigest()
    case = {"name": name, "params": params, "files": [f for f, _ in files],
            "ref": {k: js[k] for k in ("reads", "kmers", "count", "size", "max")},
            "dump_sha256": sha}
    if not keep_inputs:
        for p in paths:
            os.remove(p)
        case["files"] = []
    if not keep_dump:
        os.remove(dump)  # the sha256 of the dump text is the fixture; tests re-render and hash
    with open(os.path.join(cdir, "case.json"), "w") as fh:
        json.dump(case, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print("%-22s reads %6d kmers %8d count %7d size %8d sha %s" %
          (name, js["reads"], js["kmers"], js["count"], js["size"], sha[:12]))
    return case


def fa(seqs, gz=False):
    return lambda p: O.write_reads_file(p, seqs, fmt=2, gz=gz)


def fq(seqs, gz=False):
    return lambda p: O.write_reads_file(p, seqs, fmt=1, gz=gz)


def main():
    O.build()
    if not O.have_ref():
        sys.exit("oracle/_ref/ref_dbg missing: /root/reference not available here")
    rng = random.Random(20261003)
    base = dict(k=31, max_read_len=250, threads=1, init_hash_size=0.0001, load_factor=0.7,
                max_double=10, buffer_num=10000, fmt=2)

    # (1) mixed-strand 150-bp reads, 1 % substitutions, sparse N, two files
    reads, _ = sample_reads(rng, 300, 4000, 150)
    run_case("mixed150_k31", [("a.fa", fa(reads[:170])), ("b.fa", fa(reads[170:]))], dict(base), keep_dump=True)
    run_case("mixed150_k17", [("a.fa", fa(reads))], dict(base, k=17))
    run_case("mixed150_k31_t4", [("a.fa", fa(reads))], dict(base, threads=4))

    # (2) reads shorter than K, exactly K, longer than -r (r = 100)
    reads2, _ = sample_reads(rng, 200, 3000, 100, var_len=True)
    reads2 += [reads2[0][:31], reads2[1][:30], reads2[2][:1], b""]
    run_case("lengths_k31_r100", [("a.fa", fa(reads2))], dict(base, max_read_len=100), keep_dump=True)

    # (3) lowercase bases and n
    reads3 = [r.lower() if i % 2 else r for i, r in enumerate(reads[:120])]
    run_case("lowercase_k31", [("a.fa", fa(reads3))], dict(base))

    # (4) poly-A / poly-T reads: key-0 node with non-zero links
    reads4 = reads[:40] + [b"A" * 150, b"T" * 150, b"A" * 40 + b"C" + b"A" * 40, b"G" + b"T" * 60, b"a" * 33]
    run_case("polyA_k31", [("a.fa", fa(reads4))], dict(base), keep_dump=True)

    # (5) saturation: the same read (and its reverse complement) > 255 times
    one = reads[5]
    reads5 = [one] * 300 + [revcomp(one.decode()).encode()] * 40 + [b"A" * 150] * 300
    run_case("saturate_k31", [("a.fa", fa(reads5))], dict(base), keep_dump=True)

    # (6) even k (palindromes, tie -> forward), k = 32 (mask wraps to all-ones), small k
    pal = [b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", b"AATTAATTAATTAATTAATTAATTAATTAATTAATT", b"GAATTCGAATTCGAATTCGAATTC"]
    run_case("even_k16", [("a.fa", fa(reads[:150] + pal))], dict(base, k=16))
    run_case("even_k32", [("a.fa", fa(reads[:150] + pal))], dict(base, k=32))
    run_case("small_k4", [("a.fa", fa(reads[:60] + pal))], dict(base, k=4, init_hash_size=0.000001), keep_dump=True)
    run_case("odd_k21", [("a.fa", fa(reads[:150]))], dict(base, k=21))

    # (7) FASTQ / gz / quality line starting with '@'
    run_case("fastq_k31", [("a.fq", fq(reads[:100]))], dict(base, fmt=1))
    run_case("fastq_gz_k31", [("a.fq.gz", fq(reads[:100], gz=True)), ("b.fq", fq(reads[100:160]))], dict(base, fmt=1))
    run_case("fasta_gz_k31", [("a.fa.gz", fa(reads[:100], gz=True))], dict(base))
    run_case("fastq_at_quality_k31", [("a.fq", lambda p: write_fastq_tricky(p, reads[:80]))], dict(base, fmt=1))

    # (8) block boundaries (-b = n, n-1, n+1) and enlarges (-i 0.00001 -> 10007 slots)
    n = 120
    for b in (n, n - 1, n + 1, 7):
        run_case("block_b%d" % b, [("a.fa", fa(reads[:n]))], dict(base, buffer_num=b))
    run_case("enlarge_b50", [("a.fa", fa(reads[:300]))], dict(base, init_hash_size=0.00001, buffer_num=50))
    run_case("enlarge_b50_t3", [("a.fa", fa(reads[:300]))], dict(base, init_hash_size=0.00001, buffer_num=50, threads=3))
    # -e 1: second overflow hits the "Memory reach the maximum allowed" branch, rest of file dropped
    run_case("enlarge_cap_e1", [("a.fa", fa(reads[:300])), ("b.fa", fa(reads[:10]))],
             dict(base, init_hash_size=0.000005, buffer_num=20, max_double=1))

    # (9) larger synthetic cases: inputs regenerated from include/dbgk_synth.h, only sha256 kept
    for name, n_reads, glen, k in (("synth_20k_k31", 20000, 100000, 31), ("synth_20k_k17", 20000, 100000, 17)):
        P = O.synth_params(glen, 150, cfg=1)

        def wr(p, P=P, n_reads=n_reads):
            import ctypes as C
            O.lib().orc_synth_write_file(C.byref(P), 0, n_reads, os.fsencode(p), 2, 0)
        c = run_case(name, [("synth.fa", wr)], dict(base, k=k, init_hash_size=0.002), keep_inputs=False)
        c["synth"] = {"genome_len": glen, "read_len": 150, "n_reads": n_reads, "cfg": 1,
                      "sub_rate": 0.005, "n_rate": 0.0001}
        with open(os.path.join(HERE, name, "case.json"), "w") as fh:
            json.dump(c, fh, indent=1, sort_keys=True)
            fh.write("\n")

    # KATs
    kat = subprocess.run([O.REF_BIN, "kat"], check=True, capture_output=True, text=True).stdout
    kat += subprocess.run([O.REF_BIN, "prime"] + [str(x) for x in (
        3, 4, 1000, 10000, 20000, 100000, 200006, 200014, 1000000, 2000006, 10000000, 20000038,
        100000000, 1000000000, 2000000014, 24, 48, 120, 168, 288, 360)],
        check=True, capture_output=True, text=True).stdout
    with open(os.path.join(HERE, "kat.txt"), "w") as fh:
        fh.write(kat)
    print("kat.txt: %d lines" % kat.count("\n"))


if __name__ == "__main__":
    main()
