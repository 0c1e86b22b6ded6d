"""CPU: the host layer's two file readers (dbg_assembly_amd/host/reads_io.h) -- the sequential one (plain or gzip'ed) and the
windowed multi-threaded one for plain files -- deliver the same records, the ones the reference's rules define
(DBG_contig/DBGgraph.cpp:244-272: a line starting with the marker announces a record, the NEXT line is its sequence, FASTQ then
skips two lines whatever they start with, other lines are ignored, a header on the last line yields an empty read)."""
import gzip
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=[0, 97, 1000], ids=["window_256MiB", "window_97B", "window_1000B"])
def exe(tmp_path_factory, request):
    """the windowed reader compiled with its production window and with windows of 97 / 1000 bytes (lines straddle windows, records
    straddle windows, windows without a newline)"""
    out = str(tmp_path_factory.mktemp("readsio") / "reads_io_test")
    extra = ["-DDBGK_READS_WINDOW=%d" % request.param] if request.param else []
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "dbg_assembly_amd", "host"), os.path.join(ROOT, "tests", "reads_io_test.cpp"),
                    "-o", out, "-lz", "-lpthread"] + extra, check=True)
    return out


def expected_records(text, fmt):
    """the reference's block loop, restated line by line"""
    marker = b"@" if fmt == 1 else b">"
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()          # the text ended with a newline: no further line
    out, state = [], 0
    for line in lines:
        if state == 0:
            if line[:1] == marker:
                state = 1
        elif state == 1:
            out.append(line)
            state = 2 if fmt == 1 else 0
        elif state == 2:
            state = 3
        else:
            state = 0
    if state == 1:
        out.append(b"")
    return out


def run(exe, path, fmt, threads):
    r = subprocess.run([exe, path, str(fmt), str(threads)], capture_output=True, check=True)
    recs = []
    for line in r.stdout.split(b"\n")[:-1]:
        n, seq = line.split(b"\t", 1)
        assert int(n) == len(seq)
        recs.append(seq)
    return recs


CASES = {
    "fasta": (2, b">r1\nACGT\n>r2\nTTGCA\n"),
    "fasta_no_final_newline": (2, b">r1\nACGT\n>r2\nTTGCA"),
    "fasta_header_last": (2, b">r1\nACGT\n>r2"),
    "fasta_header_last_newline": (2, b">r1\nACGT\n>r2\n"),
    "fasta_junk_and_empty_lines": (2, b"junk\n\n>r1\nACGT\nmore junk\n\n>r2\n\n>r3\nGG\n"),
    "fasta_sequence_starting_with_marker": (2, b">r1\n>ACGT\n>r2\nAC\n"),
    "fastq": (1, b"@r1\nACGT\n+\nIIII\n@r2\nGGC\n+\nIII\n"),
    "fastq_quality_starts_with_at": (1, b"@r1\nACGT\n+\n@III\n@r2\nGGC\n+\n@@@\n@r3\nT\n+\nI"),
    "fastq_truncated": (1, b"@r1\nACGT\n+\nIIII\n@r2\nGGC\n+"),
    "empty": (2, b""),
    "only_newlines": (2, b"\n\n\n"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_both_readers_follow_the_references_record_rules(exe, tmp_path, name):
    fmt, text = CASES[name]
    path = tmp_path / "in.txt"
    path.write_bytes(text)
    want = expected_records(text, fmt)
    assert run(exe, str(path), fmt, 0) == want
    for threads in (1, 2, 3, 7):
        assert run(exe, str(path), fmt, threads) == want
    gz = tmp_path / "in.txt.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(text)
    assert run(exe, str(gz), fmt, 0) == want
    assert subprocess.run([exe, str(gz), str(fmt), "2"]).returncode == 3   # the windowed reader declines compressed files


@pytest.mark.parametrize("fmt", [1, 2])
def test_random_files_many_threads(exe, tmp_path, fmt):
    rng = random.Random(fmt)
    parts = []
    for i in range(6000):
        seq = bytes(rng.choice(b"ACGTN") for _ in range(rng.choice([0, 1, 30, 100, 150, 151])))
        if fmt == 1:
            qual = bytes(rng.choice(b"@I+>#") for _ in range(len(seq)))
            parts.append(b"@r%d\n" % i + seq + b"\n+\n" + qual + b"\n")
        else:
            parts.append(b">r%d\n" % i + seq + b"\n")
        if rng.random() < 0.01:
            parts.append(rng.choice([b"\n", b"stray line\n", b"+\n"]))
    text = b"".join(parts)
    path = tmp_path / "big.txt"
    path.write_bytes(text)
    want = expected_records(text, fmt)
    assert run(exe, str(path), fmt, 0) == want
    for threads in (1, 4, 13):
        assert run(exe, str(path), fmt, threads) == want
