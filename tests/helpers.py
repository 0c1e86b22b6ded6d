"""Shared helpers for the parity tests (test infrastructure)."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def dump_text(nodes, total_reads, total_kmers, count):
    """Render the canonical dump exactly as oracle/ref_driver.cpp prints it."""
    head = "#reads %d kmers %d count %d\n" % (total_reads, total_kmers, count)
    body = "".join("%d\t%08x\t%08x\n" % (int(k), int(l), int(r))
                   for k, l, r in zip(nodes["kmer"].tolist(), nodes["l_link"].tolist(), nodes["r_link"].tolist()))
    return head + body


def dump_sha256(nodes, total_reads, total_kmers, count):
    return hashlib.sha256(dump_text(nodes, total_reads, total_kmers, count).encode()).hexdigest()


def case_files(case):
    return [os.path.join(GOLDEN, case["name"], f) for f in case["files"]]


def case_reads(case, oracle):
    """-> list of (bases, offsets), one per input file of the golden case (regenerates synth cases)"""
    if "synth" in case:
        s = case["synth"]
        P = oracle.synth_params(s["genome_len"], s["read_len"], s["sub_rate"], s["n_rate"], s["cfg"])
        return [oracle.synth_reads(P, 0, s["n_reads"])]
    return [oracle.read_sequences(p, case["params"]["fmt"]) for p in case_files(case)]
