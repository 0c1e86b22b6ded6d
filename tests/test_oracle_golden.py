"""CPU: the oracle (oracle/dbg_oracle.c) against golden vectors made by the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases, golden_case_ids
from helpers import case_files, case_reads, dump_sha256, dump_text


def test_kat_table(oracle):
    L = oracle.lib()
    n = 0
    for line in open(os.path.join(GOLDEN, "kat.txt")):
        t = line.rstrip("\n").split("\t")
        if t[0] == "seq2bit":
            seq = t[1].encode()
            assert L.orc_seq2bit(seq, len(seq)) == int(t[2])
            assert L.orc_rev_com_kbit(int(t[2]), len(seq)) == int(t[4])
            buf = bytes(len(seq) + 1)
            L.orc_bit2seq(int(t[2]), len(seq), buf)
            assert buf[:len(seq)].decode() == t[6]
        elif t[0] == "hash_code":
            assert L.orc_hash_code(int(t[1])) == int(t[2])
        elif t[0] == "get_next_kmer_depth":
            assert L.orc_get_next_kmer_depth(int(t[1], 16), int(t[2])) == int(t[3])
        elif t[0] == "pow_integer":
            assert L.orc_pow_integer(int(t[1]), int(t[2])) == int(t[3])
        elif t[0] == "is_prime":
            assert L.orc_is_prime(int(t[1])) == int(t[2])
        elif t[0] == "find_next_prime":
            assert L.orc_find_next_prime(int(t[1])) == int(t[2])
        else:
            raise AssertionError(line)
        n += 1
    assert n >= 50


@pytest.mark.parametrize("case", golden_cases(), ids=golden_case_ids())
def test_graph_matches_reference(oracle, case):
    p = case["params"]
    kw = dict(k=p["k"], max_read_len=p["max_read_len"], init_hash_size=p["init_hash_size"],
              load_factor=p["load_factor"], max_double=p["max_double"], buffer_num=p["buffer_num"], fmt=p["fmt"])
    for threads in sorted({1, p["threads"]}):
        if "synth" in case:
            res = oracle.build_graph(files_mem=case_reads(case, oracle), threads=threads, **kw)
        else:
            res = oracle.build_graph(files=case_files(case), threads=threads, **kw)
        ref = case["ref"]
        assert (res.total_reads, res.total_kmers, res.count) == (ref["reads"], ref["kmers"], ref["count"])
        assert (res.size, res.max) == (ref["size"], ref["max"])
        assert dump_sha256(res.nodes, res.total_reads, res.total_kmers, res.count) == case["dump_sha256"]
        dpath = os.path.join(GOLDEN, case["name"], "dump.txt")
        if os.path.exists(dpath):
            assert dump_text(res.nodes, res.total_reads, res.total_kmers, res.count) == open(dpath).read()


def test_file_reader_equals_mem_path(oracle):
    """orc_read_sequences + add_file_mem must equal add_file (same block structure)"""
    case = [c for c in golden_cases() if c["name"] == "fastq_gz_k31"][0]
    p = case["params"]
    mem = [oracle.read_sequences(f, p["fmt"]) for f in case_files(case)]
    a = oracle.build_graph(files_mem=mem, k=p["k"], init_hash_size=p["init_hash_size"], buffer_num=p["buffer_num"])
    assert dump_sha256(a.nodes, a.total_reads, a.total_kmers, a.count) == case["dump_sha256"]


def test_host_table_checker(oracle):
    case = [c for c in golden_cases() if c["name"] == "enlarge_b50"][0]
    p = case["params"]
    res = oracle.build_graph(files=case_files(case), k=p["k"], init_hash_size=p["init_hash_size"],
                             buffer_num=p["buffer_num"], want_table=True)
    assert res.double_times >= 1
    assert oracle.check_host_table(res.table, res.nul_flag, res.size, res.count) == 0
    bad = res.table.copy()
    occ = np.flatnonzero(bad["kmer"] != 0)
    bad["kmer"][occ[0]] ^= np.uint64(1 << 40)  # moves the key off its probe chain (or duplicates)
    assert oracle.check_host_table(bad, res.nul_flag, res.size, res.count) != 0


def test_parse_read_conventions(oracle):
    """appendix of SURVEY.md: N is A, tie -> forward, neighbour beyond -r is never used"""
    km, lb, rb = oracle.parse_read(b"ACGTN", 5, 250)
    assert km.tolist() == [oracle.lib().orc_seq2bit(b"ACGTA", 5)] and lb.tolist() == [4] and rb.tolist() == [4]
    km, lb, rb = oracle.parse_read(b"AATT", 4)  # palindrome: fwd == rc -> forward branch
    assert km.tolist() == [15] and (lb.tolist(), rb.tolist()) == ([4], [4])
    km, lb, rb = oracle.parse_read(b"CAATTG", 4)
    assert km[1] == 15 and lb[1] == 1 and rb[1] == 2  # forward: left = C, right = G
    km, lb, rb = oracle.parse_read(b"ACGTACGTAC", 4, 6)  # trimmed at -r 6
    assert len(km) == 3 and rb[2] == 4 or lb[2] == 4
    assert oracle.parse_read(b"ACG", 4)[0].size == 0


def test_link_stats_histogram(oracle):
    case = [c for c in golden_cases() if c["name"] == "saturate_k31"][0]
    res = oracle.build_graph(files=case_files(case), k=31, init_hash_size=0.0001)
    st = oracle.link_stats(res.nodes, cutoff=2)
    assert st.total_nodes == res.count
    assert sum(st.depth_stat) == 8 * res.count
    assert st.depth_stat[255] > 0  # saturated counters present


def test_kmer_links_restatement_known_answers(oracle):
    """calculate_kmer_links (contig.cpp:107-181) restated in oracle_py.kmer_links, worked by hand from the reference's text:
    link number = counters > cutoff (at most 3), base = the first base with the largest such counter, linear = one link on each
    side, deleted = none at all; tips have exactly one link in total, branches a side with more than one"""
    nodes = np.zeros(8, dtype=oracle.NODE_DTYPE)
    flags = np.zeros(2, dtype=np.uint8)
    #            slot 0: A=5 C=3 G=9 T=0 | T=3      slot 1: empty      slot 2: linear (C=7 | G=4)   slot 3: nothing above 2
    nodes[0] = (11, 0x05030900, 0x00000003)
    nodes[2] = (12, 0x00070000, 0x00000400)
    nodes[3] = (13, 0x02020202, 0x01000002)
    nodes[4] = (14, 0x09090000, 0x00000000)   # two equal maxima: the FIRST base wins (strict <); a tip? no: l_num = 2 -> branch
    nodes[5] = (15, 0x00000000, 0xFF000000)   # one link in total: tip
    nodes[6] = (16, 0x03030303, 0x03030303)   # four above the cutoff: the 2-bit field stops at 3
    for i in (0, 2, 3, 4, 5, 6):
        flags[i >> 3] |= 128 >> (i & 7)
    rec, dele, tips, branches = oracle.kmer_links(nodes, flags, 2)
    f = lambda ln, lb, rn, rb, lin=0: ln | lb << 2 | rn << 4 | rb << 6 | lin << 8
    assert list(rec) == [f(3, 2, 1, 3), 0, f(1, 1, 1, 2, 1), f(0, 0, 0, 0), f(2, 0, 0, 0), f(0, 0, 1, 0), f(3, 0, 3, 0), 0]
    assert list(dele) == [128 >> 3, 0]
    assert list(tips) == [5] and list(branches) == [0, 4, 6]
