"""GPU (-m gpu): the debruijn_contig command line / build_debruijn_graph() host layer end to end:
files (FASTA/FASTQ/.gz) -> GPU -> host KmerSet -> canonical dump, against the golden fixtures of
the real reference."""
import hashlib
import os
import re
import subprocess

import pytest

from conftest import golden_cases, golden_case_ids
from helpers import case_files, KMER_FREQ_HEADER, KMER_FREQ_ROWS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "dbg_assembly_amd", "bin", "debruijn_contig")

FILE_CASES = [c for c in golden_cases() if c["files"]]


def run_cli(tmp_path, case, extra_env=None):
    p = case["params"]
    libf = tmp_path / "reads.lib"
    libf.write_text("\n".join(case_files(case)) + "\n")
    dump = tmp_path / "dump.txt"
    prefix = tmp_path / "out"
    env = dict(os.environ, DBGK_DUMP=str(dump))
    env.update(extra_env or {})
    cmd = [CLI, "-k", str(p["k"]), "-r", str(p["max_read_len"]), "-f", str(p["fmt"]), "-t", "4",
           "-i", repr(p["init_hash_size"]), "-l", repr(p["load_factor"]), "-e", str(p["max_double"]),
           "-b", str(p["buffer_num"]), "-o", str(prefix), str(libf)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r, dump, prefix


# PARTITION (what the CLI runs) on every case; DIRECT on the cases that exercise the host layer's own logic (block schedule, enlarges,
# the -e cap, formats, trimming, saturation) -- the DIRECT engine itself meets every golden in tests/test_gpu_parity.py
DIRECT_CLI_CASES = {"block_b120", "enlarge_b50", "enlarge_b50_t3", "enlarge_cap_e1", "even_k32", "fasta_gz_k31", "fastq_at_quality_k31",
                    "lengths_k31_r100", "mixed150_k31", "polyA_k31", "saturate_k31", "small_k4"}
CLI_MATRIX = [(c, 2) for c in FILE_CASES] + [(c, 1) for c in FILE_CASES if c["name"] in DIRECT_CLI_CASES]


@pytest.mark.parametrize("case,engine", CLI_MATRIX, ids=["%s-%s" % (c["name"], "partition" if e == 2 else "direct") for c, e in CLI_MATRIX])
def test_cli_matches_reference(tmp_path, oracle, case, engine):
    """build_debruijn_graph() behind the reference's command line, both engines (PARTITION is what the
    CLI runs by default: records streamed through the record store, regions built in LDS).  Includes the
    reference's block schedule: enlarges decided after every full -b block, and the rest of a file dropped
    at the -e cap (enlarge_cap_e1, DBGgraph.cpp:337-351)."""
    assert os.path.exists(CLI), "debruijn_contig not built (python -c 'import __graft_entry__ as g; g.build()')"
    r, dump, prefix = run_cli(tmp_path, case, {"DBGK_ENGINE": str(engine)})
    ref = case["ref"]
    log = r.stderr
    if case["name"] == "enlarge_cap_e1":
        assert "Alert message: Memory reach the maximum allowed, program have loaded" in log
    assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"]
    assert re.search(r"^count:\t%d$" % ref["count"], log, re.M)
    assert re.search(r"^array_size:\t%d$" % ref["size"], log, re.M)       # same doubling chain as the reference
    assert re.search(r"^max_cutoff:\t%d$" % ref["max"], log, re.M)
    assert "Total number of reads loaded into memory: %d" % ref["reads"] in log
    assert "Total number of kmers loaded into memory: %d" % ref["kmers"] in log
    # first pass of the contig stage on the device == OUR restatement of contig.cpp:119-181 (parity unpinned: the
    # reference's contig.cpp cannot be compiled here; the file FORMAT is pinned by tests/test_dropin_link.py)
    _, nodes = oracle.parse_dump(str(dump))
    st = oracle.link_stats(nodes, 2)
    rows = open(str(prefix) + ".contig.kmer.freq").read().splitlines()
    assert rows[0] == KMER_FREQ_HEADER and len(rows) == 1 + KMER_FREQ_ROWS
    assert [int(x.split("\t")[1]) for x in rows[1:]] == list(st.depth_stat)[1:]
    assert re.search(r"Total kmer nodes number:\s+%d" % st.total_nodes, log)
    assert re.search(r"Used tip kmer nodes:\s+%d\t" % st.tip_nodes, log)


EARLY_CASES = [c for c in FILE_CASES if c["name"] in ("enlarge_b50", "enlarge_cap_e1", "mixed150_k31", "polyA_k31", "lengths_k31_r100")]


@pytest.mark.parametrize("case", EARLY_CASES, ids=[c["name"] for c in EARLY_CASES])
def test_cli_early_host_table_and_export_through_occupied_nodes(tmp_path, case):
    """The two host-table shortcuts of round 4 at sizes the goldens have: the array allocated (and touched) at the size -i asks for
    when the run starts -- kept when the reference's doubling schedule ends there, dropped when it enlarges (enlarge_* cases) -- and
    the table leaving the device as occupied nodes + occupancy bits.  Same dump, same final array size as the reference."""
    r, dump, _ = run_cli(tmp_path, case, {"DBGK_TEST_HOOKS": "early_table_min=0,export_compact_min=0,early_threads=3"})
    ref = case["ref"]
    assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"]
    assert re.search(r"^count:\t%d$" % ref["count"], r.stderr, re.M)
    assert re.search(r"^array_size:\t%d$" % ref["size"], r.stderr, re.M)


def test_cli_small_batches_and_device_resize(tmp_path):
    """DIRECT engine: 1 MiB host batches + a tiny initial table force several device-side enlarges"""
    case = [c for c in golden_cases() if c["name"] == "enlarge_b50"][0]
    r, dump, _ = run_cli(tmp_path, case, {"DBGK_BATCH_BYTES": "1048576", "DBGK_ENGINE": "1"})
    assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"]
    assert "Enlarge device hash array size" in r.stderr


@pytest.mark.parametrize("name", ["mixed150_k31", "enlarge_b50", "polyA_k31", "saturate_k31", "fastq_gz_k31"])
def test_cli_streams_through_a_small_record_store(tmp_path, name):
    """PARTITION engine with a record store far smaller than the input and 4 KiB host batches: the input
    goes through many flush rounds (regions loaded back into LDS, new records merged, written out again)."""
    case = [c for c in golden_cases() if c["name"] == name][0]
    r, dump, _ = run_cli(tmp_path, case, {"DBGK_BATCH_BYTES": "4096", "DBGK_ENGINE": "2", "DBGK_STORE_KMERS": "3000"})
    assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"]


@pytest.mark.parametrize("name", ["mixed150_k31", "fastq_gz_k31", "lowercase_k31", "enlarge_b50"])
def test_cli_ascii_hand_over_equals_the_packed_one(tmp_path, name):
    """the reader threads pack the reads to 2 bits per base by default (every other CLI test); DBGK_HOST_ASCII=1 hands the raw bytes
    over and the GPU packs them: same graph, with many small batches as well"""
    case = [c for c in golden_cases() if c["name"] == name][0]
    for extra in ({"DBGK_HOST_ASCII": "1"}, {"DBGK_HOST_ASCII": "1", "DBGK_BATCH_BYTES": "4096"}, {"DBGK_BATCH_BYTES": "4099"},
                  {"DBGK_BATCH_BYTES": "4099", "DBGK_TEST_HOOKS": "no_zero_copy=1"}, {"DBGK_TEST_HOOKS": "parse_sequential=1", "DBGK_BATCH_BYTES": "5000"}):
        r, dump, _ = run_cli(tmp_path, case, extra)
        assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"], extra
        assert "none of ACGTNacgtn" not in r.stderr


@pytest.mark.parametrize("engine", [2, 1], ids=["partition", "direct"])
def test_cli_reads_with_bytes_outside_the_alphabet(tmp_path, oracle, engine):
    """IUPAC codes, gaps, bytes >= 128 in a FASTA file: the reference reads out of bounds on them (seqKmer.cpp:9-19,
    DBGgraph.cpp:71-73); here they are read as 'A' -- the graph is the oracle's graph of the reads with those bytes replaced
    by 'A' -- and one warning line says how many there were; packed and ASCII hand-over, plain and gzip'ed file alike"""
    import gzip
    import random
    rng = random.Random(5)
    genome = "".join(rng.choice("ACGT") for _ in range(4000)).encode()
    reads, n_other = [], 0
    for _ in range(1200):
        s0 = rng.randint(0, len(genome) - 150)
        r = bytearray(genome[s0:s0 + 150])
        for j in range(150):
            if rng.random() < 0.02:
                r[j] = rng.choice(b"RYKMSWBDHVrykm-*.?\x80\xff")
                n_other += 1
        reads.append(bytes(r))
    clean = [bytes(c if c in b"ACGTNacgtn" else ord("A") for c in r) for r in reads]
    cb, co = oracle.pack_reads(clean)
    ref = oracle.build_graph(files_mem=[(cb, co)], k=31, max_read_len=250, init_hash_size=0.001, threads=1)
    plain, gz = tmp_path / "reads.fa", tmp_path / "reads.fa.gz"
    body = b"".join(b">r%d\n%s\n" % (i, r) for i, r in enumerate(reads))
    plain.write_bytes(body)
    with gzip.open(gz, "wb") as fh:
        fh.write(body)
    for path, extra in ((plain, {}), (plain, {"DBGK_HOST_ASCII": "1"}), (gz, {}), (gz, {"DBGK_HOST_ASCII": "1"})):
        libf = tmp_path / "reads.lib"
        libf.write_text(str(path) + "\n")
        dump = tmp_path / "dump.txt"
        env = dict(os.environ, DBGK_DUMP=str(dump), DBGK_ENGINE=str(engine), DBGK_BATCH_BYTES="30000")
        env.update(extra)
        r = subprocess.run([CLI, "-k", "31", "-f", "2", "-t", "4", "-i", "0.001", "-o", str(tmp_path / "out"), str(libf)], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "Alert message: %d sequence bytes are none of ACGTNacgtn; they were read as A (like N)" % n_other in r.stderr, (path, extra)
        _, nodes = oracle.parse_dump(str(dump))
        assert len(nodes) == ref.count and all((nodes[f] == ref.nodes[f]).all() for f in ("kmer", "l_link", "r_link")), (path, extra)


LAYOUT_CASES = [c for c in FILE_CASES if c["name"] in ("mixed150_k31", "enlarge_b50", "block_b7", "polyA_k31", "fastq_gz_k31",
                                                        "lengths_k31_r100", "even_k32", "saturate_k31")]


@pytest.mark.parametrize("case", LAYOUT_CASES, ids=[c["name"] for c in LAYOUT_CASES])
def test_cli_reference_layout_mode(tmp_path, oracle, case):
    """DBGK_LAYOUT=ref: the host KmerSet is slot-for-slot the table the reference builds at -t 1
    (same size after the same enlarges, every node in the same slot, same nul_flag) -- checked
    against the oracle's sequential path, which tests/test_oracle_vs_ref_live.py pins to the real
    reference's layout."""
    p = case["params"]
    img = tmp_path / "table.img"
    r, dump, _ = run_cli(tmp_path, case, {"DBGK_LAYOUT": "ref", "DBGK_DUMP_TABLE": str(img), "DBGK_BATCH_BYTES": "1048576"})
    assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"]
    size, count, array, flags = oracle.read_table_image(str(img))
    res = oracle.build_graph(files=case_files(case), k=p["k"], max_read_len=p["max_read_len"], threads=1,
                             init_hash_size=p["init_hash_size"], load_factor=p["load_factor"], max_double=p["max_double"],
                             buffer_num=p["buffer_num"], fmt=p["fmt"], want_table=True)
    assert (size, count) == (res.size, res.count)
    import numpy as np
    assert np.array_equal(array, res.table)
    assert np.array_equal(flags, res.nul_flag)


MULTI_CASES = [c for c in FILE_CASES if c["name"] in ("mixed150_k31", "enlarge_b50", "enlarge_cap_e1", "polyA_k31", "fastq_gz_k31",
                                                       "lengths_k31_r100", "saturate_k31", "block_b7")]


@pytest.mark.parametrize("gpus,store", [("0,0", None), ("0,0,0", "2500")])
@pytest.mark.parametrize("case", MULTI_CASES, ids=[c["name"] for c in MULTI_CASES])
def test_cli_multi_gpu_path(tmp_path, case, gpus, store):
    """build_debruijn_graph() over several GPU shards of one table (host/DBGgraph.cpp with DBGK_GPU_LIST /
    DBGK_GPUS -> dbgk_comm_*): same dump, same table size schedule (incl. the -e cap) as the reference.
    The shards share the one GPU of the test box; with a small store the job takes many exchange rounds."""
    env = {"DBGK_GPU_LIST": gpus}
    if store:
        env.update({"DBGK_STORE_KMERS": store, "DBGK_BATCH_BYTES": "4096"})
    r, dump, _ = run_cli(tmp_path, case, env)
    assert "GPU shards" in r.stderr
    ref = case["ref"]
    assert hashlib.sha256(dump.read_bytes()).hexdigest() == case["dump_sha256"]
    assert re.search(r"^array_size:\t%d$" % ref["size"], r.stderr, re.M)


@pytest.fixture(scope="module")
def cfg2_fasta(tmp_path_factory, oracle):
    """the FULL BASELINE cfg2 workload as a 1.6 GB one-line FASTA file, written ONCE for the tests that run the command line on it"""
    import ctypes as C
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg2_full.json")))
    fa = str(tmp_path_factory.mktemp("cfg2") / "reads.fa")
    PO = oracle.synth_params(gold["genome_len"], 150, cfg=2)
    oracle.lib().orc_synth_write_file(C.byref(PO), 0, gold["n_reads"], os.fsencode(fa), 2, 0)
    yield gold, fa
    os.remove(fa)


@pytest.mark.parametrize("store", [None, "300000000"], ids=["one_build", "four_flush_rounds"])
def test_cli_full_size_cfg2_equals_oracle_golden(tmp_path, cfg2_fasta, store):
    """The FULL BASELINE cfg2 workload (10 M x 150 bp reads as a 1.6 GB one-line FASTA file) through the command line:
    node count, totals and `<prefix>.contig.kmer.freq` (DepthStat rows 1..255) == tests/golden/cfg2_full.json, which the
    CPU oracle computed at full size.  Second run: a record store of 300 M occurrences, i.e. the input streams through
    four flush rounds of the PARTITION engine (incremental region builds over the whole 9.6 GB table); it also hands the
    reads over as ASCII (DBGK_HOST_ASCII=1), the first run 2 bits per base (the default)."""
    gold, fa = cfg2_fasta
    lib = tmp_path / "reads.lib"
    lib.write_text(fa + "\n")
    prefix = tmp_path / "out"
    env = dict(os.environ, DBGK_TIMINGS="1")
    if store:
        env["DBGK_STORE_KMERS"] = store
        env["DBGK_HOST_ASCII"] = "1"
    r = subprocess.run([CLI, "-k", "31", "-f", "2", "-i", "0.6", "-t", "16", "-o", str(prefix), str(lib)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    log = r.stderr
    assert re.search(r"^count:\t%d$" % gold["count"], log, re.M)
    assert "Total number of reads loaded into memory: %d" % gold["total_reads"] in log
    assert "Total number of kmers loaded into memory: %d" % gold["total_kmers"] in log
    assert "engine partition" in log
    rows = open(str(prefix) + ".contig.kmer.freq").read().splitlines()
    assert rows[0] == KMER_FREQ_HEADER and [int(x.split("\t")[1]) for x in rows[1:]] == gold["depth_stat"][1:]


@pytest.mark.parametrize("k,env", [(63, {}), (33, {"DBGK_WIDE_PASSES": "3"}), (47, {"DBGK_GPU_LIST": "0,0,0"}), (63, {"DBGK_TEST_HOOKS": "wide_direct=1"})],
                         ids=["k63", "k33_three_passes", "k47_three_shards", "k63_atomic_kernels"])
def test_cli_k_above_32_emits_the_128_bit_graph_PARITY_UNPINNED(tmp_path, k, env):
    """`debruijn_contig -k 33..63` (this build only: the reference stops at 31): files -> WIDE engine -> host KmerSet128 ->
    sorted node dump + <prefix>.contig.kmer.freq, against the independent checker (tests/wide_checker.py); through 16-byte
    records in one pass, in three passes over the input files, over three in-process GPU shards, and through the atomic kernels"""
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import wide_checker as W
    from test_wide_checker import _reads
    rng = random.Random(1000 + k)
    reads = [r for r in _reads(rng, 300) if r]   # (an empty line would not be a read of a one-line FASTA file)
    fa = tmp_path / "reads.fa"
    with open(fa, "wb") as fh:
        for i, r in enumerate(reads):
            fh.write(b">r%d\n" % i + r + b"\n")
    libf = tmp_path / "reads.lib"
    libf.write_text(str(fa) + "\n")
    dump, prefix = tmp_path / "dump.txt", tmp_path / "out"
    run_env = dict(os.environ, DBGK_DUMP=str(dump), DBGK_BATCH_BYTES="20000")
    run_env.update(env)
    cmd = [CLI, "-k", str(k), "-r", "120", "-f", "2", "-t", "4", "-i", "0.0672", "-o", str(prefix), str(libf)]
    r = subprocess.run(cmd, env=run_env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    nodes, total = W.build(reads, k, 120)
    want = W.as_sorted_nodes(nodes)
    lines = dump.read_text().splitlines()
    assert lines[0] == "#reads %d kmers %d count %d" % (len(reads), total, len(want))
    got = [tuple(int(x, 16) if i >= 2 else int(x) for i, x in enumerate(l.split("\t"))) for l in lines[1:]]
    assert got == [(int(n["kmer_hi"]), int(n["kmer_lo"]), int(n["l_link"]), int(n["r_link"])) for n in want]
    assert "Total number of kmers loaded into memory: %d" % total in r.stderr
    assert re.search(r"^count: %d$" % len(want), r.stderr, re.M)
    if "DBGK_WIDE_PASSES" in env:
        assert "Pass 3 of 3 over the input" in r.stderr
    # <prefix>.contig.kmer.freq: DepthStat[1..255] of all 8 counters of every node
    depth = [0] * 256
    for n in want:
        for w in (int(n["l_link"]), int(n["r_link"])):
            for s in (24, 16, 8, 0):
                depth[(w >> s) & 0xFF] += 1
    rows = (tmp_path / "out.contig.kmer.freq").read_text().splitlines()
    assert rows[0] == KMER_FREQ_HEADER and len(rows) == 1 + KMER_FREQ_ROWS
    assert [int(x.split("\t")[1]) for x in rows[1:]] == depth[1:]


@pytest.mark.parametrize("name,cutoff,gpus", [("mixed150_k31", 2, ""), ("saturate_k31", 5, ""), ("fastq_gz_k31", 0, ""), ("mixed150_k31", 2, "0,0,0")])
def test_cli_first_pass_of_the_contig_stage_on_the_gpu_PARITY_UNPINNED(tmp_path, oracle, name, cutoff, gpus):
    """DBGK_LINKS=1: build_debruijn_graph() hands over, with the table, what calculate_kmer_links (contig.cpp:107-181) would
    compute from it -- KmerLink records, del_flag, tip / branch lists in slot order (dbgk_export_host_table_links) -- checked
    against the restatement of those lines applied to the table image the CLI wrote (parity unpinned: contig.cpp needs Boost)"""
    import numpy as np
    case = [c for c in golden_cases() if c["name"] == name][0]
    p = case["params"]
    libf = tmp_path / "reads.lib"
    libf.write_text("\n".join(case_files(case)) + "\n")
    img, lk, prefix = tmp_path / "table.img", tmp_path / "links.txt", tmp_path / "out"
    env = dict(os.environ, DBGK_LINKS="1", DBGK_DUMP_TABLE=str(img), DBGK_DUMP_LINKS=str(lk))
    if gpus:   # three GPU shards of one table: dbgk_comm_export_host_table_links
        env["DBGK_GPU_LIST"] = gpus
    cmd = [CLI, "-k", str(p["k"]), "-r", str(p["max_read_len"]), "-f", str(p["fmt"]), "-t", "4", "-i", repr(p["init_hash_size"]),
           "-l", repr(p["load_factor"]), "-e", str(p["max_double"]), "-b", str(p["buffer_num"]), "-D", str(cutoff), "-o", str(prefix), str(libf)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "First pass of the contig stage done on the GPU" in r.stderr
    size, count, array, flags = oracle.read_table_image(str(img))
    assert count == case["ref"]["count"]
    rec, dele, tips, branches = oracle.kmer_links(array, flags, cutoff)
    lines = lk.read_text().splitlines()
    assert lines[0] == "#size %d tips %d branches %d" % (size, len(tips), len(branches))
    occ = np.flatnonzero(np.unpackbits(flags)[:size])
    dbits = np.unpackbits(dele)[:size]
    want = ["K\t%d\t%04x\t%d" % (i, rec[i], dbits[i]) for i in occ] + ["T\t%d" % t for t in tips] + ["B\t%d" % b for b in branches]
    assert lines[1:] == want
