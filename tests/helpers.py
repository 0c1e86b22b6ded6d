"""Shared helpers for the parity tests (test infrastructure)."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# `<prefix>.contig.kmer.freq` (DBG_contig/contig.cpp:199-202): header line, then DepthStat[1..255]
KMER_FREQ_HEADER = "Kmer_depth\tAppear_times"
KMER_FREQ_ROWS = 255


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


def push_with_reference_schedule(g, files, params, capi):
    """Drive a Graph the way the reference's block loop would (DBG_contig/DBGgraph.cpp:226-356): blocks of
    buffer_num reads; after every FULL block `count > max` leads to one enlarge while doublings are left
    (kmerSet.cpp:132-145), otherwise the rest of the file is dropped.  Returns the table size the reference
    ends with.  (Host logic restated for the library-level tests; the C++ host layer has its own.)"""
    lf = np.float32(params["load_factor"])
    lf = np.float32(0.25) if lf <= 0 else (np.float32(0.75) if lf >= 1 else lf)
    want = int(params["init_hash_size"] * 1000000000)
    size = 3 if want < 3 else capi.find_next_prime_ref(want)
    cutoff = lambda sz: int(np.float32(sz) * lf)
    doublings, B = 0, params["buffer_num"]
    for bases, offsets in files:
        n = len(offsets) - 1
        for r0 in range(0, n, B):
            r1 = min(r0 + B, n)
            lo, hi = int(offsets[r0]), int(offsets[r1])
            g.push_reads(bases[lo:hi], offsets[r0:r1 + 1] - offsets[r0])
            if r1 - r0 < B:
                break
            g.flush()
            count = int(g.refresh_stats().count) - 1
            if count > cutoff(size):
                if doublings >= params["max_double"]:
                    break
                while True:
                    size = capi.find_next_prime_ref(size * 2)
                    if np.float32(size) * lf >= np.float32(count + 1):
                        break
                doublings += 1
    return size


def hooks_value(current, **kv):
    """DBGK_TEST_HOOKS syntax (include/dbgk_env.h): "name=value,name=value"; merge kv into `current` (None removes a hook)"""
    d = dict(item.split("=", 1) for item in (current or "").split(",") if item)
    for k, v in kv.items():
        if v is None:
            d.pop(k, None)
        else:
            d[k] = str(v)
    return ",".join("%s=%s" % kv for kv in d.items())


def set_hooks(monkeypatch=None, **kv):
    """force code paths of the library through DBGK_TEST_HOOKS (read at every use): set_hooks(monkeypatch, l1_linear=1, export_full=None)"""
    import os
    val = hooks_value(os.environ.get("DBGK_TEST_HOOKS"), **kv)
    if monkeypatch is not None:
        if val:
            monkeypatch.setenv("DBGK_TEST_HOOKS", val)
        else:
            monkeypatch.delenv("DBGK_TEST_HOOKS", raising=False)
    elif val:
        os.environ["DBGK_TEST_HOOKS"] = val
    else:
        os.environ.pop("DBGK_TEST_HOOKS", None)
