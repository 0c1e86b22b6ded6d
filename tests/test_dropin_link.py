"""CPU (build container only; skipped where /root/reference is absent): guards of the drop-in boundary.

1. The reference's UNCHANGED consumer -- DBG_contig/main.cpp, contig.cpp, global_aligning.cpp -- compiles
   against dbg_assembly_amd/host/{seqKmer,kmerSet,DBGgraph}.h and links with libdbgasm_host.so + libdbgk.so
   (same globals, struct layouts and function names).  The sources
   are reached through symlinks made in a temporary directory at run time, so that their `#include "kmerSet.h"`
   resolves to THIS build's headers; nothing of the reference is copied or stored.  contig.h needs
   boost/lexical_cast.hpp, which this image lacks: the test writes a throw-away stand-in into the temporary
   directory.  This is a LINK / ABI guard, not a parity pin -- a build that needs a stand-in header pins
   nothing, and the consumer is only asked for its usage text.
2. The format of `<prefix>.contig.kmer.freq` that tests/test_gpu_cli.py asserts for the GPU build is the
   format of the file the reference's own test directory holds (contig.cpp:199-202: header, rows 1..255, row 0
   omitted).  Format only: the producer of that file's numbers (contig.cpp:119-181) cannot be compiled here,
   so the DepthStat restatement in oracle/ stays "parity unpinned".
"""
import os
import subprocess

import pytest

from helpers import KMER_FREQ_HEADER, KMER_FREQ_ROWS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DBG_contig"
HOST = os.path.join(ROOT, "dbg_assembly_amd", "host")
LIBDIR = os.path.join(ROOT, "dbg_assembly_amd", "lib")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")

BOOST_STANDIN = r"""// tests-only stand-in (the image has no Boost): boost::lexical_cast<std::string>(integer | double),
// double printed with 17 significant digits like Boost does (contig.cpp:1006 "avgDepth: 22.271739130434781")
#pragma once
#include <cstdio>
#include <string>
#include <type_traits>
namespace boost {
template <class To, class From> To lexical_cast(const From &v)
{
	static_assert(std::is_same<To, std::string>::value, "stand-in: only lexical_cast<std::string>");
	char buf[64];
	if (std::is_floating_point<From>::value) snprintf(buf, sizeof buf, "%.17g", (double)v);
	else if (std::is_signed<From>::value) snprintf(buf, sizeof buf, "%lld", (long long)v);
	else snprintf(buf, sizeof buf, "%llu", (unsigned long long)v);
	return std::string(buf);
}
}
"""


def test_reference_consumer_compiles_and_links_against_the_host_layer(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libdbgasm_host.so")):
        subprocess.run(["make", "-s", "-C", HOST], check=True)
    src = tmp_path / "src"
    src.mkdir()
    for name in ("main.cpp", "contig.cpp", "contig.h", "global_aligning.cpp", "global_aligning.h"):
        os.symlink(os.path.join(REF, name), src / name)   # links, not copies: quote-includes resolve next to the link
    inc = tmp_path / "standin" / "boost"
    (inc / "algorithm").mkdir(parents=True)
    (inc / "lexical_cast.hpp").write_text(BOOST_STANDIN)
    (inc / "algorithm" / "string.hpp").write_text("#pragma once\n")
    objs = []
    for name in ("main.cpp", "contig.cpp", "global_aligning.cpp"):
        obj = str(tmp_path / (name + ".o"))
        subprocess.run(["g++", "-O1", "-w", "-std=c++17", "-I" + HOST, "-I" + os.path.join(ROOT, "include"),
                        "-I" + str(tmp_path / "standin"), "-c", str(src / name), "-o", obj], check=True)
        objs.append(obj)
    exe = str(tmp_path / "debruijn_contig_ref_consumer")
    subprocess.run(["g++", "-o", exe] + objs + ["-L" + LIBDIR, "-ldbgasm_host", "-ldbgk", "-lz", "-lpthread",
                                                "-Wl,-rpath," + LIBDIR], check=True)
    # every symbol the consumer takes from the graph stage is resolved by the host layer
    nm = subprocess.run(["nm", "-u", "-C", exe], check=True, capture_output=True, text=True).stdout
    for sym in ("build_debruijn_graph", "exist_kmerset", "get_next_kmer_depth", "get_rev_com_kbit", "memset_parallel", "reading_file_list"):
        assert sym in nm, sym   # undefined in the consumer = taken from libdbgasm_host.so
    nm_all = subprocess.run(["nm", "-C", exe], check=True, capture_output=True, text=True).stdout
    for sym in ("kset", "KmerHeadMaskVal", "KmerSize"):   # data: copy-relocated into the executable
        assert any(line.split()[-1] == sym for line in nm_all.splitlines() if line.split()), sym
    r = subprocess.run([exe, "-h"], capture_output=True, text=True, timeout=60)   # the reference's own usage text, no GPU touched
    assert r.returncode == 0 and "debruijn_contig" in r.stdout and "-k <int>" in r.stdout


def test_kmer_freq_file_format_matches_the_reference_fixture():
    path = "/root/reference/test/02.build_contig/Ecoli_corrected_reads.contig.kmer.freq"
    rows = open(path).read().splitlines()
    assert rows[0] == KMER_FREQ_HEADER and len(rows) == 1 + KMER_FREQ_ROWS
    for i, row in enumerate(rows[1:], start=1):   # rows 1..255, DepthStat[0] is not written (contig.cpp:200)
        depth, times = row.split("\t")
        assert int(depth) == i and int(times) >= 0
