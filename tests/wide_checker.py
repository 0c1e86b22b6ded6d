"""TEST INFRASTRUCTURE ONLY -- an INDEPENDENT checker for the 128-bit key path (k-mers of up to 63 bases).

PARITY UNPINNED above k = 32 (the reference stops at 31: `uint64_t kmer`, DBG_contig/kmerSet.h:71; "max 31",
main.cpp:100).  oracle/wide_oracle.cpp and the HIP kernels both compile include/dbgk_wide.h, so a mistake in that
header (a wrong neighbour complement, a wrong shift) would pass every comparison between the two.  This module
restates the rules a THIRD time and shares nothing with that header: no bit tricks, no C, no ctypes -- reads are
Python strings, the reverse complement is a reversed string with the letters swapped, a key is a Python int built
digit by digit, canonical is min() of two ints, a counter is min(255, n).  What it restates:

  DBGgraph.cpp:51-53   reads shorter than K are skipped
  DBGgraph.cpp:63      readlen = min(size, maxReadLen); windows j = 0 .. readlen - K
  seqKmer.cpp:9-19     A a N n -> 0, C c -> 1, G g -> 2, T t -> 3
  DBGgraph.cpp:76-97   kbit <= rc_kbit -> forward with (left, right) = (read[j-1] if j > 0, read[j+K] if j < readlen-K);
                       else the reverse complement with right = comp(read[j-1]), left = comp(read[j+K]); a missing side
                       adds nothing
  DBGgraph.cpp:177-196 per node eight counters (left A C G T, right A C G T), +1 per sighting, stopping at 255
  DBGgraph.cpp:101     Kmer_total_num += size - K + 1 with the UNtrimmed size
  DBGgraph.cpp:153-164,:418  key 0 is kept aside and always present in the result

At k <= 32 its output must equal the real reference's dumps (tests/golden, tests/test_wide_checker.py); above that it
is the yardstick for BOTH oracle/wide_oracle.cpp and the GPU engine.
"""
import numpy as np

CODE = {"A": 0, "a": 0, "N": 0, "n": 0, "C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3}
LETTER = "ACGT"
COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}

NODE32_DTYPE = np.dtype([("kmer_hi", "<u8"), ("kmer_lo", "<u8"), ("l_link", "<u4"), ("r_link", "<u4"), ("reserved", "<u8")])


def _value(word):
    """the k letters of `word` as a base-4 number, first letter most significant (seq2bit, seqKmer.cpp:34-41)"""
    v = 0
    for ch in word:
        v = v * 4 + LETTER.index(ch)
    return v


def observations(reads, k, max_read_len=250):
    """every window of every read as (canonical key, index of the left neighbour's letter or None, of the right one or None);
    second result: Kmer_total_num"""
    out = []
    total = 0
    for read in reads:
        if isinstance(read, (bytes, bytearray, np.ndarray)):
            read = bytes(read).decode("latin-1")
        if len(read) < k:
            continue
        total += len(read) - k + 1
        # N counts as A, lower case as upper case; so does every byte outside ACGTNacgtn (the reference reads out of bounds on
        # those, seqKmer.cpp:9-19 + DBGgraph.cpp:71-73: this build's rule, counted by other_bytes below)
        norm = "".join(LETTER[CODE.get(ch, 0)] for ch in read)
        readlen = min(len(norm), max_read_len)
        for j in range(0, readlen - k + 1):
            word = norm[j:j + k]
            back = "".join(COMPLEMENT[ch] for ch in reversed(word))
            left = norm[j - 1] if j > 0 else None
            right = norm[j + k] if j < readlen - k else None
            fwd, rev = _value(word), _value(back)
            if fwd <= rev:
                key, lbase, rbase = fwd, left, right
            else:   # seen from the other strand: what followed the window now precedes it, complemented
                key = rev
                lbase = COMPLEMENT[right] if right is not None else None
                rbase = COMPLEMENT[left] if left is not None else None
            out.append((key, None if lbase is None else LETTER.index(lbase), None if rbase is None else LETTER.index(rbase)))
    return out, total


def other_bytes(reads):
    """how many bytes of the reads are none of ACGTNacgtn (dbgk_stats.other_bytes)"""
    n = 0
    for read in reads:
        if isinstance(read, (bytes, bytearray, np.ndarray)):
            read = bytes(read).decode("latin-1")
        n += sum(1 for ch in read if ch not in CODE)
    return n


def build(reads, k, max_read_len=250):
    """reads: iterable of bytes / str.  -> (dict key -> [[lA, lC, lG, lT], [rA, rC, rG, rT]] with saturated counters,
    Kmer_total_num); key 0 is always present"""
    nodes = {0: [[0, 0, 0, 0], [0, 0, 0, 0]]}
    obs, total = observations(reads, k, max_read_len)
    for key, lb, rb in obs:
        node = nodes.setdefault(key, [[0, 0, 0, 0], [0, 0, 0, 0]])
        if lb is not None:
            node[0][lb] = min(255, node[0][lb] + 1)
        if rb is not None:
            node[1][rb] = min(255, node[1][rb] + 1)
    return nodes, total


def link_word(counters):
    """four counters -> the reference's link word: A in bits 31..24, C 23..16, G 15..8, T 7..0 (kmerSet.cpp:56)"""
    a, c, g, t = counters
    return a * 16777216 + c * 65536 + g * 256 + t


def as_sorted_nodes(nodes):
    """the dict of build() as the array dbgk_wide_export_sorted returns: sorted by key, the key-0 node first"""
    out = np.zeros(len(nodes), dtype=NODE32_DTYPE)
    for i, key in enumerate(sorted(nodes)):
        out[i]["kmer_hi"] = key // 18446744073709551616
        out[i]["kmer_lo"] = key % 18446744073709551616
        out[i]["l_link"] = link_word(nodes[key][0])
        out[i]["r_link"] = link_word(nodes[key][1])
    return out


def split_reads(bases, offsets):
    raw = bytes(np.asarray(bases, dtype=np.uint8))
    return [raw[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]
