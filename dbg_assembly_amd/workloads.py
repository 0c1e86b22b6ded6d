"""Synthetic workload shapes shared by bench.py and the tests (plumbing: numpy only, no compute of the hot path).

cfg2t -- "trimmed" cfg2: the input debruijn_contig really gets is quality-trimmed, error-corrected reads of MIXED lengths (the
reference's own recorded run feeds 742 648 reads with 158 315 081 k-mers at -r 250: 213 windows per read, mean length 243 of 250,
test/02.build_contig/Ecoli_corrected_reads.contig.log:437-438).  cfg2t keeps cfg2's 150-base reads (same generator, same genome,
same seeds) and trims 30 % of them to a length drawn uniformly from [60, 149]: read i keeps its FIRST len_i bases.  The lengths are
a pure function of the read index (splitmix64), so any slice of the workload can be regenerated anywhere.
"""
import numpy as np

TRIM_FRACTION = 0.30
TRIM_MIN = 60
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def trimmed_lengths(first_read, n_reads, read_len=150, seed=0x7A11ED):
    """lengths of reads [first_read, first_read + n_reads) of the trimmed workload"""
    with np.errstate(over="ignore"):
        idx = np.arange(first_read, first_read + n_reads, dtype=np.uint64)
        h = _splitmix64(idx ^ np.uint64(seed))
        trimmed = (h & np.uint64(0xFFFF)) < np.uint64(int(TRIM_FRACTION * 65536))
        span = np.uint64(read_len - TRIM_MIN)
        short = np.uint64(TRIM_MIN) + ((h >> np.uint64(16)) % span)
    return np.where(trimmed, short, np.uint64(read_len)).astype(np.uint64)


def trim_reads(bases, read_len, lengths):
    """bases of n fixed-length reads (back to back) -> (bases of the trimmed reads, offsets)"""
    n = len(lengths)
    assert len(bases) == n * read_len
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lengths, out=offsets[1:])
    out = np.empty(int(offsets[-1]), dtype=np.uint8)
    pos = np.arange(read_len, dtype=np.uint64)[None, :]
    for a in range(0, n, 1 << 20):   # a million reads at a time (bounded temporaries)
        b = min(n, a + (1 << 20))
        keep = (pos < lengths[a:b, None]).reshape(-1)
        out[int(offsets[a]):int(offsets[b])] = bases[a * read_len:b * read_len][keep]
    return out, offsets
