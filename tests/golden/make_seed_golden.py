"""Generate the seed-index golden vectors with the REAL reference code
(oracle/_ref/ref_seed = link_scaffold/{kmerSet,map_func,seqKmer,gzstream}.cpp + oracle/ref_seed_driver.cpp).
Needs /root/reference at BUILD time only; the fixtures it writes (inputs + dumps) are data.

    python tests/golden/make_seed_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle_py as orc  # noqa: E402


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(list(alphabet), size=n))


def cases():
    rng = np.random.default_rng(20260301)
    out = {}
    g = rand_seq(rng, 3000)
    # scaffold-like contigs: N gaps, a repeat shared between contigs, lower-case, a reverse-complement copy
    comp = str.maketrans("ACGT", "TGCA")
    c0 = g[:800] + "N" * 25 + g[900:1500]
    c1 = g[1400:2200].lower() + "NNN" + g[100:300]                  # overlaps c0, repeats part of it
    c2 = g[2200:3000][::-1].translate(comp) + "N" + g[2500:2600]    # reverse strand + forward copy
    out["seed_k31_scaffolds"] = (31, [c0, c1, c2])
    out["seed_k17_polyA_pal"] = (17, ["A" * 40, "ACGT" * 12, "T" * 30 + "N" + "ACGTTGCA" * 5, rand_seq(rng, 200, "ACGTn")])
    out["seed_k16_even_pal"] = (16, ["ACGTACGTACGTACGTACGT", "AATTAATTAATTAATTAATT" * 2, rand_seq(rng, 300), "GC" * 20])
    out["seed_k32_max"] = (32, [rand_seq(rng, 500), "A" * 70, rand_seq(rng, 200) + "NN" + rand_seq(rng, 31) + "N" + rand_seq(rng, 32) + "N" + rand_seq(rng, 90)])
    out["seed_k11_dense"] = (11, [rand_seq(rng, 4000, "AC"), rand_seq(rng, 4000, "ACGT"), "N" * 10 + rand_seq(rng, 50) + "N" * 3 + rand_seq(rng, 10) + "N"])
    return out


def main():
    assert orc.have_ref_seed(), "build oracle/_ref/ref_seed first (make -C oracle ref)"
    for name, (k, contigs) in cases().items():
        fa = os.path.join(HERE, name + ".fa")
        orc.write_contig_fasta(fa, contigs, width=70 if "scaffolds" in name else 0)
        meta, nodes = orc.ref_seed(fa, k, os.path.join(HERE, name + ".dump"))
        print(name, "k", k, meta, "unique", int(nodes["freq"].sum()))


if __name__ == "__main__":
    main()
