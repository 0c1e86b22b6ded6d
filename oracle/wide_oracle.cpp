// TEST INFRASTRUCTURE ONLY -- CPU restatement of the WIDE key path (k-mers of up to 63 bases, 128-bit keys).
//
// PARITY UNPINNED for k > 32: the reference stops at k = 31 (`uint64_t kmer`, DBG_contig/kmerSet.h:71; "max 31",
// main.cpp:100) and holds no fixture for longer k-mers.  The rules are the reference's, carried to 128 bits, and
// are written down ONCE in include/dbgk_wide.h (canonical pick, neighbour bases, 128-bit hash, digest); this file
// and the HIP kernels both use that header, with different extraction code around it: here the reference's own
// loop shape (first window by seq2bit, then the rolling update of DBGgraph.cpp:64-74, in unsigned __int128),
// sequential, one read after the other; there a 192-bit register window per lane.  What anchors the path:
// instantiated at k <= 31 this restatement must equal the pinned oracle (oracle/dbg_oracle.c) on every golden
// fixture, node for node (tests/test_wide.py).
//
// Because the rule definitions are SHARED with the kernels, a mistake in that header would pass every comparison between the two:
// tests/wide_checker.py restates the rules a third time in pure Python (strings and ints, nothing shared) and pins THIS file at
// k in {33, 47, 62, 63} and the real reference's dumps at k <= 32 (tests/test_wide_checker.py).
//
// Only tests/ may call this.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dbgk_wide.h"

using dbgk_wide::Key128;
typedef unsigned __int128 u128;

namespace {

struct Obs {
	uint64_t hi, lo;
	uint8_t lb, rb;
};

inline bool key_less(const Obs &a, const Obs &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

}  // namespace

extern "C" {

// build_debruijn_graph for 128-bit keys on reads held in memory (bases back to back, offsets[n_reads + 1]).
// *out_nodes: malloc'ed array of *n_nodes nodes sorted by (kmer_hi, kmer_lo), the key-0 node first (always present,
// DBGgraph.cpp:418).  total_kmers: sum of (len - k + 1) over reads with len >= k, untrimmed (DBGgraph.cpp:101).
int orcw_build(const char *bases, const uint64_t *offsets, uint64_t n_reads, int k, int max_read_len, dbgk_node32 **out_nodes,
               uint64_t *n_nodes, uint64_t *total_kmers)
{
	if (k < 1 || k > 63 || !out_nodes || !n_nodes) return -1;
	const u128 mask = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1); // KmerHeadMaskVal (DBGgraph.cpp:371)
	std::vector<Obs> obs;
	uint64_t tot = 0;
	for (uint64_t r = 0; r < n_reads; r++) {
		const char *seq = bases + offsets[r];
		const int64_t len = (int64_t)(offsets[r + 1] - offsets[r]);
		if (len < k) continue;                                    // DBGgraph.cpp:51-53
		const int64_t readlen = len > max_read_len ? max_read_len : len; // :63
		tot += (uint64_t)(len - k + 1);                            // :101 (untrimmed)
		u128 fwd = 0;
		for (int64_t j = 0; j + k <= readlen; j++) {
			if (j == 0) {
				for (int i = 0; i < k; i++) fwd = (fwd << 2) | dbgk_wide::base_code((unsigned char)seq[i]); // seq2bit, seqKmer.cpp:34-41
			} else {
				fwd = ((fwd << 2) | dbgk_wide::base_code((unsigned char)seq[j + k - 1])) & mask;             // :71-72
			}
			const uint32_t left = j > 0 ? dbgk_wide::base_code((unsigned char)seq[j - 1]) : 4u;              // :82-83
			const uint32_t right = j < readlen - k ? dbgk_wide::base_code((unsigned char)seq[j + k]) : 4u;   // :87-88
			const dbgk_wide::Observation o = dbgk_wide::canonical(Key128{(uint64_t)(fwd >> 64), (uint64_t)fwd}, k, left, right);
			obs.push_back(Obs{o.key.hi, o.key.lo, (uint8_t)o.lb, (uint8_t)o.rb});
		}
	}
	std::stable_sort(obs.begin(), obs.end(), key_less);
	std::vector<dbgk_node32> nodes;
	nodes.push_back(dbgk_node32{0, 0, 0, 0, 0}); // the key-0 node exists even without poly-A reads
	size_t i = 0;
	while (i < obs.size()) {
		size_t j = i;
		uint64_t links = 0;
		for (; j < obs.size() && obs[j].hi == obs[i].hi && obs[j].lo == obs[i].lo; j++) links = dbgk_wide::observe(links, obs[j].lb, obs[j].rb);
		const dbgk_node32 nd = {obs[i].hi, obs[i].lo, (uint32_t)links, (uint32_t)(links >> 32), 0};
		if ((nd.kmer_hi | nd.kmer_lo) == 0) nodes[0] = nd;
		else nodes.push_back(nd);
		i = j;
	}
	dbgk_node32 *out = (dbgk_node32 *)malloc(nodes.size() * sizeof(dbgk_node32));
	if (!out) return -2;
	memcpy(out, nodes.data(), nodes.size() * sizeof(dbgk_node32));
	*out_nodes = out;
	*n_nodes = nodes.size();
	if (total_kmers) *total_kmers = tot;
	return 0;
}

void orcw_free(void *p) { free(p); }

uint64_t orcw_digest(const dbgk_node32 *nodes, uint64_t n)
{
	uint64_t sum = 0;
	for (uint64_t i = 0; i < n; i++)
		sum += dbgk_wide::node_digest(Key128{nodes[i].kmer_hi, nodes[i].kmer_lo}, (uint64_t)nodes[i].l_link | ((uint64_t)nodes[i].r_link << 32));
	return sum;
}

// the consumer-side invariants (exist_kmerset kmerSet.cpp:280-302) for a table of 32-byte nodes:
// 0 ok; 1 a flagged slot is unreachable from its key's home without crossing a clear flag; 2 an unflagged slot is not
// all-zero; 3 number of flags != expect_count; 4 the same key twice
int orcw_check_host_table(const dbgk_node32 *array, const uint8_t *nul_flag, uint64_t size, uint64_t expect_count)
{
	auto flagged = [&](uint64_t i) { return (nul_flag[i >> 3] & (uint8_t)(128u >> (i & 7u))) != 0; };
	uint64_t count = 0;
	for (uint64_t i = 0; i < size; i++) {
		if (!flagged(i)) {
			if (array[i].kmer_hi | array[i].kmer_lo | array[i].l_link | array[i].r_link | array[i].reserved) return 2;
			continue;
		}
		count++;
		uint64_t hc = dbgk_wide::hash128(Key128{array[i].kmer_hi, array[i].kmer_lo}) % size;
		for (uint64_t steps = 0;; steps++) {
			if (steps > size || !flagged(hc)) return 1;
			if (hc == i) break;
			if (array[hc].kmer_hi == array[i].kmer_hi && array[hc].kmer_lo == array[i].kmer_lo) return 4;
			hc = (hc + 1 == size) ? 0 : hc + 1;
		}
	}
	return count == expect_count ? 0 : 3;
}

}  // extern "C"
