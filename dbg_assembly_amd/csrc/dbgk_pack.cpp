// dbgk_pack.cpp -- host side of the 2-bit read format: ASCII bases -> 2-bit codes, 16 bases per 32-bit word, base 0 of a
// word in bits 31..30 (what pack16_ascii makes of the same 16 bytes on the device, dbgk_device.h).
//
// The code of a letter is the reference's alphabet[] (DBG_contig/seqKmer.cpp:9-19): A a N n -> 0, C c -> 1, G g -> 2,
// T t -> 3, i.e. ((c >> 1) ^ (c >> 2)) & 3 for those ten letters.  Every OTHER byte is read as 'A' and counted -- the rule of
// every engine of this library (include/dbgk.h, dbgk_stats.other_bytes); the reference itself reads out of bounds on them
// (alphabet[] gives 4, KmerRCOrVal[4], DBGgraph.cpp:71-73).
//
// No GPU code here: plain C++ with an AVX2 body chosen at run time (the build box and the GPU box differ).
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <immintrin.h>

#include "dbgk.h"

namespace {

struct Lut {
	uint8_t code[256];
	uint8_t other[256];
	Lut()
	{
		for (int c = 0; c < 256; c++) {
			const int u = c & 0xDF;
			const bool letter = u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N';
			code[c] = letter ? (uint8_t)(((c >> 1) ^ (c >> 2)) & 3) : 0;
			other[c] = letter ? 0 : 1;
		}
	}
};
const Lut g_lut;

// n <= 16 bases -> the top 2n bits of a word (the rest 0)
inline uint32_t pack_scalar(const unsigned char *s, uint32_t n, uint64_t &other)
{
	uint32_t w = 0;
	for (uint32_t i = 0; i < n; i++) {
		w |= (uint32_t)g_lut.code[s[i]] << (30u - 2u * i);
		other += g_lut.other[s[i]];
	}
	return w;
}

void pack_words_scalar(const unsigned char *s, uint64_t n_words, uint32_t *out, uint64_t &other)
{
	for (uint64_t w = 0; w < n_words; w++) out[w] = pack_scalar(s + 16 * w, 16, other);
}

__attribute__((target("avx2,popcnt"))) void pack_words_avx2(const unsigned char *s, uint64_t n_words, uint32_t *out, uint64_t &other)
{
	const __m256i m3 = _mm256_set1_epi8(3), mDF = _mm256_set1_epi8((char)0xDF);
	const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T'),
	              cN = _mm256_set1_epi8('N');
	const __m256i mul1 = _mm256_set1_epi16(0x0104);    // (b0, b1) -> 4 * b0 + b1
	const __m256i mul2 = _mm256_set1_epi32(0x00010010); // (p0, p1) -> 16 * p0 + p1
	const __m256i gather = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
	                                        12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
	uint64_t bad = 0;
	uint64_t w = 0;
	for (; w + 2 <= n_words; w += 2) {
		const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + 16 * w));
		const __m256i u = _mm256_and_si256(x, mDF);
		const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
		                                                   _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT))),
		                                   _mm256_cmpeq_epi8(u, cN));
		bad += 32u - (uint32_t)_mm_popcnt_u32((uint32_t)_mm256_movemask_epi8(ok));
		__m256i c = _mm256_xor_si256(_mm256_srli_epi16(x, 1), _mm256_srli_epi16(x, 2)); // bits 1..0 of every byte: the code
		c = _mm256_and_si256(_mm256_and_si256(c, m3), ok);                                // other bytes -> 'A'
		const __m256i p = _mm256_madd_epi16(_mm256_maddubs_epi16(c, mul1), mul2);         // one byte (4 bases, MSB first) per dword
		const __m256i g = _mm256_shuffle_epi8(p, gather);                                 // 16 bases per 128-bit half, as one dword
		out[w] = (uint32_t)_mm256_extract_epi32(g, 0);
		out[w + 1] = (uint32_t)_mm256_extract_epi32(g, 4);
	}
	other += bad;
	if (w < n_words) out[w] = pack_scalar(s + 16 * w, 16, other);
}

typedef void (*pack_fn)(const unsigned char *, uint64_t, uint32_t *, uint64_t &);
pack_fn choose()
{
	__builtin_cpu_init();
	return (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt")) ? pack_words_avx2 : pack_words_scalar;
}
const pack_fn g_pack_words = choose();

} // namespace

extern "C" int dbgk_pack_bases(const char *bases, uint64_t n_bases, uint32_t *packed, uint64_t first_base, uint64_t *other_bytes)
{
	if ((n_bases && !bases) || !packed) return DBGK_ERR_ARG;
	const unsigned char *s = reinterpret_cast<const unsigned char *>(bases);
	uint64_t other = 0, done = 0, at = first_base;
	// a first word that starts in the middle: OR-ed in (another call may own its other part)
	if (n_bases && (at & 15u)) {
		const uint32_t room = 16u - (uint32_t)(at & 15u), n = (uint32_t)(n_bases < room ? n_bases : room);
		const uint32_t w = pack_scalar(s, n, other) >> (2u * (uint32_t)(at & 15u));
		__atomic_fetch_or(&packed[at >> 4], w, __ATOMIC_RELAXED);
		done = n;
		at += n;
	}
	const uint64_t full = (n_bases - done) >> 4;
	if (full) g_pack_words(s + done, full, packed + (at >> 4), other);
	done += full * 16;
	at += full * 16;
	if (done < n_bases) { // a last word that ends in the middle: OR-ed in as well
		const uint32_t w = pack_scalar(s + done, (uint32_t)(n_bases - done), other);
		__atomic_fetch_or(&packed[at >> 4], w, __ATOMIC_RELAXED);
	}
	if (other_bytes) *other_bytes += other;
	return DBGK_OK;
}

// n_reads sequences packed back to back from base position first_base on: what one reader thread does with its share of a batch.
// The bytes are gathered into a small scratch buffer (cache resident) so that the word packer runs on 16-base groups whatever the
// read lengths are; only the stream's first and last word can be shared with a neighbour (OR-ed in, as in dbgk_pack_bases).
extern "C" int dbgk_pack_reads(const dbgk_read_ref *reads, uint64_t n_reads, uint32_t *packed, uint64_t first_base, uint64_t *other_bytes)
{
	if ((n_reads && !reads) || !packed) return DBGK_ERR_ARG;
	constexpr size_t kScratch = 32768; // a multiple of 16
	unsigned char scratch[kScratch];
	size_t fill = 0;
	uint64_t at = first_base, other = 0;
	bool head_done = (at & 15u) == 0;
	auto flush = [&](bool last) {
		size_t from = 0;
		if (!head_done) { // the stream starts inside a word: bring it to the word boundary first
			const uint32_t room = 16u - (uint32_t)(at & 15u);
			if (fill < room && !last) return; // (keep gathering)
			const uint32_t n = (uint32_t)(fill < room ? fill : room);
			const uint32_t w = pack_scalar(scratch, n, other) >> (2u * (uint32_t)(at & 15u));
			__atomic_fetch_or(&packed[at >> 4], w, __ATOMIC_RELAXED);
			at += n;
			from = n;
			head_done = (at & 15u) == 0;
		}
		const size_t full = (fill - from) >> 4;
		if (full) {
			g_pack_words(scratch + from, full, packed + (at >> 4), other);
			at += full * 16;
			from += full * 16;
		}
		if (last && from < fill) {
			const uint32_t w = pack_scalar(scratch + from, (uint32_t)(fill - from), other);
			__atomic_fetch_or(&packed[at >> 4], w, __ATOMIC_RELAXED);
			at += fill - from;
			from = fill;
		}
		memmove(scratch, scratch + from, fill - from);
		fill -= from;
	};
	for (uint64_t i = 0; i < n_reads; i++) {
		const unsigned char *s = reinterpret_cast<const unsigned char *>(reads[i].seq);
		size_t left = reads[i].len;
		while (left) {
			const size_t take = left < kScratch - fill ? left : kScratch - fill;
			memcpy(scratch + fill, s, take);
			fill += take;
			s += take;
			left -= take;
			if (fill == kScratch) flush(false);
		}
	}
	flush(true);
	if (other_bytes) *other_bytes += other;
	return DBGK_OK;
}

extern "C" int dbgk_unpack_bases(const uint32_t *packed, uint64_t first_base, uint64_t n_bases, char *bases)
{
	if ((n_bases && !bases) || (n_bases && !packed)) return DBGK_ERR_ARG;
	for (uint64_t i = 0; i < n_bases; i++) {
		const uint64_t p = first_base + i;
		bases[i] = "ACGT"[(packed[p >> 4] >> (30u - 2u * (uint32_t)(p & 15u))) & 3u];
	}
	return DBGK_OK;
}

// dst words [0, n_words) = the 2-bit stream of src starting at base first_base (any alignment): what a batch cut out of the middle
// of a packed buffer looks like when it has to start on a word of its own
extern "C" void dbgk_internal_shift_packed(const uint32_t *src, uint64_t first_base, uint64_t n_words, uint64_t src_words, uint32_t *dst)
{
	const uint64_t w0 = first_base >> 4;
	const uint32_t sh = 2u * (uint32_t)(first_base & 15u);
	if (sh == 0) {
		memcpy(dst, src + w0, n_words * 4);
		return;
	}
	for (uint64_t i = 0; i < n_words; i++) {
		const uint32_t hi = src[w0 + i], lo = (w0 + i + 1 < src_words) ? src[w0 + i + 1] : 0u;
		dst[i] = (hi << sh) | (lo >> (32u - sh));
	}
}
