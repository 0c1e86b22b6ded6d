// kmerSet.cpp -- host-side maintenance of the k-mer hash set (see kmerSet.h).
// Semantics follow /root/reference/DBG_contig/kmerSet.cpp; line references are to that file.
#include "kmerSet.h"

#include <sys/mman.h>

#include <thread>
#include <vector>

uint8_t BitOrVal[8] = {0x80, 0x40, 0x20, 0x10, 0x08, 0x04, 0x02, 0x01};
uint32_t BitAddVal[4] = {1u << 24, 1u << 16, 1u << 8, 1u};

static inline float clamp_load_factor(float lf)
{
	if (lf <= 0) return 0.25f;   // :110-111
	if (lf >= 1) return 0.75f;
	return lf;
}

// (uint64_t)(size * load_factor) with the product formed in FLOAT, as the reference does (:114,:145);
// e.g. 100000007 * 0.7f -> 70000008
static inline uint64_t cutoff_for(uint64_t size, float lf) { return (uint64_t)((float)size * lf); }

void free_hash(KmerSet *set)
{
	if (!set) return;
	free(set->array);
	free(set->nul_flag);
	free(set->del_flag);
	free(set);
}

int is_prime(uint64_t num)
{
	// Trial division by odd numbers BELOW a float-precision square root (strict '<'), :72-81.
	// Deliberately keeps the reference's blind spots (9, 15, 25, 49 ... pass) because the table
	// size -- hence every slot index -- must match what the reference would pick for the same -i.
	if (num < 4) return 1;
	if (!(num & 1u)) return 0;
	const uint64_t bound = (uint64_t)sqrtf((float)num);
	for (uint64_t d = 3; d < bound; d += 2)
		if (num % d == 0) return 0;
	return 1;
}

uint64_t find_next_prime(uint64_t num)
{
	num |= 1u;  // even -> next odd (:87)
	while (!is_prime(num)) num += 2;
	return num;
}

void *thread_memset(void *paras)
{
	THREAD *t = static_cast<THREAD *>(paras);
	memset(t->pointer, t->value, t->memsize);
	return NULL;
}

void *memset_parallel(void *pointer, int value, uint64_t memsize, int threadNum)
{
	// The reference always writes zeros whatever `value` says (:371) and its only callers pass 0;
	// keep that contract.  Split into threadNum slices plus the remainder.
	(void)value;
	if (threadNum < 1) threadNum = 1;
	const uint64_t slice = memsize / (uint64_t)threadNum;
	std::vector<std::thread> workers;
	char *p = static_cast<char *>(pointer);
	for (int i = 0; i < threadNum && slice; i++) workers.emplace_back([=] { memset(p + (uint64_t)i * slice, 0, slice); });
	memset(p + slice * (uint64_t)threadNum, 0, memsize - slice * (uint64_t)threadNum);
	for (auto &w : workers) w.join();
	return pointer;
}

void *kmerset_alloc(size_t bytes, bool zero)
{
	if (bytes < (4u << 20)) return zero ? calloc(bytes ? bytes : 1, 1) : malloc(bytes ? bytes : 1);
	void *p = NULL;
	if (posix_memalign(&p, 2u << 20, bytes) != 0) return NULL;
	(void)madvise(p, bytes, MADV_HUGEPAGE); // (a hint: without transparent huge pages this is an ordinary aligned allocation)
	if (zero) memset_parallel(p, 0, bytes, 8);
	return p;
}

KmerSet *adopt_kmerset(uint64_t size, float load_factor, uint64_t count, uint64_t count_conflict,
                       KmerNode *array, uint8_t *nul_flag, uint8_t *del_flag)
{
	KmerSet *s = static_cast<KmerSet *>(malloc(sizeof(KmerSet)));  // free_hash() releases it with free()
	if (!s) return NULL;
	s->e_size = sizeof(KmerNode);
	s->size = size;
	s->count = count;
	s->count_conflict = count_conflict;
	s->load_factor = clamp_load_factor(load_factor);
	s->max = cutoff_for(size, s->load_factor);
	s->iter_ptr = 0;
	s->array = array;
	s->nul_flag = nul_flag;
	s->del_flag = del_flag;
	return s;
}

KmerSet *init_kmerset_parallel(uint64_t init_size, float load_factor, int threadNum)
{
	const uint64_t size = init_size < 3 ? 3 : find_next_prime(init_size);  // :103-104
	KmerNode *array = static_cast<KmerNode *>(kmerset_alloc(size * sizeof(KmerNode), false));
	uint8_t *nul = static_cast<uint8_t *>(kmerset_alloc(size / 8 + 1, true));
	uint8_t *del = static_cast<uint8_t *>(kmerset_alloc(size / 8 + 1, true));
	if (!array || !nul || !del) {
		free(array), free(nul), free(del);
		return NULL;
	}
	memset_parallel(array, 0, size * sizeof(KmerNode), threadNum);
	return adopt_kmerset(size, load_factor, 0, 0, array, nul, del);
}

void enlarge_kmerset_parallel(KmerSet *set, uint64_t num, int threadNum)
{
	// :132-189.  Size grows along find_next_prime(2*size) until size*load_factor (float) covers
	// count+num; entries are then re-seated IN PLACE, scanning old slots in index order: an entry
	// lifted from its slot is put on the first free slot of its new probe chain, and if a
	// not-yet-moved old entry occupies that slot it is lifted next.
	const uint64_t old_size = set->size;
	uint64_t new_size = old_size;
	do {
		new_size = find_next_prime(new_size * 2);
	} while ((float)new_size * set->load_factor < (float)(set->count + num));

	set->array = static_cast<KmerNode *>(realloc(set->array, new_size * sizeof(KmerNode)));
	memset_parallel(set->array + old_size, 0, (new_size - old_size) * sizeof(KmerNode), threadNum);
	set->size = new_size;
	set->max = cutoff_for(new_size, set->load_factor);

	uint8_t *was_filled = set->nul_flag, *moved = set->del_flag;
	set->nul_flag = static_cast<uint8_t *>(calloc(new_size / 8 + 1, 1));
	set->del_flag = static_cast<uint8_t *>(calloc(new_size / 8 + 1, 1));

	for (uint64_t i = 0; i < old_size; i++) {
		if (is_entity_null(was_filled, i) || is_entity_delete(moved, i)) continue;
		KmerNode carry = set->array[i];
		memset(&set->array[i], 0, sizeof(KmerNode));
		set_entity_delete(moved, i);
		for (;;) {
			uint64_t slot = hash_code(carry.kmer) % new_size;
			while (!is_entity_null(set->nul_flag, slot)) slot = (slot + 1) % new_size;
			set_entity_fill(set->nul_flag, slot);
			const bool displaces = slot < old_size && !is_entity_null(was_filled, slot) && !is_entity_delete(moved, slot);
			if (!displaces) {
				set->array[slot] = carry;
				break;
			}
			std::swap(carry, set->array[slot]);
			set_entity_delete(moved, slot);
		}
	}
	free(was_filled);
	free(moved);
}

int add_node_to_kmerset(KmerSet *set, KmerNode *e)
{
	// :253-273 -- first slot whose null flag is clear on the key's probe chain
	for (uint64_t slot = hash_code(e->kmer) % set->size;; slot = (slot + 1 == set->size) ? 0 : slot + 1) {
		if (is_entity_null(set->nul_flag, slot)) {
			set->array[slot] = *e;
			set_entity_fill(set->nul_flag, slot);
			set->count++;
			return 1;
		}
		set->count_conflict++;
	}
}

uint64_t exist_kmerset(KmerSet *set, uint64_t kmer)
{
	// :280-302
	for (uint64_t slot = hash_code(kmer) % set->size;; slot = (slot + 1 == set->size) ? 0 : slot + 1) {
		if (is_entity_null(set->nul_flag, slot)) return set->size;
		if (set->array[slot].kmer == kmer) return is_entity_delete(set->del_flag, slot) ? set->size : slot;
	}
}

int delete_kmerset(KmerSet *set, uint64_t kmer)
{
	const uint64_t idx = exist_kmerset(set, kmer);
	if (idx == set->size) return 0;
	set_entity_delete(set->del_flag, idx);
	set->count--;
	return 1;
}

void print_kmerset_entity(KmerSet *set)
{
	cout << "\narray_id\thash_kmer\thash_val\n";
	for (uint64_t i = 0; i < set->size; i++)
		if (!is_entity_null(set->nul_flag, i) && !is_entity_delete(set->del_flag, i)) cout << i << "\t" << set->array[i].kmer << "\n";
}

void print_kmerset_parameter(KmerSet *set)
{
	// same labels as the reference's log (test/02.build_contig/Ecoli_corrected_reads.contig.log:440-447)
	cerr << "\nKmerset hash parameters:" << endl;
	cerr << "element_size:\t" << set->e_size << endl;
	cerr << "array_size:\t" << set->size << endl;
	cerr << "load_factor:\t" << set->load_factor << endl;
	cerr << "max_cutoff:\t" << set->max << endl;
	cerr << "iter_ptr:\t" << set->iter_ptr << endl;
	cerr << "count:\t" << set->count << endl;
	cerr << "conflict:\t" << set->count_conflict << endl;
}

uint8_t get_next_kmer_depth(uint32_t link, uint8_t base) { return (uint8_t)(link >> ((3 - base) * 8)); }

// ---- 128-bit keys (k = 33..63; this build only, see kmerSet.h) -------------------------------------------------------------
uint64_t exist_kmerset128(KmerSet128 *set, uint64_t kmer_hi, uint64_t kmer_lo)
{
	uint64_t hc = hash_code128(kmer_hi, kmer_lo) % set->size; // exist_kmerset's walk (kmerSet.cpp:280-302)
	for (;;) {
		if (is_entity_null(set->nul_flag, hc)) return set->size;
		if (set->array[hc].kmer_hi == kmer_hi && set->array[hc].kmer_lo == kmer_lo) return is_entity_delete(set->del_flag, hc) ? set->size : hc;
		hc = (hc + 1 == set->size) ? 0 : hc + 1;
	}
}

void free_hash128(KmerSet128 *set)
{
	if (!set) return;
	free(set->array);
	free(set->nul_flag);
	free(set->del_flag);
	free(set);
}

KmerSet128 *adopt_kmerset128(uint64_t size, float load_factor, uint64_t count, uint64_t count_conflict, KmerNode32 *array, uint8_t *nul_flag,
                             uint8_t *del_flag)
{
	KmerSet128 *set = static_cast<KmerSet128 *>(malloc(sizeof(KmerSet128)));
	if (!set) return NULL;
	set->e_size = sizeof(KmerNode32);
	set->size = size;
	set->count = count;
	set->count_conflict = count_conflict;
	set->load_factor = load_factor <= 0 ? 0.25f : (load_factor >= 1 ? 0.75f : load_factor);
	set->max = (uint64_t)((float)size * set->load_factor);
	set->iter_ptr = 0;
	set->array = array;
	set->nul_flag = nul_flag;
	set->del_flag = del_flag;
	return set;
}

