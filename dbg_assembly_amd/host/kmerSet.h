// kmerSet.h -- the host-visible k-mer hash set (the de Bruijn graph container).
//
// Source- and layout-compatible with the reference's DBG_contig/kmerSet.h:53-208: KmerNode is the
// same 16-byte record, KmerSet the same 80-byte control block, array/nul_flag/del_flag are
// malloc()ed so free_hash()/realloc() keep working, and a key lives on the linear-probe chain that
// starts at hash_code(key) % size.  In this build the table is FILLED by the GPU
// (DBGgraph.cpp -> include/dbgk.h) and handed over in exactly this layout; the functions below are
// the host-side maintenance and lookup API the consumer (DBG_contig/contig.cpp) calls afterwards.
#ifndef DBGK_HOST_KMERSET_H_
#define DBGK_HOST_KMERSET_H_

#include <inttypes.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <iostream>

using namespace std;

extern uint8_t BitOrVal[8];    // flag masks 128 >> i                      (kmerSet.cpp:53)
extern uint32_t BitAddVal[4];  // +1 on the counter byte of base A,C,G,T   (kmerSet.cpp:56)

typedef struct {               // kmerSet.h:70-75 -- must stay 16 bytes, identical to dbgk_node
	uint64_t kmer;
	uint32_t l_link;           // four saturating 8-bit counters, A in bits 31..24 ... T in 7..0
	uint32_t r_link;
} KmerNode;

typedef struct {               // kmerSet.h:79-84 (argument block of thread_memset)
	void *pointer;
	uint64_t memsize;
	int value;
} THREAD;

typedef struct {               // kmerSet.h:88-99 -- field order and types fixed by the consumer
	uint32_t e_size;
	uint64_t size;
	uint64_t count;
	uint64_t count_conflict;
	uint64_t max;
	float load_factor;
	uint64_t iter_ptr;
	KmerNode *array;
	uint8_t *nul_flag;
	uint8_t *del_flag;
} KmerSet;

static_assert(sizeof(KmerNode) == 16, "KmerNode must be 16 bytes");

// 64-bit integer mix used as the table hash (kmerSet.h:105-116); the GPU kernels compute the same
// function (csrc/dbgk_device.h) so that exist_kmerset() finds what the device stored.
inline uint64_t hash_code(uint64_t kmer)
{
	uint64_t h = kmer;
	h = h + ~(h << 32);
	h = h ^ (h >> 22);
	h = h + ~(h << 13);
	h = h ^ (h >> 8);
	h = h + (h << 3);
	h = h ^ (h >> 15);
	h = h + ~(h << 27);
	h = h ^ (h >> 31);
	return h;
}

inline int hash_equal(uint64_t kmer, KmerNode *b) { return kmer == b->kmer; }

// flag bit of slot idx: byte idx/8, mask 128 >> (idx % 8)  (kmerSet.h:144-169)
inline int is_entity_null(uint8_t *nul_flag, uint64_t idx) { return (nul_flag[idx >> 3] & (0x80u >> (idx & 7u))) ? 0 : 1; }
inline void set_entity_fill(uint8_t *nul_flag, uint64_t idx) { nul_flag[idx >> 3] |= BitOrVal[idx & 7u]; }
inline int is_entity_delete(uint8_t *del_flag, uint64_t idx) { return (del_flag[idx >> 3] & (0x80u >> (idx & 7u))) ? 1 : 0; }
inline void set_entity_delete(uint8_t *del_flag, uint64_t idx) { del_flag[idx >> 3] |= BitOrVal[idx & 7u]; }

void free_hash(KmerSet *set);
int is_prime(uint64_t num);                 // the reference's test, float sqrt bound included (kmerSet.cpp:72-81)
uint64_t find_next_prime(uint64_t num);
KmerSet *init_kmerset_parallel(uint64_t init_size, float load_factor, int threadNum);
void enlarge_kmerset_parallel(KmerSet *set, uint64_t num, int threadNum);
int add_node_to_kmerset(KmerSet *set, KmerNode *e);
uint64_t exist_kmerset(KmerSet *set, uint64_t kmer);   // slot index, or set->size when absent/deleted
int delete_kmerset(KmerSet *set, uint64_t kmer);
void print_kmerset_entity(KmerSet *set);
void print_kmerset_parameter(KmerSet *set);
uint8_t get_next_kmer_depth(uint32_t link, uint8_t base);
void *thread_memset(void *paras);
void *memset_parallel(void *pointer, int value, uint64_t memsize, int threadNum);

// ---- k-mers of 33..63 bases (this build only: the reference stops at k = 31, DBG_contig/main.cpp:100) --------------------
// The same container carried to 128-bit keys: a 32-byte node {kmer_hi, kmer_lo, l_link, r_link, reserved} (layout of
// dbgk_node32, include/dbgk_wide.h), the same control block, the same flag arrays; a key lives on the linear-probe chain that
// starts at hash_code128(hi, lo) % size, where hash_code128 is hash_code(lo) for hi == 0 -- the reference's slot for every
// k <= 32 -- and hash_code(lo ^ hash_code(hi)) otherwise.  PARITY UNPINNED: nothing in the reference defines these.
typedef struct {
	uint64_t kmer_hi;          // bases 0 .. k-33 of the k-mer (2 bits each, first base most significant)
	uint64_t kmer_lo;          // the last 32 bases
	uint32_t l_link;           // as KmerNode
	uint32_t r_link;
	uint64_t reserved;
} KmerNode32;

typedef struct {
	uint32_t e_size;           // 32
	uint64_t size;
	uint64_t count;
	uint64_t count_conflict;
	uint64_t max;
	float load_factor;
	uint64_t iter_ptr;
	KmerNode32 *array;
	uint8_t *nul_flag;
	uint8_t *del_flag;
} KmerSet128;

static_assert(sizeof(KmerNode32) == 32, "KmerNode32 must be 32 bytes");

inline uint64_t hash_code128(uint64_t hi, uint64_t lo) { return hash_code(hi ? (lo ^ hash_code(hi)) : lo); }
uint64_t exist_kmerset128(KmerSet128 *set, uint64_t kmer_hi, uint64_t kmer_lo);   // slot index, or set->size when absent/deleted (exist_kmerset's rule)
void free_hash128(KmerSet128 *set);
KmerSet128 *adopt_kmerset128(uint64_t size, float load_factor, uint64_t count, uint64_t count_conflict, KmerNode32 *array, uint8_t *nul_flag,
                             uint8_t *del_flag);

// malloc() for the big arrays of a set (free()- and realloc()-compatible, as free_hash / enlarge_kmerset_parallel need): from 4 MiB
// on, 2 MiB-aligned and advised for transparent huge pages -- a 9.6 GB table then takes 4 600 page faults to fill instead of
// 2.3 million and the process exits without unmapping it page by page (0.08 s against 0.74 s to touch it with 8 threads, 0.9 s
// less at exit: profiles/r03_hip_startup.txt).  zero = calloc semantics.
void *kmerset_alloc(size_t bytes, bool zero);

// wraps already-filled arrays (as produced by dbgk_export_host_table) into a KmerSet control block
KmerSet *adopt_kmerset(uint64_t size, float load_factor, uint64_t count, uint64_t count_conflict,
                       KmerNode *array, uint8_t *nul_flag, uint8_t *del_flag);

#endif
