/*
 * oracle_synth.c -- TEST INFRASTRUCTURE ONLY.
 * Host-side synthetic read helpers for tests and bench.py's cpu_baseline leg: fill a buffer, or
 * write a one-line FASTA/FASTQ file, from the counter-based generator of include/dbgk_synth.h.
 */
#include <stdio.h>
#include <stdlib.h>
#include <zlib.h>
#include "dbgk_synth.h"

void orc_synth_fill(const dbgk_synth_params *P, uint64_t first, uint64_t n_reads, char *out)
{
	dbgk_synth_fill_host(P, first, n_reads, out);
}

/* format 1 = FASTQ (constant quality 'I'), 2 = FASTA; gz != 0 writes through zlib. 0 on success */
int orc_synth_write_file(const dbgk_synth_params *P, uint64_t first, uint64_t n_reads,
                         const char *path, int format, int gz)
{
	const uint32_t L = P->read_len;
	char *seq = (char *)malloc((size_t)L + 1), *qual = (char *)malloc((size_t)L + 1);
	if (!seq || !qual) return -1;
	for (uint32_t j = 0; j < L; j++) qual[j] = 'I';
	seq[L] = qual[L] = 0;
	gzFile gf = NULL;
	FILE *fp = NULL;
	if (gz) gf = gzopen(path, "wb1"); else fp = fopen(path, "w");
	if (!gf && !fp) { free(seq); free(qual); return -2; }
	char line[64];
	for (uint64_t i = 0; i < n_reads; i++) {
		dbgk_synth_fill_host(P, first + i, 1, seq);
		int hn = snprintf(line, sizeof line, "%cr%llu\n", format == 1 ? '@' : '>', (unsigned long long)(first + i));
		if (gz) {
			gzwrite(gf, line, (unsigned)hn); gzwrite(gf, seq, L); gzwrite(gf, "\n", 1);
			if (format == 1) { gzwrite(gf, "+\n", 2); gzwrite(gf, qual, L); gzwrite(gf, "\n", 1); }
		} else {
			fwrite(line, 1, (size_t)hn, fp); fwrite(seq, 1, L, fp); fputc('\n', fp);
			if (format == 1) { fputs("+\n", fp); fwrite(qual, 1, L, fp); fputc('\n', fp); }
		}
	}
	if (gz) gzclose(gf); else fclose(fp);
	free(seq); free(qual);
	return 0;
}
