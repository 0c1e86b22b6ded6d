// reads_io.h -- streaming reader of (optionally gzip'ed) one-line FASTQ / FASTA files.
//
// Record rules of the reference (DBG_contig/DBGgraph.cpp:244-272, same in correct_error): a line
// whose first character is '@' (format 1) / '>' (otherwise) announces a record and the NEXT line is
// its sequence; format 1 then skips two lines ('+' and qualities, whatever they start with); any
// other line is ignored.  A header on the very last line yields an empty read.  zlib is used
// directly with large buffers (the reference's gzstream wrapper reads through 303 bytes).
#ifndef DBGK_HOST_READS_IO_H_
#define DBGK_HOST_READS_IO_H_

#include <zlib.h>
#include <cstring>
#include <string>
#include <vector>

// calls cb(sequence pointer, length) for every record; false if the file cannot be opened.  When `stop`
// is given and set by the callback, the rest of the file is not read (the reference abandons a file at
// its -e memory cap, DBG_contig/DBGgraph.cpp:346-350).
template <class Callback>
bool for_each_read_in_file(const std::string &path, int format, Callback cb, const bool *stop = nullptr)
{
	gzFile fp = gzopen(path.c_str(), "rb");
	if (!fp) return false;
	gzbuffer(fp, 1 << 20);
	const char marker = (format == 1) ? '@' : '>';
	const size_t CHUNK = 8u << 20;
	std::vector<char> buf(CHUNK + 1);
	size_t have = 0;  // bytes of an unfinished line carried over
	int state = 0;    // 0 look for header, 1 sequence line, 2/3 skip (FASTQ '+' and quality)
	bool eof = false;
	while (!eof && !(stop && *stop)) {
		if (buf.size() < have + CHUNK) buf.resize(have + CHUNK);
		const int got = gzread(fp, buf.data() + have, (unsigned)CHUNK);
		if (got <= 0) eof = true;
		const size_t end = have + (got > 0 ? (size_t)got : 0);
		size_t pos = 0;
		while (pos < end && !(stop && *stop)) {
			const size_t span = end - pos;
			if (span > ((size_t)1 << 40)) break; // (never: pos < end <= buf.size(); tells the optimiser the bound of the scan)
			const char *nl = static_cast<const char *>(memchr(buf.data() + pos, '\n', span));
			size_t line_end;
			if (nl) line_end = (size_t)(nl - buf.data());
			else if (eof) line_end = end;  // last line without a newline
			else break;
			const char *line = buf.data() + pos;
			const size_t len = line_end - pos;
			switch (state) {
				case 0: if (len && line[0] == marker) state = 1; break;
				case 1: cb(line, len); state = (format == 1) ? 2 : 0; break;
				case 2: state = 3; break;
				default: state = 0; break;
			}
			pos = line_end + 1;
		}
		have = pos < end ? end - pos : 0;
		if (have) memmove(buf.data(), buf.data() + pos, have);
	}
	if (state == 1 && !(stop && *stop)) cb("", 0);
	gzclose(fp);
	return true;
}

#endif
