// reads_io.h -- streaming reader of (optionally gzip'ed) one-line FASTQ / FASTA files.
//
// Record rules of the reference (DBG_contig/DBGgraph.cpp:244-272, same in correct_error): a line
// whose first character is '@' (format 1) / '>' (otherwise) announces a record and the NEXT line is
// its sequence; format 1 then skips two lines ('+' and qualities, whatever they start with); any
// other line is ignored.  A header on the very last line yields an empty read.  zlib is used
// directly with large buffers (the reference's gzstream wrapper reads through 303 bytes).
#ifndef DBGK_HOST_READS_IO_H_
#define DBGK_HOST_READS_IO_H_

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
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

// ---- plain (not compressed) files with several threads ---------------------------------------------------------------------
// The file is read in windows of 256 MiB into two buffers used in turn (2 MiB-aligned, huge pages advised): (1) n_threads threads
// pread() their byte ranges of the window and note the newline positions in them; (2) the SAME record rules as above run over those
// lines -- in the reader threads as well since round 4 (for_each_read_bulk explains how a slice copes with not knowing the state it is
// entered in); whatever a line begins with is judged in file order, so the result is that of the sequential reader for any input (FASTQ
// quality lines that start with '@' included) -- and the calling thread hands the records to the caller; (3) end_of_window() is called before the buffer is
// reused: the caller copies / packs the sequence bytes it was shown (again with several threads).  A line that straddles two
// windows is carried over to the front of the next buffer.  (2) and (3) of window i run while (1) of window i + 1 is under way.
class ChunkedReadsFile {
public:
	~ChunkedReadsFile() { close(); }
	// false: cannot be opened, or the file is gzip'ed (use for_each_read_in_file)
	bool open(const std::string &path)
	{
		close();
		fd_ = ::open(path.c_str(), O_RDONLY);
		if (fd_ < 0) return false;
		struct stat st;
		if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) return close(), false;
		size_ = (size_t)st.st_size;
		unsigned char magic[2] = {0, 0};
		if (size_ >= 2 && pread(fd_, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return close(), false;
		return true;
	}
	void close()
	{
		if (fd_ >= 0) ::close(fd_);
		fd_ = -1;
		size_ = 0;
	}
	size_t size() const { return size_; }

	// seconds the calling thread spent (measurements: DBGK_TIMINGS): [0] waiting for the first window, [1] handing the records to the
	// caller (its bookkeeping), [2] in end_of_window (the caller's copy / pack), [3] waiting for the window read ahead
	double spent[4] = {0, 0, 0, 0};
	bool too_long = false; // a line of 4 GiB or more was met where a sequence was expected (the caller gives up on the file)

	struct ReadRef { // == dbgk_read_ref (include/dbgk.h)
		const char *seq;
		uint32_t len;
	};

	// cb(sequence, length) per record
	template <class Callback, class EndOfWindow>
	bool for_each_read(int format, int n_threads, Callback cb, EndOfWindow end_of_window, const bool *stop = nullptr)
	{
		return for_each_read_bulk(format, n_threads, [&](const ReadRef *r, size_t n) {
			for (size_t i = 0; i < n && !(stop && *stop); i++) cb(r[i].seq, (size_t)r[i].len);
		}, end_of_window, stop);
	}

	// cb(records, n): the records of the file in order, in runs (what one reader thread found in its slice of a window).  The RECORD
	// RULES run in the reader threads too: a slice does not know the state the rules are in when its first line begins (0 look for
	// a header, 1 sequence line, 2 / 3 the two skipped lines of FASTQ), so it follows all four until they agree -- with well-formed
	// input after two to four lines -- keeping what each of them would have delivered up to there, and one common list from there on;
	// the calling thread then only picks, slice after slice, the list of the state it really arrives in.  Same records as the
	// sequential reader for ANY input (tests/test_reads_io.py: quality lines that start with '@', windows of a few hundred bytes).
	template <class BulkCallback, class EndOfWindow>
	bool for_each_read_bulk(int format, int n_threads, BulkCallback cb, EndOfWindow end_of_window, const bool *stop = nullptr)
	{
		auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		double t_mark = now();
		auto lap = [&](int i) { const double t = now(); spent[i] += t - t_mark; t_mark = t; };
		const char marker = (format == 1) ? '@' : '>';
#ifdef DBGK_READS_WINDOW
		const size_t WINDOW = DBGK_READS_WINDOW; // (tests: windows of a few hundred bytes)
#else
		const size_t WINDOW = (size_t)256 << 20;
#endif
		if (n_threads < 1) n_threads = 1;
		// TWO windows: while the calling thread hands the records of window i to the caller and the caller copies / packs its reads
		// (end_of_window), the reader threads already pread window i + 1, note its newlines and find its records.  What window i + 1
		// needs of window i -- the unfinished last line, carried to its front -- is known as soon as window i has been read.
		struct Slice {
			std::vector<uint64_t> nl;        // newline positions of the slice
			std::vector<ReadRef> head[4];    // records delivered when the slice is entered in state e, until the four agree
			std::vector<ReadRef> common;     // ... and from there on
			int exit_head[4] = {0, 1, 2, 3}; // state after the head part
			int exit_common = 0;
			bool converged = false;
			bool long_head[4] = {false, false, false, false}, long_common = false; // a line of >= 4 GiB among the sequences of that list
		};
		struct Window {
			char *buf = NULL;
			size_t cap = 0;
			size_t have = 0, take = 0; // carried-over bytes, bytes read from the file
			std::vector<Slice> slice;
			bool ok = true;
		} win[2];
		for (Window &W : win) W.slice.resize((size_t)n_threads);
		auto release = [&]() { for (Window &W : win) free(W.buf); };
		auto run_threads = [&](auto &&work) {
			std::vector<std::thread> th;
			for (int t = 1; t < n_threads; t++) th.emplace_back(work, t);
			work(0);
			for (auto &x : th) x.join();
		};
		auto read_window = [&](Window &W, size_t file_off, const char *carry, size_t carry_len) {
			W.take = std::min(WINDOW, size_ - file_off);
			W.have = carry_len;
			W.ok = grow(W.buf, W.cap, W.have + W.take);
			if (!W.ok) return;
			if (carry_len) memcpy(W.buf, carry, carry_len);
			const size_t per = (W.take + (size_t)n_threads - 1) / (size_t)n_threads;
			std::vector<char> failed((size_t)n_threads, 0);
			run_threads([&](int t) { // (1) the bytes and their newlines
				std::vector<uint64_t> &out = W.slice[(size_t)t].nl;
				out.clear();
				const size_t a = std::min(W.take, per * (size_t)t), b = std::min(W.take, a + per);
				for (size_t done = a; done < b;) { // pread may return less than asked
					const ssize_t got = pread(fd_, W.buf + W.have + done, b - done, (off_t)(file_off + done));
					if (got <= 0) { failed[(size_t)t] = 1; return; }
					done += (size_t)got;
				}
				out.reserve((b - a) / 64 + 16);
				for (const char *p = W.buf + W.have + a, *e = W.buf + W.have + b; p < e;) {
					const char *q = static_cast<const char *>(memchr(p, '\n', (size_t)(e - p)));
					if (!q) break;
					out.push_back((uint64_t)(q - W.buf));
					p = q + 1;
				}
			});
			for (char f : failed) W.ok = W.ok && !f;
			if (!W.ok) return;
			// a line of 4 GiB or more where a sequence would be: noted per speculative list (and for the common part), so that only
			// the list the file really takes can stop it -- the other three states are hypotheses the file never reaches
			run_threads([&](int t) { // (2) the record rules over the lines that END in this slice
				Slice &S = W.slice[(size_t)t];
				for (auto &h : S.head) h.clear();
				S.common.clear();
				S.converged = false;
				size_t line_start = 0; // behind the last newline of the slices in front (the carried-over bytes hold none)
				for (int u = t - 1; u >= 0; u--)
					if (!W.slice[(size_t)u].nl.empty()) { line_start = (size_t)W.slice[(size_t)u].nl.back() + 1; break; }
				int st[4] = {0, 1, 2, 3};
				int cur = 0;
				for (int e = 0; e < 4; e++) S.long_head[e] = false;
				S.long_common = false;
				auto step = [&](int &state, const char *line, size_t len, std::vector<ReadRef> &out, bool &too_long_here) {
					switch (state) { // the record rules of for_each_read_in_file
						case 0: if (len && line[0] == marker) state = 1; break;
						case 1:
							if (len >> 32) too_long_here = true;
							out.push_back(ReadRef{line, (uint32_t)len});
							state = (format == 1) ? 2 : 0;
							break;
						case 2: state = 3; break;
						default: state = 0; break;
					}
				};
				for (uint64_t pos64 : S.nl) {
					const size_t pos = (size_t)pos64, len = pos - line_start;
					const char *line = W.buf + line_start;
					if (S.converged) {
						step(cur, line, len, S.common, S.long_common);
					} else {
						for (int e = 0; e < 4; e++) step(st[e], line, len, S.head[e], S.long_head[e]);
						if (st[0] == st[1] && st[1] == st[2] && st[2] == st[3]) {
							S.converged = true;
							cur = st[0];
						}
					}
					line_start = pos + 1;
				}
				for (int e = 0; e < 4; e++) S.exit_head[e] = st[e];
				S.exit_common = cur;
			});
		};
		int state = 0;          // 0 look for header, 1 sequence line, 2/3 skip (FASTQ '+' and quality)
		size_t file_off = 0;
		int cur = 0;
		if (size_ > 0) read_window(win[0], 0, NULL, 0);
		lap(0);
		while (file_off < size_ && !(stop && *stop)) {
			Window &W = win[cur];
			if (!W.ok) return release(), false;
			const size_t end = W.have + W.take;
			// the last newline of this window: what follows it is carried over to the next one
			size_t after_last_nl = 0;
			for (int t = n_threads - 1; t >= 0; t--)
				if (!W.slice[(size_t)t].nl.empty()) { after_last_nl = (size_t)W.slice[(size_t)t].nl.back() + 1; break; }
			const size_t next_off = file_off + W.take;
			const bool more = next_off < size_;
			std::thread ahead;
			if (more) ahead = std::thread([&, next_off, after_last_nl, end]() { read_window(win[cur ^ 1], next_off, W.buf + after_last_nl, end - after_last_nl); });
			for (int t = 0; t < n_threads && !(stop && *stop); t++) {
				const Slice &S = W.slice[(size_t)t];
				const std::vector<ReadRef> &h = S.head[state];
				if (S.long_head[state] || (S.converged && S.long_common)) too_long = true; // (set BEFORE the callback that sees the line)
				if (!h.empty()) cb(h.data(), h.size());
				if (S.converged) {
					if (!S.common.empty() && !(stop && *stop)) cb(S.common.data(), S.common.size());
					state = S.exit_common;
				} else {
					state = S.exit_head[state];
				}
			}
			if (!(stop && *stop) && !more && after_last_nl < end) { // the file's last line has no newline
				const size_t len = end - after_last_nl;
				const ReadRef last{W.buf + after_last_nl, (uint32_t)len};
				switch (state) {
					case 0: if (len && W.buf[after_last_nl] == marker) state = 1; break;
					case 1: if (len >> 32) too_long = true; cb(&last, 1); state = (format == 1) ? 2 : 0; break;
					case 2: state = 3; break;
					default: state = 0; break;
				}
			}
			lap(1);
			if (!(stop && *stop)) end_of_window(); // the records shown so far are copied now: this buffer is read into again two windows on
			lap(2);
			if (ahead.joinable()) ahead.join();
			lap(3);
			file_off = next_off;
			cur ^= 1;
		}
		if (!(stop && *stop) && state == 1) { // a header on the very last line
			const ReadRef none{"", 0};
			cb(&none, 1);
			end_of_window();
		}
		release();
		return true;
	}

private:
	// (2 MiB-aligned, huge pages advised; the content is kept)
	static bool grow(char *&buf, size_t &cap, size_t want)
	{
		if (want <= cap) return true;
		const size_t c = (want + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
		void *p = NULL;
		if (posix_memalign(&p, (size_t)2 << 20, c) != 0) return false;
		(void)madvise(p, c, MADV_HUGEPAGE);
		if (buf) memcpy(p, buf, cap);
		free(buf);
		buf = static_cast<char *>(p);
		cap = c;
		return true;
	}
	size_t size_ = 0;
	int fd_ = -1;
};

#endif
