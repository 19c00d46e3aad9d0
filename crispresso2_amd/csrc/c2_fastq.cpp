// c2_fastq.cpp -- FASTQ ingest + exact de-duplication on the host.
//
// Replaces the first pass of process_fastq (reference CRISPResso2/CRISPRessoCORE.py:1820-1849): a Python readline loop
// over the (optionally gzip'ed) FASTQ that builds `variantCache[sequence] = number of copies`.  Same semantics:
//   * text mode with universal newlines: a line ends at "\n", "\r\n" or a lone "\r" (open(..., 'rt') / gzip.open(x, 'rt'));
//   * records are taken four lines at a time starting from any non-empty first line; nothing is validated;
//   * the sequence is line 2 with str.strip() applied (ASCII whitespace incl. \x0b \x0c \x1c-\x1f); no upper-casing;
//   * a record cut short by the end of the file still counts (its sequence may be empty);
//   * unique sequences keep first-seen order (Python dict order), counts are exact.
// Output: the unique sequences packed back to back (the byte arena + offsets the align kernels take) and their counts.
// No GPU involved; .gz input (concatenated members too, like Python's gzip module) is inflated by zlib, or -- when the text
// fits in memory -- into one buffer by all threads (BGZF) / libdeflate, and then parsed like a plain file.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sched.h>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <memory>
#include <algorithm>
#include <atomic>
#include <tuple>
#include <dlfcn.h>
#include <immintrin.h>
#include <functional>

#include "crispresso2_amd.h"
#include "c2_gz_parallel.h"

// Byte arena that never value-initialises what it hands out (a std::vector<uint8_t>::resize to the size of all unique reads is a
// serial zero-fill of hundreds of megabytes -- and of their page faults -- right before the threads overwrite every byte).
struct ByteBuf {
    uint8_t* p = nullptr; size_t n = 0, cap = 0; bool mapped = false;   // mapped: an anonymous mapping with huge pages asked for
    static constexpr size_t MAP_FROM = (size_t)8 << 20;                 // (the page faults of a few hundred megabytes of 4 KiB pages cost more than copying the bytes)
    ByteBuf() = default;
    ByteBuf(const ByteBuf&) = delete;
    ByteBuf& operator=(const ByteBuf&) = delete;
    ~ByteBuf() { release(); }
    void release() { if (p) { if (mapped) munmap(p, cap); else free(p); } p = nullptr; n = cap = 0; mapped = false; }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void reserve(size_t want) {
        if (want <= cap) return;
        size_t c = cap ? cap : 4096;
        while (c < want) c += c / 2 + 4096;
        if (c >= MAP_FROM) {
            c = (c + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            void* q = mapped ? mremap(p, cap, c, MREMAP_MAYMOVE) : mmap(nullptr, c, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (q == MAP_FAILED) throw std::bad_alloc();
            madvise(q, c, MADV_HUGEPAGE);
            if (!mapped) { if (n) memcpy(q, p, n); free(p); }
            p = (uint8_t*)q; cap = c; mapped = true;
            return;
        }
        uint8_t* q = (uint8_t*)realloc(p, c);
        if (!q) throw std::bad_alloc();
        p = q; cap = c;
    }
    void resize(size_t want) { reserve(want); n = want; }              // new bytes are NOT initialised
    void swap(ByteBuf& o) { std::swap(p, o.p); std::swap(n, o.n); std::swap(cap, o.cap); std::swap(mapped, o.mapped); }
    void append(const uint8_t* a, size_t len) { if (len) { reserve(n + len); memcpy(p + n, a, len); n += len; } }
};

struct c2_fastq {
    ByteBuf arena;
    std::vector<uint64_t> offsets;     // n_unique + 1
    std::vector<uint32_t> counts;      // n_unique
    uint64_t n_reads = 0;
    uint64_t nonempty_lines = 0;       // what `grep -c .` counts in the parsed text: '\n'-terminated lines with at least one byte (get_n_reads_fastq)
    std::vector<uint8_t> aux;          // paired input: the quality pair "q1 q2[::-1]" of every entry, back to back
    std::vector<uint64_t> aux_offsets; // n_unique + 1 (empty for single-file input)
};

namespace {

thread_local std::string g_fastq_error;

// Helper threads that release large scratch tables off the caller's path.  They are never detached: a process that exits (or
// unloads the library) right after an ingest would otherwise run static destructors while such a thread is still inside
// free() / munmap().  At most one is outstanding -- starting the next one joins the previous -- and the registry's destructor
// (library unload / exit) joins the last.
struct HelperThreads {
    std::mutex m;
    std::vector<std::thread> t;
    template <class F> void run(F&& f) {
        std::lock_guard<std::mutex> lk(m);
        for (auto& x : t) if (x.joinable()) x.join();
        t.clear();
        t.emplace_back(std::forward<F>(f));
    }
    ~HelperThreads() { for (auto& x : t) if (x.joinable()) x.join(); }
};
static HelperThreads g_helpers;


inline bool py_space(uint8_t c) { return (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x20); }

// 64-bit hash of a byte string; quality only matters for table occupancy (equality is always decided by the bytes).  Four
// independent multiply-fold lanes over 32-byte blocks -- one lane is a chain of dependent multiplies, ~5 cycles per 8 bytes; four
// of them keep the multiplier busy (a 250-base read: ~12 ns instead of ~40) -- then the tail word by word.
inline uint64_t hash_bytes(const uint8_t* p, size_t n) {
    const uint64_t K0 = 0xc4ceb9fe1a85ec53ull, K1 = 0xff51afd7ed558ccdull, K2 = 0x9e3779b97f4a7c15ull, K3 = 0xd6e8feb86659fd93ull;
    uint64_t h0 = K2 ^ (uint64_t)n * K1, h1 = K0, h2 = K1, h3 = K3;
    while (n >= 32) {
        uint64_t w0, w1, w2, w3;
        memcpy(&w0, p, 8); memcpy(&w1, p + 8, 8); memcpy(&w2, p + 16, 8); memcpy(&w3, p + 24, 8);
        h0 = (h0 ^ w0) * K0; h0 ^= h0 >> 29;
        h1 = (h1 ^ w1) * K1; h1 ^= h1 >> 31;
        h2 = (h2 ^ w2) * K3; h2 ^= h2 >> 30;
        h3 = (h3 ^ w3) * K2; h3 ^= h3 >> 28;
        p += 32; n -= 32;
    }
    uint64_t h = h0 ^ (h1 * K3) ^ ((h2 << 21) | (h2 >> 43)) ^ (h3 * K0);
    while (n >= 8) {
        uint64_t w; memcpy(&w, p, 8);
        h = (h ^ w) * K0; h ^= h >> 29;
        p += 8; n -= 8;
    }
    uint64_t w = 0;
    if (n) memcpy(&w, p, n);
    h = (h ^ w) * K1; h ^= h >> 32;
    return h;
}

struct Dedup {
    c2_fastq* R;
    std::vector<uint32_t> table;       // open addressing, 0 = empty, else index + 1
    std::vector<uint64_t> hashes;      // per unique sequence
    uint64_t mask = 0;
    explicit Dedup(c2_fastq* r) : R(r) { table.assign(1u << 16, 0); mask = table.size() - 1; R->offsets.push_back(0); }
    void grow() {
        std::vector<uint32_t> t(table.size() * 2, 0);
        const uint64_t m = t.size() - 1;
        for (uint32_t k = 0; k < (uint32_t)hashes.size(); ++k) {
            uint64_t pos = hashes[k] & m;
            while (t[pos]) pos = (pos + 1) & m;
            t[pos] = k + 1;
        }
        table.swap(t); mask = m;
    }
    // index of a sequence already in the table, or -1
    int64_t find(const uint8_t* s, size_t n) const {
        const uint64_t h = hash_bytes(s, n);
        uint64_t pos = h & mask;
        while (table[pos]) {
            const uint32_t k = table[pos] - 1;
            if (hashes[k] == h) {
                const uint64_t o = R->offsets[k];
                if (R->offsets[k + 1] - o == n && (n == 0 || memcmp(R->arena.data() + o, s, n) == 0)) return (int64_t)k;
            }
            pos = (pos + 1) & mask;
        }
        return -1;
    }
    bool add(const uint8_t* s, size_t n, uint32_t copies = 1, bool* fresh = nullptr) {
        const uint64_t h = hash_bytes(s, n);
        uint64_t pos = h & mask;
        if (fresh) *fresh = false;
        while (table[pos]) {
            const uint32_t k = table[pos] - 1;
            if (hashes[k] == h) {
                const uint64_t o = R->offsets[k];
                if (R->offsets[k + 1] - o == n && (n == 0 || memcmp(R->arena.data() + o, s, n) == 0)) { R->counts[k] += copies; return true; }
            }
            pos = (pos + 1) & mask;
        }
        if (hashes.size() >= 0xfffffffeull) return false;
        if (fresh) *fresh = true;
        table[pos] = (uint32_t)hashes.size() + 1;
        hashes.push_back(h);
        R->arena.append(s, n);
        R->offsets.push_back((uint64_t)R->arena.size());
        R->counts.push_back(copies);
        if (hashes.size() * 2 > table.size()) grow();
        return true;
    }
};

// Line splitter with Python's universal-newline rule, fed block by block.
struct Lines {
    Dedup& D;
    std::string cur;                   // the line being assembled (without its terminator)
    bool pending_cr = false;           // last block ended in '\r': a following '\n' belongs to the same terminator
    int line_in_record = 0;            // 0 id, 1 sequence, 2 plus, 3 quality
    bool in_record = false;
    bool ok = true;
    explicit Lines(Dedup& d) : D(d) {}
    // a complete line (terminated, or the unterminated tail of the file when `at_eof`)
    void line(const char* p, size_t n, bool terminated) {
        // readline() returns '' only at EOF: an unterminated empty tail is "no line"
        if (!terminated && n == 0) return;
        if (!in_record) { in_record = true; line_in_record = 0; }     // any non-empty readline() result starts a record
        if (line_in_record == 1) {
            const uint8_t* s = (const uint8_t*)p; size_t len = n;
            while (len && py_space(s[0])) { ++s; --len; }
            while (len && py_space(s[len - 1])) --len;
            if (!D.add(s, len)) ok = false;
            ++D.R->n_reads;
        }
        if (++line_in_record == 4) in_record = false;
    }
    char last_byte = '\n';             // for the `grep -c .` count: a non-newline byte right after a '\n' (or at the start) opens a counted line
    void feed(const char* buf, size_t n) {
        {
            uint64_t ne = 0; char prev = last_byte;
            for (size_t q = 0; q < n; ++q) { const char x = buf[q]; ne += (x != '\n') & (prev == '\n'); prev = x; }
            D.R->nonempty_lines += ne; last_byte = prev;
        }
        size_t i = 0;
        if (pending_cr) { pending_cr = false; if (n && buf[0] == '\n') i = 1; }
        if (memchr(buf + i, '\r', n - i) == nullptr) {
            // fast path (no carriage return in this block): lines end at '\n', found with memchr; the three lines of a
            // record that are not the sequence are skipped without looking at their bytes
            size_t start = i;
            while (start < n) {
                const char* nl = (const char*)memchr(buf + start, '\n', n - start);
                if (!nl) break;
                const size_t end = (size_t)(nl - buf);
                if (cur.empty()) line(buf + start, end - start, true);
                else { cur.append(buf + start, end - start); line(cur.data(), cur.size(), true); cur.clear(); }
                start = end + 1;
            }
            if (start < n) {
                // keep only what a later line() call needs: the bytes of a sequence line, one marker byte otherwise
                if (in_record && line_in_record == 1) cur.append(buf + start, n - start);
                else if (cur.empty()) cur.push_back('x');
            }
            return;
        }
        size_t start = i;
        for (; i < n; ++i) {
            const char c = buf[i];
            if (c != '\n' && c != '\r') continue;
            if (cur.empty()) line(buf + start, i - start, true);
            else { cur.append(buf + start, i - start); line(cur.data(), cur.size(), true); cur.clear(); }
            if (c == '\r') { if (i + 1 < n) { if (buf[i + 1] == '\n') ++i; } else pending_cr = true; }
            start = i + 1;
        }
        if (start < n) cur.append(buf + start, n - start);
    }
    void finish() {
        if (!cur.empty()) { line(cur.data(), cur.size(), false); cur.clear(); }
        // a record whose sequence line never came: the reference's readline() returned '' and ''.strip() was counted
        if (in_record && line_in_record == 1) { if (!D.add((const uint8_t*)"", 0)) ok = false; ++D.R->n_reads; }
    }
};


// ---- plain (not gzip'ed) files: the mapped file is cut into byte ranges parsed by one thread each.  The reference frames
// records by LINE NUMBER (four readline() calls per record from the top of the file, whatever the lines contain), so
// the ranges first count their line terminators; a prefix sum gives every range the number of the first line that starts
// inside it, and line numbers 1 mod 4 are sequence lines.  Per-range tables are merged in file order, which reproduces
// the global first-seen order.
inline bool term_end(const char* b, size_t n, size_t p) {      // does a line terminator END at byte p?
    return b[p] == '\n' || (b[p] == '\r' && (p + 1 >= n || b[p + 1] != '\n'));
}

// The no-carriage-return case of count_terminators, 32 bytes per step (AVX2): newline mask m; a `grep -c .` line starts where a byte
// is not '\n' and its predecessor is.  -> false if a '\r' turned up (the caller then takes the general loop).
__attribute__((target("avx2,popcnt")))
static bool count_newlines_avx2(const char* b, size_t lo, size_t hi, char prev, uint64_t& c_out, uint64_t& ne_out) {
    const __m256i NL = _mm256_set1_epi8('\n'), CR = _mm256_set1_epi8('\r');
    uint64_t c = 0, ne = 0;
    unsigned carry = prev == '\n' ? 1u : 0u, any_cr = 0;
    size_t i = lo;
    for (; i + 32 <= hi; i += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i*)(b + i));
        const unsigned m = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, NL));
        any_cr |= (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, CR));
        c += (uint64_t)__builtin_popcount(m);
        ne += (uint64_t)__builtin_popcount(~m & ((m << 1) | carry));
        carry = m >> 31;
    }
    char pv = carry ? '\n' : 'x';
    for (; i < hi; ++i) { const char x = b[i]; any_cr |= (x == '\r'); c += (x == '\n'); ne += (x != '\n') & (pv == '\n'); pv = x; }
    c_out = c; ne_out = ne;
    return any_cr == 0;
}

// -> line terminators that end in [lo, hi); *nonempty += lines of `grep -c .` that START there ('\n' is its only terminator)
uint64_t count_terminators(const char* b, size_t n, size_t lo, size_t hi, uint64_t* nonempty) {
    uint64_t c = 0, ne = 0;
    char prev = lo ? b[lo - 1] : '\n';
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (have_avx2 && hi > lo) {
        uint64_t c1 = 0, ne1 = 0;
        if (count_newlines_avx2(b, lo, hi, prev, c1, ne1)) { *nonempty += ne1; return c1; }
    }
    if (memchr(b + lo, '\r', hi - lo) == nullptr) {
        // no carriage return in the range: terminators are the '\n' bytes.  Both counts from the bytes themselves (b[i - 1] read from
        // memory, no loop-carried state), so the compiler turns the loop into wide compares + horizontal adds
        size_t i = lo;
        if (i < hi) { const char x = b[i]; c += (x == '\n'); ne += (x != '\n') & (prev == '\n'); ++i; }
        const char* q = b;
        uint64_t c2 = 0, ne2 = 0;
        for (; i < hi; ++i) { c2 += (uint64_t)(q[i] == '\n'); ne2 += (uint64_t)((q[i] != '\n') & (q[i - 1] == '\n')); }
        *nonempty += ne + ne2;
        return c + c2;
    }
    for (size_t i = lo; i < hi; ++i) { const char x = b[i]; c += term_end(b, n, i) ? 1 : 0; ne += (x != '\n') & (prev == '\n'); prev = x; }
    *nonempty += ne;
    return c;
}

struct RangeResult { c2_fastq R; Dedup D; uint64_t n_seq = 0; bool ok = true; RangeResult() : D(&R) {} };

// lines that START in [lo, hi); `line_no` = number of the first of them
void parse_range(const char* b, size_t n, size_t lo, size_t hi, uint64_t line_no, RangeResult* out) {
    Dedup& D = out->D;
    size_t pos = lo;
    if (lo > 0 && !term_end(b, n, lo - 1)) {                    // lo is inside a line that started earlier: skip to its end
        while (pos < n && !term_end(b, n, pos)) ++pos;
        ++pos;
    }
    const bool has_cr = memchr(b + lo, '\r', (hi < n ? hi : n) - lo) != nullptr;
    while (pos < hi && pos < n) {
        size_t end;                                              // first byte of the terminator, or n
        if (!has_cr) {
            const char* nl = (const char*)memchr(b + pos, '\n', n - pos);
            end = nl ? (size_t)(nl - b) : n;
            // a '\r' may still sit beyond hi, inside a line that starts here
            if (end > hi) { const char* cr = (const char*)memchr(b + pos, '\r', end - pos); if (cr) end = (size_t)(cr - b); }
        } else {
            end = pos;
            while (end < n && b[end] != '\n' && b[end] != '\r') ++end;
        }
        if ((line_no & 3) == 1) {
            const uint8_t* s = (const uint8_t*)b + pos; size_t len = end - pos;
            while (len && py_space(s[0])) { ++s; --len; }
            while (len && py_space(s[len - 1])) --len;
            if (!D.add(s, len)) { out->ok = false; return; }
            ++out->n_seq;
        }
        ++line_no;
        if (end >= n) break;
        pos = end + ((b[end] == '\r' && end + 1 < n && b[end + 1] == '\n') ? 2 : 1);
    }
}

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

#include "c2_fastq_stream.h"

// plain files: the ranges read the page cache through a mapping (fault-around maps 16 pages per fault) or through pread() into the
// threads' buffers (no page-table work, but the kernel copies every page).  C2_FASTQ_SOURCE=mmap|pread; measured per round.
bool plain_source_is_mapped() {
    const char* e = getenv("C2_FASTQ_SOURCE");
    if (e && !strcmp(e, "pread")) return false;
    if (e && !strcmp(e, "mmap")) return true;
    return true;
}

size_t stream_range_bytes() {
    if (const char* e = getenv("C2_FASTQ_RANGE_BYTES")) { const long long v = atoll(e); if (v > 0) return (size_t)v; }
    return (size_t)4 << 20;
}

// the finished stream's result -> R (arena moved, not copied)
int stream_into(FastqStream& S, c2_fastq* R) {
    R->counts.assign(S.offsets.size() - 1, 0);
    if (!S.counts_into(R->counts.data())) return C2_E_TOO_LARGE;
    R->arena.swap(S.arena);
    R->offsets.swap(S.offsets);
    R->n_reads = S.n_reads;
    R->nonempty_lines += S.nonempty_lines;
    return 0;
}

// text in memory (inflated .gz, filtered records) -> R, on `threads` threads
int parse_plain_parallel(const char* b, size_t n, c2_fastq* R, unsigned threads) {
    const bool trace = getenv("C2_FASTQ_TRACE") != nullptr;
    const double T0 = now_s();
    FastqStream S;
    S.mem = b; S.n = n;
    if (!S.init(threads, stream_range_bytes())) { g_fastq_error = S.err; return C2_E_INVALID; }
    while (!S.done) if (!S.next()) { g_fastq_error = S.err; return S.overflow ? C2_E_TOO_LARGE : C2_E_INVALID; }
    const int rc = stream_into(S, R);
    if (trace) fprintf(stderr, "c2_fastq: %u threads, %zu bytes in memory, %.3f s\n", threads, n, now_s() - T0);
    return rc;
}


// ---- paired input (process_paired_fastq, CRISPRessoCORE.py:1296-1334): two files read in lockstep ---------------------
// readline() of a text-mode file object with universal newlines, over zlib (gzread passes plain files through)
struct TextReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false, err = false;
    ~TextReader() { if (f) gzclose(f); }
    bool open(const char* path) {
        f = gzopen(path, "rb");
        if (!f) return false;
        gzbuffer(f, 1u << 20);
        buf.resize(1u << 20);
        return true;
    }
    bool fill() {
        if (eof) return false;
        const int g = gzread(f, buf.data(), (unsigned)buf.size());
        if (g <= 0) {
            // a gzip stream that stops before its end-of-stream marker is gzread's Z_BUF_ERROR with a return of 0: Python's
            // gzip module raises EOFError when the reader gets there
            int errnum = Z_OK;
            gzerror(f, &errnum);
            eof = true; err = g < 0 || errnum == Z_BUF_ERROR; pos = end = 0;
            return false;
        }
        pos = 0; end = (size_t)g;
        return true;
    }
    // false = readline() returned '' (end of file); otherwise `line` holds the line without its terminator
    bool readline(std::string& line) {
        line.clear();
        bool any = false;
        for (;;) {
            if (pos == end && !fill()) return any;
            size_t i = pos;
            while (i < end && buf[i] != '\n' && buf[i] != '\r') ++i;
            if (i > pos) { line.append(buf.data() + pos, i - pos); any = true; }
            if (i == end) { pos = end; continue; }
            const char c = buf[i];
            pos = i + 1;
            if (c == '\r') {
                if (pos == end) fill();
                if (pos < end && buf[pos] == '\n') ++pos;
            }
            return true;
        }
    }
};

inline void py_strip(std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && py_space((uint8_t)s[a])) ++a;
    while (b > a && py_space((uint8_t)s[b - 1])) --b;
    if (a || b != s.size()) s = s.substr(a, b - a);
}

// CRISPRessoShared.reverse_complement (CRISPRessoShared.py:399-403): upper-case, then A<->T C<->G, N _ - unchanged; any
// other character is a KeyError there -> false here
inline bool reverse_complement_into(const std::string& s, std::string& out) {
    out.resize(s.size());
    for (size_t k = 0; k < s.size(); ++k) {
        char c = s[s.size() - 1 - k];
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        switch (c) {
            case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break;
            case 'N': case '_': case '-': break;
            default: return false;
        }
        out[k] = c;
    }
    return true;
}

// One pass over the two files with the reference's statements: record = 4 readline() calls per file, the loop ends when
// either file's id line is ''.  visit(key = seq1 + '+' + rc(seq2), quals = qual1 + ' ' + qual2[::-1]) -> false stops.
template <class Visit>
int for_each_pair(const char* path1, const char* path2, uint64_t* n_pairs, Visit visit) {
    TextReader r1, r2;
    if (!r1.open(path1)) { g_fastq_error = std::string("cannot open ") + path1; return C2_E_INVALID; }
    if (!r2.open(path2)) { g_fastq_error = std::string("cannot open ") + path2; return C2_E_INVALID; }
    std::string id1, id2, s1, s2, skip, q1, q2, rc2, key, quals;
    uint64_t n = 0;
    bool h1 = r1.readline(id1), h2 = r2.readline(id2);
    while (h1 && h2) {
        r1.readline(s1); py_strip(s1);
        r1.readline(skip);
        r1.readline(q1); py_strip(q1);
        r2.readline(s2); py_strip(s2);
        if (!reverse_complement_into(s2, rc2)) {
            g_fastq_error = "KeyError: reverse_complement of a read with a character outside ACGTN_- (pair " + std::to_string(n) + ")";
            return C2_E_INVALID;
        }
        r2.readline(skip);
        r2.readline(q2); py_strip(q2);
        key.assign(s1); key.push_back('+'); key.append(rc2);
        quals.assign(q1); quals.push_back(' '); quals.append(q2.rbegin(), q2.rend());
        ++n;
        if (!visit(key, quals)) return C2_E_TOO_LARGE;
        h1 = r1.readline(id1); h2 = r2.readline(id2);
    }
    if (r1.err || r2.err) { g_fastq_error = "read error in the paired FASTQ input"; return C2_E_INVALID; }
    *n_pairs = n;
    return 0;
}

}  // namespace

// ---- gzip'ed files, whole-buffer route ------------------------------------------------------------------------------
// zlib's streaming inflate (~0.7 GB/s of text) is what bounds a .gz run end to end, so when the inflated text fits in
// memory the file is inflated into ONE buffer and handed to the parallel parser of the plain route:
//   * BGZF input (bgzip / htslib: every member names its own compressed size in a 'BC' extra subfield, <= 64 KiB of text
//     each) -- the members are located from the headers alone and inflated by all threads at once;
//   * any other clean sequence of gzip members -- libdeflate's whole-buffer inflate, member after member (the image ships
//     libdeflate.so.0 without headers, so it is bound at run time; absent -> the streaming route below).
// Anything unusual (zero padding, trailing bytes that are no gzip member, a damaged or truncated member, not enough
// memory) returns false and the streaming zlib route handles the file from the start, so the accepted inputs and the
// errors are the streaming route's.  C2_FASTQ_GZ = stream | zlib | auto (tests: force a route).
namespace {

struct Deflate {
    void* lib = nullptr;
    void* (*alloc)() = nullptr;
    void (*release)(void*) = nullptr;
    int (*gunzip)(void*, const void*, size_t, void*, size_t, size_t*, size_t*) = nullptr;   // libdeflate_gzip_decompress_ex
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;                               // libdeflate_crc32 (carry-less multiply: ~10x zlib 1.2.11's)
    Deflate() {
        lib = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        alloc = (void* (*)())dlsym(lib, "libdeflate_alloc_decompressor");
        release = (void (*)(void*))dlsym(lib, "libdeflate_free_decompressor");
        gunzip = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(lib, "libdeflate_gzip_decompress_ex");
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(lib, "libdeflate_crc32");
    }
    bool ok() const { return alloc && release && gunzip; }
};
const Deflate& deflate_lib() { static Deflate d; return d; }

inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

size_t inflate_budget() {                                      // bytes of inflated text the whole-buffer route may hold
    if (const char* e = getenv("C2_FASTQ_INFLATE_MAX")) return (size_t)strtoull(e, nullptr, 10);
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    return (pages > 0 && psz > 0) ? (size_t)pages * (size_t)psz / 2 : (size_t)4 << 30;
}

// one complete gzip member -> exactly n_out bytes (zlib checks the CRC and the length)
bool zlib_gunzip_member(const uint8_t* in, size_t n_in, uint8_t* out, size_t n_out) {
    z_stream s;
    memset(&s, 0, sizeof s);
    if (inflateInit2(&s, 15 + 16) != Z_OK) return false;
    uint8_t dummy = 0;
    s.next_in = const_cast<Bytef*>(in); s.avail_in = (uInt)n_in;
    s.next_out = n_out ? out : &dummy; s.avail_out = (uInt)n_out;
    const int rc = inflate(&s, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && s.avail_in == 0 && s.total_out == n_out;
    inflateEnd(&s);
    return ok;
}

struct BgzfBlock { size_t at, len, out_at; uint32_t out_len; };

bool bgzf_blocks(const uint8_t* b, size_t n, std::vector<BgzfBlock>& blocks, size_t& total) {
    size_t p = 0;
    total = 0;
    while (p < n) {
        if (n - p < 26 || b[p] != 0x1f || b[p + 1] != 0x8b || b[p + 2] != 8 || b[p + 3] != 4) return false;   // FLG = FEXTRA only
        const size_t xlen = (size_t)b[p + 10] | ((size_t)b[p + 11] << 8);
        if (12 + xlen + 8 > n - p) return false;
        long bsize = -1;
        for (size_t q = p + 12, qe = p + 12 + xlen; q + 4 <= qe;) {
            const size_t slen = (size_t)b[q + 2] | ((size_t)b[q + 3] << 8);
            if (b[q] == 'B' && b[q + 1] == 'C' && slen == 2 && q + 6 <= qe) bsize = (long)b[q + 4] | ((long)b[q + 5] << 8);
            q += 4 + slen;
        }
        if (bsize < 0) return false;
        const size_t len = (size_t)bsize + 1;
        if (len < 12 + xlen + 8 || len > n - p) return false;
        const uint32_t isize = le32(b + p + len - 4);
        if (isize > (1u << 16)) return false;                 // a BGZF block holds at most 64 KiB
        blocks.push_back(BgzfBlock{p, len, total, isize});
        total += isize;
        p += len;
    }
    return !blocks.empty();
}

// the inflated text: anonymous pages (transparent huge pages asked for -- first-touch faults of 4 KiB pages cost as much
// as the inflate itself), grown in place by mremap
struct TextBuf {
    char* p = nullptr;
    size_t cap = 0;
    TextBuf() {}
    TextBuf(const TextBuf&) = delete;
    TextBuf& operator=(const TextBuf&) = delete;
    ~TextBuf() { if (p) munmap(p, cap); }
    char* get() const { return p; }
    bool reserve(size_t n) {
        n = (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (n <= cap) return true;
        void* q = p ? mremap(p, cap, n, MREMAP_MAYMOVE) : mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) return false;
        p = (char*)q; cap = n;
        madvise(p, cap, MADV_HUGEPAGE);
        return true;
    }
};

bool inflate_bgzf(const uint8_t* b, size_t n, bool use_libdeflate, TextBuf& text, size_t& n_text, unsigned threads) {
    std::vector<BgzfBlock> blocks;
    size_t total = 0;
    if (!bgzf_blocks(b, n, blocks, total) || total > inflate_budget()) return false;
    if (!text.reserve(total ? total : 1)) return false;
    const Deflate& L = deflate_lib();
    const bool fast = use_libdeflate && L.ok();
    std::atomic<size_t> next(0);
    std::atomic<bool> good(true);
    const size_t CHUNK = 64;                                   // blocks per grab (<= 4 MiB of text)
    auto work = [&] {
        void* d = fast ? L.alloc() : nullptr;
        if (fast && !d) { good = false; return; }
        for (;;) {
            const size_t k0 = next.fetch_add(CHUNK);
            if (k0 >= blocks.size() || !good.load(std::memory_order_relaxed)) break;
            const size_t k1 = std::min(blocks.size(), k0 + CHUNK);
            for (size_t k = k0; k < k1; ++k) {
                const BgzfBlock& B = blocks[k];
                uint8_t* o = (uint8_t*)text.get() + B.out_at;
                bool ok;
                if (fast) {
                    size_t ui = 0, uo = 0;
                    uint8_t dummy = 0;
                    ok = L.gunzip(d, b + B.at, B.len, B.out_len ? o : &dummy, B.out_len, &ui, &uo) == 0 && ui == B.len && uo == B.out_len;
                } else {
                    ok = zlib_gunzip_member(b + B.at, B.len, o, B.out_len);
                }
                if (!ok) { good = false; break; }
            }
        }
        if (d) L.release(d);
    };
    if (threads > blocks.size() / CHUNK + 1) threads = (unsigned)(blocks.size() / CHUNK + 1);
    if (threads <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    n_text = total;
    return good;
}

bool inflate_members(const uint8_t* b, size_t n, TextBuf& text, size_t& n_text) {
    const Deflate& L = deflate_lib();
    if (!L.ok() || n < 18) return false;
    const size_t budget = inflate_budget();
    size_t cap = std::max<size_t>((size_t)le32(b + n - 4), 4 * n) + 4096;     // ISIZE of the last member: exact for a one-member file < 4 GiB
    if (cap > budget) return false;
    void* d = text.reserve(cap) ? L.alloc() : nullptr;
    if (!d) return false;
    cap = text.cap;
    size_t p = 0, o = 0;
    bool ok = true;
    while (p < n) {
        if (n - p < 18 || b[p] != 0x1f || b[p + 1] != 0x8b) { ok = false; break; }
        size_t ui = 0, uo = 0;
        const int rc = L.gunzip(d, b + p, n - p, text.get() + o, cap - o, &ui, &uo);
        if (rc == 3) {                                         // LIBDEFLATE_INSUFFICIENT_SPACE: grow, inflate this member again
            if (cap > budget / 2) { ok = false; break; }
            if (!text.reserve(cap * 2)) { ok = false; break; }
            cap = text.cap;
            continue;
        }
        if (rc != 0 || ui == 0) { ok = false; break; }
        p += ui; o += uo;
    }
    L.release(d);
    n_text = o;
    return ok;
}

// ONE ordinary gzip member (gzip / pigz output, no BGZF index) on all threads: c2_gz_parallel.h.  Only worth it from a few megabytes
// up; C2_GZ_PARALLEL=0 switches it off, C2_GZ_PARALLEL_MIN / C2_GZ_PARALLEL_CHUNK move its thresholds (tests).  false -> the serial routes.
thread_local c2gz::Stats g_gz_stats;
bool inflate_single_parallel(const uint8_t* b, size_t n, TextBuf& text, size_t& n_text, unsigned threads) {
    if (const char* e = getenv("C2_GZ_PARALLEL")) if (!strcmp(e, "0")) return false;
    size_t min_bytes = (size_t)4 << 20, chunk = 0;
    if (const char* e = getenv("C2_GZ_PARALLEL_MIN")) min_bytes = (size_t)strtoull(e, nullptr, 10);
    if (const char* e = getenv("C2_GZ_PARALLEL_CHUNK")) chunk = (size_t)strtoull(e, nullptr, 10);
    if (n < min_bytes || threads < 2) return false;
    if (!chunk) {                                                  // ~4 segments per thread, 256 KiB .. 8 MiB each
        chunk = n / ((size_t)threads * 4u);
        if (chunk < ((size_t)256 << 10)) chunk = (size_t)256 << 10;
        if (chunk > ((size_t)8 << 20)) chunk = (size_t)8 << 20;
    }
    const size_t budget = inflate_budget();
    auto room = [&](size_t total) -> uint8_t* {
        if (total > budget || !text.reserve(total ? total : 1)) return nullptr;
        return (uint8_t*)text.get();
    };
    return c2gz::inflate_single_member(b, n, threads, chunk, room, n_text, g_gz_stats, deflate_lib().crc);
}

// CPUs this process may actually use: the hardware threads, cut to the cgroup's CPU bandwidth quota (cpu.max "<quota> <period>", or
// cgroup v1's cfs files) and to the affinity mask.  Running more threads than the quota does not add throughput -- the group is
// throttled for the rest of every period once the quota is spent -- it only adds stalls of up to a period (100 ms).
unsigned usable_cpus() {
    static unsigned cached = 0;
    if (cached) return cached;
    unsigned n = std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && (unsigned)c < n) n = (unsigned)c; }
    auto read_two = [](const char* path, long long& a, long long& b) {
        FILE* f = fopen(path, "r");
        if (!f) return false;
        char x[64] = {0}, y[64] = {0};
        const int got = fscanf(f, "%63s %63s", x, y);
        fclose(f);
        if (got < 1 || !strcmp(x, "max")) return false;
        a = atoll(x); b = got >= 2 ? atoll(y) : 0;
        return a > 0;
    };
    long long quota = 0, period = 0;
    if (read_two("/sys/fs/cgroup/cpu.max", quota, period) && period > 0) {
        const unsigned q = (unsigned)((quota + period - 1) / period);
        if (q >= 1 && q < n) n = q;
    } else {
        long long q1 = 0, p1 = 0, dummy = 0;
        if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", q1, dummy) && read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", p1, dummy) && p1 > 0) {
            const unsigned q = (unsigned)((q1 + p1 - 1) / p1);
            if (q >= 1 && q < n) n = q;
        }
    }
    cached = n;
    return n;
}

unsigned plain_threads(size_t n) {
    unsigned threads = usable_cpus();
    const unsigned by_size = (unsigned)(n / (4u << 20)) + 1;              // at least 4 MiB per thread
    if (threads > by_size) threads = by_size;
    unsigned cap = 128;                                                    // (memory-bound work: SMT siblings add little)
    if (const char* e = getenv("C2_FASTQ_THREADS")) { threads = (unsigned)atoi(e); cap = 256; }   // tests / measurements: any count on any size
    if (threads > cap) threads = cap;
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = (unsigned)n;
    return threads;
}

// 1 = parsed into R through the whole-buffer route, 0 = not applicable (use the streaming route), < 0 = error code
int fastq_unique_gz_whole(const uint8_t* m, size_t n, c2_fastq* R) {
    const char* route = getenv("C2_FASTQ_GZ");
    if (route && !strcmp(route, "stream")) return 0;
    const bool use_libdeflate = !(route && !strcmp(route, "zlib"));
    const bool trace = getenv("C2_FASTQ_TRACE") != nullptr;
    if (n < 18) return 0;
    const double T0 = now_s();
    TextBuf text;
    size_t n_text = 0;
    unsigned hw = usable_cpus();
    if (hw < 1) hw = 1;
    if (hw > 64) hw = 64;
    const char* how = "bgzf";
    bool ok = inflate_bgzf(m, n, use_libdeflate, text, n_text, hw);
    if (!ok && use_libdeflate) { how = "one member, all threads"; ok = inflate_single_parallel(m, n, text, n_text, hw); }
    if (!ok && use_libdeflate) { how = "libdeflate"; ok = inflate_members(m, n, text, n_text); }
    if (!ok) return 0;
    if (trace) fprintf(stderr, "c2_fastq: gz whole-buffer route (%s), %zu -> %zu bytes in %.3f s\n", how, n, n_text, now_s() - T0);
    if (n_text == 0) { R->offsets.push_back(0); return 1; }
    const int rc = parse_plain_parallel(text.get(), n_text, R, plain_threads(n_text));
    return rc ? rc : 1;
}

// The streaming route: Python's gzip reader (gzip.py, _GzipReader) over the mapped file with zlib's inflate -- members back
// to back; zero bytes after a member are skipped; anything else where a member should start is BadGzipFile("Not a gzipped
// file") there; input that ends inside a member is EOFError there; a wrong CRC / length is BadGzipFile.  All errors here.
// (zlib's own gzread ignores trailing bytes and reports a truncated stream as end of file.)
struct GzMembers {
    const uint8_t* b = nullptr;
    size_t n = 0, pos = 0;
    z_stream zs;
    bool open = false, done = false;
    std::string err;
    ~GzMembers() { if (open) inflateEnd(&zs); }
    bool init(const uint8_t* b_, size_t n_) {
        b = b_; n = n_;
        memset(&zs, 0, sizeof zs);
        open = inflateInit2(&zs, 15 + 16) == Z_OK;
        if (!open) err = "zlib: inflateInit2 failed";
        return open;
    }
    // up to cap bytes of text into out; 0 = end of the data, -1 = error (err says which)
    long read(char* out, size_t cap) {
        size_t got = 0;
        while (got < cap && !done) {
            if (zs.avail_in == 0) {
                const size_t take = std::min<size_t>(n - pos, (size_t)1 << 30);
                if (take == 0) { err = "Compressed file ended before the end-of-stream marker was reached"; return -1; }
                zs.next_in = const_cast<Bytef*>(b + pos); zs.avail_in = (uInt)take; pos += take;
            }
            const size_t room = std::min<size_t>(cap - got, (size_t)1 << 30);
            zs.next_out = (Bytef*)out + got; zs.avail_out = (uInt)room;
            const int rc = inflate(&zs, Z_NO_FLUSH);
            got += room - zs.avail_out;
            if (rc == Z_STREAM_END) {
                size_t p = pos - zs.avail_in;                  // first byte after this member
                while (p < n && b[p] == 0) ++p;
                if (p == n) { done = true; break; }
                if (n - p < 2 || b[p] != 0x1f || b[p + 1] != 0x8b) { err = "Not a gzipped file (bytes after the last member)"; return -1; }
                inflateReset(&zs);
                pos = p; zs.avail_in = 0;
            } else if (rc == Z_BUF_ERROR) {
                if (zs.avail_in != 0 && zs.avail_out != 0) { err = "zlib: no progress"; return -1; }
            } else if (rc != Z_OK) {
                err = zs.msg ? zs.msg : "invalid gzip data";
                return -1;
            }
        }
        return (long)got;
    }
};

// ---- read filter fused into the ingest (filterFastqs.py, called from CRISPRessoCORE.py:3696-3717) ----------------------
// The reference rewrites the FASTQ before it reads it: records are four '\n'-terminated lines in BINARY mode, each with
// bytes.rstrip() applied (space \t \n \r \x0b \x0c), the loop stops at the first empty id line; a read is kept if the mean
// and / or the minimum of its qualities (uint8 arithmetic: byte - 33 wraps) reach the thresholds, and bases whose quality is
// below min_bp_qual_or_N become 'N'.  Here the surviving records are written into a memory buffer with exactly the bytes
// of that intermediate file and the buffer goes to the parallel parser -- no file in between.  The order of the tests and
// the failures are the reference's, per combination of options (filterFastqs.py:128-226): minimum of an empty quality
// line (numpy: zero-size array), sequence and quality lines of different length when bases are masked (numpy: boolean
// index mismatch), and the read-only buffer of run_mBP_mBPN (min_single_bp_quality + min_bp_quality_or_N without
// min_average_read_quality raises for the first read that passes).
inline bool bytes_space(uint8_t c) { return c == ' ' || (c >= 0x09 && c <= 0x0d); }

struct TextSource {                                             // the whole text of a FASTQ in memory: mapped plain file or inflated .gz
    void* mapped = nullptr; size_t mapped_n = 0;
    TextBuf inflated;
    const char* p = nullptr; size_t n = 0;
    ~TextSource() { if (mapped) munmap(mapped, mapped_n); }
    bool open_path(const char* path, std::string& err) {
        const int fd = open(path, O_RDONLY);
        struct stat st;
        if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); err = std::string("cannot open ") + path; return false; }
        mapped_n = (size_t)st.st_size;
        if (mapped_n == 0) { close(fd); p = ""; n = 0; return true; }
        mapped = mmap(nullptr, mapped_n, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (mapped == MAP_FAILED) { mapped = nullptr; err = std::string("cannot map ") + path; return false; }
        const uint8_t* b = (const uint8_t*)mapped;
        if (!(mapped_n >= 2 && b[0] == 0x1f && b[1] == 0x8b)) { p = (const char*)mapped; n = mapped_n; return true; }
        const char* route = getenv("C2_FASTQ_GZ");
        const bool stream_only = route && !strcmp(route, "stream");
        const bool use_libdeflate = !(route && !strcmp(route, "zlib"));
        unsigned hw = usable_cpus();
        if (hw < 1) hw = 1;
        if (hw > 64) hw = 64;
        size_t got = 0;
        bool ok = !stream_only && mapped_n >= 18 && inflate_bgzf(b, mapped_n, use_libdeflate, inflated, got, hw);
        if (!ok && !stream_only && use_libdeflate && mapped_n >= 18) ok = inflate_single_parallel(b, mapped_n, inflated, got, hw);
        if (!ok && !stream_only && use_libdeflate && mapped_n >= 18) ok = inflate_members(b, mapped_n, inflated, got);
        if (!ok) {
            GzMembers gm;
            if (!gm.init(b, mapped_n)) { err = gm.err; return false; }
            got = 0;
            for (;;) {
                if (got + ((size_t)8 << 20) > inflate_budget() || !inflated.reserve(got + ((size_t)8 << 20))) {
                    err = std::string("the text of ") + path + " does not fit the in-memory budget of the fused read filter (C2_FASTQ_INFLATE_MAX)";
                    return false;
                }
                const long g = gm.read(inflated.get() + got, inflated.cap - got);
                if (g < 0) { err = std::string("read error in ") + path + ": " + gm.err; return false; }
                if (g == 0) break;
                got += (size_t)g;
            }
        }
        p = inflated.get() ? inflated.get() : ""; n = got;
        return true;
    }
};

// One '\n'-terminated line range of the text: the records whose id line STARTS in [lo, hi) (line numbers 0 mod 4 from the
// top of the file; a record's other lines may lie beyond hi).  Stops at its first empty id line (`stop_rec`) or failure
// (`err_rec`, `err`); the caller orders these events over the ranges as the serial loop would meet them.
struct FilterRange {
    TextBuf out;                                                 // (anonymous huge pages; no zero-fill pass as a growing std::vector would add)
    size_t len = 0;
    uint64_t nonempty_lines = 0, stop_rec = UINT64_MAX, err_rec = UINT64_MAX;
    std::string err;
};

void filter_fastq_range(const char* b, size_t n, size_t lo, size_t hi, uint64_t first_line, int min_bp, int min_av, int min_bpn, FilterRange& R) {
    // first line start at or after lo
    size_t pos = lo;
    if (lo > 0 && b[lo - 1] != '\n') {
        const char* e = (const char*)memchr(b + lo, '\n', n - lo);
        if (!e) return;                                          // no line starts in this range
        pos = (size_t)(e - b) + 1;
    }
    if (pos >= hi) return;
    uint64_t line = first_line;
    if (!R.out.reserve((hi - lo) + 4096)) { R.err_rec = 0; R.err = "out of memory"; return; }
    struct Line { size_t a, z, next; };
    auto read_line = [&](size_t at, Line& L) {                   // readline().rstrip() at byte `at`
        if (at >= n) { L.a = L.z = L.next = n; return; }
        const char* e = (const char*)memchr(b + at, '\n', n - at);
        size_t end = e ? (size_t)(e - b) : n;
        L.a = at;
        L.next = e ? end + 1 : n;
        while (end > L.a && bytes_space((uint8_t)b[end - 1])) --end;
        L.z = end;
    };
    bool active = true;                                          // false once this range hit its stop / failure: only lines are counted then
    while (pos < n && pos < hi) {
        const char* e = (const char*)memchr(b + pos, '\n', n - pos);
        const size_t end = e ? (size_t)(e - b) : n;
        R.nonempty_lines += end > pos;                           // `grep -c .` of get_n_reads_fastq (CRISPRessoShared.py:743-748)
        if (active && (line & 3) == 0) {
            const uint64_t rec = line >> 2;
            Line id, sq, pl, ql;
            read_line(pos, id);
            if (id.z == id.a) { R.stop_rec = rec; active = false; }
            else {
                read_line(id.next, sq); read_line(sq.next, pl); read_line(pl.next, ql);
                const size_t nq = ql.z - ql.a, ns = sq.z - sq.a;
                const uint8_t* q = (const uint8_t*)b + ql.a;
                const char* failure = nullptr;
                bool keep = true;
                auto min_ok = [&]() {
                    if (nq == 0) { failure = "ValueError: zero-size array to reduction operation minimum which has no identity (empty quality line)"; return false; }
                    unsigned mn = 255;
                    for (size_t k = 0; k < nq; ++k) { const unsigned v = (uint8_t)(q[k] - 33); if (v < mn) mn = v; }
                    return (int)mn >= min_bp;
                };
                auto mean_ok = [&]() {
                    if (nq == 0) return false;                   // numpy: mean of an empty slice is nan, nan >= x is False
                    uint64_t sum = 0;
                    for (size_t k = 0; k < nq; ++k) sum += (uint8_t)(q[k] - 33);
                    return (double)sum / (double)nq >= (double)min_av;
                };
                if (min_bp > 0 && min_av > 0 && min_bpn <= 0) {  // run_mBP_mRQ: mean first
                    keep = mean_ok();
                    if (keep) keep = min_ok();
                } else {
                    if (min_bp > 0) keep = min_ok();
                    if (keep && !failure && min_av > 0) keep = mean_ok();
                }
                if (!failure && keep && min_bpn > 0) {
                    if (min_bp > 0 && min_av <= 0) failure = "ValueError: assignment destination is read-only (run_mBP_mBPN masks a numpy.frombuffer view)";
                    // (numpy accepts a boolean index of size 0 for any array: a record with an EMPTY quality line -- a file cut short
                    // inside a record -- is written unmasked by run_mBPN and the run goes on)
                    else if (nq != 0 && ns != nq) failure = "IndexError: boolean index did not match indexed array (sequence and quality lines differ in length)";
                }
                if (failure) {
                    R.err_rec = rec;
                    R.err = "filterFastqs, record " + std::to_string(rec) + ": " + failure;
                    active = false;
                } else if (keep) {
                    const size_t need = (id.z - id.a) + ns + (pl.z - pl.a) + nq + 4;
                    if (R.len + need > R.out.cap && !R.out.reserve(R.len + need + ((size_t)1 << 20))) {
                        R.err_rec = rec; R.err = "out of memory"; active = false; ++line; pos = end + 1; continue;
                    }
                    char* o = R.out.get() + R.len;
                    R.len += need;
                    memcpy(o, b + id.a, id.z - id.a); o += id.z - id.a; *o++ = '\n';
                    memcpy(o, b + sq.a, ns);
                    if (min_bpn > 0) for (size_t k = 0; k < ns && k < nq; ++k) if ((int)(uint8_t)(q[k] - 33) < min_bpn) o[k] = 'N';
                    o += ns; *o++ = '\n';
                    memcpy(o, b + pl.a, pl.z - pl.a); o += pl.z - pl.a; *o++ = '\n';
                    memcpy(o, b + ql.a, nq); o += nq; *o++ = '\n';
                }
            }
        }
        ++line;
        pos = end + 1;
    }
}

// -> 0 ok; C2_E_INVALID with err set for the reference's failures.  The text is cut into byte ranges filtered by one thread each
// (line numbers from a parallel count of '\n', as in the plain parser); the ranges' outputs are concatenated up to the first
// empty id line, and a failure counts only if the serial loop would have reached it.
int filter_fastq_text(const char* b, size_t n, int min_bp, int min_av, int min_bpn, TextBuf& out, size_t& n_out, uint64_t& nonempty_lines, std::string& err) {
    nonempty_lines = 0;
    n_out = 0;
    unsigned threads = n ? plain_threads(n) : 1;
    if (threads < 1) threads = 1;
    std::vector<size_t> cut(threads + 1);
    for (unsigned t = 0; t <= threads; ++t) cut[t] = (size_t)((unsigned __int128)n * t / threads);
    std::vector<uint64_t> newlines(threads, 0);
    auto run = [&](auto fn) {
        if (threads == 1) { fn(0u); return; }
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(fn, t);
        for (auto& th : pool) th.join();
    };
    run([&](unsigned t) {
        uint64_t c = 0;
        for (const char* p = b + cut[t], *e = b + cut[t + 1]; p < e;) {
            const char* q = (const char*)memchr(p, '\n', (size_t)(e - p));
            if (!q) break;
            ++c; p = q + 1;
        }
        newlines[t] = c;
    });
    std::vector<FilterRange> R(threads);
    std::vector<uint64_t> first_line(threads, 0);                // number of the first line that STARTS at or after cut[t]
    {
        uint64_t before = 0;                                     // '\n' bytes in [0, cut[t])
        for (unsigned t = 0; t < threads; ++t) {
            // the line containing byte cut[t] has number `before`; if it started earlier the first line starting here is the next one
            first_line[t] = (cut[t] == 0 || b[cut[t] - 1] == '\n') ? before : before + 1;
            before += newlines[t];
        }
    }
    run([&](unsigned t) { filter_fastq_range(b, n, cut[t], cut[t + 1], first_line[t], min_bp, min_av, min_bpn, R[t]); });
    size_t total = 0;
    unsigned last = threads;                                     // ranges [0, last) contribute their output
    for (unsigned t = 0; t < threads; ++t) {
        nonempty_lines += R[t].nonempty_lines;
    }
    for (unsigned t = 0; t < threads; ++t) {
        if (R[t].err_rec != UINT64_MAX && R[t].err_rec < R[t].stop_rec) { err = R[t].err; return C2_E_INVALID; }
        total += R[t].len;
        if (R[t].stop_rec != UINT64_MAX) { last = t + 1; break; }
    }
    if (!out.reserve(total + 4096)) { err = "out of memory"; return C2_E_INVALID; }
    for (unsigned t = 0; t < last && t < threads; ++t) {
        if (R[t].len) memcpy(out.get() + n_out, R[t].out.get(), R[t].len);
        n_out += R[t].len;
    }
    return 0;
}

}  // namespace


extern "C" {

const char* c2_fastq_last_error(void) { return g_fastq_error.c_str(); }

int c2_fastq_unique(const char* path, c2_fastq** out) {
    if (!path || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    // gzip'ed input (magic 1f 8b) goes through zlib; anything else is read directly
    bool gz = false;
    {
        FILE* probe = fopen(path, "rb");
        if (!probe) { g_fastq_error = std::string("cannot open ") + path; return C2_E_INVALID; }
        unsigned char magic[2] = {0, 0};
        gz = fread(magic, 1, 2, probe) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        fclose(probe);
    }
    c2_fastq* R = new c2_fastq;
    if (!gz) {
        // plain text: pread() by the stream engine's threads, chunk after chunk (c2_fastq_stream.h)
        FastqStream S;
        S.fd = open(path, O_RDONLY);
        struct stat st;
        if (S.fd < 0 || fstat(S.fd, &st) != 0) { g_fastq_error = std::string("cannot open ") + path; delete R; return C2_E_INVALID; }
        S.n = (size_t)st.st_size;
        int rc = 0;
        struct Unmap { void* p = nullptr; size_t n = 0; ~Unmap() { if (p) munmap(p, n); } } mapped;
        if (S.n > 0 && plain_source_is_mapped()) {
            void* m = mmap(nullptr, S.n, PROT_READ, MAP_PRIVATE, S.fd, 0);
            if (m != MAP_FAILED) { mapped.p = m; mapped.n = S.n; S.mem = (const char*)m; }
        }
        if (S.n > 0) {
            const bool trace = getenv("C2_FASTQ_TRACE") != nullptr;
            const double T0 = now_s();
            const unsigned threads = plain_threads(S.n);
            if (!S.init(threads, stream_range_bytes())) { g_fastq_error = S.err; delete R; return C2_E_INVALID; }
            while (!S.done) if (!S.next()) { g_fastq_error = S.err; delete R; return S.overflow ? C2_E_TOO_LARGE : C2_E_INVALID; }
            rc = stream_into(S, R);
            if (trace) fprintf(stderr, "c2_fastq: %u threads, %zu bytes, %u chunks, %.3f s (load+count %.3f, parse %.3f, grow %.3f, insert %.3f, survivors %.3f, "
                               "copy %.3f, re-point %.3f, serial %.3f)\n", threads, S.n, S.chunk_no / 2, now_s() - T0, S.phase_s[0], S.phase_s[1], S.phase_s[2],
                               S.phase_s[3], S.phase_s[4], S.phase_s[5], S.phase_s[6], S.phase_s[7]);
        } else {
            R->offsets.push_back(0);
        }
        if (rc) { g_fastq_error = "more than 2^32 - 2 copies of one sequence"; delete R; return rc; }
        *out = R;
        return 0;
    }
    // gzip: the file is mapped; whole-buffer route when it applies (BGZF on all threads, or libdeflate) ...
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); g_fastq_error = std::string("cannot open ") + path; delete R; return C2_E_INVALID; }
    const size_t n = (size_t)st.st_size;
    void* mapped = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (mapped == MAP_FAILED) { g_fastq_error = std::string("cannot map ") + path; delete R; return C2_E_INVALID; }
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{mapped, n};
    const uint8_t* m = (const uint8_t*)mapped;
    {
        const int w = fastq_unique_gz_whole(m, n, R);
        if (w < 0) { g_fastq_error = "more than 2^32 - 2 unique sequences"; delete R; return w; }
        if (w == 1) { *out = R; return 0; }
        delete R; R = new c2_fastq;
    }
    // ... else zlib inflates on one thread while this thread splits lines and de-duplicates the previous block (two buffers)
    Dedup D(R);
    Lines L(D);
    GzMembers gm;
    if (!gm.init(m, n)) { g_fastq_error = gm.err; delete R; return C2_E_INVALID; }
    std::vector<char> bufs[2] = {std::vector<char>(8u << 20), std::vector<char>(8u << 20)};
    long got[2] = {0, 0};
    std::mutex mu;
    std::condition_variable cv;
    int filled[2] = {0, 0};                                   // 0 free, 1 holds data (got[] valid; got <= 0 ends the stream)
    std::thread producer([&] {
        for (int k = 0;; k ^= 1) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] == 0; }); }
            const long g = gm.read(bufs[k].data(), bufs[k].size());
            { std::lock_guard<std::mutex> lk(mu); got[k] = g; filled[k] = 1; }
            cv.notify_all();
            if (g <= 0) return;
        }
    });
    long last = 0;
    for (int k = 0;; k ^= 1) {
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return filled[k] == 1; }); }
        last = got[k];
        if (last <= 0) break;
        if (L.ok) L.feed(bufs[k].data(), (size_t)last);
        { std::lock_guard<std::mutex> lk(mu); filled[k] = 0; }
        cv.notify_all();
    }
    producer.join();
    if (last < 0) { g_fastq_error = std::string("read error in ") + path + ": " + gm.err; delete R; return C2_E_INVALID; }
    L.finish();
    if (!L.ok) { g_fastq_error = "more than 2^32 - 2 unique sequences"; delete R; return C2_E_TOO_LARGE; }
    *out = R;
    return 0;
}

int c2_fastq_unique_filtered(const char* path, int32_t min_bp_qual_in_read, int32_t min_av_read_qual, int32_t min_bp_qual_or_N,
                             c2_fastq** out, uint64_t* nonempty_lines_in_input) {
    if (!path || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    if (min_bp_qual_in_read <= 0 && min_av_read_qual <= 0 && min_bp_qual_or_N <= 0) {
        g_fastq_error = "no filter requested (the reference exits with 'Finished -- No modifications requested')";
        return C2_E_INVALID;
    }
    TextSource src;
    std::string err;
    if (!src.open_path(path, err)) { g_fastq_error = err; return C2_E_INVALID; }
    TextBuf filtered;
    size_t n_f = 0;
    uint64_t lines = 0;
    const int rc = filter_fastq_text(src.p, src.n, min_bp_qual_in_read, min_av_read_qual, min_bp_qual_or_N, filtered, n_f, lines, err);
    if (rc) { g_fastq_error = err; return rc; }
    if (nonempty_lines_in_input) *nonempty_lines_in_input = lines;
    c2_fastq* R = new c2_fastq;
    if (n_f == 0) { R->offsets.push_back(0); *out = R; return 0; }
    const int prc = parse_plain_parallel(filtered.get(), n_f, R, plain_threads(n_f));
    if (prc) { g_fastq_error = "more than 2^32 - 2 unique sequences"; delete R; return prc; }
    *out = R;
    return 0;
}

// ---- the same ingest, chunk by chunk: after every c2_fastq_stream_next the unique reads seen so far (arena bytes, offsets, in
// first-seen order) are final, so the caller can hand the new ones to the GPU while the next chunk is parsed
// (pipeline.quantify_fastq).  The arena pointer never moves; the offsets pointer is valid until the next call.
}  // extern "C"
struct c2_fastq_stream {
    FastqStream S;
    TextSource src;                // .gz input / input of the read filter: the whole text in memory
    TextBuf filtered;              // output of the read filter
    uint64_t lines_input = 0;      // non-empty lines of the text in front of the filter
    void* plain_map = nullptr; size_t plain_map_n = 0;       // a plain file read through a mapping
    ~c2_fastq_stream() { if (plain_map) munmap(plain_map, plain_map_n); }
};
extern "C" {

int c2_fastq_stream_open(const char* path, int32_t min_bp_qual_in_read, int32_t min_av_read_qual, int32_t min_bp_qual_or_N, c2_fastq_stream** out) {
    if (!path || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    std::unique_ptr<c2_fastq_stream> H(new c2_fastq_stream);
    const bool filter = min_bp_qual_in_read > 0 || min_av_read_qual > 0 || min_bp_qual_or_N > 0;
    bool gz = false;
    {
        FILE* probe = fopen(path, "rb");
        if (!probe) { g_fastq_error = std::string("cannot open ") + path; return C2_E_INVALID; }
        unsigned char magic[2] = {0, 0};
        gz = fread(magic, 1, 2, probe) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        fclose(probe);
    }
    if (!gz && !filter) {
        H->S.fd = open(path, O_RDONLY);
        struct stat st;
        if (H->S.fd < 0 || fstat(H->S.fd, &st) != 0) { g_fastq_error = std::string("cannot open ") + path; return C2_E_INVALID; }
        H->S.n = (size_t)st.st_size;
        if (H->S.n > 0 && plain_source_is_mapped()) {
            void* m = mmap(nullptr, H->S.n, PROT_READ, MAP_PRIVATE, H->S.fd, 0);
            if (m != MAP_FAILED) { H->plain_map = m; H->plain_map_n = H->S.n; H->S.mem = (const char*)m; }
        }
    } else {
        std::string err;
        if (!H->src.open_path(path, err)) { g_fastq_error = err; return C2_E_INVALID; }
        H->S.mem = H->src.p; H->S.n = H->src.n;
        if (filter) {
            size_t n_f = 0;
            const int rc = filter_fastq_text(H->src.p, H->src.n, min_bp_qual_in_read, min_av_read_qual, min_bp_qual_or_N, H->filtered, n_f, H->lines_input, err);
            if (rc) { g_fastq_error = err; return rc; }
            H->S.mem = n_f ? H->filtered.get() : ""; H->S.n = n_f;
        }
    }
    if (!H->S.init(H->S.n ? plain_threads(H->S.n) : 1, stream_range_bytes())) { g_fastq_error = H->S.err; return C2_E_INVALID; }
    if (H->S.n == 0) H->S.done = true;
    *out = H.release();
    return 0;
}

int c2_fastq_stream_next(c2_fastq_stream* h, uint64_t* n_unique, uint64_t* arena_bytes, int32_t* done) {
    if (!h) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    if (!h->S.done && !h->S.next()) { g_fastq_error = h->S.err; return h->S.overflow ? C2_E_TOO_LARGE : C2_E_INVALID; }
    if (n_unique) *n_unique = (uint64_t)h->S.offsets.size() - 1;
    if (arena_bytes) *arena_bytes = h->S.offsets.back();
    if (done) *done = h->S.done ? 1 : 0;
    return 0;
}

const uint8_t* c2_fastq_stream_arena(const c2_fastq_stream* h) { return h ? h->S.arena.data() : nullptr; }
const uint64_t* c2_fastq_stream_offsets(const c2_fastq_stream* h) { return h ? h->S.offsets.data() : nullptr; }
uint64_t c2_fastq_stream_text_bytes(const c2_fastq_stream* h) { return h ? (uint64_t)h->S.n : 0; }
const uint8_t* c2_fastq_stream_text(const c2_fastq_stream* h) { return h ? (const uint8_t*)h->S.mem : nullptr; }
uint64_t c2_fastq_stream_n_reads(const c2_fastq_stream* h) { return h ? h->S.n_reads : 0; }
uint64_t c2_fastq_stream_nonempty_lines(const c2_fastq_stream* h) { return h ? h->S.nonempty_lines : 0; }
uint64_t c2_fastq_stream_nonempty_lines_input(const c2_fastq_stream* h) { return h ? h->lines_input : 0; }

int c2_fastq_stream_counts(c2_fastq_stream* h, uint32_t* out, uint64_t n) {
    if (!h || (!out && n)) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    if (n != (uint64_t)h->S.offsets.size() - 1) { g_fastq_error = "counts: n differs from the number of unique reads"; return C2_E_INVALID; }
    if (!h->S.counts_into(out)) { g_fastq_error = "more than 2^32 - 1 copies of one sequence"; return C2_E_TOO_LARGE; }
    return 0;
}

int c2_fastq_stream_rc_partners(c2_fastq_stream* h, int64_t* partner, uint64_t n) {
    if (!h || (!partner && n)) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    if (!h->S.done) { g_fastq_error = "rc_partners: the stream is not exhausted yet"; return C2_E_STATE; }
    if (n != (uint64_t)h->S.offsets.size() - 1) { g_fastq_error = "rc_partners: n differs from the number of unique reads"; return C2_E_INVALID; }
    h->S.rc_partners_into(partner);
    return 0;
}

void c2_fastq_stream_close(c2_fastq_stream* h) { delete h; }

// ---- BGZF input, member range by member range (for a caller that uploads the text as it is inflated: fastq_device.py) ----
}  // extern "C"
struct c2_bgzf {
    void* mapped = nullptr; size_t mapped_n = 0;
    std::vector<BgzfBlock> blocks;
    std::vector<uint64_t> text_at;                                   // text offset of every block, + the total
    size_t total = 0;
    // c2_gzseg_open: the "blocks" are the segments of ONE ordinary gzip member (c2_gz_parallel.h's passes A - C are done; c2_bgzf_inflate runs pass D)
    bool one_member = false;
    c2gz::Plan plan;
    std::vector<uint32_t> seg_crc;
    std::vector<uint8_t> seg_done;
    std::mutex crc_lock;
    size_t n_seg_done = 0;
    ~c2_bgzf() { if (mapped) munmap(mapped, mapped_n); }
};
extern "C" {

int c2_bgzf_open(const char* path, c2_bgzf** out) {
    if (!path || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    std::unique_ptr<c2_bgzf> H(new c2_bgzf);
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); g_fastq_error = std::string("cannot open ") + path; return C2_E_INVALID; }
    H->mapped_n = (size_t)st.st_size;
    if (H->mapped_n < 18) { close(fd); g_fastq_error = "not a BGZF file"; return C2_E_INVALID; }
    H->mapped = mmap(nullptr, H->mapped_n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (H->mapped == MAP_FAILED) { H->mapped = nullptr; g_fastq_error = std::string("cannot map ") + path; return C2_E_INVALID; }
    if (!bgzf_blocks((const uint8_t*)H->mapped, H->mapped_n, H->blocks, H->total)) { g_fastq_error = "not a BGZF file"; return C2_E_INVALID; }
    H->text_at.reserve(H->blocks.size() + 1);
    for (const BgzfBlock& B : H->blocks) H->text_at.push_back((uint64_t)B.out_at);
    H->text_at.push_back((uint64_t)H->total);
    *out = H.release();
    return 0;
}
// An ordinary one-member .gz as something c2_bgzf_inflate can fill ranges of: the member is cut into segments at deflate block starts found by
// search, sizes and windows come from a first decode (c2_gz_parallel.h, passes A - C); c2_bgzf_n_blocks / _text_offsets / _inflate / _close then work
// on the segments as they do on BGZF members.  The member's CRC-32 is checked when the last segment has been inflated (c2_bgzf_inflate fails then,
// like gzip.py raises at the end of a member).  C2_E_INVALID "not applicable ...": the file routes' serial inflate is the way.
int c2_gzseg_open(const char* path, int32_t threads, uint64_t chunk_bytes, c2_bgzf** out) {
    if (!path || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    if (const char* e = getenv("C2_GZ_PARALLEL")) if (!strcmp(e, "0")) { g_fastq_error = "c2_gzseg_open: not applicable (C2_GZ_PARALLEL=0)"; return C2_E_INVALID; }
    std::unique_ptr<c2_bgzf> H(new c2_bgzf);
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); g_fastq_error = std::string("cannot open ") + path; return C2_E_INVALID; }
    H->mapped_n = (size_t)st.st_size;
    size_t min_bytes = (size_t)4 << 20;
    if (const char* e = getenv("C2_GZ_PARALLEL_MIN")) min_bytes = (size_t)strtoull(e, nullptr, 10);
    if (H->mapped_n < 18 || H->mapped_n < min_bytes) { close(fd); g_fastq_error = "c2_gzseg_open: not applicable (a small file)"; return C2_E_INVALID; }
    H->mapped = mmap(nullptr, H->mapped_n, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (H->mapped == MAP_FAILED) { H->mapped = nullptr; g_fastq_error = std::string("cannot map ") + path; return C2_E_INVALID; }
    unsigned th = threads > 0 ? (unsigned)threads : usable_cpus();
    if (th > 64) th = 64;
    size_t chunk = (size_t)chunk_bytes;
    if (const char* e = getenv("C2_GZ_PARALLEL_CHUNK")) chunk = (size_t)strtoull(e, nullptr, 10);
    if (!chunk) {                                                  // segments of a few dozen megabytes of text: several per upload buffer
        chunk = H->mapped_n / ((size_t)th * 16u);
        if (chunk < ((size_t)128 << 10)) chunk = (size_t)128 << 10;
        if (chunk > ((size_t)2 << 20)) chunk = (size_t)2 << 20;
    }
    if (!c2gz::plan_single_member((const uint8_t*)H->mapped, H->mapped_n, th, chunk, H->plan, g_gz_stats)) {
        g_fastq_error = std::string("c2_gzseg_open: not applicable (") + g_gz_stats.why + ")";
        return C2_E_INVALID;
    }
    const size_t S = H->plan.segments();
    H->one_member = true;
    H->total = (size_t)H->plan.total();
    H->blocks.resize(S);
    for (size_t k = 0; k < S; ++k) H->blocks[k] = BgzfBlock{0, 0, (size_t)H->plan.off[k], 0};
    H->text_at = H->plan.off;
    H->seg_crc.assign(S, 0); H->seg_done.assign(S, 0);
    *out = H.release();
    return 0;
}
uint64_t c2_bgzf_n_blocks(const c2_bgzf* h) { return h ? (uint64_t)h->blocks.size() : 0; }
const uint64_t* c2_bgzf_text_offsets(const c2_bgzf* h) { return h ? h->text_at.data() : nullptr; }

// blocks [b0, b1) inflated into dst (their text back to back; cap >= text_offsets[b1] - text_offsets[b0]) on up to `threads` threads;
// every member's CRC and length are checked
int c2_bgzf_inflate(c2_bgzf* h, uint64_t b0, uint64_t b1, uint8_t* dst, uint64_t cap, int32_t threads) {
    if (!h || !dst || b0 > b1 || b1 > h->blocks.size()) { g_fastq_error = "bad argument"; return C2_E_INVALID; }
    if (b0 == b1) return 0;
    const uint64_t base = h->text_at[b0];
    if (h->text_at[b1] - base > cap) { g_fastq_error = "c2_bgzf_inflate: destination too small"; return C2_E_INVALID; }
    const uint8_t* b = (const uint8_t*)h->mapped;
    const Deflate& L = deflate_lib();
    if (h->one_member) {                                            // segments of one gzip member: pass D of c2_gz_parallel.h
        std::atomic<bool> ok(true);
        unsigned T1 = threads > 0 ? (unsigned)threads : usable_cpus();
        const bool ran = c2gz::on_threads(T1, (size_t)(b1 - b0), [&](size_t i) {
            const size_t k = (size_t)b0 + i;
            uint32_t crc = 0;
            if (!ok.load(std::memory_order_relaxed)) return;
            if (!c2gz::inflate_segment(b, h->mapped_n, h->plan, k, dst + (h->text_at[k] - base), crc, L.crc)) { ok = false; return; }
            std::lock_guard<std::mutex> g(h->crc_lock);
            h->seg_crc[k] = crc;
            if (!h->seg_done[k]) { h->seg_done[k] = 1; ++h->n_seg_done; }
        });
        if (!ran) { g_fastq_error = "c2_bgzf_inflate: out of memory in a worker"; return C2_E_NOMEM; }
        if (!ok) { g_fastq_error = "c2_bgzf_inflate: invalid gzip data (the second pass differs from the first)"; return C2_E_INVALID; }
        // (the member's CRC-32 can only be checked once EVERY segment has been inflated: a caller that stops early has not verified anything, and on a
        //  failure here the ranges handed out before are to be discarded -- fastq_device.IngestSource does: it raises, nothing of the file is counted)
        bool all_done;
        { std::lock_guard<std::mutex> g(h->crc_lock); all_done = h->n_seg_done == h->blocks.size(); }
        if (all_done && !c2gz::crc_matches(h->plan, h->seg_crc)) {
            g_fastq_error = "c2_bgzf_inflate: CRC check failed";   // (gzip.py: BadGzipFile("CRC check failed ...") at the member's end)
            return C2_E_INVALID;
        }
        return 0;
    }
    const char* route = getenv("C2_FASTQ_GZ");
    const bool fast = L.ok() && !(route && !strcmp(route, "zlib"));
    std::atomic<size_t> next((size_t)b0);
    std::atomic<bool> good(true);
    const size_t CHUNK = 16;
    auto work = [&] {
        void* d = fast ? L.alloc() : nullptr;
        if (fast && !d) { good = false; return; }
        for (;;) {
            const size_t k0 = next.fetch_add(CHUNK);
            if (k0 >= b1 || !good.load(std::memory_order_relaxed)) break;
            const size_t k1 = std::min((size_t)b1, k0 + CHUNK);
            for (size_t k = k0; k < k1; ++k) {
                const BgzfBlock& B = h->blocks[k];
                uint8_t* o = dst + (B.out_at - base);
                bool ok;
                if (fast) {
                    size_t ui = 0, uo = 0;
                    uint8_t dummy = 0;
                    ok = L.gunzip(d, b + B.at, B.len, B.out_len ? o : &dummy, B.out_len, &ui, &uo) == 0 && ui == B.len && uo == B.out_len;
                } else {
                    ok = zlib_gunzip_member(b + B.at, B.len, o, B.out_len);
                }
                if (!ok) { good = false; break; }
            }
        }
        if (d) L.release(d);
    };
    unsigned T = threads > 0 ? (unsigned)threads : usable_cpus();
    if (T > (b1 - b0) / CHUNK + 1) T = (unsigned)((b1 - b0) / CHUNK + 1);
    if (T <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    if (!good) { g_fastq_error = "c2_bgzf_inflate: a damaged member"; return C2_E_INVALID; }
    return 0;
}
void c2_bgzf_close(c2_bgzf* h) { delete h; }

static void gz_stats_out(const c2gz::Stats& st, uint64_t* stats8) {
    stats8[0] = st.segments; stats8[1] = st.blocks_found; stats8[2] = st.bytes_out; stats8[3] = st.fell_back;
    stats8[4] = (uint64_t)(st.t_find * 1e6); stats8[5] = (uint64_t)(st.t_pass1 * 1e6); stats8[6] = (uint64_t)(st.t_windows * 1e6); stats8[7] = (uint64_t)(st.t_pass2 * 1e6);
}
// what the file routes' last attempt on this thread did (all zero: no .gz file opened yet, or the route was not tried)
void c2_gz_parallel_last(uint64_t* stats8) { if (stats8) gz_stats_out(g_gz_stats, stats8); }

// ---- one ordinary gzip member on all threads (c2_gz_parallel.h), as an entry of its own: tests and tools call it with their own segment size ----
int c2_gz_inflate_parallel(const uint8_t* gz, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* n_out, int32_t threads, uint64_t chunk_bytes,
                           uint64_t* stats8) {
    if (!gz || !n_out || (!dst && cap)) { g_fastq_error = "c2_gz_inflate_parallel: null argument"; return C2_E_INVALID; }
    unsigned th = threads > 0 ? (unsigned)threads : usable_cpus();
    if (th > 256) th = 256;
    size_t chunk = (size_t)chunk_bytes;
    if (!chunk) {
        chunk = (size_t)n / ((size_t)th * 4u);
        if (chunk < ((size_t)256 << 10)) chunk = (size_t)256 << 10;
        if (chunk > ((size_t)8 << 20)) chunk = (size_t)8 << 20;
    }
    uint64_t needed = 0;
    bool too_small = false;
    auto room = [&](size_t total) -> uint8_t* {
        needed = total;
        if (total > cap) { too_small = true; return nullptr; }
        return dst ? dst : (uint8_t*)"";
    };
    c2gz::Stats st;
    size_t got = 0;
    const bool ok = c2gz::inflate_single_member(gz, (size_t)n, th, chunk, room, got, st, deflate_lib().crc);
    if (stats8) gz_stats_out(st, stats8);
    if (too_small) { *n_out = needed; g_fastq_error = "c2_gz_inflate_parallel: destination too small"; return C2_E_OVERFLOW; }
    if (!ok) { *n_out = 0; g_fastq_error = std::string("c2_gz_inflate_parallel: not applicable (") + st.why + "): inflate the file serially"; return C2_E_INVALID; }
    *n_out = got;
    return 0;
}

// ---- host-side helpers of the read -> reference bookkeeping that sits between ingest and the kernels ----------------

// Strand plan of get_new_variant_object (CRISPRessoCORE.py:656-687) for every read against one reference:
// found_fw / found_rc = how many of the first n_seeds forward / reverse-complement seeds occur in the read (Python `in`);
// 0 = forward only (found_fw > seed_min and found_rc == 0), 1 = reverse complement only (found_fw == 0 and found_rc >
// seed_min), 2 = both.
int c2_strand_plan(const uint8_t* arena, const uint64_t* offsets, uint64_t n, const char* const* fw_seeds, const char* const* rc_seeds,
                   int32_t n_seeds, int32_t seed_min, uint8_t* out_plan) {
    if (!offsets || !out_plan || n_seeds < 0 || (n_seeds && (!fw_seeds || !rc_seeds))) { g_fastq_error = "bad argument"; return C2_E_INVALID; }
    std::vector<std::string> fw, rc;
    for (int q = 0; q < n_seeds; ++q) { fw.emplace_back(fw_seeds[q]); rc.emplace_back(rc_seeds[q]); }
    unsigned threads = usable_cpus();
    if (threads > 64) threads = 64;
    if (threads < 1 || n < 4096) threads = 1;
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            const void* s = arena + offsets[i];
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
            int found_fw = 0, found_rc = 0;
            for (int q = 0; q < n_seeds; ++q) {
                if (fw[q].empty() || (len >= fw[q].size() && memmem(s, len, fw[q].data(), fw[q].size()))) ++found_fw;
                if (rc[q].empty() || (len >= rc[q].size() && memmem(s, len, rc[q].data(), rc[q].size()))) ++found_rc;
            }
            out_plan[i] = (found_fw > seed_min && found_rc == 0) ? 0 : (found_fw == 0 && found_rc > seed_min) ? 1 : 2;
        }
    };
    if (threads == 1) { work(0, n); return 0; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work, n * t / threads, n * (t + 1) / threads);
    for (auto& th : pool) th.join();
    return 0;
}

// The reverse-complement merge of the aggregation loop (CRISPRessoCORE.py:3970-3975), in variantCache order: for every
// aligned read with a non-zero count, if reverse_complement(read) (upper-cased, ACGTN_- only; anything else is a KeyError
// in the reference and is skipped here) is an aligned read with a non-zero count, that count is added to this read and
// zeroed there -- a read equal to its own reverse complement therefore doubles, as it does in the reference.
int c2_merge_reverse_complements(const uint8_t* arena, const uint64_t* offsets, uint64_t n, const uint8_t* aligned, int64_t* counts) {
    if (!offsets || !aligned || !counts) { g_fastq_error = "bad argument"; return C2_E_INVALID; }
    // 1. hashes of the aligned reads (threads), 2. table hash -> index (serial inserts, cheap), 3. for every read the index of
    // the read that equals its reverse complement, or -1 (threads; the table is read-only by then), 4. the reference's
    // sequential count transfer over that partner array.
    unsigned threads = usable_cpus();
    if (threads > 64) threads = 64;
    if (threads < 1 || n < 8192) threads = 1;
    auto run = [&](auto&& fn) {
        if (threads == 1) { fn((uint64_t)0, n); return; }
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(fn, n * t / threads, n * (t + 1) / threads);
        for (auto& th : pool) th.join();
    };
    std::vector<uint64_t> hs(n, 0);
    run([&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i)
            if (aligned[i]) hs[i] = hash_bytes(arena + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
    });
    uint64_t cap = 1024;
    while (cap < 2 * n + 2) cap <<= 1;
    std::vector<uint64_t> table(cap, 0);                         // index + 1
    const uint64_t mask = cap - 1;
    for (uint64_t i = 0; i < n; ++i) {
        if (!aligned[i]) continue;
        uint64_t pos = hs[i] & mask;
        while (table[pos]) pos = (pos + 1) & mask;
        table[pos] = i + 1;
    }
    std::vector<int64_t> partner(n, -1);
    run([&](uint64_t lo, uint64_t hi) {
        std::vector<uint8_t> rc;
        for (uint64_t i = lo; i < hi; ++i) {
            if (!aligned[i]) continue;
            const uint8_t* s = arena + offsets[i];
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
            rc.resize(len);
            bool ok = true;
            for (size_t k = 0; k < len && ok; ++k) {
                uint8_t c = s[len - 1 - k];
                if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
                switch (c) {
                    case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break;
                    case 'N': case '_': case '-': break;
                    default: ok = false;
                }
                rc[k] = c;
            }
            if (!ok) continue;
            uint64_t pos = hash_bytes(rc.data(), len) & mask;
            while (table[pos]) {
                const uint64_t j = table[pos] - 1;
                if ((size_t)(offsets[j + 1] - offsets[j]) == len && (len == 0 || memcmp(arena + offsets[j], rc.data(), len) == 0)) { partner[i] = (int64_t)j; break; }
                pos = (pos + 1) & mask;
            }
        }
    });
    for (uint64_t i = 0; i < n; ++i) {
        if (!aligned[i] || counts[i] == 0 || partner[i] < 0) continue;
        const int64_t j = partner[i];
        if (counts[j] > 0) { const int64_t c = counts[i] + counts[j]; counts[j] = 0; counts[i] = c; }
    }
    return 0;
}

// The same merge in two steps, so that the expensive one -- which read equals the reverse complement of which -- does not wait for
// the alignments (a host thread runs it while the device aligns): c2_rc_partners looks the partner up among ALL reads, and
// c2_merge_counts_with_partners applies the reference's sequential transfer, skipping partners that are not aligned (the
// reference's cache holds the aligned reads only).  Same result as c2_merge_reverse_complements.
int c2_rc_partners(const uint8_t* arena, const uint64_t* offsets, uint64_t n, int64_t* partner) {
    if (!offsets || !partner) { g_fastq_error = "bad argument"; return C2_E_INVALID; }
    unsigned threads = usable_cpus();
    if (const char* e = getenv("C2_HOST_THREADS")) threads = (unsigned)atoi(e);
    if (threads > 64) threads = 64;
    if (threads < 1 || n < 8192) threads = 1;
    auto run = [&](auto&& fn) {
        if (threads == 1) { fn((uint64_t)0, n); return; }
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back(fn, n * t / threads, n * (t + 1) / threads);
        for (auto& th : pool) th.join();
    };
    std::vector<uint64_t> hs(n, 0);
    run([&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) hs[i] = hash_bytes(arena + offsets[i], (size_t)(offsets[i + 1] - offsets[i]));
    });
    uint64_t cap = 1024;
    while (cap < 2 * n + 2) cap <<= 1;
    const uint64_t mask = cap - 1;
    // lock-free parallel inserts: a slot is claimed with a compare-and-swap (the reads are distinct, so nothing is ever updated)
    std::unique_ptr<std::atomic<uint64_t>[]> table(new std::atomic<uint64_t>[cap]);
    run([&](uint64_t lo, uint64_t hi) {
        const uint64_t a = cap * lo / (n ? n : 1), b = (hi == n) ? cap : cap * hi / (n ? n : 1);
        for (uint64_t q = a; q < b; ++q) table[q].store(0, std::memory_order_relaxed);
    });
    run([&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            uint64_t pos = hs[i] & mask;
            for (;;) {
                uint64_t seen = table[pos].load(std::memory_order_relaxed);
                if (seen == 0 && table[pos].compare_exchange_strong(seen, i + 1, std::memory_order_relaxed)) break;
                pos = (pos + 1) & mask;
            }
        }
    });
    run([&](uint64_t lo, uint64_t hi) {
        std::vector<uint8_t> rc;
        for (uint64_t i = lo; i < hi; ++i) {
            partner[i] = -1;
            const uint8_t* s = arena + offsets[i];
            const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
            rc.resize(len);
            bool ok = true;
            for (size_t k = 0; k < len && ok; ++k) {
                uint8_t c = s[len - 1 - k];
                if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
                switch (c) {
                    case 'A': c = 'T'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; case 'T': c = 'A'; break;
                    case 'N': case '_': case '-': break;
                    default: ok = false;
                }
                rc[k] = c;
            }
            if (!ok) continue;
            uint64_t pos = hash_bytes(rc.data(), len) & mask;
            for (;;) {
                const uint64_t e = table[pos].load(std::memory_order_relaxed);
                if (!e) break;
                const uint64_t j = e - 1;
                if ((size_t)(offsets[j + 1] - offsets[j]) == len && (len == 0 || memcmp(arena + offsets[j], rc.data(), len) == 0)) { partner[i] = (int64_t)j; break; }
                pos = (pos + 1) & mask;
            }
        }
    });
    return 0;
}

// The reads idx[0..m) of an arena, packed back to back (a read may repeat): out_offsets[m + 1], out_arena sized by the caller from
// the lengths.  The count route uses it for the (read, reference) pairs that are aligned on both strands (CRISPRessoCORE.py:675-687);
// numpy's fancy indexing needs an 8-byte index per BYTE for the same thing.
int c2_gather_reads(const uint8_t* arena, const uint64_t* offsets, const int64_t* idx, uint64_t m, uint8_t* out_arena, uint64_t* out_offsets) {
    if (!offsets || !idx || !out_offsets || (m && !out_arena && false)) { g_fastq_error = "bad argument"; return C2_E_INVALID; }
    out_offsets[0] = 0;
    for (uint64_t k = 0; k < m; ++k) out_offsets[k + 1] = out_offsets[k] + (offsets[idx[k] + 1] - offsets[idx[k]]);
    if (!out_arena) return 0;                                        // (first call: sizes only)
    unsigned threads = usable_cpus();
    if (threads > 32) threads = 32;
    if (threads < 1 || m < 8192) threads = 1;
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t k = lo; k < hi; ++k) {
            const uint64_t a = offsets[idx[k]], len = offsets[idx[k] + 1] - a;
            if (len) memcpy(out_arena + out_offsets[k], arena + a, (size_t)len);
        }
    };
    if (threads == 1) { work(0, m); return 0; }
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(work, m * t / threads, m * (t + 1) / threads);
    for (auto& th : pool) th.join();
    return 0;
}

int c2_merge_counts_with_partners(uint64_t n, const uint8_t* aligned, const int64_t* partner, int64_t* counts) {
    if (!aligned || !partner || !counts) { g_fastq_error = "bad argument"; return C2_E_INVALID; }
    for (uint64_t i = 0; i < n; ++i) {
        if (!aligned[i] || counts[i] == 0 || partner[i] < 0) continue;
        const int64_t j = partner[i];
        if ((uint64_t)j >= n) { g_fastq_error = "partner index out of range"; return C2_E_INVALID; }
        if (aligned[j] && counts[j] > 0) { const int64_t c = counts[i] + counts[j]; counts[j] = 0; counts[i] = c; }
    }
    return 0;
}

// ---- paired input -----------------------------------------------------------------------------------------------------
// First pass of process_paired_fastq's n_processes > 1 route (CRISPRessoCORE.py:1296-1334): variantCache[seq1 + '+' +
// reverse_complement(seq2)] = [copies, qual1 + ' ' + qual2[::-1] of the FIRST occurrence], keys in first-seen order.
int c2_fastq_unique_paired(const char* path1, const char* path2, c2_fastq** out) {
    if (!path1 || !path2 || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    std::unique_ptr<c2_fastq> R(new c2_fastq);
    Dedup D(R.get());
    R->aux_offsets.push_back(0);
    uint64_t n = 0;
    const int rc = for_each_pair(path1, path2, &n, [&](const std::string& key, const std::string& quals) {
        bool fresh = false;
        if (!D.add((const uint8_t*)key.data(), key.size(), 1, &fresh)) return false;
        if (fresh) {
            R->aux.insert(R->aux.end(), quals.begin(), quals.end());
            R->aux_offsets.push_back((uint64_t)R->aux.size());
        }
        return true;
    });
    if (rc == C2_E_TOO_LARGE) g_fastq_error = "more than 2^32 - 2 unique read pairs";
    if (rc) return rc;
    R->n_reads = n;
    *out = R.release();
    return 0;
}

// Second pass (CRISPRessoCORE.py:1452-1513): every occurrence, in file order, of the pairs whose key is selected
// (selected[k] != 0 for key k of `uniq`): out->counts[j] = k, entry j of out's aux arena = that occurrence's own quality pair.
// (out's sequence arena stays empty; the keys are in `uniq`.)
int c2_fastq_paired_occurrences(const char* path1, const char* path2, const c2_fastq* uniq, const uint8_t* selected, c2_fastq** out) {
    if (!path1 || !path2 || !uniq || !selected || !out) { g_fastq_error = "NULL argument"; return C2_E_INVALID; }
    *out = nullptr;
    c2_fastq T;                                               // a table over the keys of `uniq`
    Dedup D(&T);
    const uint64_t nu = (uint64_t)uniq->counts.size();
    for (uint64_t k = 0; k < nu; ++k)
        if (!D.add(uniq->arena.data() + uniq->offsets[k], (size_t)(uniq->offsets[k + 1] - uniq->offsets[k]))) { g_fastq_error = "too many keys"; return C2_E_TOO_LARGE; }
    std::unique_ptr<c2_fastq> R(new c2_fastq);
    R->offsets.push_back(0);
    R->aux_offsets.push_back(0);
    uint64_t n = 0;
    const int rc = for_each_pair(path1, path2, &n, [&](const std::string& key, const std::string& quals) {
        const int64_t k = D.find((const uint8_t*)key.data(), key.size());
        if (k >= 0 && selected[k]) {
            R->counts.push_back((uint32_t)k);
            R->offsets.push_back(0);
            R->aux.insert(R->aux.end(), quals.begin(), quals.end());
            R->aux_offsets.push_back((uint64_t)R->aux.size());
        }
        return true;
    });
    if (rc) return rc;
    R->n_reads = n;
    *out = R.release();
    return 0;
}

uint64_t c2_fastq_aux_bytes(const c2_fastq* r) { return r ? (uint64_t)r->aux.size() : 0; }
const uint8_t* c2_fastq_aux(const c2_fastq* r) { return r ? r->aux.data() : nullptr; }
const uint64_t* c2_fastq_aux_offsets(const c2_fastq* r) { return (r && !r->aux_offsets.empty()) ? r->aux_offsets.data() : nullptr; }

uint64_t c2_fastq_n_unique(const c2_fastq* r) { return r ? (uint64_t)r->counts.size() : 0; }
uint64_t c2_fastq_n_reads(const c2_fastq* r) { return r ? r->n_reads : 0; }
uint64_t c2_fastq_nonempty_lines(const c2_fastq* r) { return r ? r->nonempty_lines : 0; }
uint64_t c2_fastq_arena_bytes(const c2_fastq* r) { return r ? (uint64_t)r->arena.size() : 0; }
const uint8_t* c2_fastq_arena(const c2_fastq* r) { return r ? r->arena.data() : nullptr; }
const uint64_t* c2_fastq_offsets(const c2_fastq* r) { return r ? r->offsets.data() : nullptr; }
const uint32_t* c2_fastq_counts(const c2_fastq* r) { return r ? r->counts.data() : nullptr; }
void c2_fastq_free(c2_fastq* r) { delete r; }

}  // extern "C"
