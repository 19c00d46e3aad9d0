// c2_gz_parallel.h -- ONE gzip member (an ordinary `gzip -6 reads.fastq`) inflated by all host threads.
//
// What it replaces: the reference opens a .gz input with gzip.open(..., 'rt') (CRISPResso2/CRISPRessoCORE.py:1820-1823) -- one zlib stream,
// one thread.  A deflate stream has no index: a block starts at any BIT, its Huffman tables are in its header, and its matches reach up
// to 32 KiB back into text the previous blocks produced.  The scheme here is the two-pass one of pugz / rapidgzip, cut down to what a
// whole-file inflate needs:
//
//   A  the file is cut into segments of ~chunk bytes; each thread looks for the first DYNAMIC block header at or behind its cut
//      (bit by bit: BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a complete code-length code, complete literal/length and distance
//      codes, the block decodes to its end-of-block symbol, and what follows is a plausible header again);
//   B  every segment is decoded from its block to the next segment's block with 16-bit symbols into a 64 Ki-symbol ring: a byte, or
//      "the byte at position i of the 32 KiB in front of this segment" (the ring starts out holding those markers, so a match needs
//      no special case).  Nothing is kept but the number of bytes produced and the last 32 Ki symbols;
//   C  in file order: the 32 KiB window in front of segment k + 1 = segment k's last symbols with the markers looked up in segment
//      k's own window (a few microseconds each); the segments' offsets in the text are the prefix sums of their sizes;
//   D  every segment is decoded again, now as bytes straight into its place in the text, matches that reach in front of the segment
//      reading its window; CRC-32 per segment;
//   E  the CRCs are combined and compared with the member's trailer (CRC32, ISIZE), as gzip.py does for every member.
//
// A decode that does not land EXACTLY on the next segment's block, any invalid code, a second member, trailing bytes, a CRC that
// differs: the function returns false and the caller's serial route (libdeflate / zlib) inflates the file from the start -- so the
// accepted inputs, the text and the errors are that route's; this one can only be faster, never different.
#pragma once
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <functional>
#include <memory>
#include <new>
#include <thread>
#include <vector>
#include <zlib.h>
#include <immintrin.h>

namespace c2gz {

struct Stats {                                                  // of the last call (c2_gz_inflate_parallel reports them)
    uint64_t segments = 0, blocks_found = 0, bytes_out = 0, fell_back = 0;
    double t_find = 0, t_pass1 = 0, t_windows = 0, t_pass2 = 0;
    const char* why = "";
};

// ---- bits, least significant first (RFC 1951 3.1.1) -------------------------------------------------------------------------------
struct Bits {
    const uint8_t* b = nullptr; size_t n = 0, pos = 0;          // pos: the next byte to load
    uint64_t buf = 0; int cnt = 0;                               // cnt bits of buf are valid; cnt < 0: the stream was read past its end
    void init(const uint8_t* base, size_t bytes, uint64_t bit) {
        b = base; n = bytes; pos = (size_t)(bit >> 3); buf = 0; cnt = 0;
        refill();
        const int skip = (int)(bit & 7u);
        buf >>= skip; cnt -= skip;
    }
    inline void refill() {                                       // >= 56 valid bits afterwards, unless the input ends
        if (cnt < 0) return;                                     // (read past the end already: the caller's next check fails)
        if (pos + 8 <= n) {
            uint64_t w;
            memcpy(&w, b + pos, 8);
            buf |= w << cnt;                                     // (cnt <= 63 here: callers refill only when bits were consumed; cnt < 0 is checked by them)
            const int adv = (63 - cnt) >> 3;
            pos += (size_t)adv; cnt += adv * 8;
        } else {
            while (cnt <= 56 && pos < n) { buf |= (uint64_t)b[pos++] << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
    inline void drop(int k) { buf >>= k; cnt -= k; }
    inline uint32_t take(int k) { const uint32_t v = peek(k); drop(k); return v; }
    uint64_t bitpos() const { return (uint64_t)pos * 8u - (uint64_t)(int64_t)cnt; }
};

// ---- Huffman tables ----------------------------------------------------------------------------------------------------------------
// entry: bits 0-4 code length to consume | bits 8-12 extra bits (or: index bits of a subtable) | bits 13-15 kind | bits 16-31 value
enum : uint32_t { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
inline uint32_t mk(uint32_t kind, uint32_t value, uint32_t extra, uint32_t nbits) { return (value << 16) | (kind << 13) | (extra << 8) | nbits; }
inline uint32_t e_kind(uint32_t e) { return (e >> 13) & 7u; }
inline uint32_t e_val(uint32_t e) { return e >> 16; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 31u; }
inline uint32_t e_bits(uint32_t e) { return e & 31u; }

constexpr int LIT_P = 11, DIST_P = 8;
constexpr int LIT_CAP = (1 << LIT_P) + 288 * 16, DIST_CAP = (1 << DIST_P) + 32 * 128;
struct Tables { uint32_t lit[LIT_CAP]; uint32_t dist[DIST_CAP]; };

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t lit_entry(int sym, int len) {
    if (sym < 256) return mk(K_LIT, (uint32_t)sym, 0, (uint32_t)len);
    if (sym == 256) return mk(K_EOB, 0, 0, (uint32_t)len);
    if (sym <= 285) return mk(K_BASE, LEN_BASE[sym - 257], LEN_EXTRA[sym - 257], (uint32_t)len);
    return mk(K_BAD, 0, 0, (uint32_t)len);                      // 286, 287: in the fixed code, never valid in data
}
inline uint32_t dist_entry(int sym, int len) {
    if (sym < 30) return mk(K_BASE, DIST_BASE[sym], DIST_EXTRA[sym], (uint32_t)len);
    return mk(K_BAD, 0, 0, (uint32_t)len);
}

// Kraft sum of a set of code lengths: 0 = complete, > 0 = incomplete, < 0 = over-subscribed.  zlib (inftrees.c) takes a complete set, or
// an incomplete one only when every code is one bit long (a single code) -- and never for the code-length code.
inline int kraft_left(const int* count) {
    int left = 1;
    for (int len = 1; len <= 15; ++len) { left <<= 1; left -= count[len]; if (left < 0) return -1; }
    return left;
}

// canonical code (RFC 1951 3.2.2) -> lookup by the next bits of the stream (P index bits, subtables for longer codes)
inline bool build_table(const uint8_t* lens, int n, int P, uint32_t* tab, int cap, bool is_dist) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    int maxlen = 0, codes = 0;
    for (int len = 1; len <= 15; ++len) if (count[len]) { maxlen = len; codes += count[len]; }
    const int left = kraft_left(count);
    if (left < 0) return false;
    if (left > 0 && !(maxlen == 1 || (is_dist && codes == 0))) return false;
    unsigned next[16];
    {
        unsigned code = 0;
        for (int len = 1; len <= 15; ++len) { code = (code + (unsigned)count[len - 1]) << 1; next[len] = code; }
    }
    const int PN = 1 << P;
    const uint32_t bad = mk(K_BAD, 0, 0, 1);
    for (int i = 0; i < PN; ++i) tab[i] = bad;
    auto rev = [](unsigned code, int len) { unsigned r = 0; for (int k = 0; k < len; ++k) { r = (r << 1) | (code & 1u); code >>= 1; } return r; };
    uint8_t submax[1 << LIT_P];
    bool any_long = maxlen > P;
    if (any_long) memset(submax, 0, (size_t)PN);
    unsigned codes_of[288];
    for (int s = 0; s < n; ++s) {
        const int len = lens[s];
        if (!len) continue;
        const unsigned r = rev(next[len]++, len);
        codes_of[s] = r;
        if (len <= P) {
            const uint32_t e = is_dist ? dist_entry(s, len) : lit_entry(s, len);
            for (unsigned j = r; j < (unsigned)PN; j += 1u << len) tab[j] = e;
        } else {
            const unsigned pre = r & (unsigned)(PN - 1);
            if (submax[pre] < len) submax[pre] = (uint8_t)len;
        }
    }
    if (any_long) {
        int used = PN;
        for (int s = 0; s < n; ++s) {
            const int len = lens[s];
            if (len <= P) continue;
            const unsigned r = codes_of[s], pre = r & (unsigned)(PN - 1);
            const int sb = submax[pre] - P;
            if (e_kind(tab[pre]) != K_SUB) {
                if (used + (1 << sb) > cap) return false;
                tab[pre] = mk(K_SUB, (uint32_t)used, (uint32_t)sb, 0);
                for (int j = 0; j < (1 << sb); ++j) tab[used + j] = bad;
                used += 1 << sb;
            }
            const uint32_t e = is_dist ? dist_entry(s, len) : lit_entry(s, len);
            const unsigned base = e_val(tab[pre]);
            for (unsigned j = r >> P; j < (1u << sb); j += 1u << (len - P)) tab[base + j] = e;
        }
    }
    return true;
}

inline const Tables& fixed_tables() {                            // BTYPE 1 (RFC 1951 3.2.6)
    static const std::unique_ptr<Tables> T = [] {
        std::unique_ptr<Tables> t(new Tables);
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build_table(l, 288, LIT_P, t->lit, LIT_CAP, false);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, DIST_P, t->dist, DIST_CAP, true);
        return t;
    }();
    return *T;
}

// the header of a dynamic block (RFC 1951 3.2.7), the 3 header bits already taken -> the two sets of code lengths.  zlib's checks.
inline bool read_dynamic_lengths(Bits& br, uint8_t* lens /*[320]*/, int& nlen, int& ndist) {
    br.refill();
    nlen = (int)br.take(5) + 257; ndist = (int)br.take(5) + 1;
    const int ncode = (int)br.take(4) + 4;
    if (nlen > 286 || ndist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    br.refill();
    for (int i = 0; i < ncode; ++i) {
        if (i == 14) br.refill();
        cl[order[i]] = (uint8_t)br.take(3);
    }
    if (br.cnt < 0) return false;
    int count[16] = {0};
    for (int i = 0; i < 19; ++i) count[cl[i]]++;
    count[0] = 0;
    if (kraft_left(count) != 0) return false;                    // (inflate_table(CODES, ...): complete, no exception)
    uint8_t pre[128];                                            // 7-bit lookup: (length << 5) | symbol
    {
        unsigned next[8], code = 0;
        for (int len = 1; len <= 7; ++len) { code = (code + (unsigned)count[len - 1]) << 1; next[len] = code; }
        memset(pre, 0, sizeof pre);
        for (int s = 0; s < 19; ++s) {
            const int len = cl[s];
            if (!len) continue;
            unsigned c = next[len]++, r = 0;
            for (int k = 0; k < len; ++k) { r = (r << 1) | (c & 1u); c >>= 1; }
            for (unsigned j = r; j < 128u; j += 1u << len) pre[j] = (uint8_t)((len << 5) | s);
        }
    }
    const int total = nlen + ndist;
    int i = 0;
    while (i < total) {
        br.refill();
        if (br.cnt < 0) return false;
        const uint8_t e = pre[br.peek(7)];
        const int len = e >> 5, sym = e & 31;
        if (!len) return false;
        br.drop(len);
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)br.take(2); }
        else if (sym == 17) rep = 3 + (int)br.take(3);
        else rep = 11 + (int)br.take(7);
        if (i + rep > total) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (br.cnt < 0) return false;
    if (lens[256] == 0) return false;                            // "invalid code -- missing end-of-block"
    return true;
}

inline bool complete_or_single(const uint8_t* lens, int n, bool is_dist) {
    int count[16] = {0}, maxlen = 0, codes = 0;
    for (int i = 0; i < n; ++i) count[lens[i]]++;
    count[0] = 0;
    for (int len = 1; len <= 15; ++len) if (count[len]) { maxlen = len; codes += count[len]; }
    const int left = kraft_left(count);
    return left == 0 || (left > 0 && (maxlen == 1 || (is_dist && codes == 0)));
}

// ---- where the decoded text goes --------------------------------------------------------------------------------------------------
// pass B: a ring of 16-bit symbols, 0..255 = that byte, 0x8000 | i = byte i of the unknown 32 KiB in front of the segment
struct Ring {
    static constexpr int N = 65536, SLACK = 16;
    uint16_t* r;
    uint64_t pos = 0;
    bool known_empty;                                            // the first segment: nothing lies in front of it, a match that reaches there is an error
    explicit Ring(bool first) : r(new uint16_t[N + SLACK]), known_empty(first) {
        memset(r, 0, sizeof(uint16_t) * (N + SLACK));
        for (int i = 0; i < 32768; ++i) r[32768 + i] = (uint16_t)(0x8000 | i);     // positions -32768 .. -1
    }
    ~Ring() { delete[] r; }
    Ring(const Ring&) = delete;
    Ring& operator=(const Ring&) = delete;
    inline bool lit(uint8_t c) { r[pos & 0xffffu] = c; ++pos; return true; }
    inline bool bytes(const uint8_t* p, size_t k) { for (size_t i = 0; i < k; ++i) { r[pos & 0xffffu] = p[i]; ++pos; } return true; }
    inline bool match(int len, int dist) {
        if (known_empty && (uint64_t)dist > pos) return false;
        const size_t d = (size_t)(pos & 0xffffu), s = (size_t)((pos - (uint64_t)dist) & 0xffffu);
        if (d + (size_t)len <= (size_t)N && s + (size_t)len <= (size_t)N) {
            uint16_t* o = r + d; const uint16_t* in = r + s;
            if (dist >= 8) {                                     // eight symbols at a time (reads and writes up to 7 behind the match: the slack, or slots no match can reach any more)
                for (int i = 0; i < len; i += 8) { __m128i v = _mm_loadu_si128((const __m128i*)(in + i)); _mm_storeu_si128((__m128i*)(o + i), v); }
            } else if (dist == 1) {
                const uint16_t v = in[0];
                for (int i = 0; i < len; ++i) o[i] = v;
            } else {
                for (int i = 0; i < len; ++i) o[i] = in[i];
            }
        } else {
            for (int i = 0; i < len; ++i) r[(pos + (uint64_t)i) & 0xffffu] = r[(pos + (uint64_t)i - (uint64_t)dist) & 0xffffu];
        }
        pos += (uint64_t)len;
        return true;
    }
    inline void block_done() {}
    void tail(uint16_t* out /*[32768]*/) const { for (int i = 0; i < 32768; ++i) out[i] = r[(pos - 32768u + (uint64_t)i) & 0xffffu]; }
};

// pass A's trial decode: counts
struct Null {
    uint64_t pos = 0;
    inline bool lit(uint8_t) { ++pos; return true; }
    inline bool bytes(const uint8_t*, size_t k) { pos += k; return true; }
    inline bool match(int len, int) { pos += (uint64_t)len; return true; }
    inline void block_done() {}
};

// pass D: bytes into the segment's place in the text; `window` = the 32 KiB in front of it (the last `have` bytes of it exist)
typedef uint32_t (*crc_fn)(uint32_t, const void*, size_t);
inline uint32_t zlib_crc(uint32_t c, const void* p, size_t len) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t a = 0; a < len; a += (size_t)1 << 30) c = (uint32_t)crc32(c, b + a, (uInt)((len - a) < ((size_t)1 << 30) ? (len - a) : ((size_t)1 << 30)));
    return c;
}
struct Linear {
    uint8_t* base; uint8_t* out; uint8_t* end;
    const uint8_t* window; size_t have;
    crc_fn crc_of = zlib_crc; uint32_t crc = 0; uint8_t* crc_upto = nullptr;      // CRC-32 of base .. crc_upto, taken block by block while the bytes are in cache
    inline void block_done() { if (out > crc_upto) { crc = crc_of(crc, crc_upto, (size_t)(out - crc_upto)); crc_upto = out; } }
    inline bool lit(uint8_t c) { if (out >= end) return false; *out++ = c; return true; }
    inline bool bytes(const uint8_t* p, size_t k) { if ((size_t)(end - out) < k) return false; memcpy(out, p, k); out += k; return true; }
    inline bool match(int len, int dist) {
        if ((size_t)(end - out) < (size_t)len) return false;
        const size_t done = (size_t)(out - base);
        if ((size_t)dist > done) {                               // reaches in front of the segment (only in its first 32 KiB)
            const size_t back = (size_t)dist - done;
            if (back > have) return false;
            for (int i = 0; i < len; ++i) {
                const size_t at = done + (size_t)i;
                out[i] = at < (size_t)dist ? window[32768 - ((size_t)dist - at)] : base[at - (size_t)dist];
            }
            out += len;
            return true;
        }
        const uint8_t* in = out - dist;
        if (dist >= 16 && (size_t)(end - out) >= (size_t)len + 16) {
            for (int i = 0; i < len; i += 16) { __m128i v = _mm_loadu_si128((const __m128i*)(in + i)); _mm_storeu_si128((__m128i*)(out + i), v); }
        } else if (dist == 1) {
            memset(out, in[0], (size_t)len);
        } else if (dist >= 8 && (size_t)(end - out) >= (size_t)len + 8) {
            for (int i = 0; i < len; i += 8) { uint64_t v; memcpy(&v, in + i, 8); memcpy(out + i, &v, 8); }
        } else {
            for (int i = 0; i < len; ++i) out[i] = in[i];
        }
        out += len;
        return true;
    }
};

// ---- blocks ------------------------------------------------------------------------------------------------------------------------
enum { R_ERROR = 0, R_STOP = 1, R_FINAL = 2, R_BLOCKS = 3 };

// the symbols of one Huffman block up to its end-of-block symbol
template <class Sink>
inline bool inflate_symbols(Bits& br, Sink& sink, const uint32_t* lit, const uint32_t* dist, uint64_t max_symbols = ~0ull) {
    for (uint64_t nsym = 0;; ++nsym) {
        br.refill();
        if (br.cnt < 0 || nsym > max_symbols) return false;
        uint32_t e = lit[br.peek(LIT_P)];
        if (e_kind(e) == K_SUB) e = lit[e_val(e) + ((uint32_t)(br.buf >> LIT_P) & ((1u << e_extra(e)) - 1u))];
        br.drop((int)e_bits(e));
        uint32_t k = e_kind(e);
        if (k == K_LIT) {
            // up to two more codes from the same 56 bits (15 each); a length found here has its bits refilled below
            if (!sink.lit((uint8_t)e_val(e))) return false;
            e = lit[br.peek(LIT_P)];
            if (e_kind(e) == K_SUB) e = lit[e_val(e) + ((uint32_t)(br.buf >> LIT_P) & ((1u << e_extra(e)) - 1u))];
            if (e_kind(e) != K_LIT) continue;                     // (not consumed: the next round looks it up again behind a refill)
            br.drop((int)e_bits(e));
            if (!sink.lit((uint8_t)e_val(e))) return false;
            e = lit[br.peek(LIT_P)];
            if (e_kind(e) == K_SUB) e = lit[e_val(e) + ((uint32_t)(br.buf >> LIT_P) & ((1u << e_extra(e)) - 1u))];
            if (e_kind(e) != K_LIT) continue;
            br.drop((int)e_bits(e));
            if (!sink.lit((uint8_t)e_val(e))) return false;
            continue;
        }
        if (k == K_EOB) return br.cnt >= 0;
        if (k != K_BASE) return false;
        const int len = (int)e_val(e) + (int)br.take((int)e_extra(e));          // <= 15 + 5 bits taken so far: 36+ left
        uint32_t d = dist[br.peek(DIST_P)];
        if (e_kind(d) == K_SUB) d = dist[e_val(d) + ((uint32_t)(br.buf >> DIST_P) & ((1u << e_extra(d)) - 1u))];
        br.drop((int)e_bits(d));
        if (e_kind(d) != K_BASE) return false;
        const int dd = (int)e_val(d) + (int)br.take((int)e_extra(d));           // 15 + 13 more: still inside the 56 bits of the refill
        if (br.cnt < 0) return false;
        if (!sink.match(len, dd)) return false;
    }
}

// blocks from the reader's position until the bit `stop` (a block boundary: R_STOP), or through the final block (R_FINAL), or `max_blocks`
template <class Sink>
inline int inflate_blocks(Bits& br, Sink& sink, uint64_t stop, Tables& T, uint64_t max_blocks = ~0ull, uint64_t max_symbols = ~0ull) {
    for (uint64_t nb = 0;; ++nb) {
        const uint64_t at = br.bitpos();
        if (at == stop) return R_STOP;
        if (at > stop) return R_ERROR;
        if (nb >= max_blocks) return R_BLOCKS;
        br.refill();
        if (br.cnt < 3) return R_ERROR;
        const uint32_t h = br.take(3);
        const bool final = h & 1u;
        const uint32_t type = h >> 1;
        if (type == 0) {
            br.drop(br.cnt & 7);                                 // to the byte boundary
            br.refill();
            if (br.cnt < 32) return R_ERROR;
            const uint32_t len = br.take(16), nlen = br.take(16);
            if ((len ^ 0xffffu) != nlen) return R_ERROR;
            // the bytes of the block lie at the reader's byte position: whole bytes are in the buffer, give them back
            const size_t byte_at = (size_t)(br.bitpos() >> 3);
            if (byte_at + len > br.n) return R_ERROR;
            if (!sink.bytes(br.b + byte_at, len)) return R_ERROR;
            br.init(br.b, br.n, ((uint64_t)byte_at + len) * 8u);
        } else if (type == 1) {
            const Tables& F = fixed_tables();
            if (!inflate_symbols(br, sink, F.lit, F.dist, max_symbols)) return R_ERROR;
        } else if (type == 2) {
            uint8_t lens[320];
            int nlen, ndist;
            if (!read_dynamic_lengths(br, lens, nlen, ndist)) return R_ERROR;
            if (!build_table(lens, nlen, LIT_P, T.lit, LIT_CAP, false)) return R_ERROR;
            if (!build_table(lens + nlen, ndist, DIST_P, T.dist, DIST_CAP, true)) return R_ERROR;
            if (!inflate_symbols(br, sink, T.lit, T.dist, max_symbols)) return R_ERROR;
        } else return R_ERROR;
        sink.block_done();
        if (final) return R_FINAL;
    }
}

// ---- pass A: the first dynamic block that starts in bits [from, to) ----------------------------------------------------------------
inline bool plausible_header(const uint8_t* b, size_t n, uint64_t bit) {
    Bits br;
    br.init(b, n, bit);
    if (br.cnt < 3) return false;
    const uint32_t h = br.take(3), type = h >> 1;
    if (type == 3) return false;
    if (type == 1) return true;
    if (type == 0) {
        br.drop(br.cnt & 7);
        br.refill();
        if (br.cnt < 32) return false;
        const uint32_t len = br.take(16), nlen = br.take(16);
        return (len ^ 0xffffu) == nlen;
    }
    uint8_t lens[320];
    int nlen, ndist;
    if (!read_dynamic_lengths(br, lens, nlen, ndist)) return false;
    return complete_or_single(lens, nlen, false) && complete_or_single(lens + nlen, ndist, true);
}

inline bool find_block(const uint8_t* b, size_t n, uint64_t from, uint64_t to, Tables& T, uint64_t& found) {
    const uint64_t last = (uint64_t)n * 8u;
    if (to > last) to = last;
    for (uint64_t p = from; p < to; ++p) {
        const size_t at = (size_t)(p >> 3);
        if (at + 16 > n) return false;
        uint64_t w0, w1;
        memcpy(&w0, b + at, 8); memcpy(&w1, b + at + 8, 8);
        const int sh = (int)(p & 7u);
        const uint64_t w = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;       // 64 bits from p
        if ((w & 7u) != 4u) continue;                            // BFINAL 0, BTYPE 2
        if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
        {   // the code-length code must be complete: 3-bit lengths from bit 17 (the first 15 of them are in w; the rest, if any, are checked below)
            const int ncode = (int)((w >> 13) & 15u) + 4;
            int left = 128, seen = ncode < 15 ? ncode : 15;      // in units of 2^-7
            uint64_t v = w >> 17;
            for (int i = 0; i < seen; ++i) { const int l = (int)(v & 7u); v >>= 3; if (l) left -= 128 >> l; }
            if (left < 0) continue;
            if (ncode <= 15 && left != 0) continue;
        }
        Bits br;
        br.init(b, n, p + 3);
        uint8_t lens[320];
        int nlen, ndist;
        if (!read_dynamic_lengths(br, lens, nlen, ndist)) continue;
        if (!complete_or_single(lens, nlen, false) || !complete_or_single(lens + nlen, ndist, true)) continue;
        // the block decodes to its end, and a header follows
        br.init(b, n, p);
        Null sink;
        const int rc = inflate_blocks(br, sink, ~0ull, T, 1, (uint64_t)1 << 22);
        if (rc == R_ERROR || rc == R_FINAL) continue;
        if (!plausible_header(b, n, br.bitpos())) continue;
        found = p;
        return true;
    }
    return false;
}

// ---- the member's frame (RFC 1952) -------------------------------------------------------------------------------------------------
inline bool member_header(const uint8_t* b, size_t n, size_t& data_at) {
    if (n < 18 || b[0] != 0x1f || b[1] != 0x8b || b[2] != 8) return false;
    const uint8_t flg = b[3];
    if (flg & 0xe0u) return false;
    size_t p = 10;
    if (flg & 4u) { if (p + 2 > n) return false; const size_t x = (size_t)b[p] | ((size_t)b[p + 1] << 8); p += 2 + x; }
    if (flg & 8u) { while (p < n && b[p]) ++p; ++p; }
    if (flg & 16u) { while (p < n && b[p]) ++p; ++p; }
    if (flg & 2u) p += 2;
    if (p + 8 >= n) return false;
    data_at = p;
    return true;
}

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

inline double now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// f(item) for every item, items handed out one at a time.  Nothing escapes (these run behind extern "C" entries: ADVICE r05): a thread that cannot
// be created (a pid / thread limit of the cgroup) is done without -- the threads that did start and the caller share the work --; an exception
// inside f (std::bad_alloc of a worker's tables under memory pressure) ends the hand-out and makes the call return false: the caller takes the
// serial route, as for everything else this file declines.
template <class F>
inline bool on_threads(unsigned threads, size_t items, F&& f) {
    std::atomic<size_t> next(0);
    std::atomic<bool> failed(false);
    auto work = [&] {
        try { for (;;) { const size_t k = next.fetch_add(1); if (k >= items) break; f(k); } }
        catch (...) { failed = true; next.store(items); }
    };
    if (threads > items) threads = (unsigned)items;
    if (threads <= 1) { work(); return !failed; }
    std::vector<std::thread> pool;
    try {
        pool.reserve(threads);
        for (unsigned t = 0; t + 1 < threads; ++t) pool.emplace_back(work);
    } catch (...) {}                                                // (fewer threads than asked for)
    work();
    for (auto& th : pool) th.join();
    return !failed;
}

// What passes A - C leave behind: where every segment starts (bit), where its text goes, and the 32 KiB in front of it.
struct Plan {
    std::vector<uint64_t> seg;                                  // S first bits
    std::vector<uint64_t> off;                                  // S + 1 text offsets (off[S] = the size of the text)
    std::vector<std::unique_ptr<uint8_t[]>> win;                // per segment (not the first): the 32 KiB of text in front of it
    uint32_t want_crc = 0;                                      // the member's trailer
    size_t segments() const { return seg.size(); }
    uint64_t total() const { return off.empty() ? 0 : off.back(); }
};

// passes A - C over one gzip member that is the whole file.  false: not applicable / not sure.
inline bool plan_single_member(const uint8_t* b, size_t n, unsigned threads, size_t chunk, Plan& P, Stats& st)
{
    st = Stats();
    auto no = [&](const char* why) { st.why = why; st.fell_back = 1; return false; };
    size_t data_at = 0;
    if (!member_header(b, n, data_at)) return no("not a plain gzip member header");
    if (chunk < (size_t)32 << 10) chunk = (size_t)32 << 10;
    const size_t n_cuts = (n - data_at) / chunk;
    if (threads < 2 || n_cuts < 2) return no("too small to cut");
    double t0 = now();
    // A: block starts
    std::vector<uint64_t> start(n_cuts, ~0ull);
    start[0] = (uint64_t)data_at * 8u;
    const uint64_t scan_bits = (uint64_t)(chunk < ((size_t)1 << 20) ? chunk : ((size_t)1 << 20)) * 8u;
    if (!on_threads(threads, n_cuts - 1, [&](size_t i) {
        const size_t k = i + 1;
        std::unique_ptr<Tables> T(new Tables);
        const uint64_t from = ((uint64_t)data_at + (uint64_t)k * chunk) * 8u;
        uint64_t f = 0;
        if (find_block(b, n - 8, from, from + scan_bits, *T, f)) start[k] = f;
    })) return no("no memory for the workers' tables");
    std::vector<uint64_t>& seg = P.seg;
    seg.clear();
    for (size_t k = 0; k < n_cuts; ++k) if (start[k] != ~0ull) seg.push_back(start[k]);
    const size_t S = seg.size();
    st.segments = S; st.blocks_found = S - 1;
    st.t_find = now() - t0; t0 = now();
    if (S < 2) return no("no block boundary found");
    // B: sizes and last symbols
    struct Seg { uint64_t produced = 0, end_bit = 0; int rc = R_ERROR; std::unique_ptr<uint16_t[]> tail; };
    std::vector<Seg> segs(S);
    std::atomic<bool> good(true);
    if (!on_threads(threads, S, [&](size_t k) {
        if (!good.load(std::memory_order_relaxed)) return;
        std::unique_ptr<Tables> T(new Tables);
        Ring ring(k == 0);
        Bits br;
        br.init(b, n - 8, seg[k]);
        const uint64_t stop = k + 1 < S ? seg[k + 1] : ~0ull;
        const int rc = inflate_blocks(br, ring, stop, *T);
        segs[k].rc = rc; segs[k].produced = ring.pos; segs[k].end_bit = br.bitpos();
        if (rc != (k + 1 < S ? R_STOP : R_FINAL)) { good = false; return; }
        segs[k].tail.reset(new uint16_t[32768]);
        ring.tail(segs[k].tail.get());
    })) return no("no memory for the workers' tables");
    st.t_pass1 = now() - t0; t0 = now();
    if (!good) return no("a segment did not end on the next segment's block");
    // the trailer follows the final block's last byte, and the file ends behind it
    const size_t trailer_at = (size_t)((segs[S - 1].end_bit + 7u) >> 3);
    if (trailer_at + 8 != n) return no("bytes behind the member");
    // C: offsets and windows
    std::vector<uint64_t>& off = P.off;
    off.assign(S + 1, 0);
    for (size_t k = 0; k < S; ++k) off[k + 1] = off[k] + segs[k].produced;
    const uint64_t total = off[S];
    if ((uint32_t)total != rd32(b + n - 4)) return no("ISIZE differs");
    P.want_crc = rd32(b + n - 8);
    std::vector<std::unique_ptr<uint8_t[]>>& win = P.win;
    win.clear();
    win.resize(S);
    for (size_t k = 1; k < S; ++k) {
        win[k].reset(new (std::nothrow) uint8_t[32768]);
        if (!win[k]) return no("no memory for the segments' windows");
        const uint16_t* t = segs[k - 1].tail.get();
        const uint8_t* prev = win[k - 1].get();
        const uint64_t have_prev = off[k - 1] < 32768u ? off[k - 1] : 32768u;    // bytes that exist in front of segment k - 1
        for (int i = 0; i < 32768; ++i) {
            const uint16_t s = t[i];
            if (s < 256u) { win[k][i] = (uint8_t)s; continue; }
            const unsigned at = s & 0x7fffu;
            // a marker: byte `at` of segment k - 1's window.  In front of the text (the tail of a short segment): never read, 0
            if (!prev || 32768u - at > have_prev) { win[k][i] = 0; if ((uint64_t)(32768 - i) <= off[k]) return no("a match reaches in front of the text"); continue; }
            win[k][i] = prev[at];
        }
        segs[k - 1].tail.reset();
    }
    st.t_windows = now() - t0;
    return true;
}

// pass D for one segment: its text to dst (off[k + 1] - off[k] bytes), its CRC-32.  false: the second pass differs from the first.
inline bool inflate_segment(const uint8_t* b, size_t n, const Plan& P, size_t k, uint8_t* dst, uint32_t& crc_out, crc_fn fast_crc = nullptr)
{
    const size_t S = P.segments();
    std::unique_ptr<Tables> T(new Tables);
    Linear lin;
    lin.base = lin.out = dst; lin.end = dst + (P.off[k + 1] - P.off[k]);
    lin.window = P.win[k].get(); lin.have = (size_t)(P.off[k] < 32768u ? P.off[k] : 32768u);
    lin.crc_upto = lin.base; lin.crc = 0;
    if (fast_crc) lin.crc_of = fast_crc;
    Bits br;
    br.init(b, n - 8, P.seg[k]);
    const uint64_t stop = k + 1 < S ? P.seg[k + 1] : ~0ull;
    const int rc = inflate_blocks(br, lin, stop, *T);
    if (rc != (k + 1 < S ? R_STOP : R_FINAL) || lin.out != lin.end) return false;
    lin.block_done();
    crc_out = lin.crc;
    return true;
}

// pass E: the segments' CRCs against the member's trailer
inline bool crc_matches(const Plan& P, const std::vector<uint32_t>& crc)
{
    const size_t S = P.segments();
    uint32_t c = crc[0];
    for (size_t k = 1; k < S; ++k) c = (uint32_t)crc32_combine(c, crc[k], (z_off_t)(P.off[k + 1] - P.off[k]));
    return c == P.want_crc;
}

// One gzip member that is the whole file -> its text.  `room(total)` returns where `total` bytes may be written (or nullptr: no room).
// false: not applicable / not sure -- nothing may be assumed about the destination; the caller inflates the file its serial way.
inline bool inflate_single_member(const uint8_t* b, size_t n, unsigned threads, size_t chunk, const std::function<uint8_t*(size_t)>& room,
                                  size_t& n_text, Stats& st, crc_fn fast_crc = nullptr)
{
    Plan P;
    if (!plan_single_member(b, n, threads, chunk, P, st)) return false;
    auto no = [&](const char* why) { st.why = why; st.fell_back = 1; return false; };
    const size_t S = P.segments();
    const uint64_t total = P.total();
    double t0 = now();
    uint8_t* text = room((size_t)total);
    if (!text && total) return no("no room for the text");
    std::vector<uint32_t> crc(S, 0);
    std::atomic<bool> good(true);
    if (!on_threads(threads, S, [&](size_t k) {
        if (!good.load(std::memory_order_relaxed)) return;
        if (!inflate_segment(b, n, P, k, text + P.off[k], crc[k], fast_crc)) good = false;
    })) return no("no memory for the workers' tables");
    st.t_pass2 = now() - t0;
    if (!good) return no("the second pass differs from the first");
    if (!crc_matches(P, crc)) return no("CRC-32 differs");
    n_text = (size_t)total;
    st.bytes_out = total;
    return true;
}

}  // namespace c2gz
