// c2_alleles_host.h -- host side of the allele table (c2_allele_table_* of include/crispresso2_amd.h), written once over a small backend
// interface: c2_api_alleles.hip instantiates it with HIP (hipMalloc, streams, hipLaunchKernelGGL, rocPRIM's merge sort and scan); the
// test-only wave emulator (tests/emu) with host memory, its fiber launcher and std::stable_sort -- so the chunking, the writer threads,
// the %Reads text and the around-cut sums are the same code on both.
//
// A backend B provides:
//   void* dalloc(size_t); void dfree(void*); void* halloc(size_t) (page-locked); void hfree(void*);
//   bool h2d(void* d, const void* h, size_t); bool d2h(void* h, const void* d, size_t) (both complete on return); bool zero(void* d, size_t); bool sync();
//   bool jobs(const c2_allele_jobs_args&); bool iota(uint32_t* d_out, uint64_t n); bool reads(rows, order, m, out); bool probe(const c2_allele_probe_args&);
//   bool lengths(const c2_allele_text_args&); bool emit(const c2_allele_text_args&); bool fetch(const c2_allele_fetch_args&);
//   bool window(const c2_allele_window_args&); bool group(const c2_allele_group_args&);            (enqueue one kernel)
//   bool scan(const uint32_t* d_in, uint64_t* d_out, uint64_t n);    exclusive sum of n values (uint32 -> uint64)
//   bool sort_rows(c2_allele_row_less, uint32_t* d_keys_in, uint32_t* d_keys_out, uint64_t n); bool sort_keys(c2_allele_key_less, ...);
//   std::string error;
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <charconv>
#include <memory>
#include <string>
#include <vector>
#include <thread>
#include <zlib.h>
#include <time.h>
#include <mutex>
#include <condition_variable>
#include <algorithm>

// repr() of a Python float: the shortest digit string that round-trips (std::to_chars without a precision gives exactly that), laid out by
// CPython's format_float_short for 'r': exponent form when the decimal point would sit before 1e-4 or from 1e16 on, else positional with
// ".0" added to an integer value.
inline int c2_py_float_repr(const double v, char* out) {
    if (v != v) { memcpy(out, "nan", 3); return 3; }
    if (v == 1.0 / 0.0) { memcpy(out, "inf", 3); return 3; }
    if (v == -1.0 / 0.0) { memcpy(out, "-inf", 4); return 4; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);
    // [-]d[.ddd]e[+-]XX
    const char* p = sci;
    int n = 0;
    if (*p == '-') { out[n++] = '-'; ++p; }
    char digits[24]; int nd = 0;
    while (p < r.ptr && *p != 'e') { if (*p != '.') digits[nd++] = *p; ++p; }
    ++p;                                                                      // 'e'
    int ex = 0; bool neg = false;
    if (*p == '-') { neg = true; ++p; } else if (*p == '+') ++p;
    while (p < r.ptr) ex = ex * 10 + (*p++ - '0');
    if (neg) ex = -ex;
    const int decpt = ex + 1;                                                 // digits * 10^(decpt - nd): position of the decimal point
    if (decpt <= -4 || decpt > 16) {
        out[n++] = digits[0];
        if (nd > 1) { out[n++] = '.'; memcpy(out + n, digits + 1, nd - 1); n += nd - 1; }
        out[n++] = 'e';
        int e10 = decpt - 1;
        out[n++] = e10 < 0 ? '-' : '+';
        if (e10 < 0) e10 = -e10;
        char eb[8]; int ne = 0;
        do { eb[ne++] = (char)('0' + e10 % 10); e10 /= 10; } while (e10);
        if (ne < 2) eb[ne++] = '0';
        while (ne) out[n++] = eb[--ne];
        return n;
    }
    if (decpt <= 0) {
        out[n++] = '0'; out[n++] = '.';
        for (int k = 0; k < -decpt; ++k) out[n++] = '0';
        memcpy(out + n, digits, nd); n += nd;
        return n;
    }
    if (decpt >= nd) {
        memcpy(out + n, digits, nd); n += nd;
        for (int k = nd; k < decpt; ++k) out[n++] = '0';
        out[n++] = '.'; out[n++] = '0';
        return n;
    }
    memcpy(out + n, digits, decpt); n += decpt;
    out[n++] = '.';
    memcpy(out + n, digits + decpt, nd - decpt); n += nd - decpt;
    return n;
}

// pwrite of [buf, buf + n) at file offset off, cut into slices for `threads` threads (tmpfs / page-cache writes are memcpy + page
// allocation: they scale with threads)
inline bool c2_pwrite_parallel(const int fd, const uint8_t* buf, const size_t n, const uint64_t off, int threads, std::string& err) {
    if (threads < 1) threads = 1;
    const size_t slice = std::max<size_t>(((n + (size_t)threads - 1) / (size_t)threads + 0xfffff) & ~(size_t)0xfffff, (size_t)1 << 20);
    std::vector<std::thread> pool;
    std::mutex mu;
    bool ok = true;
    auto work = [&](const size_t a, const size_t z) {
        size_t p = a;
        while (p < z) {
            const ssize_t w = pwrite(fd, buf + p, z - p, (off_t)(off + p));
            if (w < 0) { if (errno == EINTR) continue; std::lock_guard<std::mutex> lk(mu); ok = false; err = std::string("pwrite: ") + strerror(errno); return; }
            p += (size_t)w;
        }
    };
    for (size_t a = slice; a < n; a += slice) pool.emplace_back(work, a, std::min(n, a + slice));
    work(0, std::min(n, slice));
    for (auto& t : pool) t.join();
    return ok;
}

// ---- Alleles_frequency_table.zip (CRISPRessoCORE.py:4531: `zipfile.ZipFile(.., 'w', ZIP_DEFLATED, allowZip64=True).write(txt)`, then the .txt is
// removed): ONE deflate stream made by all threads.  A slice of the text is compressed by its own thread from a fresh window and ended with a sync
// flush -- an empty stored block, which byte-aligns it -- so the slices' outputs concatenate into one valid stream (the pigz scheme); the stream
// is closed by an empty final block.  CRC-32 per slice, combined.  The container is written by hand: local header (patched when the sizes
// are known), the data, central directory, end record -- zip64 records when the text is 4 GiB or more.
struct c2_zip_out {
    int fd = -1; uint64_t pos = 0, data_start = 0, usize = 0, csize = 0; uint32_t crc = 0; bool zip64 = false; int level = 1;
    std::string name; uint16_t dos_time = 0, dos_date = 0;
};
inline void c2_le16(std::string& o, const uint32_t v) { o.push_back((char)(v & 0xff)); o.push_back((char)((v >> 8) & 0xff)); }
inline void c2_le32(std::string& o, const uint32_t v) { c2_le16(o, v & 0xffffu); c2_le16(o, v >> 16); }
inline void c2_le64(std::string& o, const uint64_t v) { c2_le32(o, (uint32_t)(v & 0xffffffffull)); c2_le32(o, (uint32_t)(v >> 32)); }
inline std::string c2_zip_local_header(const c2_zip_out& Z) {
    std::string h;
    c2_le32(h, 0x04034b50u); c2_le16(h, Z.zip64 ? 45 : 20); c2_le16(h, 0); c2_le16(h, 8); c2_le16(h, Z.dos_time); c2_le16(h, Z.dos_date);
    c2_le32(h, Z.crc); c2_le32(h, Z.zip64 ? 0xffffffffu : (uint32_t)Z.csize); c2_le32(h, Z.zip64 ? 0xffffffffu : (uint32_t)Z.usize);
    c2_le16(h, (uint32_t)Z.name.size()); c2_le16(h, Z.zip64 ? 20 : 0);
    h += Z.name;
    if (Z.zip64) { c2_le16(h, 1); c2_le16(h, 16); c2_le64(h, Z.usize); c2_le64(h, Z.csize); }
    return h;
}
inline bool c2_zip_begin(c2_zip_out& Z, const int fd, const char* member, const uint64_t text_bytes, const int level, std::string& err) {
    Z.fd = fd; Z.name = member; Z.level = level < 0 ? 1 : (level > 9 ? 9 : level);
    // zip64 records when the text OR what deflate can make of it at worst may reach 4 GiB (stored blocks: 5 bytes per 16 KiB + a sync flush per slice --
    // a table within 1/3000 of 4 GiB whose text does not compress would otherwise overflow the 32-bit compressed size; c2_zip_end checks again).
    // The knob: the zip64 records on a small table, for the tests.
    Z.zip64 = text_bytes + text_bytes / 3000u + (1u << 20) >= 0xffffffffull || getenv("C2_ZIP_FORCE_ZIP64");
    time_t now = time(nullptr); struct tm tmv; localtime_r(&now, &tmv);
    const int yr = tmv.tm_year + 1900 < 1980 ? 1980 : tmv.tm_year + 1900;
    Z.dos_time = (uint16_t)((tmv.tm_hour << 11) | (tmv.tm_min << 5) | (tmv.tm_sec >> 1));
    Z.dos_date = (uint16_t)(((yr - 1980) << 9) | ((tmv.tm_mon + 1) << 5) | tmv.tm_mday);
    const std::string h = c2_zip_local_header(Z);                   // (a placeholder of the final size)
    if (!c2_pwrite_parallel(fd, (const uint8_t*)h.data(), h.size(), 0, 1, err)) return false;
    Z.pos = Z.data_start = h.size();
    return true;
}
// the text [buf, buf + n) joins the member: `threads` slices compressed side by side, written in order
inline bool c2_zip_append(c2_zip_out& Z, const uint8_t* buf, const size_t n, int threads, std::string& err) {
    if (n == 0) return true;
    if (threads < 1) threads = 1;
    const size_t slice = std::max<size_t>((n + (size_t)threads - 1) / (size_t)threads, (size_t)1 << 20);
    const size_t ns = (n + slice - 1) / slice;
    std::vector<std::vector<uint8_t>> out(ns);
    std::vector<uint32_t> crcs(ns, 0);
    std::vector<char> ok(ns, 1);
    auto work = [&](const size_t k) {
        const size_t a = k * slice, z = std::min(n, a + slice);
        z_stream zs; memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, Z.level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { ok[k] = 0; return; }
        out[k].resize((size_t)deflateBound(&zs, (uLong)(z - a)) + 64);
        size_t done = 0, produced = 0;
        while (true) {                                              // (avail_in is 32 bits wide: feed it in pieces)
            const size_t piece = std::min<size_t>(z - a - done, (size_t)1 << 30);
            zs.next_in = (Bytef*)(buf + a + done); zs.avail_in = (uInt)piece;
            const bool last = done + piece == z - a;
            do {
                if (produced == out[k].size()) out[k].resize(out[k].size() * 2);
                zs.next_out = out[k].data() + produced; zs.avail_out = (uInt)std::min<size_t>(out[k].size() - produced, (size_t)1 << 30);
                const uInt before = zs.avail_out;
                const int rc = deflate(&zs, last ? Z_SYNC_FLUSH : Z_NO_FLUSH);
                if (rc != Z_OK && rc != Z_BUF_ERROR) { ok[k] = 0; deflateEnd(&zs); return; }
                produced += before - zs.avail_out;
            } while (zs.avail_in > 0 || zs.avail_out == 0);
            done += piece;
            if (last) break;
        }
        deflateEnd(&zs);
        out[k].resize(produced);
        uint32_t c = 0;
        for (size_t p = a; p < z; p += (size_t)1 << 30) c = (uint32_t)crc32(c, buf + p, (uInt)std::min<size_t>(z - p, (size_t)1 << 30));
        crcs[k] = c;
    };
    {   // (nothing escapes: a slice whose thread cannot be created is compressed by this thread; std::bad_alloc inside a slice fails the call)
        auto guarded = [&](const size_t k) { try { work(k); } catch (...) { ok[k] = 0; } };
        std::vector<std::thread> pool;
        std::vector<size_t> mine{0};
        try { pool.reserve(ns); } catch (...) {}
        for (size_t k = 1; k < ns; ++k) {
            try { pool.emplace_back(guarded, k); } catch (...) { mine.push_back(k); }
        }
        for (const size_t k : mine) guarded(k);
        for (auto& t : pool) t.join();
    }
    for (size_t k = 0; k < ns; ++k) {
        if (!ok[k]) { err = "deflate failed (or out of memory)"; return false; }
        if (!c2_pwrite_parallel(Z.fd, out[k].data(), out[k].size(), Z.pos, 1, err)) return false;
        Z.pos += out[k].size(); Z.csize += out[k].size();
        const size_t a = k * slice, z = std::min(n, a + slice);
        Z.crc = (uint32_t)crc32_combine(Z.crc, crcs[k], (z_off_t)(z - a));
        Z.usize += z - a;
    }
    return true;
}
inline bool c2_zip_end(c2_zip_out& Z, std::string& err, uint64_t* file_bytes) {
    const uint8_t fin[2] = {0x03, 0x00};                            // an empty final block (fixed Huffman: end-of-block and nothing else) closes the stream
    if (!c2_pwrite_parallel(Z.fd, fin, 2, Z.pos, 1, err)) return false;
    Z.pos += 2; Z.csize += 2;
    if (!Z.zip64 && (Z.csize >= 0xffffffffull || Z.usize >= 0xffffffffull)) { err = "the member outgrew the 32-bit zip fields (c2_zip_begin was given a smaller text size)"; return false; }
    const std::string lh = c2_zip_local_header(Z);
    if (!c2_pwrite_parallel(Z.fd, (const uint8_t*)lh.data(), lh.size(), 0, 1, err)) return false;
    std::string cd;
    c2_le32(cd, 0x02014b50u); c2_le16(cd, (3u << 8) | (Z.zip64 ? 45u : 20u)); c2_le16(cd, Z.zip64 ? 45 : 20); c2_le16(cd, 0); c2_le16(cd, 8);
    c2_le16(cd, Z.dos_time); c2_le16(cd, Z.dos_date); c2_le32(cd, Z.crc);
    c2_le32(cd, Z.zip64 ? 0xffffffffu : (uint32_t)Z.csize); c2_le32(cd, Z.zip64 ? 0xffffffffu : (uint32_t)Z.usize);
    c2_le16(cd, (uint32_t)Z.name.size()); c2_le16(cd, Z.zip64 ? 20 : 0); c2_le16(cd, 0); c2_le16(cd, 0); c2_le16(cd, 0);
    c2_le32(cd, 0x81a40000u);                                      // a regular file, rw-r--r--
    c2_le32(cd, 0);                                                // the local header's offset
    cd += Z.name;
    if (Z.zip64) { c2_le16(cd, 1); c2_le16(cd, 16); c2_le64(cd, Z.usize); c2_le64(cd, Z.csize); }
    const uint64_t cd_off = Z.pos, cd_size = cd.size();
    std::string tail = cd;
    if (Z.zip64 || cd_off >= 0xffffffffull) {
        c2_le32(tail, 0x06064b50u); c2_le64(tail, 44); c2_le16(tail, 45); c2_le16(tail, 45); c2_le32(tail, 0); c2_le32(tail, 0);
        c2_le64(tail, 1); c2_le64(tail, 1); c2_le64(tail, cd_size); c2_le64(tail, cd_off);
        c2_le32(tail, 0x07064b50u); c2_le32(tail, 0); c2_le64(tail, cd_off + cd_size); c2_le32(tail, 1);
    }
    c2_le32(tail, 0x06054b50u); c2_le16(tail, 0); c2_le16(tail, 0); c2_le16(tail, 1); c2_le16(tail, 1);
    c2_le32(tail, (uint32_t)cd_size); c2_le32(tail, cd_off >= 0xffffffffull ? 0xffffffffu : (uint32_t)cd_off); c2_le16(tail, 0);
    if (!c2_pwrite_parallel(Z.fd, (const uint8_t*)tail.data(), tail.size(), Z.pos, 1, err)) return false;
    Z.pos += tail.size();
    if (ftruncate(Z.fd, (off_t)Z.pos) != 0) { /* (nothing behind it anyway) */ }
    if (file_bytes) *file_bytes = Z.pos;
    return true;
}

template <class B>
struct c2a_table {
    B be;
    c2_allele_src S;
    c2_allele_strings X;
    uint64_t m = 0;                       // rows
    c2_allele_row* d_rows = nullptr;
    uint32_t* d_order = nullptr;          // sorted position -> row
    std::string err;
    explicit c2a_table(const B& b) : be(b) {}
    ~c2a_table() { be.dfree(d_rows); be.dfree(d_order); }
};

// device buffers of one call, freed on every way out
template <class B>
struct c2a_scratch {
    B& be; std::vector<void*> dev, host;
    explicit c2a_scratch(B& b) : be(b) {}
    ~c2a_scratch() { for (void* p : dev) be.dfree(p); for (void* p : host) be.hfree(p); }
    template <class T> T* d(const size_t n) { void* p = be.dalloc(std::max<size_t>(n, 1) * sizeof(T)); if (p) dev.push_back(p); return (T*)p; }
    template <class T> T* h(const size_t n) { void* p = be.halloc(std::max<size_t>(n, 1) * sizeof(T)); if (p) host.push_back(p); return (T*)p; }
};

#define C2A_TRY(t, cond, what) do { if (!(cond)) { (t)->err = (t)->be.error.empty() ? std::string(what) : (std::string(what) + ": " + (t)->be.error); return C2_E_DEVICE; } } while (0)

template <class B>
int c2a_build(const B& backend, const c2_allele_src& src, c2a_table<B>** out, std::string& err)
{
    if (src.n_reads == 0) { auto* t = new c2a_table<B>(backend); t->S = src; *out = t; return 0; }
    if (!src.d_aln_read1 || !src.d_aln_ref1 || !src.d_records1 || !src.d_member || !src.d_flags || !src.d_counts || src.n_refs <= 0 ||
        (src.d_use2 && src.d_slot2 && (!src.d_aln_read2 || !src.d_aln_ref2 || !src.d_records2)) || (src.d_scaffold_hit && (src.scaffold_ref < 0 || src.scaffold_ref >= src.n_refs)) ||
        (src.stride1 % 16u) || (src.d_aln_read2 && (src.stride2 % 16u))) { err = "bad c2_allele_src"; return C2_E_INVALID; }
    if (src.n_reads * (uint64_t)src.n_refs >= 0x80000000ull) { err = "n_reads * n_refs must stay below 2^31"; return C2_E_TOO_LARGE; }
    std::unique_ptr<c2a_table<B>> t(new c2a_table<B>(backend));
    t->S = src;
    t->X = c2_allele_strings{src.d_aln_read1, src.d_aln_ref1, src.d_aln_read2, src.d_aln_ref2, src.stride1, src.stride2};
    B& be = t->be;
    c2a_scratch<B> tmp(be);
    const uint64_t n = src.n_reads;
    uint32_t* d_nj = tmp.template d<uint32_t>(n);
    uint64_t* d_off = tmp.template d<uint64_t>(n + 1);
    if (!d_nj || !d_off) { err = "device memory: " + be.error; return C2_E_NOMEM; }
    c2_allele_jobs_args J;
    J.S = src; J.offsets = nullptr; J.njobs = d_nj; J.rows = nullptr;
    uint64_t last_off = 0; uint32_t last_n = 0;
    if (!be.jobs(J) || !be.scan(d_nj, d_off, n) || !be.d2h(&last_off, d_off + (n - 1), 8) || !be.d2h(&last_n, d_nj + (n - 1), 4)) { err = "allele rows: " + be.error; return C2_E_DEVICE; }
    const uint64_t m = last_off + last_n;
    if (m >= 0x80000000ull) { err = "the table would have 2^31 rows or more"; return C2_E_TOO_LARGE; }
    t->m = m;
    if (m) {
        t->d_rows = (c2_allele_row*)be.dalloc(m * sizeof(c2_allele_row));
        t->d_order = (uint32_t*)be.dalloc(m * 4);
        uint32_t* d_iota = tmp.template d<uint32_t>(m);
        if (!t->d_rows || !t->d_order || !d_iota) { err = "device memory: " + be.error; return C2_E_NOMEM; }
        J.offsets = d_off; J.rows = t->d_rows;
        c2_allele_row_less less{t->X, t->d_rows};
        if (!be.jobs(J) || !be.iota(d_iota, m) || !be.sort_rows(less, d_iota, t->d_order, m) || !be.sync()) { err = "allele rows: " + be.error; return C2_E_DEVICE; }
    }
    *out = t.release();
    return 0;
}

// runs of equal #Reads in sorted order and the text of their %Reads = #Reads / n_total * 100
struct c2a_runs { std::vector<uint32_t> start, off; std::vector<uint8_t> len, blob; };
inline void c2a_make_runs(const uint32_t* reads, const uint64_t m, const int64_t n_total, c2a_runs& R) {
    char buf[40];
    for (uint64_t q = 0; q < m; ++q)
        if (q == 0 || reads[q] != reads[q - 1]) {
            R.start.push_back((uint32_t)q);
            const double pct = (double)reads[q] / (double)n_total * 100;
            const int n = c2_py_float_repr(pct, buf);
            R.off.push_back((uint32_t)R.blob.size());
            R.len.push_back((uint8_t)n);
            R.blob.insert(R.blob.end(), buf, buf + n);
        }
}

template <class B>
int c2a_write(c2a_table<B>* t, const char* path, const char* const* labels, const int64_t n_total, const char* const* probes, int threads, uint64_t* bytes_written,
              const char* zip_member = nullptr, const int zip_level = 1, uint64_t* zip_bytes = nullptr)
{
    // zip_member: `path` becomes a zip archive whose one member of that name is the table (what the reference leaves behind, CRISPRessoCORE.py:4531)
    B& be = t->be;
    const uint64_t m = t->m;
    std::string head = "Aligned_Sequence\tReference_Sequence\tReference_Name\tRead_Status\tn_deleted\tn_inserted\tn_mutated\t#Reads\t%Reads";
    if (probes) head += "\tcontains dsODN\tcontains dsODN fragment";
    head += "\n";
    const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) { t->err = std::string("open ") + path + ": " + strerror(errno); return C2_E_INVALID; }
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    c2_zip_out Z;
    if (!zip_member && !c2_pwrite_parallel(fd, (const uint8_t*)head.data(), head.size(), 0, 1, t->err)) return C2_E_INVALID;
    if (bytes_written) *bytes_written = head.size();
    if (m == 0) {
        if (zip_member && !(c2_zip_begin(Z, fd, zip_member, head.size(), zip_level, t->err) && c2_zip_append(Z, (const uint8_t*)head.data(), head.size(), 1, t->err) &&
                            c2_zip_end(Z, t->err, zip_bytes))) return C2_E_INVALID;
        return 0;
    }
    if (n_total <= 0) { t->err = "n_total must be positive"; return C2_E_INVALID; }
    if (threads < 1) threads = 1;
    c2a_scratch<B> tmp(be);
    const int k = t->S.n_refs, nl = C2_ALLELE_LABELS(k);
    // tables: labels, runs of equal #Reads with their %Reads text
    std::vector<uint32_t> lab_off(nl + 1, 0); std::vector<uint8_t> lab_blob;
    for (int l = 0; l < nl; ++l) { const char* s = labels && labels[l] ? labels[l] : ""; lab_blob.insert(lab_blob.end(), s, s + strlen(s)); lab_off[l + 1] = (uint32_t)lab_blob.size(); }
    uint32_t* d_reads = tmp.template d<uint32_t>(m);
    std::vector<uint32_t> h_reads(m);
    C2A_TRY(t, d_reads && be.reads(t->d_rows, t->d_order, m, d_reads) && be.d2h(h_reads.data(), d_reads, m * 4), "sorted #Reads");
    c2a_runs R;
    c2a_make_runs(h_reads.data(), m, n_total, R);
    std::vector<uint32_t>().swap(h_reads);
    c2_allele_text_args A;
    memset(&A, 0, sizeof A);
    A.X = t->X; A.rows = t->d_rows; A.order = t->d_order; A.m = m; A.n_runs = (uint32_t)R.start.size();
    uint32_t* d_run_start = tmp.template d<uint32_t>(R.start.size());
    uint32_t* d_pct_off = tmp.template d<uint32_t>(R.off.size());
    uint8_t* d_pct_len = tmp.template d<uint8_t>(R.len.size());
    uint8_t* d_pct_blob = tmp.template d<uint8_t>(R.blob.size());
    uint32_t* d_lab_off = tmp.template d<uint32_t>(lab_off.size());
    uint8_t* d_lab_blob = tmp.template d<uint8_t>(lab_blob.size() + 1);
    uint32_t* d_len = tmp.template d<uint32_t>(m);
    uint64_t* d_off = tmp.template d<uint64_t>(m + 1);
    C2A_TRY(t, d_run_start && d_pct_off && d_pct_len && d_pct_blob && d_lab_off && d_lab_blob && d_len && d_off, "device memory");
    C2A_TRY(t, be.h2d(d_run_start, R.start.data(), R.start.size() * 4) && be.h2d(d_pct_off, R.off.data(), R.off.size() * 4) && be.h2d(d_pct_len, R.len.data(), R.len.size()) &&
               be.h2d(d_pct_blob, R.blob.data(), R.blob.size()) && be.h2d(d_lab_off, lab_off.data(), lab_off.size() * 4) &&
               (lab_blob.empty() || be.h2d(d_lab_blob, lab_blob.data(), lab_blob.size())), "tables to the device");
    A.run_start = d_run_start; A.pct_off = d_pct_off; A.pct_len = d_pct_len; A.pct_blob = d_pct_blob; A.label_off = d_lab_off; A.label_blob = d_lab_blob;
    if (probes) {
        c2_allele_probe_args P;
        memset(&P, 0, sizeof P);
        std::vector<uint8_t> pb;
        for (int p = 0; p < 4; ++p) { const char* s = probes[p] ? probes[p] : ""; P.probe_off[p] = (uint32_t)pb.size(); pb.insert(pb.end(), s, s + strlen(s)); }
        P.probe_off[4] = (uint32_t)pb.size();
        uint8_t* d_pb = tmp.template d<uint8_t>(pb.size() + 1);
        uint8_t* d_bits = tmp.template d<uint8_t>(m);
        C2A_TRY(t, d_pb && d_bits && (pb.empty() || be.h2d(d_pb, pb.data(), pb.size())), "probes to the device");
        P.X = t->X; P.rows = t->d_rows; P.order = t->d_order; P.m = m; P.probe_blob = d_pb; P.probe_bits = d_bits;
        C2A_TRY(t, be.probe(P), "c2_allele_probe_kernel");
        A.probe_bits = d_bits;
    }
    A.lengths = d_len;
    C2A_TRY(t, be.lengths(A) && be.scan(d_len, d_off, m), "line lengths");
    std::vector<uint64_t> off(m + 1);
    uint32_t last_len = 0;
    C2A_TRY(t, be.d2h(off.data(), d_off, m * 8) && be.d2h(&last_len, d_len + (m - 1), 4), "line offsets");
    off[m] = off[m - 1] + last_len;
    const uint64_t total = off[m];
    if (zip_member) {
        if (!(c2_zip_begin(Z, fd, zip_member, head.size() + total, zip_level, t->err) && c2_zip_append(Z, (const uint8_t*)head.data(), head.size(), 1, t->err))) return C2_E_INVALID;
    } else if (ftruncate(fd, (off_t)(head.size() + total)) != 0) { /* (a file system without it: pwrite extends the file) */ }
    // chunks of whole lines, at most `chunk` bytes each (one line always fits: a chunk holds at least one)
    uint64_t chunk = (uint64_t)64 << 20;
    if (const char* e = getenv("C2_ALLELE_CHUNK_BYTES")) { const long long v = atoll(e); if (v > 0) chunk = (uint64_t)v; }
    uint64_t longest = 0;
    std::vector<uint64_t> cuts{0};
    for (uint64_t q = 0; q < m;) {
        uint64_t lo = q + 1, hi = m;                                          // the largest q1 with off[q1] - off[q] <= chunk (at least q + 1)
        while (lo < hi) { const uint64_t mid = (lo + hi + 1) >> 1; if (off[mid] - off[q] <= chunk) lo = mid; else hi = mid - 1; }
        longest = std::max(longest, off[lo] - off[q]);
        cuts.push_back(lo);
        q = lo;
    }
    uint8_t* d_buf[2] = {tmp.template d<uint8_t>(longest), tmp.template d<uint8_t>(longest)};
    uint8_t* h_buf[2] = {tmp.template h<uint8_t>(longest), tmp.template h<uint8_t>(longest)};
    C2A_TRY(t, d_buf[0] && d_buf[1] && h_buf[0] && h_buf[1], "chunk buffers");
    A.offsets = d_off; A.lengths = nullptr;
    // the writer thread puts chunk c on disk while the device forms chunk c + 1 and the link carries it
    std::mutex mu; std::condition_variable cv;
    int pending[2] = {0, 0};                                                  // 1: chunk in h_buf[i] waits to be written
    uint64_t w_off[2] = {0, 0}, w_len[2] = {0, 0};
    bool done = false, failed = false; std::string werr;
    std::thread writer([&] {
        for (int i = 0;; i ^= 1) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return pending[i] || done; });
            if (!pending[i]) return;
            lk.unlock();
            std::string e;
            const bool ok = zip_member ? c2_zip_append(Z, h_buf[i], w_len[i], threads, e)       // (chunks arrive in order: one writer thread)
                                       : c2_pwrite_parallel(fd, h_buf[i], w_len[i], head.size() + w_off[i], threads, e);
            lk.lock();
            if (!ok && !failed) { failed = true; werr = e; }
            pending[i] = 0;
            cv.notify_all();
        }
    });
    bool dev_ok = true;
    for (size_t c = 0; c + 1 < cuts.size() && dev_ok; ++c) {
        const int i = (int)(c & 1);
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !pending[i]; }); if (failed) break; }
        A.q0 = cuts[c]; A.q1 = cuts[c + 1]; A.out = d_buf[i];
        const uint64_t nb = off[A.q1] - off[A.q0];
        dev_ok = be.emit(A) && be.d2h(h_buf[i], d_buf[i], nb);
        if (!dev_ok) break;
        { std::lock_guard<std::mutex> lk(mu); w_off[i] = off[A.q0]; w_len[i] = nb; pending[i] = 1; }
        cv.notify_all();
    }
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !pending[0] && !pending[1]; }); done = true; }
    cv.notify_all();
    writer.join();
    C2A_TRY(t, dev_ok, "c2_allele_emit_kernel");
    if (failed) { t->err = werr; return C2_E_INVALID; }
    if (zip_member && !c2_zip_end(Z, t->err, zip_bytes)) return C2_E_INVALID;
    if (bytes_written) *bytes_written = head.size() + total;
    return 0;
}

template <class B>
int c2a_fetch(c2a_table<B>* t, c2_allele_row* rows, uint8_t* aligned, uint8_t* reference, const uint32_t stride)
{
    B& be = t->be;
    const uint64_t m = t->m;
    if (m == 0) return 0;
    if ((aligned || reference) && stride == 0) { t->err = "stride is 0"; return C2_E_INVALID; }
    c2a_scratch<B> tmp(be);
    // in slabs, so that the staging stays small next to a table of millions of rows
    const uint64_t slab = std::max<uint64_t>(1, std::min<uint64_t>(m, ((uint64_t)256 << 20) / std::max<uint32_t>(stride, 24)));
    c2_allele_row* d_rows = rows ? tmp.template d<c2_allele_row>(slab) : nullptr;
    uint8_t* d_a = aligned ? tmp.template d<uint8_t>(slab * stride) : nullptr;
    uint8_t* d_f = reference ? tmp.template d<uint8_t>(slab * stride) : nullptr;
    C2A_TRY(t, (!rows || d_rows) && (!aligned || d_a) && (!reference || d_f), "device memory");
    for (uint64_t q0 = 0; q0 < m; q0 += slab) {
        const uint64_t mm = std::min(slab, m - q0);
        c2_allele_fetch_args F;
        F.X = t->X; F.rows = t->d_rows; F.order = t->d_order + q0; F.m = mm; F.out_rows = d_rows; F.out_a = d_a; F.out_f = d_f; F.stride = stride;
        C2A_TRY(t, be.fetch(F), "c2_allele_fetch_kernel");
        C2A_TRY(t, (!rows || be.d2h(rows + q0, d_rows, mm * sizeof(c2_allele_row))) && (!aligned || be.d2h(aligned + q0 * stride, d_a, mm * stride)) &&
                   (!reference || be.d2h(reference + q0 * stride, d_f, mm * stride)), "rows to the host");
    }
    return 0;
}

// <ref>Alleles_frequency_table_around_<guide>.txt: windows and groups on the device; the sums in table order (pandas' groupby().sum() adds
// float64 with Kahan compensation: `y = x - c; t = s + y; c = t - s - y; s = t` per group, rows in table order), the final order
// (#Reads descending over groups that are already in key order: a stable counting sort) and the text on the host.
template <class B>
int c2a_around_cut_write(c2a_table<B>* t, const int32_t label, const int32_t cut_point, const int32_t ref_len, const int32_t plot_window_size,
                         const int64_t n_total, const char* path, int threads, uint64_t* n_groups)
{
    B& be = t->be;
    const uint64_t m = t->m;
    if (n_groups) *n_groups = 0;
    if (cut_point < 0 || plot_window_size < 0 || ref_len <= 0) { t->err = "bad window"; return C2_E_INVALID; }
    // plots/data_prep.py:285-301
    const int left = cut_point - plot_window_size + 1 >= 0 ? plot_window_size : cut_point + 1;
    const int right = cut_point + plot_window_size < ref_len ? plot_window_size : ref_len - cut_point - 1;
    if (left + right > C2_ALLELE_MAX_WINDOW || right < 0) { t->err = "window wider than C2_ALLELE_MAX_WINDOW columns"; return C2_E_TOO_LARGE; }
    if (threads < 1) threads = 1;
    const std::string head = "Aligned_Sequence\tReference_Sequence\tUnedited\tn_deleted\tn_inserted\tn_mutated\t#Reads\t%Reads\n";
    const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) { t->err = std::string("open ") + path + ": " + strerror(errno); return C2_E_INVALID; }
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    if (!c2_pwrite_parallel(fd, (const uint8_t*)head.data(), head.size(), 0, 1, t->err)) return C2_E_INVALID;
    if (m == 0) return 0;
    if (n_total <= 0) { t->err = "n_total must be positive"; return C2_E_INVALID; }
    c2a_scratch<B> tmp(be);
    const uint32_t W = (uint32_t)(left + right), KB = (2u * W + 7u + 7u) & ~7u;
    c2_allele_window_args A;
    memset(&A, 0, sizeof A);
    A.X = t->X; A.rows = t->d_rows; A.order = t->d_order; A.m = m; A.label = label; A.cut_point = cut_point; A.left = left; A.right = right; A.W = W; A.key_bytes = KB;
    uint32_t* d_flag = tmp.template d<uint32_t>(m);
    uint64_t* d_sub = tmp.template d<uint64_t>(m + 1);
    uint32_t* d_err = tmp.template d<uint32_t>(1);
    C2A_TRY(t, d_flag && d_sub && d_err && be.zero(d_err, 4), "device memory");
    A.flag = d_flag;
    uint64_t last_sub = 0; uint32_t last_flag = 0;
    C2A_TRY(t, be.window(A) && be.scan(d_flag, d_sub, m) && be.d2h(&last_sub, d_sub + (m - 1), 8) && be.d2h(&last_flag, d_flag + (m - 1), 4), "rows of the reference");
    const uint64_t ms = last_sub + last_flag;
    if (ms == 0) return 0;
    uint8_t* d_keys = tmp.template d<uint8_t>(ms * KB);
    uint32_t* d_sreads = tmp.template d<uint32_t>(ms);
    uint32_t* d_iota = tmp.template d<uint32_t>(ms);
    uint32_t* d_perm = tmp.template d<uint32_t>(ms);
    uint32_t* d_head = tmp.template d<uint32_t>(ms);
    uint64_t* d_hscan = tmp.template d<uint64_t>(ms + 1);
    uint32_t* d_gid = tmp.template d<uint32_t>(ms);
    C2A_TRY(t, d_keys && d_sreads && d_iota && d_perm && d_head && d_hscan && d_gid, "device memory");
    A.sub_index = d_sub; A.keys = d_keys; A.sub_reads = d_sreads; A.error = d_err;
    uint32_t h_err = 0;
    C2A_TRY(t, be.window(A) && be.d2h(&h_err, d_err, 4), "c2_allele_window_kernel");
    if (h_err) { t->err = std::to_string(cut_point) + " is not in list"; return C2_E_INVALID; }
    c2_allele_key_less less{d_keys, KB};
    c2_allele_group_args G;
    memset(&G, 0, sizeof G);
    G.keys = d_keys; G.key_bytes = KB; G.ms = ms; G.perm = d_perm; G.head = d_head;
    uint64_t last_hs = 0; uint32_t last_h = 0;
    C2A_TRY(t, be.iota(d_iota, ms) && be.sort_keys(less, d_iota, d_perm, ms) && be.group(G) && be.scan(d_head, d_hscan, ms) &&
               be.d2h(&last_hs, d_hscan + (ms - 1), 8) && be.d2h(&last_h, d_head + (ms - 1), 4), "window groups");
    const uint64_t ng = last_hs + last_h;
    uint8_t* d_gkeys = tmp.template d<uint8_t>(ng * KB);
    C2A_TRY(t, d_gkeys, "device memory");
    G.head_scan = d_hscan; G.gid = d_gid; G.gkeys = d_gkeys;
    std::vector<uint32_t> gid(ms), sreads(ms);
    std::vector<uint8_t> gkeys(ng * KB);
    C2A_TRY(t, be.group(G) && be.d2h(gid.data(), d_gid, ms * 4) && be.d2h(sreads.data(), d_sreads, ms * 4) && be.d2h(gkeys.data(), d_gkeys, ng * KB), "groups to the host");
    // sums in table order
    std::vector<uint64_t> g_reads(ng, 0);
    std::vector<double> g_sum(ng, 0.0), g_comp(ng, 0.0);
    for (uint64_t s = 0; s < ms; ++s) {
        const uint32_t g = gid[s];
        g_reads[g] += sreads[s];
        const double x = (double)sreads[s] / (double)n_total * 100;
        volatile double y = x - g_comp[g];                                    // (volatile: the compensation must not be simplified away)
        volatile double tt = g_sum[g] + y;
        g_comp[g] = (tt - g_sum[g]) - y;
        g_sum[g] = tt;
    }
    // #Reads descending, stable over the key order: LSD radix sort of the groups on 16-bit digits of ~reads
    std::vector<uint32_t> ord(ng), ord2(ng);
    for (uint64_t g = 0; g < ng; ++g) ord[g] = (uint32_t)g;
    uint64_t max_reads = 0;
    for (uint64_t g = 0; g < ng; ++g) max_reads = std::max(max_reads, g_reads[g]);
    for (int shift = 0; shift < 64 && (shift == 0 || (max_reads >> shift)); shift += 16) {
        std::vector<uint64_t> cnt(65537, 0);
        for (uint64_t g = 0; g < ng; ++g) cnt[65535 - ((g_reads[ord[g]] >> shift) & 0xffff) + 1]++;
        for (int b = 0; b < 65536; ++b) cnt[b + 1] += cnt[b];
        for (uint64_t g = 0; g < ng; ++g) ord2[cnt[65535 - ((g_reads[ord[g]] >> shift) & 0xffff)]++] = ord[g];
        ord.swap(ord2);
    }
    // text: threads format ranges of groups, then write them at their offsets
    const int nt = (int)std::min<uint64_t>((uint64_t)threads, std::max<uint64_t>(1, ng / 4096));
    std::vector<std::string> parts(nt);
    auto fmt = [&](const int ti) {
        std::string& o = parts[ti];
        const uint64_t g0 = ng * (uint64_t)ti / (uint64_t)nt, g1 = ng * (uint64_t)(ti + 1) / (uint64_t)nt;
        o.reserve((g1 - g0) * (2 * W + 48));
        char buf[64];
        for (uint64_t j = g0; j < g1; ++j) {
            const uint32_t g = ord[j];
            const uint8_t* key = gkeys.data() + (uint64_t)g * KB;
            o.append((const char*)key, strnlen((const char*)key, W)); o.push_back('\t');
            o.append((const char*)key + W, strnlen((const char*)key + W, W)); o.push_back('\t');
            o.append(key[2 * W] ? "True" : "False");
            for (int f = 0; f < 3; ++f) { const unsigned v = ((unsigned)key[2 * W + 1 + 2 * f] << 8) | key[2 * W + 2 + 2 * f]; o.push_back('\t'); o.append(std::to_string(v)); }
            o.push_back('\t'); o.append(std::to_string(g_reads[g]));
            o.push_back('\t'); o.append(buf, c2_py_float_repr(g_sum[g], buf));
            o.push_back('\n');
        }
    };
    { std::vector<std::thread> pool; for (int ti = 1; ti < nt; ++ti) pool.emplace_back(fmt, ti); fmt(0); for (auto& th : pool) th.join(); }
    uint64_t at = head.size();
    std::vector<uint64_t> where(nt);
    for (int ti = 0; ti < nt; ++ti) { where[ti] = at; at += parts[ti].size(); }
    {
        std::vector<std::thread> pool; std::vector<std::string> errs(nt); std::vector<char> oks(nt, 1);
        auto wr = [&](const int ti) { oks[ti] = c2_pwrite_parallel(fd, (const uint8_t*)parts[ti].data(), parts[ti].size(), where[ti], 1, errs[ti]) ? 1 : 0; };
        for (int ti = 1; ti < nt; ++ti) pool.emplace_back(wr, ti);
        wr(0);
        for (auto& th : pool) th.join();
        for (int ti = 0; ti < nt; ++ti) if (!oks[ti]) { t->err = errs[ti]; return C2_E_INVALID; }
    }
    if (n_groups) *n_groups = ng;
    return 0;
}
