// Streaming FASTQ ingest + exact de-duplication (host code; included by c2_fastq.cpp inside its anonymous namespace).
//
// Same semantics as the reference's readline loop (CRISPRessoCORE.py:1820-1849; see c2_fastq.cpp's header): records are four
// consecutive lines from the top of the text whatever they contain, universal newlines, str.strip() on the sequence line, unique
// sequences in first-seen order with their multiplicities.  What is different from a whole-file parse is HOW the text is walked:
//
//   * the text is consumed in CHUNKS of (threads x range_bytes); after every chunk the unique reads seen so far -- arena, offsets, in
//     first-seen order -- are final, so the caller can send the new ones to the GPU while the next chunk is parsed
//     (pipeline.quantify_fastq; SURVEY 8d "stream in batches");
//   * a plain file is pread() into per-thread buffers of range_bytes (reused chunk after chunk, so they stay in the core's L2 / L3
//     share) instead of being mapped: no page-table work for gigabytes of page cache (mapping + first touch was ~40 % of the old ingest);
//   * the reference frames records by LINE NUMBER, so every range first counts its terminators (from the buffer it just filled), a
//     prefix sum over the chunk gives each range the number of its first line, and the range is parsed from the still-cached buffer;
//   * de-duplication is two-level: a per-thread table that persists over the chunks (hot sequences -- the unmodified amplicon is
//     ~18 % of a typical run -- never leave the thread) and ONE global lock-free table (CAS on 32-bit slots, entries in stable
//     segments) into which every thread inserts only what is new to IT; a chunk's new unique reads are ordered by (range, position
//     in range) of their first occurrence, which is the file order.
#pragma once

struct WorkPool {
    // run(fn): fn(t) for t in [0, T) on T - 1 pooled threads + the caller; returns when all are done
    unsigned T;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    std::function<void(unsigned)> job;
    uint64_t gen = 0;
    unsigned pending = 0;
    bool quit = false;
    explicit WorkPool(unsigned t) : T(t < 1 ? 1 : t) {
        for (unsigned k = 1; k < T; ++k) th.emplace_back([this, k] { loop(k); });
    }
    ~WorkPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; ++gen; }
        cv.notify_all();
        for (auto& x : th) x.join();
    }
    void loop(unsigned k) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (quit) return;
                f = job;
            }
            f(k);
            { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv_done.notify_one(); }
        }
    }
    template <class F> void run(F&& fn) {
        if (T == 1) { fn(0u); return; }
        { std::lock_guard<std::mutex> lk(mu); job = fn; pending = T - 1; ++gen; }
        cv.notify_all();
        fn(0u);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

struct StreamEntry {                      // one unique sequence of the run
    uint64_t h;
    std::atomic<uint64_t> first;          // (range << 32 | position in the range's new-unique list) of its first occurrence in the chunk that created it
    std::atomic<uint64_t> count;
    const uint8_t* src;                   // its bytes in the creating thread's buffer (valid during the creating chunk only)
    uint64_t arena_off;                   // ... and in the arena from the end of that chunk on
    uint32_t len;
    uint32_t chunk;                       // the chunk that created it
    uint32_t gidx;                        // index in first-seen order
};

struct LocalSeq { uint64_t h; const uint8_t* p; uint32_t len; uint32_t entry; uint64_t pending; };
constexpr uint32_t C2_NO_ENTRY = 0xffffffffu;

struct alignas(128) LocalSet {             // a thread's own table: persists over the chunks (own cache lines: its vectors' end pointers move with every new sequence)
    std::vector<uint32_t> slots = std::vector<uint32_t>(1u << 12, 0);
    uint64_t mask = (1u << 12) - 1;
    std::vector<LocalSeq> u;
    size_t first_new = 0;                 // u[first_new ..] were first seen by this thread in the current chunk
    void grow() {
        std::vector<uint32_t> t(slots.size() * 2, 0);
        const uint64_t m = t.size() - 1;
        for (uint32_t k = 0; k < (uint32_t)u.size(); ++k) { uint64_t pos = u[k].h & m; while (t[pos]) pos = (pos + 1) & m; t[pos] = k + 1; }
        slots.swap(t); mask = m;
    }
    void add(const uint8_t* s, uint32_t len) { add_hashed(s, len, hash_bytes(s, len)); }
    // the look-up of one sequence touches three lines nobody has in cache (slot -> u[k] -> the representative's bytes): the parser
    // keeps a few sequences in flight and asks for those lines ahead of time (stage 1, 2, 3), see FastqStream::next, phase B
    void prefetch_slot(uint64_t h) const { __builtin_prefetch(&slots[h & mask]); }
    void prefetch_entry(uint64_t h) const { const uint32_t k = slots[h & mask]; if (k) __builtin_prefetch(&u[k - 1]); }
    void prefetch_bytes(uint64_t h) const {
        const uint32_t k = slots[h & mask];
        if (!k) return;
        const LocalSeq& q = u[k - 1];
        for (uint32_t o = 0; o < q.len; o += 64) __builtin_prefetch(q.p + o);
    }
    void add_hashed(const uint8_t* s, uint32_t len, const uint64_t h) {
        uint64_t pos = h & mask;
        while (slots[pos]) {
            LocalSeq& q = u[slots[pos] - 1];
            if (q.h == h && q.len == len && (len == 0 || memcmp(q.p, s, len) == 0)) { ++q.pending; return; }
            pos = (pos + 1) & mask;
        }
        slots[pos] = (uint32_t)u.size() + 1;
        u.push_back(LocalSeq{h, s, len, C2_NO_ENTRY, 1});
        if (u.size() * 2 > slots.size()) grow();
    }
};

struct FastqStream {
    // ---- source: a plain file (pread) or text in memory (mapped / inflated / filtered)
    int fd = -1;
    const char* mem = nullptr;
    size_t n = 0;                         // bytes of text
    size_t pos = 0;                       // first byte of the next chunk
    uint64_t terms_before = 0;            // line terminators that end before `pos`
    // ---- work
    unsigned T = 1;
    size_t range_bytes = (size_t)4 << 20;
    std::unique_ptr<WorkPool> pool;
    std::vector<LocalSet> local;
    std::vector<ByteBuf> buf;             // per-thread read buffers (file source)
    uint32_t chunk_no = 0;
    // ---- global table
    static constexpr unsigned SEG_BITS = 18;
    std::vector<std::unique_ptr<StreamEntry[]>> seg;
    std::atomic<uint32_t> n_entries{0};
    std::atomic<uint32_t>* slots = nullptr;
    size_t slots_cap = 0;
    // ---- result so far
    ByteBuf arena;                        // reserved for the whole text up front: the pointer never moves
    std::vector<uint64_t> offsets{0};
    std::vector<uint32_t> entry_of;       // entry id of unique g
    uint64_t n_reads = 0, nonempty_lines = 0;
    bool done = false, overflow = false;
    std::string err;
    double phase_s[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // A load + count, B parse, (grow), C global insert, D survivors, E arena copy, F re-point, serial parts

    ~FastqStream() { if (slots) munmap((void*)slots, slots_cap * sizeof(std::atomic<uint32_t>)); if (fd >= 0) close(fd); }

    StreamEntry& entry(uint32_t id) { return seg[id >> SEG_BITS][id & ((1u << SEG_BITS) - 1)]; }
    const uint8_t* bytes_of(const StreamEntry& e) const { return e.chunk == chunk_no ? e.src : arena.data() + e.arena_off; }

    bool init(unsigned threads, size_t rb) {
        T = threads < 1 ? 1 : threads;
        range_bytes = rb < 1 ? 1 : rb;
        pool.reset(new WorkPool(T));
        local.resize(T);
        { std::vector<ByteBuf> fresh(T); buf.swap(fresh); }
        try { arena.reserve(n + 1); } catch (...) { err = "cannot reserve the arena"; return false; }     // (anonymous mapping: untouched pages cost nothing)
        return grow_slots((size_t)1 << 16);
    }

    bool grow_slots(size_t want_cap) {
        size_t cap = slots_cap ? slots_cap : ((size_t)1 << 16);
        while (cap < want_cap) cap <<= 1;
        if (cap == slots_cap) return true;
        void* q = mmap(nullptr, cap * sizeof(std::atomic<uint32_t>), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);   // zero pages
        if (q == MAP_FAILED) { err = "cannot allocate the hash table"; return false; }
        madvise(q, cap * sizeof(std::atomic<uint32_t>), MADV_HUGEPAGE);
        std::atomic<uint32_t>* ns = (std::atomic<uint32_t>*)q;
        const uint64_t m = cap - 1;
        const uint32_t ne = n_entries.load();
        if (ne) {
            pool->run([&](unsigned t) {
                for (uint32_t id = (uint32_t)((uint64_t)ne * t / T), end = (uint32_t)((uint64_t)ne * (t + 1) / T); id < end; ++id) {
                    StreamEntry& e = entry(id);
                    if (e.gidx == C2_NO_ENTRY) continue;                 // an id that was reserved and never published
                    uint64_t p = e.h & m;
                    for (;;) {
                        uint32_t exp = 0;
                        if (ns[p].load(std::memory_order_relaxed) == 0 && ns[p].compare_exchange_strong(exp, id + 1, std::memory_order_relaxed)) break;
                        p = (p + 1) & m;
                    }
                }
            });
        }
        if (slots) munmap((void*)slots, slots_cap * sizeof(std::atomic<uint32_t>));
        slots = ns; slots_cap = cap;
        return true;
    }

    // bytes [a, b) of the text for thread t: a pointer `base` such that base[x] is text byte x for x in [a, b)
    const char* view(unsigned t, size_t a, size_t b) {
        if (mem) return mem;
        ByteBuf& B = buf[t];
        B.resize(b - a);
        size_t got = 0;
        while (got < b - a) {
            const ssize_t r = pread(fd, B.data() + got, b - a - got, (off_t)(a + got));
            if (r <= 0) { memset(B.data() + got, '\n', b - a - got); break; }     // (a file that shrank under us: defined bytes, flagged below)
            got += (size_t)r;
        }
        return (const char*)B.data() - a;
    }

    // Parse the next chunk.  -> false on error (err set).  After it, offsets / arena / entry_of hold every unique read seen so far.
    bool next() {
        if (done) return true;
        const size_t lo0 = pos;
        const size_t span = (size_t)T * range_bytes;
        const size_t hi0 = (n - lo0 <= span + span / 8) ? n : lo0 + span;       // (a short tail joins the last chunk)
        ++chunk_no;
        const uint32_t cur = chunk_no;
        std::vector<size_t> cut(T + 1);
        for (unsigned t = 0; t <= T; ++t) cut[t] = lo0 + (size_t)((unsigned __int128)(hi0 - lo0) * t / T);
        std::vector<uint64_t> terms(T, 0), starts(T, 0), first(T, 0), n_seq(T, 0);
        std::vector<const char*> base(T, nullptr);
        std::vector<size_t> have(T, 0);                              // base[t][x] is valid for x in [max(cut[t], 1) - 1, have[t])
        double tp = now_s();
        auto lapse = [&](int k) { const double t = now_s(); phase_s[k] += t - tp; tp = t; };
        // A: load + count.  The view of a range reaches back one byte (is cut[t] a line start?) and forward to the end of the last
        // line that starts inside it (+ one byte, to tell "\r\n" from "\r")
        pool->run([&](unsigned t) {
            const size_t lo = cut[t], hi = cut[t + 1];
            if (lo >= hi) return;
            const size_t a = lo ? lo - 1 : 0;
            size_t b = std::min(n, hi + 512);
            const char* B = view(t, a, b);
            for (;;) {
                bool found = false;
                for (size_t x = hi - 1; x < b; ++x) if (B[x] == '\n' || B[x] == '\r') { found = x + 1 < b || b == n; break; }
                if (found || b == n) break;
                b = std::min(n, b + std::max<size_t>(65536, b - hi));
                B = view(t, a, b);
            }
            base[t] = B; have[t] = b;
            terms[t] = count_terminators(B, n, lo, hi, &starts[t]);
        });
        lapse(0);
        uint64_t before = terms_before;
        for (unsigned t = 0; t < T; ++t) {
            const size_t lo = cut[t];
            if (lo < cut[t + 1]) {
                if (lo == 0) first[t] = 0;
                else first[t] = before + (term_end(base[t], n, lo - 1) ? 0 : 1);
            }
            before += terms[t];
            nonempty_lines += starts[t];
        }
        // B: parse the lines that START in the range; sequence lines (number 1 mod 4) go to the thread's table
        pool->run([&](unsigned t) {
            const size_t lo = cut[t], hi = cut[t + 1];
            LocalSet& L = local[t];
            L.first_new = L.u.size();
            if (lo >= hi) return;
            const char* b = base[t];
            const size_t e = have[t];
            size_t p = lo;
            if (lo > 0 && !term_end(b, n, lo - 1)) {                 // lo is inside a line that started earlier: skip to its end
                while (p < e && !term_end(b, n, p)) ++p;
                ++p;
            }
            uint64_t line_no = first[t];
            const bool has_cr = memchr(b + lo, '\r', e - lo) != nullptr;
            uint64_t seqs = 0;
            constexpr unsigned RING = 12;
            struct Pending { const uint8_t* s; uint32_t len; uint64_t h; };
            Pending ring[RING];
            uint64_t rn = 0;
            while (p < hi && p < n) {
                size_t end;                                          // first byte of the terminator, or n
                if (!has_cr) {
                    const char* nl = (const char*)memchr(b + p, '\n', e - p);
                    end = nl ? (size_t)(nl - b) : e;
                } else {
                    end = p;
                    while (end < e && b[end] != '\n' && b[end] != '\r') ++end;
                }
                if ((line_no & 3) == 1) {
                    const uint8_t* s = (const uint8_t*)b + p; size_t len = end - p;
                    while (len && py_space(s[0])) { ++s; --len; }
                    while (len && py_space(s[len - 1])) --len;
                    if (len > 0xfffffff0ull) { overflow = true; return; }
                    // a ring of RING sequences in flight: hashed now (slot line requested), entry line requested RING/3 sequences
                    // later, the representative's bytes after 2 RING/3, added to the table when the ring comes round
                    Pending& slot_ = ring[rn % RING];
                    if (rn >= RING) L.add_hashed(slot_.s, slot_.len, slot_.h);
                    slot_.s = s; slot_.len = (uint32_t)len; slot_.h = hash_bytes(s, len);
                    L.prefetch_slot(slot_.h);
                    if (rn >= RING / 3) L.prefetch_entry(ring[(rn - RING / 3) % RING].h);
                    if (rn >= 2 * RING / 3) L.prefetch_bytes(ring[(rn - 2 * RING / 3) % RING].h);
                    ++rn;
                    ++seqs;
                }
                ++line_no;
                if (end >= n) break;
                p = end + ((b[end] == '\r' && end + 1 < n && b[end + 1] == '\n') ? 2 : 1);
            }
            for (uint64_t q = rn > RING ? rn - RING : 0; q < rn; ++q) { const Pending& P_ = ring[q % RING]; L.add_hashed(P_.s, P_.len, P_.h); }   // drain, in order
            n_seq[t] = seqs;
        });
        if (overflow) { err = "a sequence line of 4 GiB"; return false; }
        lapse(1);
        // room in the global table for everything this chunk can add
        size_t incoming = 0;
        for (unsigned t = 0; t < T; ++t) { incoming += local[t].u.size() - local[t].first_new; n_reads += n_seq[t]; }
        const size_t need_entries = (size_t)n_entries.load() + incoming + (size_t)T * 257;
        if (need_entries >= 0xfffffff0ull) { err = "more than 2^32 - 2 unique sequences"; overflow = true; return false; }
        while (seg.size() << SEG_BITS < need_entries) seg.emplace_back(new StreamEntry[(size_t)1 << SEG_BITS]);
        if (need_entries * 2 > slots_cap && !grow_slots(need_entries * 3)) return false;
        const uint64_t gmask = slots_cap - 1;
        lapse(2);
        // C: what is new to a thread goes to the global table (created there, or found: another thread / an earlier chunk had it);
        // the counts every thread collected in this chunk are added to the entries
        pool->run([&](unsigned t) {
            LocalSet& L = local[t];
            uint32_t spare = C2_NO_ENTRY;
            uint32_t id_next = 0, id_end = 0;                           // entry ids are taken from the shared counter 256 at a time
            const size_t nu_ = L.u.size();
            for (size_t i = 0; i < nu_; ++i) {
                // (the global slot of a sequence that is new to this thread, and the entry behind it, are lines nobody has in cache: ask
                // for them a few sequences ahead)
                if (i + 8 < nu_ && L.u[i + 8].entry == C2_NO_ENTRY) __builtin_prefetch(&slots[L.u[i + 8].h & gmask]);
                if (i + 4 < nu_ && L.u[i + 4].entry == C2_NO_ENTRY) {
                    const uint32_t s4 = slots[L.u[i + 4].h & gmask].load(std::memory_order_relaxed);
                    if (s4) __builtin_prefetch(&entry(s4 - 1));
                } else if (i + 4 < nu_ && L.u[i + 4].pending) __builtin_prefetch(&entry(L.u[i + 4].entry));
                LocalSeq& q = L.u[i];
                if (!q.pending) continue;
                if (q.entry == C2_NO_ENTRY) {
                    const uint64_t key = ((uint64_t)t << 32) | (uint64_t)(i - L.first_new);
                    uint64_t p = q.h & gmask;
                    for (;;) {
                        uint32_t s = slots[p].load(std::memory_order_acquire);
                        if (s == 0) {
                            if (spare == C2_NO_ENTRY) {
                                if (id_next == id_end) { id_next = n_entries.fetch_add(256); id_end = id_next + 256; }
                                spare = id_next++;
                            }
                            StreamEntry& E = entry(spare);
                            E.h = q.h; E.len = q.len; E.src = q.p; E.chunk = cur; E.arena_off = 0; E.gidx = 0;
                            E.first.store(key, std::memory_order_relaxed); E.count.store(0, std::memory_order_relaxed);
                            uint32_t exp = 0;
                            if (slots[p].compare_exchange_strong(exp, spare + 1, std::memory_order_release, std::memory_order_acquire)) {
                                q.entry = spare; spare = C2_NO_ENTRY; break;
                            }
                            s = exp;
                        }
                        StreamEntry& E = entry(s - 1);
                        if (E.h == q.h && E.len == q.len && (q.len == 0 || memcmp(bytes_of(E), q.p, q.len) == 0)) { q.entry = s - 1; break; }
                        p = (p + 1) & gmask;
                    }
                    StreamEntry& E = entry(q.entry);
                    if (E.chunk == cur) {                            // first occurrence in file order = smallest (range, position)
                        uint64_t seen = E.first.load(std::memory_order_relaxed);
                        while (key < seen && !E.first.compare_exchange_weak(seen, key, std::memory_order_relaxed)) {}
                    }
                }
                entry(q.entry).count.fetch_add(q.pending, std::memory_order_relaxed);
                q.pending = 0;
            }
            if (spare != C2_NO_ENTRY) entry(spare).gidx = C2_NO_ENTRY;      // reserved, never published
            for (; id_next < id_end; ++id_next) entry(id_next).gidx = C2_NO_ENTRY;
        });
        lapse(3);
        // D: the unique reads this chunk adds, per range in file order: the thread's new sequences whose entry was created in
        // this chunk with THIS occurrence as its first
        std::vector<uint64_t> n_new(T + 1, 0), new_bytes(T + 1, 0);
        pool->run([&](unsigned t) {
            LocalSet& L = local[t];
            uint64_t c = 0, by = 0;
            for (size_t i = L.first_new; i < L.u.size(); ++i) {
                const StreamEntry& E = entry(L.u[i].entry);
                if (E.chunk == cur && E.first.load(std::memory_order_relaxed) == (((uint64_t)t << 32) | (uint64_t)(i - L.first_new))) { ++c; by += E.len; }
            }
            n_new[t] = c; new_bytes[t] = by;
        });
        lapse(4);
        const uint64_t g0 = offsets.size() - 1, a0 = offsets.back();
        uint64_t gs = g0, as = a0;
        for (unsigned t = 0; t < T; ++t) { const uint64_t c = n_new[t], by = new_bytes[t]; n_new[t] = gs; new_bytes[t] = as; gs += c; as += by; }
        if (gs >= 0xfffffffeull) { err = "more than 2^32 - 2 unique sequences"; overflow = true; return false; }
        offsets.resize(gs + 1);
        entry_of.resize(gs);
        arena.resize((size_t)as);
        lapse(7);
        // E: copy them into the arena
        pool->run([&](unsigned t) {
            LocalSet& L = local[t];
            uint64_t g = n_new[t], a = new_bytes[t];
            for (size_t i = L.first_new; i < L.u.size(); ++i) {
                StreamEntry& E = entry(L.u[i].entry);
                if (!(E.chunk == cur && E.first.load(std::memory_order_relaxed) == (((uint64_t)t << 32) | (uint64_t)(i - L.first_new)))) continue;
                if (E.len) memcpy(arena.data() + a, E.src, E.len);
                E.arena_off = a; E.gidx = (uint32_t)g;
                entry_of[g] = L.u[i].entry;
                a += E.len; ++g;
                offsets[g] = a;
            }
        });
        lapse(5);
        // F: the threads' tables point into the arena from now on (their buffers are refilled by the next chunk)
        pool->run([&](unsigned t) {
            LocalSet& L = local[t];
            for (size_t i = L.first_new; i < L.u.size(); ++i) L.u[i].p = arena.data() + entry(L.u[i].entry).arena_off;
        });
        lapse(6);
        // (the chunk is closed: entries created in it now answer from the arena)
        terms_before = before;
        pos = hi0;
        ++chunk_no;                                                   // (so that bytes_of() of this chunk's entries reads the arena)
        if (pos >= n) finish();
        return true;
    }

    void finish() {
        done = true;
        // lines in the text = terminators + (1 if it does not end with one and is not empty); a record whose sequence line never came
        // still counts, with the empty sequence (readline() returned ''), last in order unless an empty sequence was seen before
        bool ends_with_term = true;
        if (n > 0) {
            char last[2] = {0, 0};
            if (mem) last[0] = mem[n - 1];
            else if (pread(fd, last, 1, (off_t)(n - 1)) != 1) last[0] = '\n';
            ends_with_term = last[0] == '\n' || last[0] == '\r';
        }
        const uint64_t lines = terms_before + ((n > 0 && !ends_with_term) ? 1 : 0);
        if ((lines & 3) == 1) {
            ++n_reads;
            const uint64_t nu = offsets.size() - 1;
            for (uint64_t g = 0; g < nu; ++g)
                if (offsets[g + 1] == offsets[g]) { entry(entry_of[g]).count.fetch_add(1); return; }
            const uint32_t id = n_entries.fetch_add(1);
            while (seg.size() << SEG_BITS <= id) seg.emplace_back(new StreamEntry[(size_t)1 << SEG_BITS]);
            StreamEntry& E = entry(id);
            E.h = hash_bytes(nullptr, 0); E.len = 0; E.src = nullptr; E.chunk = 0; E.arena_off = offsets.back(); E.gidx = (uint32_t)nu;
            E.first.store(0); E.count.store(1);
            offsets.push_back(offsets.back());
            entry_of.push_back(id);
        }
    }

    // partner[g] = index of the unique read that equals reverse_complement(read g) (CRISPRessoShared.py:399-403: upper-cased, ACGTN_-
    // only; anything else is a KeyError there and gets -1 here), or -1 -- what c2_rc_partners computes, looked up in the table the
    // ingest already built (no second table, no second hash of every read), on the stream's own threads.
    void rc_partners_into(int64_t* partner) {
        const uint64_t nu = offsets.size() - 1;
        const uint64_t gmask = slots_cap - 1;
        // complement of an upper-cased base, 0 = not in ACGTN_- (lower-case letters map like their capitals: the reference upper-cases first)
        uint8_t comp[256];
        memset(comp, 0, sizeof comp);
        comp['A'] = 'T'; comp['C'] = 'G'; comp['G'] = 'C'; comp['T'] = 'A'; comp['N'] = 'N'; comp['_'] = '_'; comp['-'] = '-';
        comp['a'] = 'T'; comp['c'] = 'G'; comp['g'] = 'C'; comp['t'] = 'A'; comp['n'] = 'N';
        pool->run([&](unsigned t) {
            // the look-up of one read misses the cache three times (slot, entry, the candidate's bytes): RING reads are in flight, each
            // stage asks for the next line of a read a few positions behind
            constexpr unsigned RING = 12;
            struct Pend { std::vector<uint8_t> rc; uint64_t h; uint64_t g; bool ok; };
            Pend ring[RING];
            const uint64_t g0 = nu * t / T, g1 = nu * (t + 1) / T;
            auto finish_one = [&](const Pend& P) {
                partner[P.g] = -1;
                if (!P.ok) return;
                const size_t len = P.rc.size();
                for (uint64_t p = P.h & gmask;; p = (p + 1) & gmask) {
                    const uint32_t e = slots[p].load(std::memory_order_relaxed);
                    if (!e) break;
                    const StreamEntry& E = entry(e - 1);
                    if (E.h == P.h && E.len == len && E.gidx != C2_NO_ENTRY && (len == 0 || memcmp(arena.data() + E.arena_off, P.rc.data(), len) == 0)) {
                        partner[P.g] = (int64_t)E.gidx; break;
                    }
                }
            };
            uint64_t rn = 0;
            for (uint64_t g = g0; g < g1; ++g, ++rn) {
                Pend& P = ring[rn % RING];
                if (rn >= RING) finish_one(P);
                const uint8_t* s = arena.data() + offsets[g];
                const size_t len = (size_t)(offsets[g + 1] - offsets[g]);
                P.rc.resize(len);
                uint8_t bad = 0xff;
                for (size_t k = 0; k < len; ++k) { const uint8_t c = comp[s[len - 1 - k]]; P.rc[k] = c; bad &= (uint8_t)(c ? 0xff : 0); }
                P.ok = bad != 0 || len == 0; P.g = g;
                P.h = P.ok ? hash_bytes(P.rc.data(), len) : 0;
                if (P.ok) __builtin_prefetch(&slots[P.h & gmask]);
                if (rn >= RING / 3) {
                    const Pend& Q = ring[(rn - RING / 3) % RING];
                    if (Q.ok) { const uint32_t e = slots[Q.h & gmask].load(std::memory_order_relaxed); if (e) __builtin_prefetch(&entry(e - 1)); }
                }
                if (rn >= 2 * RING / 3) {
                    const Pend& Q = ring[(rn - 2 * RING / 3) % RING];
                    if (Q.ok) {
                        const uint32_t e = slots[Q.h & gmask].load(std::memory_order_relaxed);
                        if (e) { const StreamEntry& E = entry(e - 1); if (E.h == Q.h) for (uint32_t o = 0; o < E.len; o += 64) __builtin_prefetch(arena.data() + E.arena_off + o); }
                    }
                }
            }
            for (uint64_t q = rn > RING ? rn - RING : 0; q < rn; ++q) finish_one(ring[q % RING]);
        });
    }

    // multiplicities of the unique reads seen so far -> out[n_unique]; false if one exceeds 32 bits
    bool counts_into(uint32_t* out) {
        const uint64_t nu = offsets.size() - 1;
        bool ok = true;
        for (uint64_t g = 0; g < nu; ++g) {
            const uint64_t c = entry(entry_of[g]).count.load(std::memory_order_relaxed);
            if (c > 0xffffffffull) ok = false;
            out[g] = (uint32_t)c;
        }
        return ok;
    }
};
