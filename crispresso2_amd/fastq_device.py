"""FASTQ text -> the run's unique reads, framed and de-duplicated ON THE DEVICE (the readline loop of process_fastq,
reference CRISPResso2/CRISPRessoCORE.py:1825-1849: `variantCache[fastq_seq] += 1` per record).

Why: on the GPU box the host may use 16 CPUs; the native chunked parser (c2_fastq_stream) then needs 0.26-0.39 s for a 10 M-read file
whose 5.16 GB of text cross the link in 0.10 s.  Here the text is uploaded as it lies in the file (host threads only copy it into
pinned staging) and four kernels do what the parser does:

  c2_fq_count_kernel   per 16 KB tile: newlines, newlines that end an empty line (`grep -c .`), "a carriage return was seen"
  c2_fq_lines_kernel   with the prefix sum of those counts: where every record's sequence line starts and ends
  c2_fq_dedup_kernel   one wavefront per record: str.strip(), hash, look-up / insert in an open-addressing table in HBM whose keys ARE
                       the bytes of the first occurrence in the text (equality = byte comparison); occurrences and first record per key
  c2_fq_gather_kernel  the unique reads, in first-seen order, back to back: the arena the align kernels read

Nothing between the chunks waits for the device: the number of complete records is kept in device memory and the de-duplication
kernel reads its range from there.  The result is exactly c2_fastq_stream's (same reads, same order, same multiplicities, same
line statistics) for text without carriage returns, without quality filters, uncompressed; anything else -- and any of the kernels'
"I cannot" flags (a line of 16 MB, more records / keys than estimated) -- is DeviceIngestUnavailable, and the caller uses the host
parser.  The product has no CPU fallback for COMPUTE; this module is an ingest route, and the host parser is the other one."""
import ctypes
import os

import numpy as np

from . import _native
from .hostcopy import to_host

TILE = 16384
CHUNK_BYTES = int(os.environ.get("C2_FQ_DEVICE_CHUNK", 256 << 20))     # upper limit; a file is cut into at least ~8 chunks (chunk_bytes)
MIN_TEXT_BYTES = int(os.environ.get("C2_FQ_DEVICE_MIN", 64 << 20))      # below this the host parser is as fast and needs no table
MAX_TEXT_BYTES = 1 << 36                                                 # 64 GiB of text resident; beyond: the host parser
_pinned = {}
_upload_lock = None                     # (the pinned upload buffers are per device, not per call: one upload at a time)


class DeviceIngestUnavailable(Exception):
    """this file goes through the host parser (the reason is the message)"""


def usable_cpus():
    """the CPUs this process may use: cgroup quota (cpu.max), affinity"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def copy_threads():
    """host threads that copy the text into pinned memory.  Not all the CPUs the container may use: under a CFS quota a process that
    exceeds it is stopped for the rest of the 100 ms period -- every thread of it, also the one that enqueues the kernels"""
    if os.environ.get("C2_FQ_COPY_THREADS"):
        return max(1, int(os.environ["C2_FQ_COPY_THREADS"]))
    return max(2, min(16, usable_cpus() // 2))


# ---- the four launches (tests replace these with the wave emulator's entries) ----
def fq_count(ctx, d_text, lo, hi, d_tile_nl, d_tile_empty, d_flags, stream):
    import ctypes
    ctx.check(ctx.lib.c2_fq_count_device(ctx.handle, ctypes.c_void_p(d_text), ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_void_p(d_tile_nl),
                                         ctypes.c_void_p(d_tile_empty), ctypes.c_void_p(d_flags), ctypes.c_void_p(stream)), "c2_fq_count_device")


def fq_lines(ctx, d_text, lo, hi, d_tile_base, d_seq_start, d_seq_end, cap, stream):
    import ctypes
    ctx.check(ctx.lib.c2_fq_lines_device(ctx.handle, ctypes.c_void_p(d_text), ctypes.c_uint64(lo), ctypes.c_uint64(hi), ctypes.c_void_p(d_tile_base),
                                         ctypes.c_void_p(d_seq_start), ctypes.c_void_p(d_seq_end), ctypes.c_uint64(cap), ctypes.c_void_p(stream)),
              "c2_fq_lines_device")


def fq_lines4(ctx, d_text, lo, hi, d_tile_base, d_seq_start, d_seq_end, d_qual_start, d_qual_end, cap, stream):
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_fq_lines4_device(ctx.handle, V(d_text), ctypes.c_uint64(lo), ctypes.c_uint64(hi), V(d_tile_base), V(d_seq_start), V(d_seq_end),
                                          V(d_qual_start), V(d_qual_end), ctypes.c_uint64(cap), V(stream or 0)), "c2_fq_lines4_device")


def fq_pair_lengths(ctx, d_text1, d_text2, lines1, lines2, n, d_s1, d_q1, d_s2, d_q2, d_key_len, d_qual_len, d_flags, stream):
    """lines1 / lines2: the four device addresses (seq_start, seq_end, qual_start, qual_end) of a text's line arrays"""
    V = ctypes.c_void_p
    L1, L2 = (V * 4)(*[V(x) for x in lines1]), (V * 4)(*[V(x) for x in lines2])
    ctx.check(ctx.lib.c2_fq_pair_lengths_device(ctx.handle, V(d_text1), V(d_text2), L1, L2, ctypes.c_uint64(n), V(d_s1), V(d_q1), V(d_s2), V(d_q2),
                                                V(d_key_len), V(d_qual_len), V(d_flags), V(stream or 0)), "c2_fq_pair_lengths_device")


def fq_pair_write(ctx, d_text1, d_text2, n, d_s1, d_q1, d_s2, d_q2, d_key_off, d_qual_off, d_key_out, d_qual_out, d_flags, stream):
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_fq_pair_write_device(ctx.handle, V(d_text1), V(d_text2), ctypes.c_uint64(n), V(d_s1), V(d_q1), V(d_s2), V(d_q2), V(d_key_off),
                                              V(d_qual_off), V(d_key_out), V(d_qual_out), V(d_flags), V(stream or 0)), "c2_fq_pair_write_device")


def fq_dedup(ctx, d_text, d_seq_start, d_seq_end, d_range, cap, d_slots, n_slots, d_count, d_first, d_slot_of, d_rinfo, d_flags, d_n_unique, stream):
    import ctypes
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_fq_dedup_device(ctx.handle, V(d_text), V(d_seq_start), V(d_seq_end), V(d_range), ctypes.c_uint64(cap), V(d_slots),
                                         ctypes.c_uint64(n_slots), V(d_count), V(d_first), V(d_slot_of), V(d_rinfo), V(d_flags), V(d_n_unique),
                                         V(stream)), "c2_fq_dedup_device")


def fq_gather(ctx, d_text, d_info, d_records, d_out_offsets, d_out, n, stream):
    import ctypes
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_fq_gather_device(ctx.handle, V(d_text), V(d_info), V(d_records or 0), V(d_out_offsets), V(d_out), ctypes.c_uint64(n),
                                          V(stream)), "c2_fq_gather_device")


def fq_rc_partner(ctx, d_text, d_info, d_records, n, d_slots, n_slots, d_partner_slot, stream):
    import ctypes
    V = ctypes.c_void_p
    ctx.check(ctx.lib.c2_fq_rc_partner_device(ctx.handle, V(d_text), V(d_info), V(d_records), ctypes.c_uint64(n), V(d_slots), ctypes.c_uint64(n_slots),
                                              V(d_partner_slot), V(stream)), "c2_fq_rc_partner_device")


def gather_reads_device(ctx, d_reads, d_off, idx, dev, stream):
    """the reads idx[...] of a device arena (d_off: int64 [n + 1]) back to back -> (uint8 tensor, int64 offsets tensor, longest read);
    _native.gather_reads for reads that never were on the host"""
    import torch
    d_idx = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(dev)
    lens = d_off[d_idx + 1] - d_off[d_idx]
    if int(lens.max().item()) >= (1 << 24):
        raise _native.NativeError("gather_reads_device: a read of 2^24 bytes or more")
    info = (d_off[d_idx] << 24) | lens
    out_off = torch.zeros(len(idx) + 1, dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=out_off[1:])
    out = torch.empty(max(int(out_off[-1].item()), 1), dtype=torch.uint8, device=dev)
    fq_gather(ctx, d_reads.data_ptr(), info.data_ptr(), None, out_off.data_ptr(), out.data_ptr(), len(idx), stream)
    return out, out_off, int(lens.max().item())


def applicable(path, filters=(0, 0, 0)):
    """-> None if the device route can take this FILE as it lies on disk, else why not (a reason that starts with "in memory:" means
    the host has to inflate / filter first and the device can frame the text it then holds: text_applicable)"""
    if os.environ.get("C2_FQ_INGEST", "auto") == "host":
        return "C2_FQ_INGEST=host"
    try:
        size = os.path.getsize(path)
        with open(path, "rb") as fh:
            magic = fh.read(2)
    except OSError as e:
        return str(e)
    if any(filters):
        return "in memory: the quality filters run on the host"
    if magic == b"\x1f\x8b" or str(path).endswith(".gz"):
        return "in memory: compressed text is inflated on the host"
    if size < MIN_TEXT_BYTES and os.environ.get("C2_FQ_INGEST", "auto") != "device":
        return "small file"
    if size > MAX_TEXT_BYTES or size == 0:
        return "text of %d bytes" % size
    return None


def size_applicable(size):
    """-> None if the device route takes a text of `size` bytes, else why not"""
    if os.environ.get("C2_FQ_INGEST", "auto") == "host":
        return "C2_FQ_INGEST=host"
    if size < MIN_TEXT_BYTES and os.environ.get("C2_FQ_INGEST", "auto") != "device":
        return "small text"
    if size > MAX_TEXT_BYTES or size == 0:
        return "text of %d bytes" % size
    return None


def text_applicable(text):
    """-> None if the device route can take this text (a uint8 array in host memory), else why not"""
    if text is None:
        return "the text is not in memory"
    return size_applicable(int(text.size))


class DeviceIngest:
    """feed(lo, hi) after bytes [lo, hi) of the text are in d_text (in order, lo a multiple of 16; the last chunk may end anywhere);
    finish() once everything was fed.  All work is enqueued on torch's current stream of `dev`; nothing waits for the device before
    finish() except poll(), which only waits for chunks fed `lag` chunks ago.
    Batches: take_batch(r1, m, max_len) hands out the m unique non-empty reads whose FIRST occurrence is a record in
    [records handed out so far, r1) -- final as soon as those records are de-duplicated, because later records have larger numbers --
    so the caller can align them while later chunks are still being uploaded."""

    def __init__(self, ctx, dev, text_bytes, est_records, d_text=None):
        import collections
        import torch
        self.ctx, self.dev, self.T = ctx, dev, int(text_bytes)
        self.cap = int(est_records)
        if self.cap >= (1 << 31) - 2:
            raise DeviceIngestUnavailable("more than 2^31 records expected")
        n_slots = 1 << 12
        while n_slots < 2 * self.cap:
            n_slots <<= 1
        if n_slots > (1 << 30):
            raise DeviceIngestUnavailable("table of more than 2^30 slots")
        self.n_slots = n_slots
        i64, i32 = torch.int64, torch.int32
        self.d_text = d_text if d_text is not None else torch.empty(max(self.T, 1), dtype=torch.uint8, device=dev)
        self.seq_start = torch.full((self.cap,), self.T, dtype=i64, device=dev)     # (a record whose newline never comes: its line
        self.seq_end = torch.full((self.cap,), self.T, dtype=i64, device=dev)       #  starts / ends where the text ends)
        self.slot_of = torch.empty(self.cap, dtype=i32, device=dev)
        self.rinfo = torch.empty(self.cap, dtype=i64, device=dev)
        self.slots = torch.zeros(n_slots, dtype=i64, device=dev)
        self.count = torch.zeros(n_slots, dtype=i32, device=dev)
        self.first = torch.full((n_slots,), -1, dtype=i32, device=dev)              # 0xffffffff
        self.flags = torch.zeros(1, dtype=i32, device=dev)
        self.stats = torch.zeros(4, dtype=i32, device=dev)                          # keys, longest key, empty keys (c2_fq_dedup_args.stats)
        self.newlines = torch.zeros(1, dtype=i64, device=dev)                       # in the text so far
        self.empty_lines = torch.zeros(1, dtype=i64, device=dev)
        self.range = torch.zeros(2, dtype=i64, device=dev)                          # records de-duplicated so far: [., range[1])
        self.fed = 0
        self.batch_r0 = 0                                                           # records below this were handed out in batches
        self.batch_u0 = 0                                                           # ... and so many unique non-empty reads
        self.batches = []                                                           # (rec, d_off) of every batch handed out
        self._snaps = collections.deque()
        self._ring = None
        self._ring_at = 0

    def _stream(self):
        import torch
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _dedup_to(self, r1):
        import torch
        self.range = torch.cat([self.range[1:2], r1.reshape(1)])
        fq_dedup(self.ctx, self.d_text.data_ptr(), self.seq_start.data_ptr(), self.seq_end.data_ptr(), self.range.data_ptr(), self.cap,
                 self.slots.data_ptr(), self.n_slots, self.count.data_ptr(), self.first.data_ptr(), self.slot_of.data_ptr(), self.rinfo.data_ptr(),
                 self.flags.data_ptr(), self.stats.data_ptr(), self._stream())

    def feed(self, lo, hi):
        import torch
        if lo != self.fed or lo % 16 or hi > self.T or hi <= lo:
            raise ValueError("chunks come in order, from a multiple of 16")
        self.fed = hi
        tiles = (hi - lo + TILE - 1) // TILE
        tile_nl = torch.empty(tiles, dtype=torch.int32, device=self.dev)
        tile_em = torch.empty(tiles, dtype=torch.int32, device=self.dev)
        s = self._stream()
        fq_count(self.ctx, self.d_text.data_ptr(), lo, hi, tile_nl.data_ptr(), tile_em.data_ptr(), self.flags.data_ptr(), s)
        nl = tile_nl.to(torch.int64)
        upto = torch.cumsum(nl, 0)
        base = (upto - nl) + self.newlines
        fq_lines(self.ctx, self.d_text.data_ptr(), lo, hi, base.data_ptr(), self.seq_start.data_ptr(), self.seq_end.data_ptr(), self.cap, s)
        self.newlines = self.newlines + upto[-1:]
        self.empty_lines = self.empty_lines + tile_em.sum(dtype=torch.int64).reshape(1)
        # a record is complete once the newline behind its sequence line (number 4 r + 1) was seen
        self._dedup_to((self.newlines + 2) // 4)

    # ---- what the host may know without waiting ----
    def mark(self):
        """after feed(): (records de-duplicated, keys, longest key, empty keys, flags) as of this chunk start their way to the host"""
        import torch
        vals = torch.cat([self.range[1:2], self.stats[:3].to(torch.int64), self.flags.to(torch.int64)])
        if self.dev.type != "cuda":
            self._snaps.append((None, vals.clone()))
            return
        if self._ring is None:
            self._ring = torch.empty((64, 5), dtype=torch.int64, pin_memory=True)
        if len(self._snaps) >= 60:
            self._snaps.popleft()[0].synchronize()
        row = self._ring[self._ring_at % 64]
        self._ring_at += 1
        row.copy_(vals, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._snaps.append((ev, row))

    def poll(self, lag=2):
        """-> the newest (records, unique non-empty reads, longest) that reached the host, or None; waits only for marks older than `lag`"""
        got = None
        while self._snaps and (len(self._snaps) > lag or self._snaps[0][0] is None or self._snaps[0][0].query()):
            ev, row = self._snaps.popleft()
            if ev is not None:
                ev.synchronize()
            got = [int(x) for x in row.tolist()]
        if got is None:
            return None
        r1, keys, longest, empty_keys, flags = got
        self._check(flags, r1, keys)
        return r1, keys - empty_keys, longest

    def _check(self, flags, r1, keys):
        if flags & 1:
            raise DeviceIngestUnavailable("carriage returns in the text")
        if flags & ~1:
            raise DeviceIngestUnavailable("device ingest gave up (flags %d: 2 = line too long, 4 = more records than estimated)" % flags)
        if r1 > self.cap or 2 * keys > self.n_slots:
            raise DeviceIngestUnavailable("more records than estimated")

    def take_batch(self, r1, m, max_len, may_wait=False):
        """-> (d_reads uint8 [m * max_len reserved], d_off int64 [m + 1]) of the m unique non-empty reads first seen in records [batch_r0, r1)"""
        import torch
        dev, ra = self.dev, self.batch_r0
        idx = torch.arange(ra, r1, dtype=torch.int64, device=dev)
        slot = self.slot_of[ra:r1].to(torch.int64).clamp_(min=0)
        info = self.rinfo[ra:r1]
        mask = (self.first[slot].to(torch.int64) == idx) & ((info & 0xffffff) > 0)
        # first occurrences, in file order; their number is known (m), so nothing waits for the device here
        # (may_wait: the caller waits for the device anyway -- torch.nonzero, which does, is the faster of the two)
        rec = (torch.nonzero(mask) if may_wait else torch.nonzero_static(mask, size=m, fill_value=0)).reshape(-1) + ra
        if int(rec.numel()) != m:
            raise _native.NativeError("device ingest: %d first occurrences where the counters say %d" % (int(rec.numel()), m))
        lens = self.rinfo[rec] & 0xffffff
        d_off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        torch.cumsum(lens, 0, out=d_off[1:])
        d_reads = torch.empty(max(m * max_len, 1), dtype=torch.uint8, device=dev)
        fq_gather(self.ctx, self.d_text.data_ptr(), self.rinfo.data_ptr(), rec.data_ptr(), d_off.data_ptr(), d_reads.data_ptr(), m, self._stream())
        self.batch_r0, self.batch_u0 = r1, self.batch_u0 + m
        self.batches.append((rec, d_off))
        return d_reads, d_off

    def finish(self, unterminated, on_batch=None):
        """unterminated: the text does not end with a newline (its last line still counts).  Hands out the last batch (everything, if
        none was taken before) and -> dict(offsets, counts, n_reads, nonempty_lines, n_unique, max_len, min_len, ...); waits for the device.
        on_batch(m, d_reads, d_off, max_len): called for the last batch, as the caller did for the earlier ones."""
        import time
        import torch
        if self.fed != self.T:
            raise ValueError("%d of %d bytes were fed" % (self.fed, self.T))
        trace = [] if os.environ.get("C2_FQ_TRACE") else None

        def lap(what):
            if trace is not None:
                if self.dev.type == "cuda":
                    torch.cuda.synchronize(self.dev)
                trace.append((what, time.perf_counter()))
        lap("queue drained")
        lines = self.newlines + (1 if unterminated else 0)
        self._dedup_to((lines + 3) // 4)                                            # readline() loop: every started group of four lines is a record
        self._snaps.clear()
        n_records, keys, longest, empty_keys, flags = [int(x) for x in torch.cat([self.range[1:2], self.stats[:3].to(torch.int64),
                                                                               self.flags.to(torch.int64)]).tolist()]
        self._check(flags, n_records, keys)
        lap("last records de-duplicated")
        nonempty = int(lines.item()) - int(self.empty_lines.item())
        m = keys - empty_keys - self.batch_u0
        last = None
        if m > 0:
            last = self.take_batch(n_records, m, longest, may_wait=True)
            if on_batch is not None:
                on_batch(m, last[0], last[1], longest)
        lap("last batch handed out")
        n = self.batch_u0
        n_empty = 0
        if empty_keys:
            first_empty = torch.nonzero((self.rinfo[:n_records] & 0xffffff) == 0)[:1].reshape(-1)
            n_empty = int(self.count[self.slot_of[first_empty].to(torch.int64)].sum().item())
        if self.batches:
            rec = torch.cat([b[0] for b in self.batches])
            lens = torch.cat([b[1][1:] - b[1][:-1] for b in self.batches])
            d_counts = self.count[self.slot_of[rec].to(torch.int64)].contiguous()          # (stays on the device for the selection kernel)
            counts = to_host(d_counts, np.int64)                                          # (32-bit over the link, widened on the host)
            lens_h = to_host(lens.to(torch.int32), np.int64)
        else:
            counts, lens_h, d_counts = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), None
        # which unique read is the reverse complement of which (the count merge asks): looked up in the table, which is still here
        rc_partner = np.zeros(0, dtype=np.int64)
        d_rc_partner = None
        if self.batches:
            slot_u = self.slot_of[rec].to(torch.int64)
            pslot = torch.empty(n, dtype=torch.int32, device=self.dev)
            fq_rc_partner(self.ctx, self.d_text.data_ptr(), self.rinfo.data_ptr(), rec.data_ptr(), n, self.slots.data_ptr(), self.n_slots,
                          pslot.data_ptr(), self._stream())
            unique_of_slot = torch.full((self.n_slots,), -1, dtype=torch.int64, device=self.dev)
            unique_of_slot[slot_u] = torch.arange(n, dtype=torch.int64, device=self.dev)
            d_rc_partner = torch.where(pslot >= 0, unique_of_slot[pslot.to(torch.int64).clamp_(min=0)], -1)     # (stays on the device for the count transfer)
            rc_partner = to_host(d_rc_partner.to(torch.int32), np.int64)
        lap("multiplicities and lengths on the host")
        off64 = np.zeros(n + 1, dtype=np.int64)
        lap("zeros")
        np.cumsum(lens_h, out=off64[1:])
        lap("cumsum")
        offsets = off64.view(np.uint64)
        mx, mn = (int(lens_h.max()), int(lens_h.min())) if n else (0, 0)
        lap("max min")
        ends = np.cumsum([int(b[0].numel()) for b in self.batches], dtype=np.int64)
        bb = [int(x) for x in np.diff(off64[np.concatenate([[0], ends])])] if self.batches else []      # bytes of every batch
        lap("batch bytes")
        out = dict(offsets=offsets, counts=counts, n_reads=n_records, n_empty_records=n_empty, nonempty_lines=nonempty, n_unique=n,
                   max_len=mx, min_len=mn, batch_bytes=bb, rc_partner=rc_partner, d_counts=d_counts, d_rc_partner=d_rc_partner)
        if last is not None and len(self.batches) == 1:
            out["d_reads"], out["d_off"] = last
        elif not self.batches:                                                      # (no non-empty sequence line at all: an empty arena, as _device_front makes)
            out["d_reads"], out["d_off"] = torch.zeros(1, dtype=torch.uint8, device=self.dev), torch.zeros(1, dtype=torch.int64, device=self.dev)
        lap("offsets")
        if trace is not None:
            out["finish_trace_ms"] = {w: round((t - trace[i][1]) * 1e3, 3) for i, (w, t) in enumerate(trace[1:])}
        return out


def chunk_bytes(size):
    """bytes per upload: large chunks keep the per-chunk host work (a dozen torch launches, a hand-over between two Python threads)
    away from the link -- 256 MB chunks upload 5 GB a third faster than 64 MB ones -- but a file should still be several chunks, so
    that the device works under the upload"""
    want = min(CHUNK_BYTES, max(16 << 20, size // 8))
    return max(TILE, want // TILE * TILE)


def estimate_records(source, size):
    """records the text is expected to hold, from the line density of its first MB, with a quarter of slack"""
    if isinstance(source, np.ndarray):
        head = source[:1 << 20].tobytes()
    else:
        with open(source, "rb") as fh:
            head = fh.read(1 << 20)
    nl = head.count(b"\n")
    per_byte = (nl + 1) / max(len(head), 1)
    return int(size * per_byte / 4.0 * 1.25) + 4096


def ingest_file(path, ctx, dev, timings=None, on_batch=None, min_batch=200_000):
    """_ingest_file; running out of device memory (the whole text, 40 bytes per record and 16 per table slot are resident) is
    DeviceIngestUnavailable like the kernels' own "I cannot": the caller takes the host parser, which holds only the unique reads."""
    import torch
    try:
        return _ingest_file(path, ctx, dev, timings, on_batch, min_batch)
    except torch.OutOfMemoryError as e:
        if dev.type == "cuda":
            torch.cuda.empty_cache()
        raise DeviceIngestUnavailable("out of device memory (%s)" % str(e).split("\n")[0])


def release_pinned():
    """give the page-locked upload buffers back (3 x the chunk size per device, at most 768 MiB; kept between runs because pinning them costs
    as much as uploading a GB).  A process that ingests one file and then lives on can call this; hostcopy.release_staging() is its twin."""
    _pinned.clear()


PINNED_CACHE_BYTES = int(os.environ.get("C2_FQ_PINNED_CACHE", 1 << 30))    # the upload buffers are kept between runs up to this size (per device); beyond, freed at once


def _ingest_file(path, ctx, dev, timings=None, on_batch=None, min_batch=200_000):
    """path: a plain FASTQ file, the text itself as a uint8 array in host memory (what the host inflated / filtered), or a
    _native.BgzfFile (its members are inflated chunk by chunk straight into the pinned upload buffers).
    The whole text -> DeviceIngest.finish()'s dict.  Host threads copy the text into three pinned buffers in turn; every chunk is framed
    and de-duplicated on the compute stream while the next one is copied and uploaded.
    on_batch(m, d_reads, d_off, max_len): called (on this thread, with the compute stream current) whenever min_batch new unique reads
    are final, and for the rest at the end -- the caller enqueues their alignments behind the de-duplication, under the upload.
    Without it the result carries one batch: d_reads / d_off of all unique reads."""
    import time
    import torch
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()
    text = path if isinstance(path, np.ndarray) else None
    bgzf = path if isinstance(path, _native.BgzfFile) else None
    size = int(text.size) if text is not None else bgzf.text_bytes if bgzf is not None else os.path.getsize(path)
    on_gpu = dev.type == "cuda"
    chunk = chunk_bytes(size)
    if bgzf is not None:
        # chunks are whole members: [cuts[c], cuts[c + 1]) in blocks, ~chunk bytes of text each (a member holds at most 64 KiB)
        offs = bgzf.text_offsets.astype(np.int64)
        cuts = [0]
        while cuts[-1] < bgzf.n_blocks:
            cuts.append(max(cuts[-1] + 1, int(np.searchsorted(offs, offs[cuts[-1]] + chunk - 65536, side="right")) - 1))
        cuts[-1] = min(cuts[-1], bgzf.n_blocks)
        chunk = int(max(offs[b] - offs[a] for a, b in zip(cuts[:-1], cuts[1:]))) if size else chunk
        head = np.empty(min(size, 1 << 20), dtype=np.uint8)
        n_head = int(np.searchsorted(offs, head.size, side="left"))             # the members that cover the first MB (for the table's size)
        if size:
            big = np.empty(int(offs[n_head]), dtype=np.uint8)
            bgzf.inflate(0, n_head, big.ctypes.data, big.size, 1)
            head = big[:head.size]
        ing = DeviceIngest(ctx, dev, size, estimate_records(head, size))
    else:
        ing = DeviceIngest(ctx, dev, size, estimate_records(path, size))

    def after_feed():
        if on_batch is None:
            return
        ing.mark()
        snap = ing.poll()
        if snap is not None:
            r1, uniq, longest = snap
            if uniq - ing.batch_u0 >= min_batch and r1 > ing.batch_r0:
                m = uniq - ing.batch_u0
                d_reads, d_off = ing.take_batch(r1, m, longest)
                on_batch(m, d_reads, d_off, longest)
    fd = os.open(path, os.O_RDONLY) if (text is None and bgzf is None) else -1
    held = []
    try:
        if bgzf is not None:
            spans = [(int(bgzf.text_offsets[a]), int(bgzf.text_offsets[b]), a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            last = None                                               # (known once the last member is inflated)
        else:
            spans = [(lo, min(size, lo + chunk), 0, 0) for lo in range(0, size, chunk)]
            last = os.pread(fd, 1, size - 1) if text is None else text[size - 1:].tobytes()

        def feed_upto(up_hi):
            """the text is resident up to up_hi: frame what is new of it, up to a multiple of 16 (all of it at the end)"""
            hi = up_hi if up_hi == size else up_hi & ~15
            if hi > ing.fed:
                ing.feed(ing.fed, hi)
                after_feed()
        if not on_gpu:                                                # (tests: the "device" is host memory)
            view = ing.d_text.numpy()
            for lo, hi, b0, b1 in spans:
                if bgzf is not None:
                    bgzf.inflate(b0, b1, view[lo:hi].ctypes.data, hi - lo, 2)
                elif text is not None:
                    view[lo:hi] = text[lo:hi]
                elif os.preadv(fd, [memoryview(view[lo:hi])], lo) != hi - lo:
                    raise OSError("short read")
                feed_upto(hi)
            if bgzf is not None:
                last = view[size - 1:size].tobytes()
            return ing.finish(last != b"\n", on_batch)
        import queue
        import threading
        global _upload_lock
        if _upload_lock is None:
            _upload_lock = threading.Lock()
        _upload_lock.acquire()
        held.append(True)
        threads = copy_threads()
        key = dev.index
        if key not in _pinned or _pinned[key][0].numel() < chunk:
            _pinned[key] = None
            _pinned[key] = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
        pins = _pinned[key]
        compute = torch.cuda.current_stream(dev)
        copy_stream = torch.cuda.Stream(device=dev)
        q = queue.Queue()
        tail_byte = []
        stop = threading.Event()                                      # set when the consumer gave up: the producer ends after the chunk in hand

        def read_into(args):
            buf, off, n = args
            if text is not None:
                np.copyto(np.frombuffer(buf, dtype=np.uint8, count=n), text[off:off + n])
                return
            while n:
                got = os.preadv(fd, [buf[:n]], off)
                if got <= 0:
                    raise OSError("short read")
                buf, off, n = buf[got:], off + got, n - got

        def producer():
            """file -> pinned buffers (all host threads) -> device, chunk after chunk; this thread only waits and enqueues copies, so
            the thread that enqueues kernels is never in its way"""
            evs = [None] * len(pins)
            try:
                with ThreadPoolExecutor(threads) as pool:
                    for c, (lo, hi, b0, b1) in enumerate(spans):
                        if stop.is_set():
                            break
                        k = c % len(pins)
                        if evs[k] is not None:
                            evs[k].synchronize()                      # the upload out of this buffer is done
                        if bgzf is not None:                          # members b0 .. b1 - 1 inflated straight into the pinned buffer (native threads)
                            bgzf.inflate(b0, b1, pins[k].data_ptr(), hi - lo, max(2, usable_cpus() - 2))
                            if hi == size:
                                tail_byte.append(bytes(pins[k][hi - lo - 1:hi - lo].numpy()))
                        else:
                            mv = memoryview(pins[k].numpy())
                            step = -(-(hi - lo) // threads)
                            step = (step + 4095) // 4096 * 4096
                            list(pool.map(read_into, [(mv[a_:min(hi - lo, a_ + step)], lo + a_, min(hi - lo, a_ + step) - a_) for a_ in range(0, hi - lo, step)]))
                        with torch.cuda.stream(copy_stream):
                            ing.d_text[lo:hi].copy_(pins[k][:hi - lo], non_blocking=True)
                            evs[k] = torch.cuda.Event()
                            evs[k].record(copy_stream)
                        q.put((lo, hi, evs[k]))
                q.put(None)
            except BaseException as e:
                q.put(e)
        th = threading.Thread(target=producer, name="c2-fastq-upload")
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                lo, hi, ev = item
                compute.wait_event(ev)
                feed_upto(hi)
        except BaseException:
            stop.set()                                                # (a carriage return, more records than estimated, ...: no point in uploading the rest)
            raise
        finally:
            th.join()
            if 3 * chunk > PINNED_CACHE_BYTES:
                _pinned.pop(key, None)
        if timings is not None:
            timings["upload_text"] = time.perf_counter() - t0
        if bgzf is not None:
            last = tail_byte[0]
        out = ing.finish(last != b"\n", on_batch)
        if timings is not None:
            timings["device_dedup_tail"] = time.perf_counter() - t0 - timings["upload_text"]
        return out
    finally:
        if held:
            _upload_lock.release()
        if fd >= 0:
            os.close(fd)


# ---- sharded ingest: every rank uploads, frames and de-duplicates ITS byte range of the text; the ranks then reconcile ----
SHARD_OVERLAP = int(os.environ.get("C2_FQ_SHARD_OVERLAP", 1 << 20))     # bytes behind a rank's range it also uploads: the line that starts in the range ends in them


def _all_gather_i64(values, dev):
    """list of ints of this rank -> int64 [world, len(values)] on the host"""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def _all_gather_var(t, sizes, dev):
    """1-D tensors of different lengths (sizes: every rank's length) -> their concatenation in rank order, on every rank"""
    import torch
    import torch.distributed as dist
    mx = int(max(sizes)) if len(sizes) else 0
    if mx == 0:
        return torch.zeros(0, dtype=t.dtype, device=dev)
    pad = torch.zeros(mx, dtype=t.dtype, device=dev)
    pad[:t.numel()] = t
    parts = [torch.empty_like(pad) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, pad)
    return torch.cat([p_[:int(n)] for p_, n in zip(parts, sizes)])


def _read_span(source, lo, hi, dst):
    """bytes [lo, hi) of the text -> dst (a uint8 numpy array of hi - lo bytes): a plain file, the text in host memory, or a BGZF file
    (only the members that cover the span are inflated)"""
    n = hi - lo
    if n <= 0:
        return
    if isinstance(source, np.ndarray):
        dst[:n] = source[lo:hi]
    elif isinstance(source, _native.BgzfFile):
        offs = source.text_offsets.astype(np.int64)
        b0 = int(np.searchsorted(offs, lo, side="right")) - 1
        b1 = int(np.searchsorted(offs, hi, side="left"))
        a0, a1 = int(offs[b0]), int(offs[b1])
        if a0 == lo and a1 == hi:
            source.inflate(b0, b1, dst.ctypes.data, n, max(2, usable_cpus() - 2))
        else:
            tmp = np.empty(a1 - a0, dtype=np.uint8)
            source.inflate(b0, b1, tmp.ctypes.data, tmp.size, max(2, usable_cpus() - 2))
            dst[:n] = tmp[lo - a0:hi - a0]
    else:
        fd = os.open(source, os.O_RDONLY)
        try:
            mv = memoryview(dst)[:n]
            got_all = 0
            while got_all < n:
                got = os.preadv(fd, [mv[got_all:]], lo + got_all)
                if got <= 0:
                    raise OSError("short read")
                got_all += got
        finally:
            os.close(fd)


def ingest_shard(source, ctx, dev, timings=None):
    """The sharded form of ingest_file (torch.distributed initialised, one process per GPU, a collective: every rank calls it with the
    same source).  Rank r takes bytes [r * per, (r + 1) * per) of the text (per: a multiple of the framing tile): it uploads them (plus
    SHARD_OVERLAP bytes behind them and 16 in front), counts their newlines, learns from an all-gather how many lines lie in front of its
    range -- records are four lines from the TOP of the file -- frames the records whose id line ends in its range and de-duplicates them
    on its device.  The ranks then all-gather their unique reads (bytes, multiplicities; in rank order, each rank's in first-seen order:
    that IS the file's first-seen order with repeats) and every rank de-duplicates the gathered list once more, adding the multiplicities up:
    the same global list, multiplicities and reverse-complement partners on every rank -- what ingest_file returns for the whole file
    (DeviceIngest.finish()'s dict; "shard_bytes": what this rank uploaded).  Raises DeviceIngestUnavailable on EVERY rank if any of them
    meets something the kernels cannot take (a carriage return, a line longer than the overlap, ...)."""
    import time
    import torch
    import torch.distributed as dist
    t0 = time.perf_counter()
    rank, world = dist.get_rank(), dist.get_world_size()
    size = int(source.size) if isinstance(source, np.ndarray) else source.text_bytes if isinstance(source, _native.BgzfFile) else os.path.getsize(source)
    per = max(TILE, -(-(-(-size // world)) // TILE) * TILE)
    lo, hi = min(size, rank * per), min(size, (rank + 1) * per)
    up_lo, up_hi = max(0, lo - 16), min(size, hi + SHARD_OVERLAP)
    i64, i32 = torch.int64, torch.int32
    why = None
    state = {}
    try:
        span = np.empty(max(up_hi - up_lo, 1), dtype=np.uint8)
        _read_span(source, up_lo, up_hi, span)
        d_span = torch.from_numpy(span).to(dev) if dev.type != "cuda" else _upload(span, dev)
        text_ptr = d_span.data_ptr() - up_lo                        # the kernels address the text by absolute position
        s = torch.cuda.current_stream(dev).cuda_stream
        flags = torch.zeros(1, dtype=i32, device=dev)
        n_tiles = max(1, (up_hi - lo + TILE - 1) // TILE)
        own_tiles = (hi - lo + TILE - 1) // TILE
        tile_nl = torch.zeros(n_tiles, dtype=i32, device=dev)
        tile_em = torch.zeros(n_tiles, dtype=i32, device=dev)
        if up_hi > lo:
            fq_count(ctx, text_ptr, lo, up_hi, tile_nl.data_ptr(), tile_em.data_ptr(), flags.data_ptr(), s)
        # (the count kernel's tiles run from lo on: the first own_tiles of them are this rank's range; hi - lo is a multiple of the tile
        #  except at the end of the text, where the overlap is empty)
        own_nl = int(tile_nl[:own_tiles].sum(dtype=i64).item()) if own_tiles else 0
        own_em = int(tile_em[:own_tiles].sum(dtype=i64).item()) if own_tiles else 0
        last_byte = int(span[size - 1 - up_lo]) if (size and up_lo <= size - 1 < up_hi) else -1
        state.update(d_span=d_span, text_ptr=text_ptr, s=s, flags=flags, tile_nl=tile_nl, own_nl=own_nl, own_em=own_em, last_byte=last_byte)
        if int(flags.item()) & 1:
            why = "carriage returns in the text"
    except Exception as e:                                             # (whatever it is: the vote below must still happen on this rank)
        why = "%s: %s" % (type(e).__name__, e)
    # ---- everybody learns everybody's line counts (and whether anybody gave up)
    g = _all_gather_i64([state.get("own_nl", 0), state.get("own_em", 0), state.get("last_byte", -1), 1 if why else 0], dev)
    if g[:, 3].any():
        raise DeviceIngestUnavailable(why or "another rank gave up on the device ingest")
    nl_before = int(g[:rank, 0].sum())
    total_nl, total_em = int(g[:, 0].sum()), int(g[:, 1].sum())
    lasts = [int(x) for x in g[:, 2] if x >= 0]
    unterminated = bool(size) and (not lasts or lasts[-1] != 0x0a)
    lines = total_nl + (1 if unterminated else 0)
    n_records = (lines + 3) // 4                                     # readline() loop: every started group of four lines is a record
    rec_lo = (nl_before + 3) // 4
    rec_hi = n_records if rank == world - 1 else min(n_records, (nl_before + state["own_nl"] + 3) // 4)
    rec_hi = max(rec_hi, rec_lo)
    n_loc = rec_hi - rec_lo
    try:
        if n_loc >= (1 << 31) - 2 or n_records >= 0xffffffff:
            raise DeviceIngestUnavailable("more than 2^31 records in a shard")
        text_ptr, s, flags, tile_nl = state["text_ptr"], state["s"], state["flags"], state["tile_nl"]
        # frame: seq_start / seq_end of the records this rank owns.  The arrays are addressed by ABSOLUTE record number (pointer moved
        # back by rec_lo entries); one guard entry in front takes the end of the line that was cut by the range's start.
        seq_start = torch.full((n_loc + 2,), size, dtype=i64, device=dev)
        seq_end = torch.full((n_loc + 2,), size, dtype=i64, device=dev)
        nl64 = tile_nl.to(i64)
        base = (torch.cumsum(nl64, 0) - nl64) + nl_before
        shift = 8 * (1 - rec_lo)
        if up_hi > lo and n_loc > 0:
            fq_lines(ctx, text_ptr, lo, up_hi, base.data_ptr(), seq_start.data_ptr() + shift, seq_end.data_ptr() + shift, rec_hi, s)
        if up_hi < size and n_loc > 0 and bool((seq_end[1:1 + n_loc] >= up_hi).any().item()):
            raise DeviceIngestUnavailable("a line longer than the shard overlap (C2_FQ_SHARD_OVERLAP)")
        # local de-duplication
        n_slots = 1 << 12
        while n_slots < 2 * max(n_loc, 1):
            n_slots <<= 1
        slots = torch.zeros(n_slots, dtype=i64, device=dev)
        count = torch.zeros(n_slots, dtype=i32, device=dev)
        first = torch.full((n_slots,), -1, dtype=i32, device=dev)
        slot_of = torch.zeros(n_loc + 1, dtype=i32, device=dev)
        rinfo = torch.zeros(n_loc + 1, dtype=i64, device=dev)
        stats = torch.zeros(4, dtype=i32, device=dev)
        rng_t = torch.tensor([rec_lo, rec_hi], dtype=i64, device=dev)
        if n_loc > 0:
            fq_dedup(ctx, text_ptr, seq_start.data_ptr() + shift, seq_end.data_ptr() + shift, rng_t.data_ptr(), rec_hi, slots.data_ptr(), n_slots,
                     count.data_ptr(), first.data_ptr(), slot_of.data_ptr() - 4 * rec_lo, rinfo.data_ptr() - 8 * rec_lo, flags.data_ptr(), stats.data_ptr(), s)
        fl = int(flags.item())
        if fl & 1:
            raise DeviceIngestUnavailable("carriage returns in the text")
        if fl & ~1:
            raise DeviceIngestUnavailable("device ingest gave up (flags %d)" % fl)
        # this rank's unique reads, in first-seen order
        so = slot_of[:n_loc].to(i64)
        idx = torch.arange(rec_lo, rec_hi, dtype=i64, device=dev)
        info = rinfo[:n_loc]
        is_first = (first[so].to(i64) & 0xffffffff) == idx
        mask = is_first & ((info & 0xffffff) > 0)
        rec = torch.nonzero(mask).reshape(-1)                        # (local record numbers)
        lens = (info[rec] & 0xffffff)
        m_loc = int(rec.numel())
        loc_counts = count[so[rec]].contiguous()
        n_empty = int(count[so[torch.nonzero(is_first & ((info & 0xffffff) == 0)).reshape(-1)]].sum().item())
        d_off = torch.zeros(m_loc + 1, dtype=i64, device=dev)
        torch.cumsum(lens, 0, out=d_off[1:])
        loc_bytes = int(d_off[-1].item())
        arena = torch.empty(max(loc_bytes, 1), dtype=torch.uint8, device=dev)
        if m_loc:
            fq_gather(ctx, text_ptr, rinfo.data_ptr(), rec.data_ptr(), d_off.data_ptr(), arena.data_ptr(), m_loc, s)
        if lens.numel() and int(lens.max().item()) >= (1 << 24):
            raise DeviceIngestUnavailable("a sequence line of 2^24 bytes or more")
    except Exception as e:
        why = "%s: %s" % (type(e).__name__, e)
        m_loc = loc_bytes = n_empty = 0
    t_local = time.perf_counter()
    # ---- reconcile: all ranks' unique reads in rank order = the file's first-seen order with repeats; de-duplicated again, multiplicities added
    g2 = _all_gather_i64([m_loc, loc_bytes, n_empty, 1 if why else 0], dev)
    if g2[:, 3].any():
        raise DeviceIngestUnavailable(why or "another rank gave up on the device ingest")
    G, GB = int(g2[:, 0].sum()), int(g2[:, 1].sum())
    if G >= (1 << 31) - 2:
        raise DeviceIngestUnavailable("more than 2^31 unique reads over the shards")
    all_lens = _all_gather_var(lens.to(i32) if m_loc else torch.zeros(0, dtype=i32, device=dev), g2[:, 0], dev).to(i64)
    all_counts = _all_gather_var(loc_counts if m_loc else torch.zeros(0, dtype=i32, device=dev), g2[:, 0], dev)
    all_bytes = _all_gather_var(arena[:loc_bytes], g2[:, 1], dev)
    del arena, d_span
    state.clear()
    g_off = torch.zeros(G + 1, dtype=i64, device=dev)
    torch.cumsum(all_lens, 0, out=g_off[1:])
    text2 = all_bytes if GB else torch.zeros(1, dtype=torch.uint8, device=dev)
    n_slots = 1 << 12
    while n_slots < 2 * max(G, 1):
        n_slots <<= 1
    slots = torch.zeros(n_slots, dtype=i64, device=dev)
    count = torch.zeros(n_slots, dtype=i32, device=dev)
    first = torch.full((n_slots,), -1, dtype=i32, device=dev)
    slot_of = torch.zeros(G + 1, dtype=i32, device=dev)
    rinfo = torch.zeros(G + 1, dtype=i64, device=dev)
    stats = torch.zeros(4, dtype=i32, device=dev)
    flags = torch.zeros(1, dtype=i32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    if G:
        seq_s, seq_e = g_off[:-1].contiguous(), g_off[1:].contiguous()
        rng_t = torch.tensor([0, G], dtype=i64, device=dev)
        fq_dedup(ctx, text2.data_ptr(), seq_s.data_ptr(), seq_e.data_ptr(), rng_t.data_ptr(), G, slots.data_ptr(), n_slots, count.data_ptr(), first.data_ptr(),
                 slot_of.data_ptr(), rinfo.data_ptr(), flags.data_ptr(), stats.data_ptr(), s)
        if int(flags.item()):
            raise _native.NativeError("sharded ingest: the second de-duplication raised flags %d" % int(flags.item()))
    so = slot_of[:G].to(i64)
    wcount = torch.zeros(n_slots, dtype=i64, device=dev)
    if G:
        wcount.index_add_(0, so, all_counts.to(i64))
    rec = torch.nonzero((first[so].to(i64) & 0xffffffff) == torch.arange(G, dtype=i64, device=dev)).reshape(-1) if G else torch.zeros(0, dtype=i64, device=dev)
    n = int(rec.numel())
    lens_u = all_lens[rec] if n else torch.zeros(0, dtype=i64, device=dev)
    cnt_u = wcount[so[rec]] if n else torch.zeros(0, dtype=i64, device=dev)
    if n and int(cnt_u.max().item()) > 0x7fffffff:
        raise OverflowError("a read multiplicity exceeds 2^31 - 1")
    d_off = torch.zeros(n + 1, dtype=i64, device=dev)
    torch.cumsum(lens_u, 0, out=d_off[1:])
    d_reads = torch.empty(max(int(d_off[-1].item()), 1), dtype=torch.uint8, device=dev)
    rc_partner, d_rc_partner = np.zeros(0, dtype=np.int64), None
    if n:
        fq_gather(ctx, text2.data_ptr(), rinfo.data_ptr(), rec.data_ptr(), d_off.data_ptr(), d_reads.data_ptr(), n, s)
        pslot = torch.empty(n, dtype=i32, device=dev)
        fq_rc_partner(ctx, text2.data_ptr(), rinfo.data_ptr(), rec.data_ptr(), n, slots.data_ptr(), n_slots, pslot.data_ptr(), s)
        unique_of_slot = torch.full((n_slots,), -1, dtype=i64, device=dev)
        unique_of_slot[so[rec]] = torch.arange(n, dtype=i64, device=dev)
        d_rc_partner = torch.where(pslot >= 0, unique_of_slot[pslot.to(i64).clamp_(min=0)], -1)
        rc_partner = to_host(d_rc_partner.to(i32), np.int64)
    d_counts = cnt_u.to(i32).contiguous() if n else None
    counts = to_host(d_counts, np.int64) if n else np.zeros(0, dtype=np.int64)
    lens_h = to_host(lens_u.to(i32), np.int64) if n else np.zeros(0, dtype=np.int64)
    off64 = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens_h, out=off64[1:])
    if timings is not None:
        timings["shard_upload_frame_dedup"] = t_local - t0
        timings["shard_reconcile"] = time.perf_counter() - t_local
    return dict(offsets=off64.view(np.uint64), counts=counts, n_reads=n_records, n_empty_records=int(g2[:, 2].sum()), nonempty_lines=lines - total_em,
                n_unique=n, max_len=int(lens_h.max()) if n else 0, min_len=int(lens_h.min()) if n else 0, batch_bytes=[int(off64[-1])] if n else [],
                rc_partner=rc_partner, d_counts=d_counts, d_rc_partner=d_rc_partner, d_reads=d_reads, d_off=d_off,
                shard_bytes=up_hi - up_lo, text_bytes=size, shard_records=n_loc, shard_unique=m_loc, gathered_unique=G, gathered_bytes=GB)


def _upload(span, dev):
    """a host byte array -> device tensor through the pinned upload buffers, chunk by chunk (the copies overlap the next chunk's memcpy)"""
    import torch
    n = int(span.size)
    out = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
    key = dev.index
    chunk = chunk_bytes(n)
    if key not in _pinned or _pinned[key][0].numel() < chunk:
        _pinned[key] = None
        _pinned[key] = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
    pins = _pinned[key]
    evs = [None] * len(pins)
    copy_stream = torch.cuda.Stream(device=dev)
    for c, a in enumerate(range(0, n, chunk)):
        z = min(n, a + chunk)
        k = c % len(pins)
        if evs[k] is not None:
            evs[k].synchronize()
        pins[k].numpy()[:z - a] = span[a:z]
        with torch.cuda.stream(copy_stream):
            out[a:z].copy_(pins[k][:z - a], non_blocking=True)
            evs[k] = torch.cuda.Event()
            evs[k].record(copy_stream)
    torch.cuda.current_stream(dev).wait_stream(copy_stream)
    copy_stream.synchronize()
    return out


# ---- paired input: two texts in HBM, record r of one with record r of the other (CRISPRessoCORE.py:1309-1334) ----
class IngestSource:
    """A FASTQ file as something the device ingest can read (_read_span): the path of a plain file, a BgzfFile (members inflated range by range
    straight into the upload buffers), or -- other gzip input, or input the read filter has to see first -- the text the host inflated /
    filtered into memory.  A context manager that closes what it opened.
        source     path / BgzfFile / uint8 array, or None when the device route does not apply (why_not says why)
        route      the name of the route for QuantResult.ingest_route
        filtered_lines_input   non-empty lines in FRONT of the read filter (the reference's N_READS_INPUT * 4), None without a filter
        host_stream()          the native chunked parser over the same input (the one that already holds the inflated text, if any): what the
                               host route parses when the device declines; None when a .gz text exceeds the in-memory budget"""
    def __init__(self, path, filters=(0, 0, 0)):
        self.path, self.filters = path, tuple(filters)
        self.source, self.route, self.filtered_lines_input, self._held, self._stream = None, "device", None, [], None
        why = applicable(path, filters)
        try:
            if why is None:
                self.source = os.fspath(path)
            elif why.startswith("in memory:"):
                bg = None
                if why.startswith("in memory: compressed"):
                    try:
                        bg = _native.BgzfFile(path)
                    except _native.NativeError:
                        bg = None                                      # (some other gzip file: inflated as a whole)
                gz_route = "device, members inflated into the upload buffers"
                if bg is None and not any(self.filters):
                    try:                                               # ONE ordinary gzip member: segments found by search (c2_gz_parallel.h), inflated the same way
                        bg = _native.GzSegFile(path)
                        gz_route = "device, one gzip member inflated segment by segment into the upload buffers"
                    except _native.NativeError:
                        bg = None                                      # (several members, a small file, ...: inflated as a whole below)
                if bg is not None:
                    self._held.append(bg)
                    why = size_applicable(bg.text_bytes)
                    if why is None:
                        self.source, self.route = bg, gz_route
                else:
                    try:
                        fq = _native.FastqStream(path, *self.filters)    # (inflates / filters: its text is in memory now)
                    except _native.NativeError as e:
                        if "in-memory budget" not in str(e):
                            raise
                        fq, why = None, str(e)                         # (a .gz whose text does not fit in memory streams through zlib on the host)
                        self._stream = False
                    if fq is not None:
                        self._held.append(fq)
                        self._stream = fq
                        text = fq.text()
                        why = text_applicable(text)
                        if why is None:
                            self.source, self.route = text, "device, text from host memory"
                            if fq.filtered:
                                self.filtered_lines_input = fq.lines_input()
        except OSError as e:
            why = "%s: %s" % (type(e).__name__, e)
        except BaseException:
            self.close()                                              # (what was opened so far; the error goes to the caller)
            raise
        self.why_not = why

    def host_stream(self):
        if self._stream is None:
            try:
                self._stream = _native.FastqStream(self.path, *self.filters)
                self._held.append(self._stream)
            except _native.NativeError as e:
                if "in-memory budget" not in str(e):
                    raise
                self._stream = False
        return self._stream or None

    def close(self):
        for h in self._held:
            try:
                h.close()
            except Exception:
                pass
        self._held = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def source_size(source):
    return int(source.size) if isinstance(source, np.ndarray) else source.text_bytes if isinstance(source, _native.BgzfFile) else os.path.getsize(source)


def upload_whole(source, dev):
    """the whole text of a source -> (uint8 device tensor, its size): chunk by chunk through the pinned upload buffers, the host threads reading
    (a plain file: pread() by all of them; BGZF: members inflated by the native threads) while the previous chunk crosses the link"""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    size = source_size(source)
    out = torch.empty(max(size, 1), dtype=torch.uint8, device=dev)
    if dev.type != "cuda":                                            # (tests: the "device" is host memory)
        _read_span(source, 0, size, out.numpy())
        return out, size
    import threading
    global _upload_lock
    if _upload_lock is None:
        _upload_lock = threading.Lock()
    with _upload_lock:
        chunk = chunk_bytes(size)
        key = dev.index
        if key not in _pinned or _pinned[key][0].numel() < chunk:
            _pinned[key] = None
            _pinned[key] = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(3)]
        pins = _pinned[key]
        evs = [None] * len(pins)
        copy_stream = torch.cuda.Stream(device=dev)
        threads = copy_threads() if isinstance(source, str) else 1
        try:
            with ThreadPoolExecutor(threads) as pool:
                for c, a in enumerate(range(0, size, chunk)):
                    z = min(size, a + chunk)
                    k = c % len(pins)
                    if evs[k] is not None:
                        evs[k].synchronize()
                    view = pins[k].numpy()
                    if threads > 1:
                        step = (-(-(z - a) // threads) + 4095) // 4096 * 4096
                        list(pool.map(lambda p0: _read_span(source, a + p0, min(z, a + p0 + step), view[p0:]), range(0, z - a, step)))
                    else:
                        _read_span(source, a, z, view)
                    with torch.cuda.stream(copy_stream):
                        out[a:z].copy_(pins[k][:z - a], non_blocking=True)
                        evs[k] = torch.cuda.Event()
                        evs[k].record(copy_stream)
            torch.cuda.current_stream(dev).wait_stream(copy_stream)
            copy_stream.synchronize()
        finally:
            if 3 * chunk > PINNED_CACHE_BYTES:
                _pinned.pop(key, None)
    return out, size


class PairIngest:
    """what ingest_pairs leaves in HBM: d_keys / d_quals = the key (seq1 + '+' + reverse_complement(seq2)) and quality pair (qual1 + ' ' + qual2[::-1])
    of EVERY record, back to back (key_off / qual_off: int64 [n_records + 1]); l1 / lq1 [n_records]: the lengths in front of the '+' / the blank;
    uniq_rec [n_unique]: the record in which every distinct key first occurs, in file order -- the order of the reference's variantCache;
    rec_key [n_records]: the index of a record's key in that list; counts [n_unique] int64: copies of every key."""
    pass


def ingest_pairs(source1, source2, ctx, dev, timings=None):
    """Two sources (see IngestSource) -> PairIngest.  Raises DeviceIngestUnavailable for what the kernels do not take or where the reference raises an
    error the host route reproduces (carriage returns; a character of read 2 that reverse_complement() does not know; a '+' inside a read or a
    blank inside a quality string, which the reference's key.split('+') / quals.split(' ') would trip over)."""
    import time
    import torch
    t0 = time.perf_counter()
    i64, i32, u8 = torch.int64, torch.int32, torch.uint8
    s = torch.cuda.current_stream(dev).cuda_stream
    flags = torch.zeros(1, dtype=i32, device=dev)
    framed = []
    for src in (source1, source2):
        d_t, size = upload_whole(src, dev)
        tiles = max(1, (size + TILE - 1) // TILE)
        tile_nl = torch.zeros(tiles, dtype=i32, device=dev)
        tile_em = torch.zeros(tiles, dtype=i32, device=dev)
        if size:
            fq_count(ctx, d_t.data_ptr(), 0, size, tile_nl.data_ptr(), tile_em.data_ptr(), flags.data_ptr(), s)
        nl = tile_nl.to(i64)
        upto = torch.cumsum(nl, 0)
        base = (upto - nl).contiguous()
        lines = int(upto[-1].item()) + (1 if size and int(d_t[size - 1].item()) != 0x0a else 0)
        recs = (lines + 3) // 4                                       # readline(): every started group of four lines is a record
        if recs >= (1 << 31) - 2:
            raise DeviceIngestUnavailable("more than 2^31 records")
        arrs = [torch.full((max(recs, 1),), size, dtype=i64, device=dev) for _ in range(4)]   # (a line that never ends: ends with the text; one that never starts: empty)
        if size:
            fq_lines4(ctx, d_t.data_ptr(), 0, size, base.data_ptr(), arrs[0].data_ptr(), arrs[1].data_ptr(), arrs[2].data_ptr(), arrs[3].data_ptr(), recs, s)
        framed.append((d_t, size, recs, arrs))
    if int(flags.item()) & 1:
        raise DeviceIngestUnavailable("carriage returns in the text")
    if timings is not None:
        torch.cuda.synchronize(dev) if dev.type == "cuda" else None
        timings["upload_and_frame"] = time.perf_counter() - t0
    (t1, _, recs1, a1), (t2, _, recs2, a2) = framed
    n = min(recs1, recs2)                                             # while (fastq1_id and fastq2_id), :1311
    P = PairIngest()
    P.n_records, P.different_lengths = n, recs1 != recs2
    z64 = lambda m: torch.zeros(m, dtype=i64, device=dev)
    s1, q1, s2, q2, klen, qlen = (z64(max(n, 1)) for _ in range(6))
    pflags = torch.zeros(1, dtype=i32, device=dev)
    if n:
        fq_pair_lengths(ctx, t1.data_ptr(), t2.data_ptr(), [x.data_ptr() for x in a1], [x.data_ptr() for x in a2], n, s1.data_ptr(), q1.data_ptr(),
                        s2.data_ptr(), q2.data_ptr(), klen.data_ptr(), qlen.data_ptr(), pflags.data_ptr(), s)
    koff, qoff = z64(n + 1), z64(n + 1)
    if n:
        torch.cumsum(klen[:n], 0, out=koff[1:])
        torch.cumsum(qlen[:n], 0, out=qoff[1:])
    kb, qb = int(koff[-1].item()), int(qoff[-1].item())
    if kb >= (1 << 40) or qb >= (1 << 40):
        raise DeviceIngestUnavailable("more than 2^40 bytes of keys")
    d_keys = torch.empty(max(kb, 1), dtype=u8, device=dev)
    d_quals = torch.empty(max(qb, 1), dtype=u8, device=dev)
    if n:
        fq_pair_write(ctx, t1.data_ptr(), t2.data_ptr(), n, s1.data_ptr(), q1.data_ptr(), s2.data_ptr(), q2.data_ptr(), koff.data_ptr(), qoff.data_ptr(),
                      d_keys.data_ptr(), d_quals.data_ptr(), pflags.data_ptr(), s)
    pf = int(pflags.item())
    del framed, t1, t2, a1, a2
    if pf & 1:
        raise DeviceIngestUnavailable("a line of 2^24 bytes or more")
    if pf & 2:
        raise DeviceIngestUnavailable("a character of read 2 outside ACGTN_-")
    if n and (int((d_keys[:kb] == 43).sum().item()) != n or int((d_quals[:qb] == 32).sum().item()) != n):
        raise DeviceIngestUnavailable("a '+' inside a read or a blank inside a quality string")
    # ---- exact de-duplication of the keys: first-seen order, copies (variantCache of :1324-1329)
    n_slots = 1 << 12
    while n_slots < 2 * n:
        n_slots <<= 1
    if n_slots > (1 << 30):
        raise DeviceIngestUnavailable("table of more than 2^30 slots")
    slots = torch.zeros(n_slots, dtype=i64, device=dev)
    count = torch.zeros(n_slots, dtype=i32, device=dev)
    first = torch.full((n_slots,), -1, dtype=i32, device=dev)
    slot_of = torch.zeros(n + 1, dtype=i32, device=dev)
    rinfo = torch.zeros(n + 1, dtype=i64, device=dev)
    dstat = torch.zeros(4, dtype=i32, device=dev)
    dflag = torch.zeros(1, dtype=i32, device=dev)
    rng = torch.tensor([0, n], dtype=i64, device=dev)
    if n:
        fq_dedup(ctx, d_keys.data_ptr(), koff[:-1].data_ptr(), koff[1:].data_ptr(), rng.data_ptr(), n, slots.data_ptr(), n_slots, count.data_ptr(),
                 first.data_ptr(), slot_of.data_ptr(), rinfo.data_ptr(), dflag.data_ptr(), dstat.data_ptr(), s)
        if int(dflag.item()):
            raise DeviceIngestUnavailable("the de-duplication of the pair keys raised flags %d" % int(dflag.item()))
    so = slot_of[:n].to(i64)
    is_first = (first.to(i64) & 0xffffffff)[so] == torch.arange(n, dtype=i64, device=dev)
    P.uniq_rec = torch.nonzero(is_first).reshape(-1)
    P.n_unique = int(P.uniq_rec.numel())
    key_of_slot = torch.zeros(n_slots, dtype=i64, device=dev)
    key_of_slot[so[P.uniq_rec]] = torch.arange(P.n_unique, dtype=i64, device=dev)
    P.rec_key = key_of_slot[so]
    P.counts = (count.to(i64) & 0xffffffff)[so[P.uniq_rec]]
    P.d_keys, P.d_quals, P.key_off, P.qual_off = d_keys, d_quals, koff, qoff
    P.l1, P.lq1 = (s1[:n] & 0xffffff), (q1[:n] & 0xffffff)
    P.key_bytes, P.qual_bytes = kb, qb
    if timings is not None:
        torch.cuda.synchronize(dev) if dev.type == "cuda" else None
        timings["pair_keys_and_dedup"] = time.perf_counter() - t0 - timings["upload_and_frame"]
    return P
