"""Deterministic synthetic amplicon-sequencing reads for the BASELINE.json configs.

Amplicon: numpy.random.default_rng(20240601 [+ amplicon id]).choice("ACGT", L); cut site c = L//2, so the reference's
setup would put gap_incentive[c+1] = 1 and include_idxs = [c, c+1] (defaults -w 1 -wc -3).
Reads are generated in independent blocks of BLOCK reads (seed sequence [20240602, amplicon id, block index]), so any prefix
of a data set is reproducible without generating the rest.  Per read, starting from the amplicon:
  * substitutions at rate 0.005 per base,
  * with p = 0.30 one deletion of min(60, Geometric(0.12)) bases placed so that it overlaps the cut site,
  * with p = 0.10 one insertion of U[1,15] random bases at the cut site,
  * with p = 0.01 three random positions become 'N',
  * random "downstream" bases are appended and the read is truncated, so EVERY read has exactly L bases
    (the DP is always L x L), forward strand.
"""
import numpy as np

BLOCK = 1 << 16
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _bases(x):
    """uint8 codes 0..3 -> ASCII 'A','C','G','T' (arithmetic: numpy's take() on uint8 indices is slow)."""
    o = (x << 1) + np.uint8(65)
    o += (x == 2).view(np.uint8) << 1
    o += (x == 3).view(np.uint8) * np.uint8(13)
    return o


def make_amplicon(L, amplicon_id=0):
    seed = 20240601 if amplicon_id == 0 else [20240601, int(amplicon_id)]
    return _bases(np.random.default_rng(seed).integers(0, 4, L, dtype=np.uint8)).tobytes().decode()


def amplicon_setup(L, amplicon_id=0):
    """-> (sequence, gap_incentive int64[L+1], include_idxs list) as CRISPRessoCORE.py:3205-3207 would build them."""
    seq = make_amplicon(L, amplicon_id)
    cut = L // 2
    g = np.zeros(L + 1, dtype=np.int64)
    g[cut + 1] = 1
    return seq, g, [cut, cut + 1]


def _block(amplicon_u8, n, block_index, amplicon_id):
    L = amplicon_u8.shape[0]
    cut = L // 2
    rng = np.random.default_rng([20240602, int(amplicon_id), int(block_index)])
    # the substituted template of every read (sparse substitutions)
    n_sub = int(rng.binomial(n * L, 0.005))
    sub_pos = rng.integers(0, n * L, n_sub)
    sub_base = _bases(rng.integers(0, 4, n_sub, dtype=np.uint8))
    has_del = rng.random(n) < 0.30
    del_len = np.minimum(60, rng.geometric(0.12, n)).astype(np.int16)
    del_len[~has_del] = 0
    del_start = np.maximum(0, cut - (rng.random(n) * (del_len + 1)).astype(np.int16)).astype(np.int16)
    has_ins = rng.random(n) < 0.10
    ins_len = rng.integers(1, 16, n).astype(np.int16)
    ins_len[~has_ins] = 0
    filler = _bases(rng.integers(0, 4, (n, L), dtype=np.uint8))   # inserted bases and downstream bases come from here
    p = np.arange(L, dtype=np.int16)[None, :]
    # undo the insertion: position in the post-deletion sequence (or -1 inside the inserted run)
    q = np.where(p < cut, p, np.where(p < cut + ins_len[:, None], np.int16(-1), p - ins_len[:, None]))
    # undo the deletion: position in the amplicon
    src = np.where(q < del_start[:, None], q, q + del_len[:, None])
    src = np.where(q < 0, np.int16(-1), src)
    tmpl = np.broadcast_to(amplicon_u8, (n, L)).copy().reshape(-1)
    tmpl[sub_pos] = sub_base
    tmpl = tmpl.reshape(n, L)
    inside = (src >= 0) & (src < L)
    reads = np.where(inside, np.take_along_axis(tmpl, np.clip(src, 0, L - 1), axis=1), filler)
    has_n = np.nonzero(rng.random(n) < 0.01)[0]
    if has_n.size:
        npos = rng.integers(0, L, (has_n.size, 3))
        reads[has_n[:, None], npos] = ord('N')
    return np.ascontiguousarray(reads, dtype=np.uint8)


def _block_job(job):
    amp_bytes, b, amplicon_id = job
    return _block(np.frombuffer(amp_bytes, dtype=np.uint8), BLOCK, b, amplicon_id)


def make_reads(L, n, amplicon_id=0, amplicon=None, first_block=0, workers=1):
    """-> uint8 [n, L] reads (row k is read k of the data set starting at block `first_block`).
    workers > 1 generates blocks in a fork()ed process pool: call it before the process touches HIP."""
    amp_s = amplicon or make_amplicon(L, amplicon_id)
    amp = np.frombuffer(amp_s.encode(), dtype=np.uint8)
    out = np.empty((n, L), dtype=np.uint8)
    nblocks = (n + BLOCK - 1) // BLOCK
    if workers > 1 and nblocks > 1:
        import multiprocessing as mp
        jobs = [(amp_s.encode(), first_block + b, amplicon_id) for b in range(nblocks)]
        with mp.get_context("fork").Pool(min(workers, nblocks)) as pool:
            for b, blk in enumerate(pool.imap(_block_job, jobs)):
                start = b * BLOCK
                m = min(BLOCK, n - start)
                out[start:start + m] = blk[:m]
        return out
    for b, start in enumerate(range(0, n, BLOCK)):
        m = min(BLOCK, n - start)
        out[start:start + m] = _block(amp, BLOCK, first_block + b, amplicon_id)[:m]
    return out


def make_variant(amplicon, kind):
    """Candidate references of config 4: 'hdr' = 6 substitutions + a 3-bp insertion near the cut; 'pe' = a 12-bp replacement."""
    L = len(amplicon)
    cut = L // 2
    s = list(amplicon)
    comp = {'A': 'C', 'C': 'G', 'G': 'T', 'T': 'A'}
    if kind == 'hdr':
        for d in (-9, -6, -3, 2, 5, 8):
            s[cut + d] = comp[s[cut + d]]
        s[cut:cut] = list('GAT')
    elif kind == 'pe':
        s[cut - 6:cut + 6] = [comp[c] for c in s[cut - 6:cut + 6]]
    else:
        raise ValueError(kind)
    return ''.join(s)


# ---- the same reads as FASTQ files (bench.py's FASTQ -> count tensors leg, tools/e2e_rate.py) ----------------------------------
def write_fastq(reads, path, chunk=1 << 20):
    """reads uint8 [n, L] -> a 4-line FASTQ file: '@r<9-digit index>', the read, '+', qualities 'I' * L.  -> bytes written"""
    n, L = reads.shape
    W = 12 + (L + 1) + 2 + (L + 1)
    with open(path, "wb") as fh:
        for a in range(0, n, chunk):
            m = min(chunk, n - a)
            rec = np.empty((m, W), dtype=np.uint8)
            rec[:, 0], rec[:, 1] = ord('@'), ord('r')
            idx = np.arange(a, a + m, dtype=np.int64)
            for d in range(9):
                rec[:, 10 - d] = (idx % 10 + 48).astype(np.uint8)
                idx //= 10
            rec[:, 11] = 10
            rec[:, 12:12 + L] = reads[a:a + m]
            rec[:, 12 + L] = 10
            rec[:, 13 + L], rec[:, 14 + L] = ord('+'), 10
            rec[:, 15 + L:15 + 2 * L] = ord('I')
            rec[:, 15 + 2 * L] = 10
            rec.tofile(fh)
    return n * W


_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_slice(job):
    """one byte range of the plain file -> its BGZF members (blocks of <= 65280 input bytes, each a gzip member whose 'BC' extra
    subfield holds the member's size - 1: the SAM specification's block format, what bgzip / htslib write)"""
    import struct
    import zlib
    path, a, b, level = job
    with open(path, "rb") as fh:
        fh.seek(a)
        data = fh.read(b - a)
    out = []
    for p in range(0, len(data), 65280):
        blk = data[p:p + 65280]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        cd = c.compress(blk) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cd) + 25) + cd +
                   struct.pack("<II", zlib.crc32(blk) & 0xffffffff, len(blk)))
    return b"".join(out)


def write_bgzf(src_path, dst_path, workers=1, level=1, slice_bytes=65280 * 256):
    """plain file -> BGZF file (fork()ed pool when workers > 1: call it before the process touches HIP).  -> bytes written"""
    import os
    size = os.path.getsize(src_path)
    jobs = [(src_path, a, min(size, a + slice_bytes), level) for a in range(0, size, slice_bytes)]
    total = 0
    with open(dst_path, "wb") as fh:
        if workers > 1 and len(jobs) > 1:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
                for part in pool.imap(_bgzf_slice, jobs):
                    fh.write(part)
                    total += len(part)
        else:
            for j in jobs:
                part = _bgzf_slice(j)
                fh.write(part)
                total += len(part)
        fh.write(_BGZF_EOF)
    return total + len(_BGZF_EOF)


def _gzip_slice(job):
    """one byte range of the plain file -> (raw deflate of it ended by a sync flush, its CRC-32, its length)"""
    import zlib
    path, a, b, level = job
    with open(path, "rb") as fh:
        fh.seek(a)
        data = fh.read(b - a)
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush(zlib.Z_SYNC_FLUSH), zlib.crc32(data) & 0xffffffff, len(data)


def _crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (not exposed by Python's zlib): CRC of A + B from CRC(A), CRC(B), len(B) -- GF(2) matrix squaring"""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    if len2 <= 0:
        return crc1
    odd = [0xedb88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def write_gzip_member(src_path, dst_path, workers=1, level=6, slice_bytes=32 << 20):
    """plain file -> ONE gzip member (what `gzip -6` / pigz give: a single deflate stream, one header, one CRC + length trailer), compressed slice by slice
    on a fork()ed pool (before HIP) -- every slice ends with a sync flush, the stream with an empty final block.  A reader sees an ordinary .gz file: no
    member boundaries, no BGZF extra fields, nothing to split it by.  -> bytes written"""
    import os
    import struct
    size = os.path.getsize(src_path)
    jobs = [(src_path, a, min(size, a + slice_bytes), level) for a in range(0, size, slice_bytes)]
    crc, total = 0, 0
    with open(dst_path, "wb") as fh:
        fh.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff")
        total += 10
        if workers > 1 and len(jobs) > 1:
            import multiprocessing as mp
            with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
                parts = pool.imap(_gzip_slice, jobs)
                for blob, c, n in parts:
                    fh.write(blob)
                    total += len(blob)
                    crc = _crc32_combine(crc, c, n)
        else:
            for j in jobs:
                blob, c, n = _gzip_slice(j)
                fh.write(blob)
                total += len(blob)
                crc = _crc32_combine(crc, c, n)
        fh.write(b"\x03\x00" + struct.pack("<II", crc, size & 0xffffffff))
        total += 10
    return total


# ---- reads shaped like the reference's own test data (bench.py's robustness legs) ------------------------------------------------
_FANC = {}


def fanc_profile():
    """crispresso2_amd/fanc_profile.json (tools/fanc_profile.py): the 213 distinct reads of the reference's tests/FANC.Cas9.fastq reduced to
    signatures -- leading overhang, read length, deletions / insertions by reference position, substitution count, or 'junk' -- with their
    multiplicities, the 223-bp FANC amplicon, its cut point and the genomic flank the reads run into behind the amplicon."""
    if not _FANC:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fanc_profile.json")) as fh:
            _FANC.update(json.load(fh))
    return _FANC


def fanc_setup():
    """-> (amplicon, gap_incentive int64[L+1], include_idxs) of the FANC run as CRISPRessoCORE.py:3205-3207 builds them (-w 1 -wc -3)"""
    p = fanc_profile()
    amp, cut = p["amplicon"], int(p["cut_point"])
    g = np.zeros(len(amp) + 1, dtype=np.int64)
    g[cut + 1] = 1
    return amp, g, [cut, cut + 1]


def _fanc_bases():
    """per template: the read it stands for, built from the amplicon (uint8), and where its amplicon part lies"""
    p = fanc_profile()
    amp = np.frombuffer(p["amplicon"].encode(), dtype=np.uint8)
    flank = np.frombuffer(p["trail_flank"].encode(), dtype=np.uint8)
    out = []
    for t in p["templates"]:
        if t["junk"]:
            out.append((None, 0, t["len"]))
            continue
        keep = np.ones(len(amp), dtype=bool)
        for pos, ln in t["dels"]:
            keep[pos:pos + ln] = False
        parts, last = [], 0
        for pos, ln in sorted(t["ins"]):
            parts.append(amp[last:pos][keep[last:pos]])
            parts.append(np.full(ln, ord('A'), dtype=np.uint8))        # (re-drawn per read)
            last = pos
        parts.append(amp[last:][keep[last:]])
        body = np.concatenate(parts)
        seq = np.concatenate([np.full(t["lead"], ord('A'), dtype=np.uint8), body, flank])
        if len(seq) < t["len"]:
            seq = np.concatenate([seq, np.full(t["len"] - len(seq), ord('A'), dtype=np.uint8)])
        out.append((seq[:t["len"]].copy(), t["lead"], min(t["len"], t["lead"] + len(body))))
    return out


FANC_W = 256                                                          # row width of the padded read matrix (longest FANC read: 250)


def _fanc_block(n, block_index):
    """-> (uint8 [n, FANC_W] reads padded with 0, int32 [n] lengths)"""
    p = fanc_profile()
    T = p["templates"]
    bases = _fanc_bases()
    rng = np.random.default_rng([20240603, int(block_index)])
    w = np.array([t["w"] for t in T], dtype=np.float64)
    which = rng.choice(len(T), n, p=w / w.sum())
    out = np.zeros((n, FANC_W), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    for ti in np.unique(which):
        rows = np.nonzero(which == ti)[0]
        t = T[int(ti)]
        seq, lead, body_end = bases[int(ti)]
        ln = t["len"]
        lens[rows] = ln
        if seq is None:                                                # junk: an unrelated read of that length
            out[rows, :ln] = _bases(rng.integers(0, 4, (len(rows), ln), dtype=np.uint8))
            continue
        out[rows, :ln] = seq
        if lead:                                                       # the overhang in front varies from read to read in the data
            out[rows, :lead] = _bases(rng.integers(0, 4, (len(rows), lead), dtype=np.uint8))
        col = lead
        for pos, iln in sorted(t["ins"]):                              # inserted bases: drawn per read
            c0 = lead + pos - sum(d[1] for d in t["dels"] if d[0] < pos) + sum(i[1] for i in t["ins"] if i[0] < pos)
            if c0 + iln <= ln:
                out[rows, c0:c0 + iln] = _bases(rng.integers(0, 4, (len(rows), iln), dtype=np.uint8))
        if t["subs"] and body_end > lead:
            pos = rng.integers(lead, body_end, (len(rows), t["subs"]))
            cur = out[rows[:, None], pos]
            new = _bases(rng.integers(0, 4, pos.shape, dtype=np.uint8))
            new = np.where(new == cur, _bases(((rng.integers(0, 4, pos.shape, dtype=np.uint8)) + 1) % 4), new)
            out[rows[:, None], pos] = new
    return out, lens


def _fanc_block_job(job):
    return _fanc_block(*job)


def make_fanc_reads(n, first_block=0, workers=1):
    """n reads drawn from the FANC profile -> (uint8 [n, FANC_W] padded with 0, int32 [n] lengths).  workers > 1: a fork()ed pool (before HIP)."""
    out = np.empty((n, FANC_W), dtype=np.uint8)
    lens = np.empty(n, dtype=np.int32)
    jobs = [(min(BLOCK, n - a), first_block + b) for b, a in enumerate(range(0, n, BLOCK))]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            it = pool.imap(_fanc_block_job, jobs)
            for b, (blk, ln) in enumerate(it):
                out[b * BLOCK:b * BLOCK + len(blk)] = blk
                lens[b * BLOCK:b * BLOCK + len(blk)] = ln
    else:
        for b, job in enumerate(jobs):
            blk, ln = _fanc_block(*job)
            out[b * BLOCK:b * BLOCK + len(blk)] = blk
            lens[b * BLOCK:b * BLOCK + len(blk)] = ln
    return out, lens


def pack_ragged(padded, lens):
    """padded uint8 [n, W] + lengths -> (uint8 arena of the reads back to back, uint64 offsets [n + 1])"""
    n, W = padded.shape
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:], dtype=np.uint64)
    mask = np.arange(W, dtype=np.int32)[None, :] < lens[:, None]
    return np.ascontiguousarray(padded[mask]), off
