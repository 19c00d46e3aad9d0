"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the first pass of process_fastq (reference
CRISPResso2/CRISPRessoCORE.py:1820-1849) and of process_paired_fastq (:1296-1334, :1452-1470): the FASTQ -> {sequence: copies} loop, line for line (text mode, readline,
strip() on the sequence and '+' lines, every non-empty first line starts a record, '' keys included).
Parity pinned: it is the reference's own statements with the logging removed, and what it feeds is pinned end to end by runs of the reference itself:
read_fastq_unique by every whole-run golden of tests/test_whole_run_tables.py (tests/golden/fanc_full_run / params_run / both_run / pe_run / ...: the
reference's main() read the same FASTQ and its result files come out byte for byte), filter_fastq by the filtered FASTQ of the reference's own
CRISPResso_on_params run (params_run.json.gz: `fastq_after_quality_filter`), the paired loop by tests/golden/paired_fastq.json.gz (the reference's
process_paired_fastq on the same two files: keys, order, counts).  tests/test_fastq_ingest.py compares the native c2_fastq_unique with it.
Never imported by the product package."""
import gzip


def read_fastq_unique(path):
    opener = (lambda x: gzip.open(x, 'rt')) if str(path).endswith('.gz') else open      # :1820-1823
    variantCache = {}
    num_reads = 0
    with opener(path) as fastq_handle:                                                    # :1825
        fastq_id = fastq_handle.readline()                                                # :1831
        while fastq_id:                                                                   # :1832
            fastq_seq = fastq_handle.readline().strip()                                   # :1836
            fastq_handle.readline().strip()                                               # :1837
            fastq_handle.readline()                                                       # :1838
            if fastq_seq in variantCache:                                                 # :1839-1844
                variantCache[fastq_seq] += 1
            else:
                variantCache[fastq_seq] = 1
            fastq_id = fastq_handle.readline()                                            # :1845
            num_reads += 1
    return variantCache, num_reads


_RC = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N', '_': '_', '-': '-'}


def reverse_complement(seq):                                                              # CRISPRessoShared.py:399-403
    return "".join([_RC[c] for c in seq.upper()[-1::-1]])


def read_paired_fastq_unique(path1, path2):
    """First pass of process_paired_fastq's n_processes > 1 route (CRISPRessoCORE.py:1296-1334), statement for statement:
    -> ({seq1 + '+' + rc(seq2): [copies, qual1 + ' ' + qual2[::-1] of the first occurrence]}, num_reads)"""
    opener = lambda p: gzip.open(p, 'rt') if str(p).endswith('.gz') else open(p)          # :1296-1304
    fastq1_file, fastq2_file = opener(path1), opener(path2)
    variantCache = {}
    num_reads = 0
    fastq1_id = fastq1_file.readline()                                                    # :1310-1311
    fastq2_id = fastq2_file.readline()
    while fastq1_id and fastq2_id:                                                        # :1312
        fastq1_seq = fastq1_file.readline().strip()                                       # :1316-1318
        fastq1_file.readline()
        fastq1_qual = fastq1_file.readline().strip()
        fastq2_seq = reverse_complement(fastq2_file.readline().strip())                   # :1320-1322
        fastq2_file.readline()
        fastq2_qual = fastq2_file.readline().strip()[::-1]
        fastq_read_key = fastq1_seq + '+' + fastq2_seq                                    # :1323-1324
        fastq_quals = fastq1_qual + ' ' + fastq2_qual
        if fastq_read_key in variantCache:                                                # :1325-1330
            variantCache[fastq_read_key][0] += 1
        else:
            variantCache[fastq_read_key] = [1, fastq_quals]
        fastq1_id = fastq1_file.readline()                                                # :1331-1333
        fastq2_id = fastq2_file.readline()
        num_reads += 1
    fastq1_file.close()
    fastq2_file.close()
    return variantCache, num_reads


def paired_occurrences(path1, path2, wanted):
    """Second pass of the same route (:1452-1470): [(key, qual1, qual2[::-1])] of every record whose key is in `wanted`."""
    opener = lambda p: gzip.open(p, 'rt') if str(p).endswith('.gz') else open(p)
    f1, f2 = opener(path1), opener(path2)
    out = []
    id1, id2 = f1.readline(), f2.readline()
    while id1 and id2:
        s1 = f1.readline().strip(); f1.readline(); q1 = f1.readline().strip()
        s2 = reverse_complement(f2.readline().strip()); f2.readline(); q2 = f2.readline().strip()[::-1]
        key = s1 + '+' + s2
        if key in wanted:
            out.append((key, q1, q2))
        id1, id2 = f1.readline(), f2.readline()
    f1.close()
    f2.close()
    return out


def filter_fastq(path_in, path_out, min_bp_qual_in_read=None, min_av_read_qual=None, min_bp_qual_or_N=None):
    """Single-file route of filterFastqs.filterFastqs (reference CRISPResso2/filterFastqs.py:29-126 dispatch, :128-226 the
    seven loops), restated as ONE loop with the statements of each loop in its own order: binary readline + rstrip, uint8
    qualities minus 33, numpy mean / min, N-masking through a boolean index -- including run_mBP_mBPN's missing .copy()
    (assignment into a read-only frombuffer view raises).  Pinned: FANC.Cas9.fastq with -q 30 gives byte for byte the
    *_filtered.fastq.gz content of the reference's CRISPResso_on_params run (tests/golden/params_run.json.gz).
    -> int(`grep -c .` / 4.0) of the input, the reference's N_READS_INPUT (CRISPRessoShared.py:743-748)."""
    import io
    import numpy
    mBP, mRQ, mBPN = bool(min_bp_qual_in_read), bool(min_av_read_qual), bool(min_bp_qual_or_N)
    if not (mBP or mRQ or mBPN):
        raise SystemExit('Finished -- No modifications requested')                      # :105
    f1_in = io.BufferedReader(gzip.open(path_in, 'rb')) if str(path_in).endswith('.gz') else open(path_in, 'rb')     # :48-57
    f1_out = gzip.open(path_out, 'wt') if str(path_out).endswith('.gz') else open(path_out, 'w')                     # :60-63
    try:
        idLine = f1_in.readline().rstrip().decode('utf-8')
        while idLine:
            seqLine = f1_in.readline().rstrip()
            plusLine = f1_in.readline().rstrip()
            qualLine = f1_in.readline().rstrip()
            npQualLine = numpy.frombuffer(qualLine, dtype=numpy.uint8) - 33
            keep = True
            if mBP and mRQ and not mBPN:                                                  # run_mBP_mRQ :167-179: mean, then min
                keep = bool(numpy.mean(npQualLine) >= min_av_read_qual)
                if keep:
                    keep = bool(numpy.min(npQualLine) >= min_bp_qual_in_read)
            else:                                                                         # every other loop: min (if asked), then mean (if asked)
                if mBP:
                    keep = bool(numpy.min(npQualLine) >= min_bp_qual_in_read)
                if keep and mRQ:
                    keep = bool(numpy.mean(npQualLine) >= min_av_read_qual)
            if keep:
                if mBPN:
                    npSeqLine = numpy.frombuffer(seqLine, 'c') if (mBP and not mRQ) else numpy.frombuffer(seqLine, 'c').copy()   # :191 vs :136, :206, :223
                    npSeqLine[npQualLine < min_bp_qual_or_N] = 'N'
                    seq_text = npSeqLine.tobytes().decode('utf-8')
                else:
                    seq_text = seqLine.decode('utf-8')
                f1_out.write("%s\n%s\n%s\n%s\n" % (idLine, seq_text, plusLine.decode('utf-8'), qualLine.decode('utf-8')))
            idLine = f1_in.readline().rstrip().decode('utf-8')
    finally:
        f1_in.close()
        f1_out.close()
    opener = gzip.open if str(path_in).endswith('.gz') else open
    with opener(path_in, 'rb') as fh:
        return int(float(sum(1 for line in fh.read().split(b'\n') if line)) / 4.0)
