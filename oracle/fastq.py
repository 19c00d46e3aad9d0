"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the first pass of process_fastq (reference
CRISPResso2/CRISPRessoCORE.py:1820-1849) and of process_paired_fastq (:1296-1334, :1452-1470): the FASTQ -> {sequence: copies} loop, line for line (text mode, readline,
strip() on the sequence and '+' lines, every non-empty first line starts a record, '' keys included).
Parity pinned: it is the reference's own statements with the logging removed; tests/test_fastq_ingest.py compares the
native c2_fastq_unique with it.  Never imported by the product package."""
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
