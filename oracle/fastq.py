"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the first pass of process_fastq (reference
CRISPResso2/CRISPRessoCORE.py:1820-1849): the FASTQ -> {sequence: copies} loop, line for line (text mode, readline,
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
