"""Wire formats on the far side of the align + classify path.

* the JSON form of a per-read variant dict -- the reference's `CRISPRessoJSONEncoder` / `CRISPRessoJSONDecoder`
  (CRISPRessoShared.py:812-900): tagged objects `{"_type": ..., "value": ...}` for ResultsSlotsDict, numpy arrays,
  DataFrames, datetimes, sets, ranges and argparse namespaces, numpy scalars as plain numbers;
* `variants_<k>.tsv`, one `sequence<TAB>json` line per unique read, which the reference's worker processes write
  (`variant_file_generator_process`, CRISPRessoCORE.py:1198-1242) and its parent merges (:1900-1955).  Here worker k is
  GPU rank k: `write_variant_file` / `merge_variant_files` keep that exchange format, byte for byte;
* the `--fastq_output` annotation (`process_fastq_write_out`, CRISPRessoCORE.py:2283-2350): every input record is written
  again with ` ALN=... ALN_SCORES=... ALN_DETAILS=... CLASS=... MODS=... DEL=... INS=... SUB=... ALN_REF=... ALN_SEQ=...`
  appended to its '+' line.

Host code only: the variant dicts come from `variants.get_new_variant_objects` (device alignments + device classifier).
"""
import argparse
import datetime
import gzip
import io
import json
import os
import re

import numpy as np

from . import CRISPRessoCOREResources

_RANGE_RE = re.compile(r'range\((\d+), (\d+)(?:, (\d+))?\)')


def _tagged(kind, value):
    return {'_type': kind, 'value': value}


class CRISPRessoJSONEncoder(json.JSONEncoder):
    """json.dumps(obj, cls=CRISPRessoJSONEncoder) gives the reference's text (CRISPRessoShared.py:812-860)."""

    def default(self, obj):
        if isinstance(obj, CRISPRessoCOREResources.ResultsSlotsDict):
            return _tagged('ResultsSlotsDict', obj.__dict__)
        if isinstance(obj, np.ndarray):
            return _tagged('np.ndarray', obj.tolist())
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, datetime.datetime):
            return _tagged('datetime.datetime', str(obj))
        if isinstance(obj, datetime.timedelta):
            return _tagged('datetime.timedelta', {'days': obj.days, 'seconds': obj.seconds, 'microseconds': obj.microseconds})
        if isinstance(obj, (set, range)):
            return _tagged(type(obj).__name__, repr(obj))
        if isinstance(obj, argparse.Namespace):
            return _tagged('argparse.Namespace', vars(obj))
        if type(obj).__name__ == 'DataFrame' and hasattr(obj, 'to_json'):     # pandas only when the caller already uses it
            return _tagged('pd.DataFrame', obj.to_json(orient='split'))
        return super().default(obj)


def _untag(obj):
    kind = obj.get('_type')
    if kind is None:
        return obj
    value = obj['value']
    if kind == 'ResultsSlotsDict':
        return CRISPRessoCOREResources.ResultsSlotsDict(**value)
    if kind == 'np.ndarray':
        return np.array(value)
    if kind == 'pd.DataFrame':
        import pandas as pd
        return pd.read_json(io.StringIO(value), orient='split')
    if kind == 'datetime.datetime':
        return datetime.datetime.fromisoformat(value)
    if kind == 'datetime.timedelta':
        return datetime.timedelta(days=value['days'], seconds=value['seconds'], microseconds=value['microseconds'])
    if kind == 'set':
        return eval(value)                                            # the reference's decoder does the same (:888)
    if kind == 'range':
        start, end, step = _RANGE_RE.match(value).groups()
        return range(int(start), int(end), int(step)) if step is not None else range(int(start), int(end))
    if kind == 'argparse.Namespace':
        return argparse.Namespace(**value)
    return obj


class CRISPRessoJSONDecoder(json.JSONDecoder):
    """Inverse of the encoder (CRISPRessoShared.py:863-900)."""

    def __init__(self, *args, **kwargs):
        kwargs.setdefault('object_hook', _untag)
        super().__init__(*args, **kwargs)


def variant_line(key, variant):
    """One line of variants_<k>.tsv (CRISPRessoCORE.py:1231-1233)."""
    return "%s\t%s\n" % (key, json.dumps(variant, cls=CRISPRessoJSONEncoder))


def write_variant_file(variants_dir, process_id, keys, variants):
    """variants_<process_id>.tsv for a slice of the unique reads (the worker side, :1222-1240).  `keys` are the
    variantCache keys (the read, or `read1+read2` for pairs), `variants` their dicts in the same order."""
    path = os.path.join(variants_dir, "variants_%d.tsv" % process_id)
    with open(path, 'w') as fh:
        chunk = []
        for k, (key, variant) in enumerate(zip(keys, variants)):
            chunk.append(variant_line(key, variant))
            if len(chunk) == 10000:
                fh.write("".join(chunk))
                chunk = []
        fh.write("".join(chunk))
    return path


def read_variant_file(path):
    """-> iterator of (key, variant dict) (the parent side's parsing, :1913-1922)."""
    with open(path, 'r') as fh:
        for line in fh:
            parts = line.strip().split('\t')
            if len(parts) != 2:
                raise ValueError("Could not parse variant from file %s: %r" % (path, line[:80]))
            yield parts[0], json.loads(parts[1], cls=CRISPRessoJSONDecoder)


def new_aln_stats():
    return dict(N_TOT_READS=0, N_CACHED_ALN=0, N_CACHED_NOTALN=0, N_COMPUTED_ALN=0, N_COMPUTED_NOTALN=0, N_GLOBAL_SUBS=0,
                N_SUBS_OUTSIDE_WINDOW=0, N_MODS_IN_WINDOW=0, N_MODS_OUTSIDE_WINDOW=0, N_READS_IRREGULAR_ENDS=0, READ_LENGTH=0)


def account_variant(st, variant, count, names):
    """The statistics one aligned variant adds (:1935-1949 / :1968-1979); `names` = the references it is counted under."""
    for name in names:
        p = variant["variant_" + name]
        if st['READ_LENGTH'] == 0:
            st['READ_LENGTH'] = len(p['aln_seq'])
        st['N_GLOBAL_SUBS'] += (p['substitution_n'] + p['substitutions_outside_window']) * count
        st['N_SUBS_OUTSIDE_WINDOW'] += p['substitutions_outside_window'] * count
        st['N_MODS_IN_WINDOW'] += p['mods_in_window'] * count
        st['N_MODS_OUTSIDE_WINDOW'] += p['mods_outside_window'] * count
        if p['irregular_ends']:
            st['N_READS_IRREGULAR_ENDS'] += count


def merge_variant_files(paths, variantCache, args):
    """The parent side of the reference's multi-process route (CRISPRessoCORE.py:1900-1985): `variantCache` maps each
    unique read to its number of copies on entry; on return aligned reads map to their variant dicts (with 'count'),
    reads that did not align are removed and returned separately.  -> (aln_stats, not_aligned_variants)."""
    st = new_aln_stats()
    not_aligned = {}
    n_unique = len(variantCache)
    for path in paths:
        for seq, variant in read_variant_file(path):
            count = variantCache[seq]
            st['N_TOT_READS'] += count
            variant['count'] = count
            if variant['best_match_score'] <= 0:
                st['N_COMPUTED_NOTALN'] += 1
                st['N_CACHED_NOTALN'] += count - 1
                not_aligned[seq] = variant
                continue
            variantCache[seq] = variant
            st['N_COMPUTED_ALN'] += 1
            st['N_CACHED_ALN'] += count - 1
            if len(variant['aln_ref_names']) == 1 or args.expand_ambiguous_alignments:      # :1935
                account_variant(st, variant, count, variant['aln_ref_names'])
    if st['N_COMPUTED_ALN'] + st['N_COMPUTED_NOTALN'] != n_unique:
        raise ValueError("Number of unique reads in the variant files does not match the number of unique reads of the fastq file")
    for seq in not_aligned:
        del variantCache[seq]
    return st, not_aligned


# ---------------------------------------------------------------------------------------------------------------------
# --fastq_output

def _scores_and_details(variant):
    return (" ALN_SCORES=" + '&'.join(str(x) for x in variant['aln_scores'])
            + " ALN_DETAILS=" + '&'.join(','.join(str(y) for y in x) for x in variant['ref_aln_details']))


def crispresso2_annotation(variant):
    """The text appended to the '+' line of a read (CRISPRessoCORE.py:2300-2342).  A read that did not align gets
    ` ALN=NA` and its scores; an aligned read gets, per reference it was assigned to (joined by '&'): the edit counts
    `D<n>;I<n>;S<n>`, deletions `start(size)`, insertions `start(size+bases)`, substitution positions, and the two
    aligned strings."""
    if variant['best_match_score'] <= 0:
        return " ALN=NA" + _scores_and_details(variant)
    mods, dels, inss, subs, aln_refs, aln_seqs = [], [], [], [], [], []
    for name in variant['aln_ref_names']:
        p = variant['variant_' + name]
        dels.append(';'.join("%s(%s)" % (c[0], s) for c, s in zip(p['deletion_coordinates'], p['deletion_sizes'])))
        cols = []
        for c, s in zip(p['insertion_coordinates'], p['insertion_sizes']):
            at = p['ref_positions'].index(c[0]) + 1                   # first alignment column of the inserted bases
            cols.append("%s(%s+%s)" % (c[0], s, p['aln_seq'][at:at + s]))
        inss.append(';'.join(cols))
        subs.append(';'.join(str(x) for x in p['substitution_positions']))
        mods.append("D%d;I%d;S%d" % (int(p['deletion_n']), int(p['insertion_n']), int(p['substitution_n'])))
        aln_refs.append(p['aln_ref'])
        aln_seqs.append(p['aln_seq'])
    return (" ALN=" + "&".join(variant['aln_ref_names']) + _scores_and_details(variant)
            + " CLASS=" + variant['class_name'] + " MODS=" + "&".join(mods) + " DEL=" + "&".join(dels)
            + " INS=" + "&".join(inss) + " SUB=" + "&".join(subs)
            + " ALN_REF=" + '&'.join(aln_refs) + " ALN_SEQ=" + '&'.join(aln_seqs))


def _open_text(path, mode='rt'):
    return gzip.open(path, mode) if path.endswith('.gz') else open(path, mode.replace('t', ''))


def write_annotated_fastq(fastq_input, fastq_output, variantCache, not_aligned_variants):
    """Second half of process_fastq_write_out (:2289-2348): the input FASTQ, record by record, with the annotation of its
    read on the '+' line, gzip'ed.  Like the reference it stores the text under 'crispresso2_annotation' in each aligned
    read's dict.  A read found in neither dict re-uses the previous read's aligned variant in the reference (a stale
    loop variable); here that is an error."""
    notes = {}
    with gzip.open(fastq_output, 'wt') as out, _open_text(fastq_input) as src:
        fastq_id = src.readline()
        while fastq_id:
            seq = src.readline().strip()
            plus = src.readline().strip()
            qual = src.readline()
            note = notes.get(seq)
            if note is None:
                if seq in not_aligned_variants:
                    note = crispresso2_annotation(not_aligned_variants[seq])
                elif seq in variantCache:
                    note = crispresso2_annotation(variantCache[seq])
                    variantCache[seq]['crispresso2_annotation'] = note
                else:
                    raise KeyError("read of %s is in neither the aligned nor the not-aligned variants: %r" % (fastq_input, seq[:60]))
                notes[seq] = note
            out.write(fastq_id + seq + "\n" + plus + note + "\n" + qual)
            fastq_id = src.readline()


# ---------------------------------------------------------------------------------------------------------------------
# --bam_output: the SAM text (process_single_fastq_write_bam_out, CRISPRessoCORE.py:2351-2515)

# CRISPRessoShared.CIGAR_LOOKUP (CRISPRessoShared.py:426-434): (read column, reference column) -> operation; any other pair
# of characters is a KeyError there and here
_CIGAR_OP = {}
for _a in "ACGTN":
    for _b in "ACGTN":
        _CIGAR_OP[(_a, _b)] = 'M'
    _CIGAR_OP[(_a, '-')] = 'I'
    _CIGAR_OP[('-', _a)] = 'D'
del _a, _b


def cigar_elements(aln_seq, aln_ref):
    """`unexplode_cigar(''.join(CIGAR_LOOKUP[x] for x in zip(aln_seq, aln_ref)))` (CRISPRessoShared.py:561-582): run-length
    elements such as ['93M', '3D', '127M']."""
    els, prev, run = [], None, 0
    for pair in zip(aln_seq, aln_ref):
        op = _CIGAR_OP[pair]
        if op == prev:
            run += 1
        else:
            if prev is not None:
                els.append(str(run) + prev)
            prev, run = op, 1
    if prev is not None:
        els.append(str(run) + prev)
    return els


_COMPLEMENT = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N', '_': '_', '-': '-'}


def sam_entry(fastq_id, seq, qual, variant, refs):
    """The eleven SAM columns + the `c2:Z:` field of one read (:2398-2496) as a list of strings.  A read that did not align
    is unmapped (flag 4); an aligned read sits on its first assigned reference's contig (`refs[name]['aln_chr']`,
    `['aln_start']`), flag 16 when read and contig strands differ; on a '-' strand contig the CIGAR elements, the read and
    the qualities are reversed (:2464-2472).  The optional field is the --fastq_output annotation behind `c2:Z:`."""
    note = "c2:Z:" + crispresso2_annotation(variant)[1:]
    if variant['best_match_score'] <= 0:
        return [fastq_id, '4', '*', '0', '0', '*', '*', '0', '0', seq, qual, note]
    first = variant['aln_ref_names'][0]
    p = variant['variant_' + first]
    els = cigar_elements(p['aln_seq'], p['aln_ref'])
    flag = '16' if p['aln_strand'] == '-' else '0'
    cigar = ''.join(els)
    if refs[first]['aln_strand'] == '-':
        flag = '0' if flag == '16' else '16'
        cigar = ''.join(els[::-1])
        seq = ''.join(_COMPLEMENT[c] for c in reversed(seq.upper()))       # CRISPRessoShared.reverse_complement (:399-403)
        qual = qual[::-1]
    return [fastq_id, flag, refs[first]['aln_chr'], str(refs[first]['aln_start']), str(int(variant['best_match_score'])), cigar,
            '*', '0', '0', seq, qual, note]


def write_annotated_sam(fastq_input, sam_output, bam_header, variantCache, not_aligned_variants, refs):
    """Second half of process_single_fastq_write_bam_out (:2385-2500): `bam_header`, then one SAM line per input record, in
    input order.  Like the reference: the id is the stripped first line without its first character and an empty id ends the
    loop; a read in neither dict writes nothing; every aligned read's dict gets its columns under 'sam_entry'.  What the
    reference does next -- `samtools sort` + `samtools index` into the .bam -- is the caller's business (samtools is an
    external program there too)."""
    with open(sam_output, 'wt') as out, _open_text(fastq_input) as src:
        out.write(bam_header)
        fastq_id = src.readline().strip()[1:]
        while fastq_id:
            seq = src.readline().strip()
            src.readline()
            qual = src.readline().strip()
            if seq in not_aligned_variants:
                out.write("\t".join(sam_entry(fastq_id, seq, qual, not_aligned_variants[seq], refs)) + "\n")
            if seq in variantCache:
                variant = variantCache[seq]
                entry = sam_entry(fastq_id, seq, qual, variant, refs)
                variant['sam_entry'] = entry
                out.write("\t".join(entry) + "\n")
            fastq_id = src.readline().strip()[1:]
