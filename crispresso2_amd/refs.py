"""Reference-amplicon records for the hot path: the subset of `refs[name]` that the reference builds in
CRISPRessoCORE.py:3205-3268 and that the align + classify path reads (sequence, gap_incentive, include_idxs,
min_aln_score, forward / reverse-complement alignment seeds)."""
import numpy as np

_COMPLEMENT = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N', '_': '_', '-': '-'}


def reverse_complement(seq):
    """CRISPRessoShared.reverse_complement (CRISPRessoShared.py:399-403): upper-case, ACGTN_- only (KeyError otherwise)."""
    return "".join([_COMPLEMENT[c] for c in seq.upper()[-1::-1]])


def alignment_seeds(seq, exclude_bp_from_left=15, exclude_bp_from_right=15, aln_seed_len=10, aln_seed_count=5):
    """Strand-detection seeds as CRISPRessoCORE.py:3209-3234 picks them: one candidate every `aln_seed_count` bases inside
    the non-excluded region; a candidate that also occurs in the reverse complement (or was already taken) slides right
    (up to 100 tries); candidates whose reverse complement occurs in the amplicon are dropped.
    -> (fw_seeds, rc_seeds)"""
    L = len(seq)
    seq_rc = reverse_complement(seq)
    fw, rc = [], []
    for start0 in range(exclude_bp_from_left, L - exclude_bp_from_right - aln_seed_len, aln_seed_count):
        start, tries = start0, 0
        cand = seq[start:start + aln_seed_len]
        while cand in seq_rc or cand in fw:
            tries += 1
            if tries > 100:
                break
            if start0 > L - aln_seed_len:
                start = 0
            start += 1
            cand = seq[start:start + aln_seed_len]
        cand_rc = reverse_complement(cand)
        if cand_rc in seq:
            continue
        if cand not in seq_rc:
            fw.append(cand)
            rc.append(cand_rc)
    return fw, rc


def make_ref(name, sequence, cut_points, include_idxs, min_aln_score=60, gap_incentive_value=1, **seed_kw):
    """One `refs[name]` record.  gap_incentive[cut+1] = --needleman_wunsch_gap_incentive for every cut point (:3205-3207)."""
    L = len(sequence)
    g = np.zeros(L + 1, dtype=int)
    for c in cut_points:
        g[c + 1] = gap_incentive_value
    fw, rc = alignment_seeds(sequence, **seed_kw)
    return {'name': name, 'sequence': sequence, 'sequence_length': L, 'min_aln_score': min_aln_score,
            'gap_incentive': g, 'sgRNA_cut_points': list(cut_points), 'include_idxs': np.array(list(include_idxs)),
            'fw_seeds': fw, 'rc_seeds': rc}
