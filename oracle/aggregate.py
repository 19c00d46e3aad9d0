"""CPU restatement of the reference's per-amplicon aggregation -- TEST INFRASTRUCTURE ONLY.

Follows CRISPRessoCORE.py:3964-4115 (the "Quantifying indels/substitutions" loop, non-coding case: no exons) and
process_fastq's aln_stats (:1974-1979), taking per-read classifier payloads (the reference's / the oracle's
find_indels_substitutions output) and read multiplicities.  Returns plain numpy vectors keyed by the reference's names.
Pinned by runs of the reference itself, not only by reading it: aggregate() by the nucleotide / modification count tables of the eight whole-run goldens
(tests/test_whole_run_tables.py: the reference's main() wrote those files from its own loop; the tables built from this function's vectors equal them byte
for byte), allele_table_text() / alleles_around_cut() by the same runs' Alleles_frequency_table.txt and ..._around_sgRNA_... files."""
from collections import Counter

import numpy as np


def aggregate(items, ref_len, ignore_substitutions=False, ignore_insertions=False, ignore_deletions=False,
              discard_indel_reads=False):
    """items: iterable of (payload dict incl. 'aln_seq', count)."""
    L = ref_len
    v = {k: np.zeros(L) for k in (
        "all_insertion_count_vectors", "all_insertion_left_count_vectors", "all_deletion_count_vectors",
        "all_substitution_count_vectors", "insertion_count_vectors", "deletion_count_vectors", "substitution_count_vectors",
        "insertion_length_vectors", "deletion_length_vectors")}
    for n in "ACGTN":
        v["all_substitution_base_vectors_" + n] = np.zeros(L)
    for n in "ACGTN-":
        v["all_base_count_vectors_" + n] = np.zeros(L)
    s = Counter()
    h = {k: Counter() for k in ("inserted_n", "deleted_n", "substituted_n", "effective_len")}
    for p, count in items:
        # aln_stats, CRISPRessoCORE.py:1974-1979 (payload fields of :734-744)
        all_ins = len(p["all_insertion_positions"]) // 2
        subs_out = len(p["all_substitution_positions"]) - len(p["substitution_positions"])
        total_mods = all_ins + len(p["all_deletion_positions"]) + len(p["all_substitution_positions"])
        in_win = p["substitution_n"] + p["deletion_n"] + p["insertion_n"]
        s["N_GLOBAL_SUBS"] += (p["substitution_n"] + subs_out) * count
        s["N_SUBS_OUTSIDE_WINDOW"] += subs_out * count
        s["N_MODS_IN_WINDOW"] += in_win * count
        s["N_MODS_OUTSIDE_WINDOW"] += (total_mods - in_win) * count
        s1, s2 = p["aln_seq"], p["aln_ref"]
        if (s1[0] == "-" or s2[0] == "-" or s1[0] != s2[0]) or (s1[-1] == "-" or s2[-1] == "-" or s1[-1] != s2[-1]):
            s["N_READS_IRREGULAR_ENDS"] += count
        s["alignments_counted"] += 1
        if discard_indel_reads and (p["deletion_n"] > 0 or p["insertion_n"] > 0):      # :3996-4000
            s["counts_discarded"] += count
            continue
        s["counts_total"] += count
        modified = ((not ignore_deletions and p["deletion_n"] > 0) or (not ignore_insertions and p["insertion_n"] > 0)
                    or (not ignore_substitutions and p["substitution_n"] > 0))          # :746-760
        s["counts_modified" if modified else "counts_unmodified"] += count
        eff = L
        has_ins = has_del = has_sub = False
        v["all_insertion_count_vectors"][p["all_insertion_positions"]] += count          # :4016 (fancy +=: repeats count once)
        v["all_insertion_left_count_vectors"][p["all_insertion_left_positions"]] += count
        if not ignore_insertions:
            h["inserted_n"][p["insertion_n"]] += count
            v["insertion_count_vectors"][p["insertion_positions"]] += count
            eff += p["insertion_n"]
            if p["insertion_n"] > 0:
                s["counts_insertion"] += count
                has_ins = True
        v["all_deletion_count_vectors"][p["all_deletion_positions"]] += count
        if not ignore_deletions:
            h["deleted_n"][p["deletion_n"]] += count
            v["deletion_count_vectors"][p["deletion_positions"]] += count
            eff -= p["deletion_n"]
            if p["deletion_n"] > 0:
                s["counts_deletion"] += count
                has_del = True
        h["effective_len"][eff] += count
        v["all_substitution_count_vectors"][p["all_substitution_positions"]] += count
        if not ignore_substitutions:
            h["substituted_n"][p["substitution_n"]] += count
            v["substitution_count_vectors"][p["substitution_positions"]] += count
            if p["substitution_n"] > 0:
                s["counts_substitution"] += count
                has_sub = True
            for nuc in "ATCGN":                                                          # :4048-4054
                locs = [q for q, b in zip(p["all_substitution_positions"], p["all_substitution_values"]) if b == nuc]
                if locs:
                    v["all_substitution_base_vectors_" + nuc][locs] += count
        if has_del:                                                                      # :4058-4072
            s["counts_insertion_and_deletion_and_substitution" if (has_ins and has_sub) else
              "counts_insertion_and_deletion" if has_ins else
              "counts_deletion_and_substitution" if has_sub else "counts_only_deletion"] += count
        elif has_ins:
            s["counts_insertion_and_substitution" if has_sub else "counts_only_insertion"] += count
        elif has_sub:
            s["counts_only_substitution"] += count
        for c, rp in zip(s1, p["ref_positions"]):                                        # :4075-4081
            if rp >= 0:
                v["all_base_count_vectors_" + c][rp] += count
        if has_ins or has_del or has_sub:                                                # :4085, :4104-4115
            for (a, b), sz in zip(p["insertion_coordinates"], p["insertion_sizes"]):
                v["insertion_length_vectors"][a] += sz * count
                v["insertion_length_vectors"][b] += sz * count
            for (a, b), sz in zip(p["deletion_coordinates"], p["deletion_sizes"]):
                v["deletion_length_vectors"][list(range(a, b))] += sz * count
    out = {k: x.astype(np.int64) for k, x in v.items()}
    out.update({k: int(s[k]) for k in (
        "counts_total", "counts_modified", "counts_unmodified", "counts_discarded", "counts_insertion", "counts_deletion",
        "counts_substitution", "counts_only_insertion", "counts_only_deletion", "counts_only_substitution",
        "counts_insertion_and_deletion", "counts_insertion_and_substitution", "counts_deletion_and_substitution",
        "counts_insertion_and_deletion_and_substitution", "N_GLOBAL_SUBS", "N_SUBS_OUTSIDE_WINDOW", "N_MODS_IN_WINDOW",
        "N_MODS_OUTSIDE_WINDOW", "N_READS_IRREGULAR_ENDS", "alignments_counted")})
    out.update({k: {int(a): int(b) for a, b in c.items() if b} for k, c in h.items()})
    return out


def alleles_around_cut(rows, ref_name, cut_point, ref_len, plot_window_size=20):
    """Restatement with pandas of CRISPRessoShared.get_dataframe_around_cut_asymmetrical (CRISPRessoShared.py:1513-1531) and
    the window of plots/data_prep.py:285-301, on allele table rows (Aligned_Sequence, Reference_Sequence, Reference_Name,
    Read_Status, n_deleted, n_inserted, n_mutated, #Reads, %Reads).  -> the text pandas' to_csv writes (CRISPRessoCORE.py:5272)."""
    import io
    import pandas as pd
    left = plot_window_size if cut_point - plot_window_size + 1 >= 0 else cut_point + 1
    right = plot_window_size if cut_point + plot_window_size < ref_len else ref_len - cut_point - 1
    df = pd.DataFrame([r for r in rows if r[2] == ref_name], columns=['Aligned_Sequence', 'Reference_Sequence', 'Reference_Name', 'Read_Status',
                                                                     'n_deleted', 'n_inserted', 'n_mutated', '#Reads', '%Reads'])

    def positions(ref_al):                                            # CRISPRessoCOREResources.pyx:105-133
        out, idx = [], 0
        for c in ref_al:
            if c != '-':
                out.append(idx)
                idx += 1
            else:
                out.append(-1 if idx == 0 else -idx)
        return out

    def cut_row(row):
        k = positions(row['Reference_Sequence']).index(cut_point)
        return (row['Aligned_Sequence'][k - left + 1:k + right + 1], row['Reference_Sequence'][k - left + 1:k + right + 1],
                row['Read_Status'] == 'UNMODIFIED', row['n_deleted'], row['n_inserted'], row['n_mutated'], row['#Reads'], row['%Reads'])
    d = pd.DataFrame(list(df.apply(cut_row, axis=1).values),
                     columns=['Aligned_Sequence', 'Reference_Sequence', 'Unedited', 'n_deleted', 'n_inserted', 'n_mutated', '#Reads', '%Reads'])
    d = d.groupby(['Aligned_Sequence', 'Reference_Sequence', 'Unedited', 'n_deleted', 'n_inserted', 'n_mutated']).sum().reset_index().set_index('Aligned_Sequence')
    d.sort_values(by=['#Reads', 'Aligned_Sequence', 'Reference_Sequence'], inplace=True, ascending=[False, True, True])
    d['Unedited'] = d['Unedited'] > 0
    buf = io.StringIO()
    d.to_csv(buf, sep='\t', header=True, index=True, index_label=d.index.name)
    return buf.getvalue()


def remap_to_first_reference(first_ref_vectors, items_by_ref, ref1_len, ref1_include_idxs):
    """Restatement of the 'all reads with respect to ref1' arrays (CRISPRessoCORE.py:4195-4264), built when an expected HDR
    amplicon or a prime-editing extension is given.  first_ref_vectors: the aggregate() result of the first reference (its
    all_* vectors are copied, :4217-4224); items_by_ref: {name of another reference: [(s1, s2, count)]} with the alignment of
    every read counted for that reference AGAINST THE FIRST reference (variant['ref_aln_details'][0]).
    -> {name: {vector name: numpy vector of length ref1_len}} for the first reference (key None) and the others."""
    import oracle
    names = ("all_insertion_count_vectors", "all_insertion_left_count_vectors", "all_deletion_count_vectors",
             "all_substitution_count_vectors")
    out = {None: {k: first_ref_vectors[k].copy() for k in names}}
    out[None]["all_indelsub_count_vectors"] = (first_ref_vectors["all_insertion_count_vectors"] + first_ref_vectors["all_deletion_count_vectors"]
                                               + first_ref_vectors["all_substitution_count_vectors"])
    for n in "ACGTN-":
        out[None]["all_base_count_vectors_" + n] = first_ref_vectors["all_base_count_vectors_" + n].copy()
    for name, items in items_by_ref.items():
        v = {k: np.zeros(ref1_len) for k in names + ("all_indelsub_count_vectors",)}
        for n in "ACGTN-":
            v["all_base_count_vectors_" + n] = np.zeros(ref1_len)
        for s1, s2, count in items:
            payload = oracle.find_indels_substitutions(s1, s2, ref1_include_idxs)                      # :4247
            v["all_insertion_count_vectors"][payload['all_insertion_positions']] += count               # :4256-4264
            v["all_insertion_left_count_vectors"][payload['all_insertion_left_positions']] += count
            v["all_indelsub_count_vectors"][payload['all_insertion_positions']] += count
            v["all_deletion_count_vectors"][payload['all_deletion_positions']] += count
            v["all_indelsub_count_vectors"][payload['all_deletion_positions']] += count
            v["all_substitution_count_vectors"][payload['all_substitution_positions']] += count
            v["all_indelsub_count_vectors"][payload['all_substitution_positions']] += count
            ref_pos = payload['ref_positions']
            for i in range(len(s1)):                                                                    # :4266-4270
                if ref_pos[i] < 0:
                    continue
                v["all_base_count_vectors_" + s1[i]][ref_pos[i]] += count
        out[name] = v
    return out


def select_best(scores, scores_rc, min_aln_scores, assign_first=False, expand=False):
    """Strand and best-reference choice of get_new_variant_object, per read, as the reference's loop does it
    (CRISPRessoCORE.py:683 strand, :697-707 best match, :710 aligned, :779-785 ambiguous reads).
    scores: [k] floats of the alignments on the seeds' strand; scores_rc: [k] floats or None per reference (the
    reverse-complement alignment of a pair aligned on both strands).  -> (best_match_names as indices, use_rc flags [k],
    aligned, counted-for indices, ambiguous)"""
    best_match_score = -1
    best = []
    use_rc = [False] * len(scores)
    for idx in range(len(scores)):
        score = scores[idx]
        if scores_rc is not None and scores_rc[idx] is not None and scores_rc[idx] > score:      # :683
            score = scores_rc[idx]
            use_rc[idx] = True
        if score > best_match_score and score > min_aln_scores[idx]:                              # :697
            best_match_score = score
            best = [idx]
        elif score == best_match_score:                                                           # :703
            best.append(idx)
    aligned = best_match_score > 0                                                                # :710
    if not aligned:
        return [], use_rc, False, [], False
    counted, ambiguous = list(best), False
    if len(best) > 1:                                                                             # :779-785
        if assign_first:
            counted = best[:1]
        elif not expand:
            counted, ambiguous = [], True
    return best, use_rc, True, counted, ambiguous


def allele_table_text(rows, n_total, dsODN=""):
    """Restatement with pandas of the allele table's frame, sort and file (CRISPRessoCORE.py:4298-4303, :4498-4530): `rows` are the
    alleles_list entries in the order the reference appends them, as tuples (Aligned_Sequence, Reference_Sequence, Reference_Name,
    Read_Status, n_deleted, n_inserted, n_mutated, #Reads).  -> the text of Alleles_frequency_table.txt"""
    import io
    import pandas as pd
    cols = ["Aligned_Sequence", "Reference_Sequence", "Reference_Name", "Read_Status", "n_deleted", "n_inserted", "n_mutated", "#Reads"]
    df = pd.DataFrame(list(rows), columns=cols)
    df['%Reads'] = df['#Reads'] / n_total * 100
    df[['n_deleted', 'n_inserted', 'n_mutated']] = df[['n_deleted', 'n_inserted', 'n_mutated']].astype(int)
    df.sort_values(by=['#Reads', 'Aligned_Sequence', 'Reference_Sequence'], inplace=True, ascending=[False, True, True])
    out_cols = cols + ['%Reads']
    if dsODN != "":
        from .fastq import reverse_complement
        df["contains dsODN fw"] = df["Aligned_Sequence"].str.find(dsODN) > 0
        df["contains dsODN rv"] = df["Aligned_Sequence"].str.find(reverse_complement(dsODN)) > 0
        df["contains dsODN"] = df["contains dsODN fw"] | df["contains dsODN rv"]
        out_cols.append("contains dsODN")
        if len(dsODN) > 6:
            sub = dsODN[3:-3]
            df["contains dsODN fragment fw"] = df["Aligned_Sequence"].str.find(sub) > 0
            df["contains dsODN fragment rv"] = df["Aligned_Sequence"].str.find(reverse_complement(sub)) > 0
            df["contains dsODN fragment"] = df["contains dsODN fragment fw"] | df["contains dsODN fragment rv"]
        out_cols.append("contains dsODN fragment")
    buf = io.StringIO()
    df.loc[:, out_cols].to_csv(buf, sep='\t', header=True, index=None)
    return buf.getvalue()
