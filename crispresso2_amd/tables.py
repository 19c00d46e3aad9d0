"""Result tables of a run, written from the reduced count tensor in the reference's own formats:

    CRISPResso_quantification_of_editing_frequency.txt            CRISPRessoCORE.py:4554-4582
    <ref>Nucleotide_frequency_table.txt, <ref>Nucleotide_percentage_table.txt,
    <ref>Quantification_window_nucleotide_{frequency,percentage}_table.txt      plots/data_prep.py:3509-3570 (pandas to_csv of float vectors)
    <ref>Modification_count_vectors.txt, <ref>Quantification_window_modification_count_vectors.txt   CRISPRessoCORE.py:4604-4609, :4668-4687
    Alleles_frequency_table.txt (unzipped)                                                           CRISPRessoCORE.py:3926-4010, :4298-4303, :4498-4509

`res` is a pipeline.QuantResult.  File names carry the reference's prefix rule: no prefix for a single amplicon named
'Reference', else '<name>.' (CRISPRessoCORE.py:4618-4640).  The reference accumulates these vectors in float64 numpy
arrays and prints them with str() / pandas' float repr, hence '235.0'; the 'Total' row is a list of Python ints.
"""
import os


def _f(x):
    """pandas.to_csv / str(numpy.float64) text of a float: the shortest round-trip repr."""
    return repr(float(x))


def ref_plot_name(ref_names, name):
    if len(ref_names) == 1 and name == "Reference":
        return ""
    return name + "."


def write_quantification_of_editing_frequency(res, ref_names, path):
    head = ("Amplicon\tUnmodified%\tModified%\tReads_in_input\tReads_aligned_all_amplicons\tReads_aligned\tUnmodified\tModified\tDiscarded\t"
            "Insertions\tDeletions\tSubstitutions\tOnly Insertions\tOnly Deletions\tOnly Substitutions\tInsertions and Deletions\t"
            "Insertions and Substitutions\tDeletions and Substitutions\tInsertions Deletions and Substitutions\n")
    with open(path, "w") as fh:
        fh.write(head)
        for name in ref_names:
            c = res.per_ref[name]
            n_aligned = c["counts_total"]
            unmod_pct = mod_pct = "NA"
            if n_aligned > 0:
                unmod_pct = round(100 * c["counts_unmodified"] / float(n_aligned), 8)
                mod_pct = round(100 * c["counts_modified"] / float(n_aligned), 8)
            vals = [unmod_pct, mod_pct, res.stats.get("N_READS_INPUT", res.stats["N_TOT_READS"]), res.stats["N_TOTAL"], n_aligned,
                    c["counts_unmodified"], c["counts_modified"], c["counts_discarded"], c["counts_insertion"], c["counts_deletion"],
                    c["counts_substitution"], c["counts_only_insertion"], c["counts_only_deletion"], c["counts_only_substitution"],
                    c["counts_insertion_and_deletion"], c["counts_insertion_and_substitution"], c["counts_deletion_and_substitution"],
                    c["counts_insertion_and_deletion_and_substitution"]]
            fh.write("\t".join([name] + [str(x) for x in vals]) + "\n")


def _write_frame(path, columns, rows, index):
    with open(path, "w") as fh:
        fh.write("\t" + "\t".join(columns) + "\n")
        for label, row in zip(index, rows):
            fh.write(label + "\t" + "\t".join(_f(x) for x in row) + "\n")


def _write_count_vectors(path, ref_seq, vectors, names):
    with open(path, "w") as fh:
        fh.write("Sequence\t" + "\t".join(list(ref_seq)) + "\n")
        for vec, nm in zip(vectors, names):
            fh.write(nm + "\t" + "\t".join(vec) + "\n")


def write_alleles_frequency_table(res, path):
    """Alleles_frequency_table.txt (the reference zips it): CRISPRessoCORE.py:4498-4509, the non-detailed columns."""
    with open(path, "w") as fh:
        fh.write("Aligned_Sequence\tReference_Sequence\tReference_Name\tRead_Status\tn_deleted\tn_inserted\tn_mutated\t#Reads\t%Reads\n")
        for a, r, name, status, dn, inn, sn, reads, pct in res.alleles():
            fh.write("%s\t%s\t%s\t%s\t%d\t%d\t%d\t%d\t%s\n" % (a, r, name, status, dn, inn, sn, reads, _f(pct)))


def write_tables(res, refs, ref_names, out_dir):
    """Writes the tables listed in the module docstring into out_dir; returns the list of file names."""
    os.makedirs(out_dir, exist_ok=True)
    written = ["CRISPResso_quantification_of_editing_frequency.txt", "Alleles_frequency_table.txt"]
    write_quantification_of_editing_frequency(res, ref_names, os.path.join(out_dir, written[0]))
    write_alleles_frequency_table(res, os.path.join(out_dir, written[1]))
    nucs = ["A", "C", "G", "T", "N", "-"]
    for name in ref_names:
        c = res.per_ref[name]
        total = c["counts_total"]
        seq = refs[name]["sequence"]
        L = len(seq)
        prefix = ref_plot_name(ref_names, name)
        fl = lambda v: [_f(x) for x in v[:L]]
        ins, dele, sub = c["insertion_count_vectors"], c["deletion_count_vectors"], c["substitution_count_vectors"]
        a_ins, a_insl = c["all_insertion_count_vectors"], c["all_insertion_left_count_vectors"]
        a_del, a_sub = c["all_deletion_count_vectors"], c["all_substitution_count_vectors"]
        tot_row = [str(total)] * L
        fn = prefix + "Quantification_window_modification_count_vectors.txt"
        _write_count_vectors(os.path.join(out_dir, fn), seq, [fl(ins), fl(dele), fl(sub), fl(ins + dele + sub), tot_row],
                             ["Insertions", "Deletions", "Substitutions", "All_modifications", "Total"])
        written.append(fn)
        fn = prefix + "Modification_count_vectors.txt"
        _write_count_vectors(os.path.join(out_dir, fn), seq, [fl(a_ins), fl(a_insl), fl(a_del), fl(a_sub), fl(a_ins + a_del + a_sub), tot_row],
                             ["Insertions", "Insertions_Left", "Deletions", "Substitutions", "All_modifications", "Total"])
        written.append(fn)
        if total < 1:                                              # plots/data_prep.py:3510
            continue
        inc = [int(x) for x in refs[name]["include_idxs"]]
        rows_all = [[float(x) for x in c["all_base_count_vectors_" + n][:L]] for n in nucs]
        rows_win = [[r[x] for x in inc] for r in rows_all]
        win_seq = [seq[x] for x in inc]
        for fn, cols, rows in ((prefix + "Quantification_window_nucleotide_frequency_table.txt", win_seq, rows_win),
                               (prefix + "Quantification_window_nucleotide_percentage_table.txt", win_seq, [[x / total for x in r] for r in rows_win]),
                               (prefix + "Nucleotide_frequency_table.txt", list(seq), rows_all),
                               (prefix + "Nucleotide_percentage_table.txt", list(seq), [[x / total for x in r] for r in rows_all])):
            _write_frame(os.path.join(out_dir, fn), cols, rows, nucs)
            written.append(fn)
    return written
