"""Result tables of a run, written from the reduced count tensor in the reference's own formats:

    CRISPResso_quantification_of_editing_frequency.txt            CRISPRessoCORE.py:4554-4582
    <ref>Nucleotide_frequency_table.txt, <ref>Nucleotide_percentage_table.txt,
    <ref>Quantification_window_nucleotide_{frequency,percentage}_table.txt      plots/data_prep.py:3509-3570 (pandas to_csv of float vectors)
    <ref>Modification_count_vectors.txt, <ref>Quantification_window_modification_count_vectors.txt   CRISPRessoCORE.py:4604-4609, :4668-4687
    Alleles_frequency_table.txt (unzipped)                                                           CRISPRessoCORE.py:3926-4010, :4298-4303, :4498-4509
    CRISPResso_mapping_statistics.txt                                                                CRISPRessoCORE.py:4591-4594
    <ref>Effect_vector_{insertion,deletion,substitution,combined}.txt  (np.savetxt '%d', '%.18e')    CRISPRessoCORE.py:4380-4390, :4600-4602, :4650-4665
    <ref>Indel_histogram.txt, <ref>Insertion_histogram.txt, <ref>Deletion_histogram.txt, <ref>Substitution_histogram.txt
                                                                                                     CRISPRessoCORE.py:4349-4377, :4750-4775
    <ref>Alleles_frequency_table_around_<guide>.txt  (when refs[name] carries 'sgRNA_orig_sequences')   CRISPRessoShared.py:1513-1531,
                                                                                                     plots/data_prep.py:285-301, CRISPRessoCORE.py:5250-5273
(the reference writes the effect vectors and histograms only when it also draws its plots; here they are always written)

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


def write_alleles_frequency_table(res, path, dsODN="", zip_member=None):
    """Alleles_frequency_table.txt (the reference zips it): CRISPRessoCORE.py:4498-4527, the non-detailed columns.  With
    --dsODN two more columns say whether the aligned read contains the oligo, or the oligo without its first and last three
    bases, on either strand -- `str.find(...) > 0`, so a match at the very start of the read does not count (:4512-4524).
    A pipeline.QuantResult has the table on the device: its text is formed there and written by native threads
    (alleles.AlleleTable.write, c2_allele_table_write).  Any other `res` (rows in host memory, res.alleles()) is printed here."""
    table = res.allele_table() if hasattr(res, "allele_table") else None
    if table is not None:
        table.write(path, res.align_ref_names, res.stats["N_TOTAL"], dsODN=dsODN, zip_member=zip_member)
        return
    if zip_member:                                                   # (rows in host memory: the text below, zipped as the reference zips it)
        import tempfile
        import zipfile
        with tempfile.TemporaryDirectory(dir=os.path.dirname(os.path.abspath(path))) as tmp:
            txt = os.path.join(tmp, zip_member)
            write_alleles_frequency_table(res, txt, dsODN=dsODN)
            with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, allowZip64=True) as z:
                z.write(txt, zip_member)
        return
    from .refs import reverse_complement
    head = "Aligned_Sequence\tReference_Sequence\tReference_Name\tRead_Status\tn_deleted\tn_inserted\tn_mutated\t#Reads\t%Reads"
    probes = []
    if dsODN != "":
        if len(dsODN) <= 6:
            raise KeyError("contains dsODN fragment")                  # the reference selects a column it never made (:4519-4524)
        head += "\tcontains dsODN\tcontains dsODN fragment"
        probes = [(dsODN, reverse_complement(dsODN)), (dsODN[3:-3], reverse_complement(dsODN[3:-3]))]
    with open(path, "w") as fh:
        fh.write(head + "\n")
        for a, r, name, status, dn, inn, sn, reads, pct in res.alleles():
            extra = "".join("\t" + ("True" if (a.find(fw) > 0 or a.find(rv) > 0) else "False") for fw, rv in probes)
            fh.write("%s\t%s\t%s\t%s\t%d\t%d\t%d\t%d\t%s%s\n" % (a, r, name, status, dn, inn, sn, reads, _f(pct), extra))


def write_mapping_statistics(res, path):
    """CRISPResso_mapping_statistics.txt (:4591-4594).  Without a preprocessing step (no trimming / merging: out of scope
    here) the reads after preprocessing are the reads in the input."""
    st = res.stats
    n_in = st.get("N_READS_INPUT", st["N_TOT_READS"])
    vals = [n_in, st.get("N_READS_AFTER_PREPROCESSING", n_in), st["N_TOTAL"], st["N_COMPUTED_ALN"], st["N_CACHED_ALN"],
            st["N_COMPUTED_NOTALN"], st["N_CACHED_NOTALN"]]
    with open(path, "w") as fh:
        fh.write("READS IN INPUTS\tREADS AFTER PREPROCESSING\tREADS ALIGNED\tN_COMPUTED_ALN\tN_CACHED_ALN\tN_COMPUTED_NOTALN\tN_CACHED_NOTALN\n")
        fh.write("\t".join(str(x) for x in vals) + "\n")


def _write_effect_vector(path, vector):
    """save_vector_to_file (:4600-4602): np.savetxt of (position, value) rows, fmt '%d' / '%.18e', '# '-commented header."""
    with open(path, "w") as fh:
        fh.write("# amplicon position\teffect\n")
        for k, x in enumerate(vector):
            fh.write("%d\t%.18e\n" % (k + 1, x))


def _histogram_bins(hist, floor):
    """x = 0 .. max(floor, largest key), y = the counts (:4349-4358; missing keys count 0)."""
    top = max(floor, max(hist.keys() or [0]))
    return list(range(top + 1)), [int(hist.get(x, 0)) for x in range(top + 1)]


def _write_histogram(path, columns, xs, ys):
    with open(path, "w") as fh:
        fh.write("%s\t%s\n" % columns)
        for x, y in zip(xs, ys):
            fh.write("%d\t%d\n" % (x, y))


def alleles_around_cut(rows, ref_name, cut_point, ref_len, plot_window_size=20):
    """Rows of <ref>Alleles_frequency_table_around_<sgRNA>.txt from the allele table rows of `QuantResult.alleles()`:
    CRISPRessoShared.get_dataframe_around_cut_asymmetrical (CRISPRessoShared.py:1513-1531) with the window of
    plots/data_prep.py:285-301 (`plot_window_size` bases either side of the cut, clipped at the amplicon's ends).  Every allele
    of THIS reference is cut down to the alignment columns around the column that holds reference base `cut_point`
    (`ref_positions.index(cut_point)`: gap columns carry negative positions and never match), alleles that coincide there
    (same two strings, same Unedited flag and edit counts) are merged -- #Reads added, %Reads added the way pandas' groupby
    adds floats (Kahan-compensated, in table order) -- and sorted by #Reads descending, then the two strings.
    -> list of (Aligned_Sequence, Reference_Sequence, Unedited, n_deleted, n_inserted, n_mutated, #Reads, %Reads)"""
    left = plot_window_size if cut_point - plot_window_size + 1 >= 0 else cut_point + 1
    right = plot_window_size if cut_point + plot_window_size < ref_len else ref_len - cut_point - 1
    groups = {}
    for a, r, name, status, dn, inn, sn, reads, pct in rows:
        if name != ref_name:
            continue
        seen, cut_idx = -1, -1
        for col, ch in enumerate(r):
            if ch != '-':
                seen += 1
                if seen == cut_point:
                    cut_idx = col
                    break
        if cut_idx < 0:
            raise ValueError("%d is not in list" % cut_point)          # ref_positions.index(cut_point)
        key = (a[cut_idx - left + 1:cut_idx + right + 1], r[cut_idx - left + 1:cut_idx + right + 1],
               status == 'UNMODIFIED', int(dn), int(inn), int(sn))
        g = groups.get(key)
        if g is None:
            g = groups[key] = [0, 0.0, 0.0]                              # #Reads, %Reads, its compensation term
        g[0] += int(reads)
        y = float(pct) - g[2]
        t = g[1] + y
        g[2] = t - g[1] - y
        g[1] = t
    out = [k + (g[0], g[1]) for k, g in sorted(groups.items())]         # groupby(sort=True) order, then a stable sort
    out.sort(key=lambda t: (-t[6], t[0], t[1]))
    return out


def slugify(value):
    """CRISPRessoShared.slugify (CRISPRessoShared.py:418-423): file-name form of a guide label."""
    import re
    import unicodedata
    value = unicodedata.normalize('NFKD', value).encode('ascii', 'ignore')
    value = re.sub(rb'[\s\'*"/\\\[\]:;|,<>?]', b'_', value).strip()
    return re.sub(rb'_{2,}', b'_', value).decode('utf-8')


def write_alleles_around_cut(rows, path):
    with open(path, "w") as fh:
        fh.write("Aligned_Sequence\tReference_Sequence\tUnedited\tn_deleted\tn_inserted\tn_mutated\t#Reads\t%Reads\n")
        for a, r, unedited, dn, inn, sn, reads, pct in rows:
            fh.write("%s\t%s\t%s\t%d\t%d\t%d\t%d\t%s\n" % (a, r, "True" if unedited else "False", dn, inn, sn, reads, _f(pct)))


def write_reads_from_all_amplicons_tables(res, refs, ref_names, out_dir):
    """<first ref>Reads_from_all_amplicons_{modification,nucleotide}_percent_table.txt (CRISPRessoCORE.py:5104-5114 from
    plots/data_prep.py:247-282, :943-982): for every amplicon with reads, the insertion / deletion / substitution counts and
    the base counts of ITS reads in the coordinates of the FIRST amplicon (`res.first_ref_view`, the arrays of
    CRISPRessoCORE.py:4195-4270), divided by that amplicon's read count.  The reference passes these fractions through text
    (numpy concatenates them with the row labels into a string array) and through pandas' to_numeric before printing, which
    changes the last digits of some values; the same two steps are taken here so that the files agree byte for byte."""
    import numpy as np
    import pandas as pd
    view = res.first_ref_view
    with_reads = [nm for nm in ref_names if res.per_ref[nm]["counts_total"] > 0]
    first = with_reads[0]
    seq = refs[first]["sequence"]
    L = len(seq)

    def as_printed(label_cells, values):
        cells = np.concatenate((label_cells, values))[len(label_cells):]
        return [repr(float(x)) for x in pd.to_numeric(pd.Series(list(cells)), errors='coerce')]
    prefix = ref_plot_name(ref_names, first)
    written = []
    fn = prefix + "Reads_from_all_amplicons_modification_percent_table.txt"
    with open(os.path.join(out_dir, fn), "w") as fh:
        fh.write("Amplicon\tModification\t" + "\t".join(seq) + "\n")
        for nm in with_reads:
            tot = float(res.per_ref[nm]["counts_total"])
            v = view[nm]
            for label, key in (("Insertions", "all_insertion_count_vectors"), ("Insertions_Left", "all_insertion_left_count_vectors"),
                               ("Deletions", "all_deletion_count_vectors"), ("Substitutions", "all_substitution_count_vectors"),
                               ("All_modifications", "all_indelsub_count_vectors")):
                fh.write("\t".join([nm, label] + as_printed([nm, label], np.array(v[key][:L]).astype(float) / tot)) + "\n")
            fh.write("\t".join([nm, "Total"] + as_printed([nm, "Total"], [res.per_ref[nm]["counts_total"]] * L)) + "\n")
    written.append(fn)
    fn = prefix + "Reads_from_all_amplicons_nucleotide_percent_table.txt"
    with open(os.path.join(out_dir, fn), "w") as fh:
        fh.write("Amplicon\tNucleotide\t" + "\t".join(seq) + "\n")
        for nm in with_reads:
            tot = float(res.per_ref[nm]["counts_total"])
            for nuc in "ACGTN-":
                fh.write("\t".join([nm, nuc] + as_printed([nm, nuc], np.array(view[nm]["all_base_count_vectors_" + nuc][:L]).astype(float) / tot)) + "\n")
    written.append(fn)
    return written


def write_tables(res, refs, ref_names, out_dir, plot_window_size=20, dsODN="", timings=None, allele_table_zip=False):
    """Writes the tables listed in the module docstring into out_dir; returns the list of file names.
    allele_table_zip: the allele frequency table as Alleles_frequency_table.zip (one member, Alleles_frequency_table.txt) and no .txt -- the state the
    reference's run ends in (CRISPRessoCORE.py:4531-4533); the device route deflates on all host threads and writes only the compressed bytes.
    timings: optional dict that receives the wall seconds of the allele table's build (rows + sort on the device), its file, the
    around-cut files, and everything else."""
    import time
    import numpy as np
    allele_rows = None
    t_start = time.perf_counter()
    t_alleles = 0.0
    os.makedirs(out_dir, exist_ok=True)
    written = ["CRISPResso_quantification_of_editing_frequency.txt", "Alleles_frequency_table.zip" if allele_table_zip else "Alleles_frequency_table.txt"]
    write_quantification_of_editing_frequency(res, ref_names, os.path.join(out_dir, written[0]))
    t0 = time.perf_counter()
    if hasattr(res, "allele_table"):
        res.allele_table()
    t1 = time.perf_counter()
    write_alleles_frequency_table(res, os.path.join(out_dir, written[1]), dsODN=dsODN, zip_member="Alleles_frequency_table.txt" if allele_table_zip else None)
    t2 = time.perf_counter()
    t_alleles += t2 - t0
    t_around = 0.0
    if timings is not None:
        timings["allele_table_build"] = t1 - t0
        timings["allele_table_write"] = t2 - t1
    if all(k in res.stats for k in ("N_TOTAL", "N_COMPUTED_ALN", "N_CACHED_ALN", "N_COMPUTED_NOTALN", "N_CACHED_NOTALN")):
        written.append("CRISPResso_mapping_statistics.txt")
        write_mapping_statistics(res, os.path.join(out_dir, written[-1]))
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
        # effect vectors: 100 * window count vector / reads of this amplicon, zeros without reads (:4380-4415)
        win = {"insertion": ins[:L], "deletion": dele[:L], "substitution": sub[:L], "combined": (ins + dele + sub)[:L]}
        for kind, vec in win.items():
            fn = prefix + "Effect_vector_%s.txt" % kind
            v = 100 * np.asarray(vec, dtype=np.float64) / total if total > 0 else np.zeros(L)
            _write_effect_vector(os.path.join(out_dir, fn), v)
            written.append(fn)
        # histograms of the per-read counts in the quantification window and of the effective length (:4349-4377, :4750-4775)
        for fn, cols, hist, sign in ((prefix + "Insertion_histogram.txt", ("ins_size", "fq"), c["inserted_n"], 1),
                                     (prefix + "Deletion_histogram.txt", ("del_size", "fq"), c["deleted_n"], -1),
                                     (prefix + "Substitution_histogram.txt", ("sub_count", "fq"), c["substituted_n"], 1)):
            xs, ys = _histogram_bins(hist, 15)
            _write_histogram(os.path.join(out_dir, fn), cols, [sign * x for x in xs], ys)
            written.append(fn)
        eff = c["effective_len"]
        lo = min(L - 15, min(eff.keys() or [0]))                   # (with no reads the reference's range starts at 0)
        hi = max(L + 15, max(eff.keys() or [0]))
        fn = prefix + "Indel_histogram.txt"
        _write_histogram(os.path.join(out_dir, fn), ("indel_size", "fq"), [x - L for x in range(lo, hi + 1)],
                         [int(eff.get(x, 0)) for x in range(lo, hi + 1)])
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
        # alleles around each guide's cut (CRISPRessoCORE.py:5250-5273); needs the guides' sequences for the file names
        guides = refs[name].get("sgRNA_orig_sequences") or []
        if guides:
            table = res.allele_table() if hasattr(res, "allele_table") else None
            if table is None and allele_rows is None:
                allele_rows = res.alleles()
            labels = refs[name].get("sgRNA_names") or [""] * len(guides)
            t_a0 = time.perf_counter()
            for cut_point, guide, label in zip(refs[name]["sgRNA_cut_points"], guides, labels):
                fn = prefix + "Alleles_frequency_table_around_" + slugify(label if label != "" else "sgRNA_" + guide) + ".txt"
                if table is not None:
                    # windows cut and merged on the device, sums and text by native code (c2_allele_table_around_cut_write)
                    aligned_to = res.align_ref_names
                    lab = aligned_to.index(name) if name in aligned_to else 3 * len(aligned_to)      # ('Scaffold-incorporated': nothing is aligned to it)
                    table.write_around_cut(os.path.join(out_dir, fn), lab, cut_point, L, plot_window_size, res.stats["N_TOTAL"])
                else:
                    write_alleles_around_cut(alleles_around_cut(allele_rows, name, cut_point, L, plot_window_size), os.path.join(out_dir, fn))
                written.append(fn)
            t_around += time.perf_counter() - t_a0
    if getattr(res, "first_ref_view", None) and any(res.per_ref[nm]["counts_total"] > 0 for nm in ref_names):
        written += write_reads_from_all_amplicons_tables(res, refs, ref_names, out_dir)
    if timings is not None:
        timings["around_cut_tables"] = t_around
        timings["other_tables"] = time.perf_counter() - t_start - t_alleles - t_around
    return written
