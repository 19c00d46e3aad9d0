"""DEV CONTAINER ONLY (reads /root/reference): the shape of the reference's own test reads, tests/FANC.Cas9.fastq (250 reads of a Cas9-edited
FANC amplicon), as a small table that crispresso2_amd/synth.py resamples to any number of reads (bench.py's `robustness.fanc_shaped` leg:
VERDICT r04 item 4 -- real amplicon reads carry a 4-base leading and a 23-base trailing overhang, vary in length, and 7 % of them are junk).

Every distinct read is aligned to the 223-bp amplicon by the oracle (the C restatement of the reference's aligner; any global aligner would do:
this is a description of the DATA) and reduced to a signature: leading overhang, read length, deletions / insertions as (reference position,
length), number of substitutions, or "junk" when more than 15 % of its aligned bases differ.  The flanks (the genomic sequence the reads run
into on either side) are the consensus of the overhangs.  -> crispresso2_amd/fanc_profile.json"""
import collections
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from crispresso2_amd import CRISPResso2Align as A  # noqa: E402

REF = os.environ.get("C2_REFERENCE_DIR", "/root/reference")
AMP = ("CGGATGTTCCAATCAGTACGCAGAGAGTCGCCGTCTCCAAGGTGAAAGCGGAAGTAGGGCCTTCGCGCACCTCATGGAATCCCTTCTGCAGCACCTGGATCGCTTTTCCGAGCTTCTGGCGGTCTCAAG"
       "CACTACCTACGTCAGCACCTGGGACCCCGCCACCGTGCGCCGGGCCTTGCAGTGGGCGCGCTACCTGCGCCACATCCATCGGCGCTTTGGTCGG")
GUIDE = "GGAATCCCTTCTGCAGCACC"


def main():
    m = A.read_matrix(os.path.join(ROOT, "crispresso2_amd", "EDNAFULL"))
    cut = AMP.index(GUIDE) + len(GUIDE) - 3 - 1                       # the reference's cut point index for a Cas9 guide (-3 from its 3' end)
    g = np.zeros(len(AMP) + 1, dtype=np.int64)
    g[cut + 1] = 1
    lines = open(os.path.join(REF, "tests", "FANC.Cas9.fastq")).read().split("\n")
    reads = [lines[i] for i in range(1, len(lines), 4) if lines[i]]
    mult = collections.Counter(reads)
    leads, trails, templates = collections.Counter(), [], []
    for rd, w in mult.items():
        s1, s2, _ = oracle.global_align(rd, AMP, m, g, -20, -2)
        lead = len(s2) - len(s2.lstrip("-"))
        trail = len(s2) - len(s2.rstrip("-"))
        c1, c2 = s1[lead:len(s1) - trail], s2[lead:len(s2) - trail]
        both = [(a, b) for a, b in zip(c1, c2) if a != "-" and b != "-"]
        subs = sum(1 for a, b in both if a != b)
        if not both or subs > 0.15 * len(both):
            templates.append({"w": w, "junk": True, "len": len(rd)})
            continue
        if lead:
            leads[s1[:lead]] += w
        if trail:
            trails.append((s1[len(s1) - trail:], w))
        dels, ins, pos = [], [], 0                                     # pos: reference position of the column
        for mt in re.finditer(r"-+|[^-]+", c1):
            pass
        i = 0
        while i < len(c1):
            if c1[i] == "-":
                j = i
                while j < len(c1) and c1[j] == "-":
                    j += 1
                dels.append([pos, j - i])
                pos += j - i
                i = j
            elif c2[i] == "-":
                j = i
                while j < len(c1) and c2[j] == "-":
                    j += 1
                ins.append([pos, j - i])
                i = j
            else:
                pos += 1
                i += 1
        ref_start = 0
        # a read that starts inside the amplicon (no leading overhang, reference bases unaligned at its start) shows as a deletion at 0
        templates.append({"w": w, "junk": False, "len": len(rd), "lead": lead, "dels": dels, "ins": ins, "subs": subs})
    lead_flank = max(leads.items(), key=lambda kv: (len(kv[0]) >= 4, kv[1]))[0]
    # trailing flank: per position the most common base over the reads that reach it
    tl = max(len(t) for t, _ in trails)
    flank = []
    for p in range(tl):
        c = collections.Counter()
        for t, w in trails:
            if len(t) > p:
                c[t[p]] += w
        flank.append(c.most_common(1)[0][0])
    out = {"source": "pinellolab/CRISPResso2 tests/FANC.Cas9.fastq (250 reads, %d distinct) against the FANC amplicon of tests/Cas9.amplicons.txt; "
                     "made by tools/fanc_profile.py" % len(mult),
           "amplicon": AMP, "guide": GUIDE, "cut_point": cut, "lead_flank": lead_flank, "trail_flank": "".join(flank), "templates": templates}
    path = os.path.join(ROOT, "crispresso2_amd", "fanc_profile.json")
    with open(path, "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    n = sum(t["w"] for t in templates)
    print("written", path, os.path.getsize(path), "bytes;", len(templates), "templates,", sum(t["w"] for t in templates if t["junk"]), "of", n, "reads junk;",
          "lead flank", lead_flank, "trail flank", len(flank), "bases")
    print("lengths", sorted(collections.Counter(t["len"] for t in templates for _ in range(t["w"])).items()))


if __name__ == "__main__":
    main()
