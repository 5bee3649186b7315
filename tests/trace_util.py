"""The reference's recorded per-record intermediates (fixtures' `records`: cmp_list2 and the identify_ambigious_diffs result of every
kept record, typing_core.py:1351-1384 / typing_common.py:1663-1955; `mpileup`: get_mpileup's tables, typing_common.py:1059-1134) against
a front end's trace lines and pileup tables -- shared by the host front end's CPU tests, the CPU emulation of the kernels and the
GPU tests of the kernels themselves."""


def fmt_record(rec):
    parts = []
    for c in rec["cmp"]:
        s = "%s:%d:%d" % (c[0], c[1], c[2])
        if c[0] != "match":
            s += ":" + c[3]
        parts.append(s)
    cl, cr, la, ra = rec["iad"]
    return "%s\t%d\t%d\t%s\t%s" % (",".join(parts), cl, cr, ";".join(sorted(la)), ";".join(sorted(ra)))


def check_trace(fx, batch):
    """batch.trace_text() == the reference's records, line for line."""
    got = batch.trace_text().splitlines()
    assert len(got) == len(fx["records"]), (len(got), len(fx["records"]))
    for k, (g, rec) in enumerate(zip(got, fx["records"])):
        assert g == fmt_record(rec), (k, g, fmt_record(rec))


def check_pileup(fx, batch):
    """batch.pileup() == the reference's get_mpileup tables: six counters and the nt_set of every backbone position."""
    nt, cnt = batch.pileup(len(fx["_locus"].backbone))
    exp_sets = fx["mpileup"]["nt_set"]
    assert len(nt) == len(exp_sets)
    for i in range(len(nt)):
        s = "".join(b for k, b in enumerate("ACGT") if nt[i] & (1 << k))
        assert s == exp_sets[i], i
        exp = fx["mpileup"]["counts"][i]
        for k, b in enumerate("ACGT"):
            assert cnt[i, k] == exp.get(b, 0), (i, b)
        assert cnt[i, 5] == exp.get("D", 0), i
        assert cnt[i, 4] == sum(v for b, v in exp.items() if b not in "ACGTD"), i
