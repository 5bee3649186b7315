"""Report summariser: the reference's `hisatgenotype_parse_results` tool over the `.report` files typing()
writes -- same entry points, same printed text, same CSV.

  build_tree / call_nuance_results   typing_common.py:1965-2030  (one report -> {'EM', 'Allele splitting', 'Assembly'})
  flatten / result_process           hisatgenotype_tools/hisatgenotype_parse_results.py:33-137

An allele name `GENE*f1:f2:f3:f4` is a path in a field tree; every node carries the summed abundance of the report lines
below it, so resolution that the EM split over sibling alleles (A*01:01:01:01 25 % + A*01:01:01:02 25 %) is reported once at
the deepest field the data supports (A*01:01:01 - Trimmed 50 %).

Host-side text processing only; nothing here touches the GPU.  Behaviour follows the reference including its edge cases
(documented where they matter): the gene is located with `str.find`, abundance lines are recognised by the substring
'abundance', everything after a line starting with 'Assembly' is taken as `name: value` assembly output.
"""
import glob
import sys


def build_tree(vlist, tree, leaf):
    """Insert the field path `vlist` with abundance `leaf` below `tree` ({'score', 'children'}); returns the updated node.
    A leaf is {'score': leaf, 'children': None}; re-inserting a path REPLACES the leaf but ADDS to every ancestor
    (typing_common.py:1965-1983)."""
    if not vlist:
        return {"score": leaf, "children": None}
    head, rest = vlist[0], vlist[1:]
    children = tree["children"]       # None below a leaf: an allele that extends a reported shorter name raises TypeError, as in the reference
    below = children[head] if head in children else {"score": 0, "children": {}}
    children[head] = build_tree(rest, below, leaf)
    tree["score"] += leaf
    return tree


def call_nuance_results(nfile):
    """Parse one report file (typing_common.py:1985-2030)."""
    datatree = {"EM": {}, "Allele splitting": {}, "Assembly": {}}
    in_assembly = False
    with open(nfile, "r") as fh:
        for raw in fh:
            line = raw.strip()
            if line.startswith("Assembly"):
                in_assembly = True
                continue
            if in_assembly:
                cut = line.find(":")
                datatree["Assembly"][line[:cut]] = line[cut + 2:]
                continue
            if "abundance" not in line:
                continue
            # "<rank> ranked <allele> (abundance: x%)" or "*** <rank> ranked <allele> (abundance: x%)"
            word = 3 if "***" in line else 2
            gene = line.split()[word].split("*")[0]
            line = line[line.find(gene):]
            if gene not in datatree["EM"]:
                datatree["EM"][gene] = []
                datatree["Allele splitting"][gene] = {"score": 0, "children": {}}
            datatree["EM"][gene].append(line)
            allele, _, percent = line.replace("(", "").replace(")", "").split()
            fields = allele.split("*")[-1].split(":")
            weight = round(float(percent[:-1]) / 100, 4)
            datatree["Allele splitting"][gene] = build_tree(fields, datatree["Allele splitting"][gene], weight)
    return datatree


def flatten(tree, prev_key="", sep="*", trim=4, cur_lvl=1):
    """Field tree -> [(name, score)] (hisatgenotype_parse_results.py:33-60).  Leaves keep their full name; an inner node
    contributes '<name> - Trimmed' when trim == 4 (every level) or at exactly the trim level; levels below `trim` are not
    descended into.  The top-level call (sep '*') returns the list sorted by (score, name length) descending; nested calls
    return dicts, so a name reached twice keeps its last score."""
    items = []
    for key, node in tree.items():
        name = prev_key + sep + key if prev_key else key
        if node["children"] is None:
            items.append((name, node["score"]))
            continue
        if trim > cur_lvl:
            items.extend(flatten(node["children"], name, ":", trim=trim, cur_lvl=cur_lvl + 1).items())
        if trim == 4 or trim == cur_lvl:
            items.append((name + " - Trimmed", node["score"]))
    if sep == ":":
        return dict(items)
    return sorted(items, key=lambda it: (it[1], len(it[0].split()[0])), reverse=True)


def result_process(args, out=None):
    """The tool's main routine (hisatgenotype_parse_results.py:62-137): every `*.report` of `args.read_dir` (default '.'),
    printed per file / analysis / gene; with `args.csv` a tab-separated table goes to `args.ofile`.  `args` needs the
    attributes read_dir, trim_level, csv, ofile (an argparse namespace like the tool's).  Returns the table rows."""
    out = out or sys.stdout
    indir = args.read_dir if args.read_dir else "."
    parsed = {}
    for path in glob.glob("%s/*.report" % indir):
        parsed[path] = call_nuance_results(path)
    rows, header = [], ["File"]
    for path, analyses in parsed.items():
        print("File: %s" % path, file=out)
        rows.append([path])
        for kind, per_gene in analyses.items():
            print("\tAnalysis - %s" % kind, file=out)
            for gene, data in per_gene.items():
                col = "%s: %s" % (kind, gene)
                if col not in header:
                    header.append(col)
                if kind == "Allele splitting":
                    print("\t\tGene: %s (score: %.2f)" % (gene, data["score"]), file=out)
                    cell, last_score = "", 0
                    for name, score in flatten(data["children"], gene, trim=args.trim_level):
                        # below 20 % is noise; a trimmed parent that only repeats the score just printed adds nothing
                        if score < 0.2 or (last_score == score and "Trimmed" in name):
                            continue
                        text = "%s (score: %.4f)" % (name, score)
                        cell += text + ","
                        print("\t\t\t" + text, file=out)
                        last_score = score
                    rows[-1].append(cell[:-1])
                    continue
                print("\t\tGene: %s" % gene, file=out)
                if isinstance(data, list):
                    rows[-1].append(",".join(data))
                    for line in data:
                        print("\t\t\t%s" % line, file=out)
                else:
                    rows[-1].append(data)
                    print("\t\t\t%s" % data, file=out)
    if args.csv:
        rows.insert(0, header)
        with open(args.ofile, "w") as fh:
            for row in rows:
                fh.write("\t".join(row) + "\n")
    return rows


def main(argv=None):
    """`python -m hisatgenotype_amd.results`: the tool's command line (--in-dir, -t/--trim, --csv, --output-file)."""
    from argparse import ArgumentParser
    ap = ArgumentParser(description="Script for simplifying HISAT-genotype results")
    ap.add_argument("--in-dir", dest="read_dir", type=str, default=".", help="Input directory (e.g. read_input)")
    ap.add_argument("-t", "--trim", dest="trim_level", type=int, default=4,
                    help="Trim allele to specific field level (example : A*01:01:01:01 trim 2 A*01:01)")
    ap.add_argument("--csv", dest="csv", action="store_true", help="Save Results as CSV dataframe")
    ap.add_argument("--output-file", dest="ofile", default="HG_report_results.csv", help="Path to the output CSV file")
    result_process(ap.parse_args(argv))


if __name__ == "__main__":
    main()
