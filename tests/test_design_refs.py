"""DESIGN.md stays auditable ("every number in DESIGN.md section 5 has a profiles/<file>:<line> beside it"): every
`profiles/<file>:<line>` it cites exists, the kernel anchors point at the kernels' definitions in the current sources, the measured
tables are exactly what tools/design_tables.py makes of the session they name, and the document keeps to its length."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def design():
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        return f.read()


def between(s, name):
    a, b = "<!-- %s:BEGIN -->" % name, "<!-- %s:END -->" % name
    return s[s.index(a) + len(a):s.index(b)].strip()


def test_every_profiles_reference_exists():
    s = design()
    refs = set(re.findall(r"profiles/([A-Za-z0-9_.]+?)(?::(\d+)(?:-(\d+))?)?[`\s,;)]", s))
    assert len(refs) > 15
    for fn, a, b in refs:
        path = os.path.join(ROOT, "profiles", fn)
        assert os.path.exists(path), "DESIGN.md cites profiles/%s, which does not exist" % fn
        if a:
            n = sum(1 for _ in open(path, errors="replace"))
            assert int(b or a) <= n, "DESIGN.md cites profiles/%s:%s, the file has %d lines" % (fn, b or a, n)


def test_kernel_anchors_and_tables_are_current():
    import design_tables
    s = design()
    assert between(s, "ANCHORS") == design_tables.anchors()
    tag = re.search(r"one session, `([^`]+)`, on", s).group(1)
    k, t, p = design_tables.tables(tag)
    assert between(s, "KERNELS") == k and between(s, "SESSION") == t and between(s, "PROFILES") == p


def test_design_is_short_and_names_every_scope_row():
    s = design()
    assert s.count("\n") <= 350
    for row in ("a1", "a7", "a11", "a12", "a16", "| b |", "| c |", "| d |", "| e |", "next-1", "next-2", "next-3", "next-4"):
        assert row in s, row


def test_tool_references_in_the_documents_exist():
    """the documents cite tools by path; a tool that was purged with its closed study must not be cited as if it were
    there.  Every `tools/...` path in DESIGN.md, README.md, profiles/ANALYSIS.md and profiles/README.md exists in the tree --
    unless the sentence says it was removed (or the path is the reference's own tools/src/..., or a glob)."""
    for doc in ("DESIGN.md", "README.md", os.path.join("profiles", "ANALYSIS.md"), os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"tools/[A-Za-z0-9_./*-]+", text):
            path = m.group(0).rstrip(".")
            if path.startswith("tools/src/") or "*" in path or path.endswith(("_", "/")):
                continue
            if os.path.exists(os.path.join(ROOT, path)):
                continue
            tail = text[m.end():m.end() + 80]
            assert "removed" in tail or "deleted" in tail or "is gone" in tail, "%s cites %s, which is not in the tree" % (doc, path)
