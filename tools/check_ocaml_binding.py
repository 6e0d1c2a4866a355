#!/usr/bin/env python3
"""Static check of integration/ocaml/hvc.ml against include/hvc_jpeg.h (the OCaml side cannot
be compiled in this image, so at least its `foreign` declarations and `structure` layouts are compared with the C
header they bind).  Fails (exit code 1, one line per finding) when

  * a `foreign "name"` names no HVC_API function of the header,
  * its arity differs from the prototype's, or an argument / return type is not the ctypes spelling of the C type,
  * a `structure "hvc_..."` module lists other fields (name, order, type, array length) than the C struct.

Functions of the header without a binding are listed (the binding names every one of them; tests/test_ocaml_binding.py
holds it to that).

    python tools/check_ocaml_binding.py [--list-unbound]
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hvc_jpeg.h")
BINDING = os.path.join(ROOT, "integration", "ocaml", "hvc.ml")

# C type (const / spaces removed) -> the ctypes spellings that bind it
C_TO_ML = {
    "int": {"int"}, "size_t": {"size_t"}, "void": {"void"}, "uint64_t": {"uint64_t"},
    "hvc_ctx*": {"ctx"}, "hvc_ctx**": {"ptr ctx"},
    "int16_t*": {"ptr int16_t"}, "uint16_t*": {"ptr uint16_t"}, "uint64_t*": {"ptr uint64_t"}, "uint32_t*": {"ptr uint32_t"},
    "uint8_t*": {"ptr char", "ptr uint8_t", "string"},  # Base_bigstring data / OCaml string for read-only bytes
    "char*": {"string", "ptr char"},  # a read-only C string / a buffer the callee fills
    "int*": {"ptr int"}, "size_t*": {"ptr size_t"}, "float*": {"ptr float"},
    "void*": {"ptr void"}, "void**": {"ptr (ptr void)"},
    "hvc_component*": {"ptr Component.t"}, "hvc_jpeg_info*": {"ptr Jpeg_info.t"},
    "hvc_batch_stats*": {"ptr Batch_stats.t"}, "hvc_slot_stats*": {"ptr Slot_stats.t"},
    "uint8_t**": {"ptr (ptr char)", "ptr string"},  # const uint8_t *const *: an array of byte strings
}
FIELD_TO_ML = {"int": "int", "size_t": "size_t", "uint16_t": "uint16_t", "double": "double", "uint64_t": "uint64_t",
               "hvc_jpeg_component": "Jpeg_component.t", "hvc_component": "Component.t"}


def strip_comments(text, ml=False):
    return re.sub(r"\(\*.*?\*\)" if ml else r"/\*.*?\*/", " ", text, flags=re.S)


def norm_ctype(t):
    t = re.sub(r"\bconst\b", "", t)
    return re.sub(r"\s+", "", t)


def header_functions(text):
    out = {}
    for m in re.finditer(r"HVC_API\s+([\w\s\*]+?)\b(hvc_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a, flags=re.S)   # type, then the parameter name
                params.append(norm_ctype(mm.group(1)))
        out[name] = (norm_ctype(ret), params)
    return out


def header_structs(text):
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\w+\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"([\w\s]+?)\s+([\w\s,\[\]]+)$", decl)
            ctype = norm_ctype(mm.group(1))
            for name in mm.group(2).split(","):
                name = name.strip()
                dims = [int(d) for d in re.findall(r"\[(\d+)\]", name)]
                n = 1
                for d in dims:
                    n *= d
                fields.append((re.sub(r"\[.*", "", name), ctype, n if dims else 0))
        out[m.group(1)] = fields
    return out


def split_arrows(sig):
    parts, depth, cur = [], 0, ""
    i = 0
    while i < len(sig):
        if sig[i] == "(":
            depth += 1
        elif sig[i] == ")":
            depth -= 1
        if depth == 0 and sig.startswith("@->", i):
            parts.append(cur.strip())
            cur = ""
            i += 3
            continue
        cur += sig[i]
        i += 1
    parts.append(cur.strip())
    return [re.sub(r"\s+", " ", p) for p in parts]


def binding_foreigns(text):
    out = []
    for m in re.finditer(r'foreign\s+"(\w+)"\s*(?:~release_runtime_lock:true\s*)?\(', text):
        start = m.end()
        depth, i = 1, start
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        parts = split_arrows(text[start:i - 1])
        assert parts[-1].startswith("returning "), (m.group(1), parts)
        out.append((m.group(1), parts[:-1], parts[-1][len("returning "):].strip()))
    return out


def binding_structs(text):
    out = {}
    for m in re.finditer(r'structure\s+"(\w+)"(.*?)let \(\) = seal t', text, flags=re.S):
        fields = []
        for f in re.finditer(r'field t "(\w+)"\s+(\(array (\d+) ([\w.]+)\)|[\w.]+)', m.group(2)):
            fields.append((f.group(1), f.group(4) or f.group(2), int(f.group(3)) if f.group(3) else 0))
        out[m.group(1)] = fields
    return out


def check():
    hdr = strip_comments(open(HEADER).read())
    ml = strip_comments(open(BINDING).read(), ml=True)
    funcs, structs = header_functions(hdr), header_structs(hdr)
    problems, bound = [], set()
    for name, args, ret in binding_foreigns(ml):
        bound.add(name)
        if name not in funcs:
            problems.append('foreign "%s": no such HVC_API function in include/hvc_jpeg.h' % name)
            continue
        cret, cargs = funcs[name]
        if args == ["void"] and not cargs:  # `void @-> returning t`: ctypes' spelling of f(void)
            args = []
        if len(args) != len(cargs):
            problems.append('foreign "%s": %d arguments, the C prototype has %d' % (name, len(args), len(cargs)))
            continue
        for k, (a, ca) in enumerate(zip(args, cargs)):
            if a not in C_TO_ML.get(ca, ()):
                problems.append('foreign "%s": argument %d is `%s`, the C prototype has `%s`' % (name, k + 1, a, ca))
        if ret not in C_TO_ML.get(cret, ()):
            problems.append('foreign "%s": returns `%s`, the C prototype returns `%s`' % (name, ret, cret))
    for sname, mfields in binding_structs(ml).items():
        if sname not in structs:
            problems.append('structure "%s": no such struct in include/hvc_jpeg.h' % sname)
            continue
        cfields = structs[sname]
        want = [(n, FIELD_TO_ML.get(t, "?" + t), k) for n, t, k in cfields]
        if want != mfields:
            for i in range(max(len(want), len(mfields))):
                w = want[i] if i < len(want) else None
                g = mfields[i] if i < len(mfields) else None
                if w != g:
                    problems.append('structure "%s": field %d is %s, the C struct has %s' % (sname, i + 1, g, w))
    return problems, sorted(set(funcs) - bound), len(bound)


PATCH = os.path.join(ROOT, "integration", "ocaml", "hvc_backend.patch")


def binding_names(text):
    """what hvc.ml defines at its top level and inside its modules: {"check", "Component.blocks_w", ...}"""
    names, module = set(), None
    for line in text.split("\n"):
        m = re.match(r"^module (\w+) = struct", line)
        if m:
            module = m.group(1)
            names.add(module)
            continue
        if re.match(r"^end\b", line):
            module = None
            continue
        m = re.match(r"^(?:  )?(?:let|type)(?: rec)? (?:\(\) |)(\w+)", line)
        if m and m.group(1) != "_":
            names.add(("%s.%s" % (module, m.group(1))) if (module and line.startswith("  ")) else m.group(1))
    return names


def check_patch(patch=None, binding=None):
    """Every `Hvc.<path>` the patch's added lines use must be defined by hvc.ml (the patch cannot be compiled here:
    at least it must not call what does not exist -- round 2's sketch did)."""
    added = "\n".join(ln[1:] for ln in open(patch or PATCH).read().split("\n") if ln.startswith("+") and not ln.startswith("+++"))
    added = strip_comments(added, ml=True)
    defined = binding_names(strip_comments(open(binding or BINDING).read(), ml=True))
    used = sorted(set(re.findall(r"\bHvc\.((?:[A-Z]\w*\.)*\w+)", added)))
    return [u for u in used if u not in defined], used


def main():
    problems, unbound, n = check()
    missing, used = check_patch()
    for u in missing:
        problems.append("hvc_backend.patch uses Hvc.%s, which hvc.ml does not define" % u)
    print("hvc_backend.patch: %d Hvc identifiers used, %d undefined" % (len(used), len(missing)))
    for p in problems:
        print("MISMATCH", p)
    print("%d foreign declarations checked against include/hvc_jpeg.h, %d mismatches, %d functions of the header unbound"
          % (n, len(problems), len(unbound)))
    if "--list-unbound" in sys.argv:
        print("\n".join(unbound))
    sys.exit(1 if problems else 0)


if __name__ == "__main__":
    main()
