"""next-4 hygiene: integration/ocaml/hvc.ml cannot be compiled here (no OCaml toolchain), so its `foreign`
declarations and `structure` layouts are at least held against include/hvc_jpeg.h by tools/check_ocaml_binding.py --
and the checker is shown to catch what it is there to catch."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_ocaml_binding", os.path.join(ROOT, "tools", "check_ocaml_binding.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)


def test_binding_matches_the_header():
    problems, unbound, n = chk.check()
    assert problems == []
    assert n >= 50
    assert unbound == []   # one `foreign` per function of the header: nothing of the ABI is out of the OCaml side's reach


def test_checker_catches_drift(tmp_path, monkeypatch):
    good = open(chk.BINDING).read()
    mutations = {
        "arity": good.replace("(ctx @-> ptr int16_t @-> size_t @-> ptr uint16_t @-> int @-> int @-> int @-> ptr char @-> size_t\n    @-> size_t @-> int @-> returning int)",
                              "(ctx @-> ptr int16_t @-> size_t @-> ptr uint16_t @-> int @-> int @-> ptr char @-> size_t\n    @-> size_t @-> int @-> returning int)"),
        "type": good.replace('foreign "hvc_quant_table" (int @-> int @-> ptr uint16_t @-> returning int)',
                             'foreign "hvc_quant_table" (int @-> int @-> ptr int16_t @-> returning int)'),
        "name": good.replace('foreign "hvc_destroy"', 'foreign "hvc_destory"'),
        "return": good.replace('foreign "hvc_strerror" (int @-> returning string)', 'foreign "hvc_strerror" (int @-> returning int)'),
        "field": good.replace('let qtab = field t "qtab" int\n  let reserved = field t "reserved" int',
                              'let reserved = field t "reserved" int\n  let qtab = field t "qtab" int'),
        "array": good.replace('field t "qtabs" (array 256 uint16_t)', 'field t "qtabs" (array 128 uint16_t)'),
    }
    for what, text in mutations.items():
        assert text != good, what
        p = tmp_path / ("hvc_%s.ml" % what)
        p.write_text(text)
        monkeypatch.setattr(chk, "BINDING", str(p))
        problems, _, _ = chk.check()
        assert problems, what


def test_patch_uses_only_what_the_binding_defines(tmp_path):
    missing, used = chk.check_patch()
    assert missing == []
    for name in ("ctx", "coefs", "coefs_ptr", "check", "decode_frames", "mem_host", "Component.t", "Component.stride",
                 "decode_frames_submit", "encode_frames_submit", "wait", "pinned_coefs", "pinned_bytes", "free_pinned_coefs",
                 "free_pinned_bytes"):
        assert name in used, name
    # ... and the check sees a name that is not there (round 2's sketch called helpers that existed nowhere)
    text = open(chk.PATCH).read().replace("Hvc.coefs_ptr record", "Hvc.coefs_pointer record")
    p = tmp_path / "bad.patch"
    p.write_text(text)
    assert chk.check_patch(str(p))[0] == ["coefs_pointer"]


REF = "/root/reference"


def test_patch_applies_to_the_reference(tmp_path):
    """integration/ocaml/hvc_backend.patch is a real unified diff against the reference's decoder.ml / decoder.mli /
    dune / plane.mli: `patch --dry-run` accepts it, and applied it leaves the functions INTEGRATION.md names.  (The
    reference exists in the build container only; its files are copied to a temporary directory and never travel.)"""
    import shutil
    import subprocess
    import pytest
    files = ["jpeg/model/src/decoder.ml", "jpeg/model/src/decoder.mli", "jpeg/model/src/encoder.ml", "jpeg/model/src/encoder.mli",
             "jpeg/model/src/dune", "common/src/plane.mli"]
    if not all(os.path.exists(os.path.join(REF, f)) for f in files):
        pytest.skip("the reference tree is not on this machine")
    for f in files:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), tmp_path / f)
    for flag in (["--dry-run"], []):
        r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch"] + flag, stdin=open(chk.PATCH), cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    ml = (tmp_path / "jpeg/model/src/decoder.ml").read_text()
    mli = (tmp_path / "jpeg/model/src/decoder.mli").read_text()
    for name in ("let decode_gpu ", "let decode_a_frame_gpu ", "module Gpu = struct", "?(decode_block = decode_block)",
                 "module Gpu_slot = struct", "let decode_frames_gpu (hvc : Hvc.ctx) (files : Bits.t list) : Frame.t list ="):
        assert name in ml, name
    # the double-buffered form is defined after what it calls (OCaml reads top to bottom): get_yuv_frame, decode_seq, Gpu
    assert ml.index("let get_yuv_frame decoder") < ml.index("let decode_frames_gpu") and ml.index("module Gpu = struct") < ml.index("module Gpu_slot")
    assert "val decode_gpu : Hvc.ctx -> t -> unit" in mli and "val decode_a_frame_gpu" in mli
    assert "val decode_frames_gpu : Hvc.ctx -> Bits.t list -> Frame.t list" in mli
    enc, enci = (tmp_path / "jpeg/model/src/encoder.ml").read_text(), (tmp_path / "jpeg/model/src/encoder.mli").read_text()
    for name in ("let encode_seq_with ~encode_block (t : t) =", "let encode_seq (t : t) = encode_seq_with ~encode_block t",
                 "let encode_seq_gpu (hvc : Hvc.ctx) (t : t) =", "let encode_420_gpu hvc ~frame ~quality ~writer =",
                 "let encode_420_frames_gpu (hvc : Hvc.ctx) ~quality jobs =", "module Gpu_slot = struct"):
        assert name in enc, name
    assert enc.index("let complete_and_write_eoi") < enc.index("let encode_420_frames_gpu") and enc.index("module Gpu = struct") < enc.index("module Gpu_slot")
    assert "val encode_420_frames_gpu : Hvc.ctx -> quality:int -> (Frame.t * Bitstream_writer.t) list -> unit" in enci
    # the exported signature of encode_seq is the reference's own (an optional argument would not match it)
    assert "val encode_seq : t -> Block.t Sequence.t" in enci and "val encode_seq_gpu : Hvc.ctx -> t -> Block.t Sequence.t" in enci
    assert "val plane : t -> Base_bigstring.t" in (tmp_path / "common/src/plane.mli").read_text()
    assert "ctypes.foreign" in (tmp_path / "jpeg/model/src/dune").read_text()
    # balanced: what the patch ADDS opens and closes its own brackets, comments and modules (a cheap stand-in for the
    # parser this image does not have), hunk by hunk
    added = "\n".join(ln[1:] for ln in open(chk.PATCH).read().split("\n") if ln.startswith("+") and not ln.startswith("+++"))
    assert added.count("(*") == added.count("*)")
    code = chk.strip_comments(added, ml=True)
    depth = 0
    for ch in code:
        depth += {"(": 1, ")": -1, "[": 1, "]": -1, "{": 1, "}": -1}.get(ch, 0)
        assert depth >= 0
    assert depth == 0
    import re
    assert len(re.findall(r"\bstruct\b", code)) == len(re.findall(r"^end\b", code, flags=re.M))
