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
    assert n >= 20
    # what the patched Decoder / Encoder of INTEGRATION.md call is bound
    for f in ("hvc_decode_frames", "hvc_jpeg_decode", "hvc_encode_frames", "hvc_jpeg_encode", "hvc_create", "hvc_destroy"):
        assert f not in unbound


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
