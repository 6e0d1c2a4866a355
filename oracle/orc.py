"""ctypes loader for the CPU oracle (oracle/hvc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)


def build(force=False):
    so = os.path.join(_DIR, "liborc.so")
    src = os.path.join(_DIR, "hvc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _DIR, "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_zigzag_inverse.restype = C.POINTER(C.c_int)
        L.orc_zigzag_forward.restype = C.POINTER(C.c_int)
        L.orc_quant_luma.restype = C.POINTER(C.c_int)
        L.orc_quant_chroma.restype = C.POINTER(C.c_int)
        L.orc_mag.restype = C.c_int64
        L.orc_mag.argtypes = [C.c_int, C.c_int64]
        L.orc_enc_size.argtypes = [C.c_int64]
        L.orc_enc_magnitude.restype = C.c_int64
        L.orc_enc_magnitude.argtypes = [C.c_int, C.c_int64]
        L.orc_decoder_create.restype = C.c_void_p
        L.orc_decoder_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        for f in ("orc_decoder_destroy", "orc_decoder_next_block", "orc_decoder_decode", "orc_decoder_ncomp",
                  "orc_decoder_width", "orc_decoder_height", "orc_decoder_frame_of_planes"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_decoder_destroy.restype = None
        L.orc_decoder_coef_record.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_decoder_coef_count.restype = C.c_int64
        L.orc_decoder_coef_count.argtypes = [C.c_void_p]
        L.orc_decoder_component_info.argtypes = [C.c_void_p, C.c_int, i64p]
        L.orc_decoder_component_array.restype = i64p
        L.orc_decoder_component_array.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_decoder_plane.restype = u8p
        L.orc_decoder_plane.argtypes = [C.c_void_p, C.c_int]
        L.orc_decoder_cropped_plane.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_encode_yuv.restype = C.c_int64
        L.orc_encode_yuv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_write_headers.restype = C.c_int64
        L.orc_write_headers.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_dequant_idct_recon.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_size_t, C.c_size_t]
        L.orc_fdct_quant.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]
        L.orc_encode_recon.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_max_difference.restype = C.c_int64
        L.orc_square_error.restype = C.c_int64
        L.orc_max_difference.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_square_error.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(i64p)


def create_code_table(lengths16, values):
    """Specification.create_code_table (tables.ml:27-45): [(length, bits, data)] in the model's list order"""
    le = np.ascontiguousarray(lengths16, dtype=np.int32)
    va = np.ascontiguousarray(values, dtype=np.int32)
    out = np.zeros(3 * 256, dtype=np.int32)
    n = lib().orc_create_code_table(_ptr(le), _ptr(va), int(va.size), _ptr(out))
    assert n >= 0
    return [tuple(int(x) for x in out[3 * i:3 * i + 3]) for i in range(n)]


def enc_table(name):
    """Tables.Encoder.dc_table / ac_table of a default specification ("dc_luma", "dc_chroma", "ac_luma", "ac_chroma") in the
    shape tests/golden/g8_code_tables.json holds them: dc [category] -> [length, bits, data]; ac [run][size] -> [length, bits,
    run, size]"""
    which = ["dc_luma", "dc_chroma", "ac_luma", "ac_chroma"].index(name)
    out = np.zeros(16 * 16 * 4, dtype=np.int32)
    rows = np.zeros(16, dtype=np.int32)
    n = lib().orc_enc_table(which, _ptr(out), _ptr(rows))
    if which < 2:
        return [[int(x) for x in out[3 * i:3 * i + 3]] for i in range(n)]
    return [[[int(x) for x in out[(16 * r + k) * 4:(16 * r + k) * 4 + 4]] for k in range(int(rows[r]))] for r in range(16)]


def zigzag_inverse():
    return np.array(lib().orc_zigzag_inverse()[:64])


def zigzag_forward():
    return np.array(lib().orc_zigzag_forward()[:64])


def quant_luma():
    return np.array(lib().orc_quant_luma()[:64])


def quant_chroma():
    return np.array(lib().orc_quant_chroma()[:64])


def quant_scale(table, q):
    t = np.ascontiguousarray(table, dtype=np.int32)
    out = np.zeros(64, dtype=np.int32)
    lib().orc_quant_scale(_ptr(t), C.c_int(q), _ptr(out))
    return out


def idct_8x8(block):
    b, p = _i64(np.array(block).reshape(64).copy())
    lib().orc_idct_8x8(p)
    return b


def fdct_8x8(block):
    b, p = _i64(np.array(block).reshape(64).copy())
    lib().orc_fdct_8x8(p)
    return b


def decode_block_summary(coefs_zz, qtab, dc_pred=0):
    """Model Component.Summary of one block: returns (dc, dequant, idct, recon)."""
    c, cp = _i64(coefs_zz)
    q, qp = _i64(qtab)
    dq, dqp = _i64(np.zeros(64))
    idc, ip = _i64(np.zeros(64))
    rc, rp = _i64(np.zeros(64))
    dc = C.c_int64(0)
    lib().orc_decode_block_summary(cp, qp, C.c_int64(dc_pred), dqp, ip, rp, C.byref(dc))
    return dc.value, dq, idc, rc


def dequant_idct_recon(coefs, qtab, bw, bh, n_planes=1, stride=None, plane_stride=None):
    """Batch block stage on the C-ABI layout.  coefs int16 [n_planes][bh][bw][64]."""
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    assert coefs.size == n_planes * bh * bw * 64
    qtab = np.ascontiguousarray(qtab, dtype=np.uint16)
    stride = stride or bw * 8
    plane_stride = plane_stride or stride * bh * 8
    out = np.zeros(n_planes * plane_stride, dtype=np.uint8)
    r = lib().orc_dequant_idct_recon(_ptr(coefs), _ptr(qtab), bw, bh, n_planes, _ptr(out), stride, plane_stride)
    if r:
        raise RuntimeError("orc_dequant_idct_recon: %d" % r)
    return out


def fdct_quant(planes, qtab, bw, bh, n_planes=1, stride=None, plane_stride=None):
    planes = np.ascontiguousarray(planes, dtype=np.uint8)
    qtab = np.ascontiguousarray(qtab, dtype=np.uint16)
    stride = stride or bw * 8
    plane_stride = plane_stride or stride * bh * 8
    out = np.zeros(n_planes * bh * bw * 64, dtype=np.int16)
    r = lib().orc_fdct_quant(_ptr(planes), stride, plane_stride, _ptr(qtab), bw, bh, n_planes, _ptr(out))
    if r:
        raise RuntimeError("orc_fdct_quant: %d" % r)
    return out


def encode_recon(planes, qtab, bw, bh, n_planes=1):
    """Encoder.encode_block with ~compute_reconstruction_error:true on tight planes uint8 [n_planes][bh*8][bw*8]:
    (quantised coefficients int16 [n][bh][bw][64], recon uint8, error uint8), encoder.ml:81-125."""
    planes = np.ascontiguousarray(planes, dtype=np.uint8)
    assert planes.size == n_planes * bh * bw * 64
    qtab = np.ascontiguousarray(qtab, dtype=np.uint16)
    coefs = np.zeros(n_planes * bh * bw * 64, dtype=np.int16)
    recon = np.zeros(planes.size, dtype=np.uint8)
    error = np.zeros(planes.size, dtype=np.uint8)
    r = lib().orc_encode_recon(_ptr(planes), bw * 8, bw * 8 * bh * 8, _ptr(qtab), bw, bh, n_planes, _ptr(coefs), _ptr(recon),
                               _ptr(error))
    if r:
        raise RuntimeError("orc_encode_recon: %d" % r)
    return coefs, recon, error


class Decoder:
    """The model's Decoder (jpeg/model/src/decoder.mli), restated."""

    def __init__(self, data: bytes):
        err = C.c_int(0)
        self._buf = bytes(data)
        self._d = lib().orc_decoder_create(self._buf, len(self._buf), C.byref(err))
        if not self._d:
            raise ValueError("orc_decoder_create failed: %d" % err.value)
        self.ncomp = lib().orc_decoder_ncomp(self._d)
        self.width = lib().orc_decoder_width(self._d)
        self.height = lib().orc_decoder_height(self._d)

    def __del__(self):
        if getattr(self, "_d", None):
            lib().orc_decoder_destroy(self._d)
            self._d = None

    def info(self, i):
        a = (C.c_int64 * 12)()
        lib().orc_decoder_component_info(self._d, i, a)
        k = ["decoded_width", "decoded_height", "actual_width", "actual_height", "x", "y", "dc_pred",
             "identifier", "hscale", "vscale", "tq"]
        return dict(zip(k, list(a)))

    def array(self, i, which):
        idx = {"coefs": 0, "dequant": 1, "idct": 2, "recon": 3, "quant_table": 4}[which]
        return np.array(lib().orc_decoder_component_array(self._d, i, idx)[:64])

    def next_block(self):
        """For_testing.Sequenced.decode: one block; returns component index or None."""
        r = lib().orc_decoder_next_block(self._d)
        if r == -1:
            return None
        if r < -1:
            raise ValueError("decode error %d" % r)
        return r

    def coef_record(self):
        """Sequenced.decode to the end: the frame's coefficient record in the C ABI's layout (component
        planes back to back, [bh][bw][64] zig-zag, DC absolute = dc_pred after the block) as int64 -- the
        model's ints, so a DC outside int16 shows as such.  Decodes the planes as a side effect."""
        out = np.zeros(lib().orc_decoder_coef_count(self._d), dtype=np.int64)
        r = lib().orc_decoder_coef_record(self._d, _ptr(out))
        if r:
            raise ValueError("decode error %d" % r)
        return out

    def decode(self):
        r = lib().orc_decoder_decode(self._d)
        if r:
            raise ValueError("decode error %d" % r)

    def plane(self, i):
        inf = self.info(i)
        n = inf["decoded_width"] * inf["decoded_height"]
        return np.ctypeslib.as_array(lib().orc_decoder_plane(self._d, i), shape=(n,)).reshape(
            inf["decoded_height"], inf["decoded_width"]).copy()

    def cropped_plane(self, i):
        inf = self.info(i)
        out = np.zeros((inf["actual_height"], inf["actual_width"]), dtype=np.uint8)
        lib().orc_decoder_cropped_plane(self._d, i, _ptr(out))
        return out

    def cropped_planes(self):
        """Array.map crop (get_decoded_planes t) (decoder.ml:399-413): every component's crop, whatever the sampling."""
        return [self.cropped_plane(i) for i in range(self.ncomp)]

    def chroma_subsampling(self):
        """Frame.infer_chroma_subsampling of the first three crops (frame.ml:42-61): 420 / 422 / 444; ValueError where
        the model raises (fewer than three components, chroma planes of two sizes, sizes it has no name for)."""
        r = lib().orc_decoder_frame_of_planes(self._d)
        if r < 0:
            raise ValueError("Frame.of_planes raises: %d" % r)
        return r

    def get_yuv_frame(self):
        """Decoder.get_yuv_frame (decoder.ml:415-420): the crops of components 0, 1, 2 -- where Frame.of_planes takes them."""
        self.chroma_subsampling()
        return [self.cropped_plane(i) for i in range(3)]


def decode_a_frame(data: bytes):
    d = Decoder(data)
    d.decode()
    return d.get_yuv_frame()


def chroma_dims(chroma, w, h):
    """common/src/frame.ml:10-24"""
    if chroma == 420:
        return w // 2, h // 2
    if chroma == 422:
        return w // 2, h
    return w, h


def encoder_plane_dims(chroma, w, h):
    """encoder.ml:451-458 padded plane size per scan component."""
    scales = {420: (2, 2, 1, 1, 1, 1), 422: (2, 2, 1, 2, 1, 2), 444: (1, 1, 1, 1, 1, 1)}[chroma]
    mh, mv = max(scales[0::2]), max(scales[1::2])
    dims = []
    for i in range(3):
        hs, vs = scales[2 * i], scales[2 * i + 1]
        ww, hh = w * hs // mh, h * vs // mv
        r = lambda v, m: (v + m - 1) // m * m
        dims.append((r(ww, 8 * hs), r(hh, 8 * vs)))
    return dims


def split_yuv(raw: bytes, w, h, chroma):
    cw, ch = chroma_dims(chroma, w, h)
    a = np.frombuffer(raw, dtype=np.uint8)
    assert a.size == w * h + 2 * cw * ch, (a.size, w, h, chroma)
    y = a[:w * h].reshape(h, w)
    u = a[w * h:w * h + cw * ch].reshape(ch, cw)
    v = a[w * h + cw * ch:].reshape(ch, cw)
    return y, u, v


def encode_yuv(y, u, v, w, h, chroma=420, quality=75, want_coefs=False):
    """Encoder.encode_420/422/444 ~frame ~quality.  Returns jpeg bytes
    (and the per-component quantised coefficient planes in C-ABI layout)."""
    y = np.ascontiguousarray(y, dtype=np.uint8)
    u = np.ascontiguousarray(u, dtype=np.uint8)
    v = np.ascontiguousarray(v, dtype=np.uint8)
    cap = 4 * w * h + 65536
    out = np.zeros(cap, dtype=np.uint8)
    coefs = None
    cp = None
    if want_coefs:
        dims = encoder_plane_dims(chroma, w, h)
        n = sum((pw // 8) * (ph // 8) * 64 for pw, ph in dims)
        coefs = np.zeros(n, dtype=np.int16)
        cp = _ptr(coefs)
    n = lib().orc_encode_yuv(_ptr(y), _ptr(u), _ptr(v), w, h, chroma, quality, _ptr(out), cap, cp)
    if n < 0:
        raise RuntimeError("orc_encode_yuv failed")
    data = out[:n].tobytes()
    if want_coefs:
        res, off = [], 0
        for pw, ph in encoder_plane_dims(chroma, w, h):
            k = (pw // 8) * (ph // 8) * 64
            res.append(coefs[off:off + k].reshape(ph // 8, pw // 8, 64))
            off += k
        return data, res
    return data


def write_headers(w, h, chroma, quality):
    out = np.zeros(4096, dtype=np.uint8)
    n = lib().orc_write_headers(w, h, chroma, quality, _ptr(out), 4096)
    return out[:n].tobytes()


def _plane_op(name, src, *args):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    return src, getattr(lib(), name)


def supersample_hv2(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros((2 * h, 2 * w), dtype=np.uint8)
    lib().orc_supersample_hv2(_ptr(src), w, h, _ptr(dst))
    return dst


def subsample_hv2(src, w, h):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((h, w), dtype=np.uint8)
    lib().orc_subsample_hv2(_ptr(src), src.shape[1], _ptr(dst), w, h)
    return dst


def supersample_h2(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros((h, 2 * w), dtype=np.uint8)
    lib().orc_supersample_h2(_ptr(src), w, h, _ptr(dst))
    return dst


def subsample_h2(src, w, h):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((h, w), dtype=np.uint8)
    lib().orc_subsample_h2(_ptr(src), src.shape[1], _ptr(dst), w, h)
    return dst


def crop_plane(src, dw, dh, x_pos=0, y_pos=0):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dh, dw), dtype=np.uint8)
    lib().orc_crop_plane(_ptr(src), src.shape[1], src.shape[0], x_pos, y_pos, _ptr(dst), dw, dh)
    return dst


PACKED = {"YUY2": 1, "UYVY": 2, "YVYU": 3}


def packed422_to_planar(which, src, w, h):
    """Packed_422.convert_to_planar (tools/src/packed_422.ml:10-23): a (2 w) x h packed plane -> (y, u, v) of a 4:2:2 frame"""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    y, u, v = np.zeros((h, w), np.uint8), np.zeros((h, w // 2), np.uint8), np.zeros((h, w // 2), np.uint8)
    lib().orc_packed422_to_planar(C.c_int(which), _ptr(src), C.c_int(w), C.c_int(h), _ptr(y), _ptr(u), _ptr(v))
    return y, u, v


def packed422_from_planar(which, y, u, v):
    """Packed_422.convert_from_planar (packed_422.ml:33-46)"""
    y, u, v = (np.ascontiguousarray(p, dtype=np.uint8) for p in (y, u, v))
    h, w = y.shape
    dst = np.zeros((h, 2 * w), np.uint8)
    lib().orc_packed422_from_planar(C.c_int(which), _ptr(y), _ptr(u), _ptr(v), C.c_int(w), C.c_int(h), _ptr(dst))
    return dst


def oconv_frame(raw, fmt_in, size_in, fmt_out, size_out, offset=(0, 0)):
    """One pass of Oconv.main's loop (tools/src/oconv.ml:111-133) over one raw frame: Oconv.input (:12-28) into a 4:4:4
    frame, Yuv.crop (tools/src/yuv.ml:42-62), Oconv.output (:38-51).  fmt: 420 / 422 / 444 or "YUY2" / "UYVY" / "YVYU".
    ValueError where Yuv.assert_is_420 / _422 raise (yuv.ml:90-116)."""
    def fits(fmt, w, h):
        if fmt == 444:
            return True
        return w % 2 == 0 and (fmt != 420 or h % 2 == 0)
    (w, h), (w2, h2) = size_in, size_out
    if not fits(fmt_in, w, h) or not fits(fmt_out, w2, h2):
        raise ValueError("Expecting a 4:2:x frame")
    raw = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else raw
    if fmt_in in PACKED:
        y, u, v = packed422_to_planar(PACKED[fmt_in], raw[:2 * w * h].reshape(h, 2 * w), w, h)
        u, v = supersample_h2(u), supersample_h2(v)
    else:
        y, u, v = split_yuv(raw, w, h, fmt_in)
        if fmt_in == 420:
            u, v = supersample_hv2(u), supersample_hv2(v)
        elif fmt_in == 422:
            u, v = supersample_h2(u), supersample_h2(v)
    y, u, v = (crop_plane(p, w2, h2, offset[0], offset[1]) for p in (y, u, v))
    if fmt_out in PACKED:
        return packed422_from_planar(PACKED[fmt_out], y, subsample_h2(u, w2 // 2, h2), subsample_h2(v, w2 // 2, h2)).tobytes()
    if fmt_out == 420:
        u, v = subsample_hv2(u, w2 // 2, h2 // 2), subsample_hv2(v, w2 // 2, h2 // 2)
    elif fmt_out == 422:
        u, v = subsample_h2(u, w2 // 2, h2), subsample_h2(v, w2 // 2, h2)
    return y.tobytes() + u.tobytes() + v.tobytes()


def max_difference(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    assert a.shape == b.shape
    return lib().orc_max_difference(_ptr(a), _ptr(b), a.size)


def square_error(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    assert a.shape == b.shape
    return lib().orc_square_error(_ptr(a), _ptr(b), a.size)


def psnr(a, b, r=255.0):
    """tools/src/ocompare.ml:54-59"""
    import math
    mse = float(square_error(a, b)) / (float(a.shape[1]) * float(a.shape[0]))
    return 10.0 * math.log10(r * r / mse)


def ocaml_float_to_string(x):
    """OCaml/Base Float.to_string: "%.15g" if it round-trips, else "%.17g"."""
    s = "%.15g" % x
    return s if float(s) == x else "%.17g" % x
