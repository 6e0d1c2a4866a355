"""ctypes binding of libhvc_jpeg.so (include/hvc_jpeg.h)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("HVC_JPEG_LIB") or os.path.join(_DIR, "libhvc_jpeg.so")  # env override: A/B experiments only
_LIB = None

HVC_MEM_HOST, HVC_MEM_DEVICE = 0, 1

# every symbol include/hvc_jpeg.h declares
SYMBOLS = [
    "hvc_create", "hvc_destroy", "hvc_strerror", "hvc_last_hip_error", "hvc_version", "hvc_set_stream",
    "hvc_synchronize", "hvc_timer_begin", "hvc_timer_end", "hvc_set_profiling", "hvc_last_kernel_ms", "hvc_kernel_ms_history", "hvc_dequant_idct_recon", "hvc_decode_frames",
    "hvc_last_wide_blocks", "hvc_fdct_quant", "hvc_encode_frames", "hvc_upsample420", "hvc_device_alloc",
    "hvc_device_free", "hvc_memcpy_h2d", "hvc_memcpy_d2h",
    "hvc_jpeg_read_header", "hvc_jpeg_entropy_decode", "hvc_jpeg_get_yuv_frame", "hvc_jpeg_decode",
    "hvc_jpeg_decode_batch", "hvc_quant_table", "hvc_jpeg_encoder_layout", "hvc_jpeg_entropy_encode",
    "hvc_jpeg_encode", "hvc_set_decode_kernel", "hvc_reset_stream", "hvc_decode_frames_yuv444", "hvc_jpeg_decode_yuv444", "hvc_compare_planes", "hvc_jpeg_encode_batch", "hvc_jpeg_decode_batch_yuv444", "hvc_jpeg_encoder_check", "hvc_huffman_encode_frames", "hvc_jpeg_header", "hvc_jpeg_encode_batch_gpu", "hvc_jpeg_entropy_decode_gpu", "hvc_jpeg_decode_batch_gpu",
    "hvc_checksum_records", "hvc_encode_frames_recon", "hvc_set_host_cpus", "hvc_get_host_cpus",
    "hvc_host_threads", "hvc_host_threads_probe", "hvc_jpeg_entropy_decode2", "hvc_jpeg_get_cropped_planes",
    "hvc_jpeg_entropy_decode_restart", "hvc_set_restart_markers",
    "hvc_subsample420", "hvc_subsample422", "hvc_upsample422", "hvc_crop_planes", "hvc_yuv_frame_bytes", "hvc_yuv_convert",
    "hvc_host_alloc", "hvc_host_free", "hvc_host_register", "hvc_host_unregister", "hvc_decode_frames_submit",
    "hvc_encode_frames_submit", "hvc_wait", "hvc_slot_query", "hvc_slot_last_stats", "hvc_huffman_code_tables",
]
HVC_SLOTS = 4       # enum { HVC_SLOTS }
HVC_E_BUSY = -12


class HvcError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().hvc_strerror(code).decode() if _LIB is not None else str(code)
        super().__init__("hvc error %d (%s) %s" % (code, msg, what))


class Component(C.Structure):
    """struct hvc_component"""
    _fields_ = [("blocks_w", C.c_int), ("blocks_h", C.c_int), ("qtab", C.c_int), ("reserved", C.c_int),
                ("coef_offset", C.c_size_t), ("plane_offset", C.c_size_t), ("stride", C.c_size_t)]


class JpegComponent(C.Structure):
    """struct hvc_jpeg_component"""
    _fields_ = [(n, C.c_int) for n in ("identifier", "hscale", "vscale", "decoded_width", "decoded_height",
                                       "actual_width", "actual_height", "dc_table", "ac_table")]


class JpegInfo(C.Structure):
    """struct hvc_jpeg_info"""
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("n_comp", C.c_int), ("n_qtabs", C.c_int),
                ("comp", JpegComponent * 4), ("layout", Component * 4), ("qtabs", (C.c_uint16 * 64) * 4),
                ("coef_count", C.c_size_t), ("pixel_bytes", C.c_size_t), ("ecs_offset", C.c_size_t)]

    def qtab_array(self):
        return np.array([[self.qtabs[t][i] for i in range(64)] for t in range(self.n_qtabs)], dtype=np.uint16)

    def planes(self, pixels):
        """views of the padded planes inside a frame's pixel record (numpy uint8)"""
        out = []
        for i in range(self.n_comp):
            c, L = self.comp[i], self.layout[i]
            out.append(pixels[L.plane_offset:L.plane_offset + c.decoded_width * c.decoded_height].reshape(
                c.decoded_height, c.decoded_width))
        return out


class BatchStats(C.Structure):
    """struct hvc_batch_stats"""
    _fields_ = [("wall_ms", C.c_double), ("entropy_ms_sum", C.c_double), ("h2d_ms_sum", C.c_double),
                ("kernel_ms_sum", C.c_double), ("d2h_ms_sum", C.c_double), ("chunks", C.c_int), ("threads", C.c_int),
                ("frames_per_chunk", C.c_int), ("coef_bytes", C.c_uint64), ("host_prep_ms_sum", C.c_double)]


PAGE = os.sysconf("SC_PAGESIZE")


def page_aligned_empty(shape, dtype=np.uint8):
    """a numpy array on whole pages of its own (mmap): what hvc_host_register takes"""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = mmap.mmap(-1, max((n + PAGE - 1) // PAGE * PAGE, PAGE))
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class SlotStats(C.Structure):
    """struct hvc_slot_stats"""
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("h2d_bytes", C.c_uint64),
                ("d2h_bytes", C.c_uint64)]


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_DIR, "csrc", f) for f in os.listdir(os.path.join(_DIR, "csrc"))]
    srcs.append(os.path.join(os.path.dirname(_DIR), "include", "hvc_jpeg.h"))
    stale = not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_DIR, "csrc")] + (["-B"] if force else []))
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise ImportError("libhvc_jpeg.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C video-coding_amd/csrc` (there is no CPU fallback)")
        if "torch" not in sys.modules:
            # A process that also uses PyTorch-ROCm must run ONE HIP runtime: torch bundles
            # its own libamdhip64 (same SONAME as /opt/rocm's).  Loading torch first makes the
            # dynamic linker resolve this library's dependency to the copy torch already
            # mapped; the other order leaves torch without a usable device.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(_SO)
        vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
        L.hvc_strerror.restype = C.c_char_p
        L.hvc_strerror.argtypes = [i]
        L.hvc_version.restype = C.c_char_p
        L.hvc_create.argtypes = [C.POINTER(vp), i]
        L.hvc_destroy.argtypes = [vp]
        L.hvc_destroy.restype = None
        L.hvc_last_hip_error.argtypes = [vp]
        L.hvc_set_stream.argtypes = [vp, vp]
        L.hvc_synchronize.argtypes = [vp]
        L.hvc_timer_begin.argtypes = [vp]
        L.hvc_timer_end.argtypes = [vp, C.POINTER(C.c_float)]
        L.hvc_set_profiling.argtypes = [vp, i]
        L.hvc_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.hvc_kernel_ms_history.argtypes = [vp, C.POINTER(C.c_float), i]
        L.hvc_dequant_idct_recon.argtypes = [vp, vp, sz, vp, i, i, i, vp, sz, sz, i]
        L.hvc_decode_frames.argtypes = [vp, vp, sz, vp, i, C.POINTER(Component), i, i, vp, sz, i]
        L.hvc_decode_frames_yuv444.argtypes = [vp, vp, sz, vp, i, C.POINTER(Component), i, i, i, i, vp, sz, i]
        L.hvc_last_wide_blocks.argtypes = [vp, C.POINTER(C.c_uint64)]
        if hasattr(L, "hvc_fdct_quant"):
            L.hvc_fdct_quant.argtypes = [vp, vp, sz, sz, vp, i, i, i, vp, sz, i]
            L.hvc_encode_frames.argtypes = [vp, vp, sz, vp, i, C.POINTER(Component), i, i, vp, sz, i]
            L.hvc_upsample420.argtypes = [vp, vp, i, i, sz, vp, sz, i, sz, sz, i]
            for f in ("hvc_subsample420", "hvc_subsample422", "hvc_upsample422"):
                getattr(L, f).argtypes = [vp, vp, i, i, sz, vp, sz, i, sz, sz, i]
            L.hvc_crop_planes.argtypes = [vp, vp, i, i, sz, i, i, vp, i, i, sz, i, sz, sz, i]
            L.hvc_yuv_frame_bytes.argtypes = [i, i, i, C.POINTER(sz)]
            L.hvc_yuv_convert.argtypes = [vp, vp, i, i, i, i, i, vp, i, i, i, i, i]
        ip = C.POINTER(JpegInfo)
        L.hvc_jpeg_read_header.argtypes = [vp, sz, ip]
        L.hvc_jpeg_entropy_decode.argtypes = [vp, sz, ip, vp]
        L.hvc_jpeg_entropy_decode_restart.argtypes = [vp, sz, ip, vp]
        L.hvc_set_restart_markers.argtypes = [vp, i]
        L.hvc_jpeg_get_yuv_frame.argtypes = [ip, vp, vp, sz, C.POINTER(sz)]
        L.hvc_jpeg_get_cropped_planes.argtypes = [ip, vp, vp, sz, C.POINTER(sz)]
        L.hvc_jpeg_entropy_decode2.argtypes = [vp, sz, ip, vp, C.POINTER(i), vp, sz, ip, vp, C.POINTER(i)]
        L.hvc_jpeg_decode.argtypes = [vp, vp, sz, ip, vp, sz]
        L.hvc_jpeg_decode_yuv444.argtypes = [vp, vp, sz, ip, vp, sz]
        L.hvc_jpeg_decode_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), i, i, i, vp, sz, i,
                                            C.POINTER(BatchStats)]
        L.hvc_quant_table.argtypes = [i, i, vp]
        L.hvc_jpeg_decode_batch_gpu.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), i, i, i, vp, sz, i, i, C.POINTER(BatchStats)]
        L.hvc_jpeg_entropy_decode_gpu.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), i, vp, sz, i, ip, C.POINTER(i)]
        L.hvc_huffman_encode_frames.argtypes = [vp, ip, vp, sz, i, vp, sz, vp, i]
        L.hvc_jpeg_header.argtypes = [ip, vp, sz, C.POINTER(sz)]
        L.hvc_jpeg_decode_batch_yuv444.argtypes = L.hvc_jpeg_decode_batch.argtypes
        L.hvc_jpeg_encode_batch.argtypes = [vp, C.POINTER(vp), i, i, i, i, i, i, i, C.POINTER(vp), C.POINTER(sz),
                                            C.POINTER(sz), C.POINTER(BatchStats)]
        L.hvc_jpeg_encode_batch_gpu.argtypes = L.hvc_jpeg_encode_batch.argtypes
        L.hvc_compare_planes.argtypes = [vp, vp, sz, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.hvc_jpeg_encoder_layout.argtypes = [i, i, i, i, ip]
        L.hvc_jpeg_entropy_encode.argtypes = [ip, vp, vp, sz, C.POINTER(sz)]
        L.hvc_jpeg_encode.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, sz, C.POINTER(sz)]
        L.hvc_checksum_records.argtypes = [vp, vp, sz, sz, i, vp, i]
        L.hvc_set_host_cpus.argtypes = [vp, C.c_char_p]
        L.hvc_get_host_cpus.argtypes = [vp, C.c_char_p, sz, C.POINTER(i)]
        L.hvc_host_threads.argtypes = [vp, C.POINTER(i), C.POINTER(C.c_uint64)]
        L.hvc_host_threads_probe.argtypes = [i]
        L.hvc_encode_frames_recon.argtypes = [vp, vp, sz, vp, i, C.POINTER(Component), i, i, vp, sz, vp, vp, i]
        L.hvc_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.hvc_host_free.argtypes = [vp, vp]
        L.hvc_host_register.argtypes = [vp, vp, sz]
        L.hvc_host_unregister.argtypes = [vp, vp]
        L.hvc_decode_frames_submit.argtypes = [vp, i, vp, sz, vp, i, C.POINTER(Component), i, i, vp, sz, i]
        L.hvc_encode_frames_submit.argtypes = [vp, i, vp, sz, vp, i, C.POINTER(Component), i, i, vp, sz, i]
        L.hvc_wait.argtypes = [vp, i]
        L.hvc_slot_query.argtypes = [vp, i, C.POINTER(i)]
        L.hvc_slot_last_stats.argtypes = [vp, i, C.POINTER(SlotStats)]
        L.hvc_huffman_code_tables.argtypes = [vp, i, i, vp]
        L.hvc_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.hvc_device_free.argtypes = [vp, vp]
        L.hvc_memcpy_h2d.argtypes = [vp, vp, vp, sz]
        L.hvc_memcpy_d2h.argtypes = [vp, vp, vp, sz]
        _LIB = L
    return _LIB


KERNEL_SRCS = ("hvc_kernels.hip", "hvc_kernels.h", "hvc_idct_spec.h")   # csrc/Makefile KERNEL_SRCS, in that order


def kernel_build_id():
    """the kernel id the loaded library reports (hvc_version: "... kernels <id>")"""
    return lib().hvc_version().decode().rsplit(" ", 1)[-1]


def kernel_source_id():
    """the same id computed from the sources in the tree (None where they are not there)"""
    import hashlib
    h = hashlib.sha256()
    try:
        for f in KERNEL_SRCS:
            with open(os.path.join(_DIR, "csrc", f), "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    return h.hexdigest()[:12]


def _chk(code, what=""):
    if code != 0:
        raise HvcError(code, what)


def _addr(x):
    """host numpy array or device torch tensor / raw int address -> (address, where)"""
    if isinstance(x, np.ndarray):
        return x.ctypes.data, HVC_MEM_HOST
    if isinstance(x, int):
        return x, HVC_MEM_DEVICE
    if hasattr(x, "data_ptr"):
        return x.data_ptr(), (HVC_MEM_DEVICE if x.is_cuda else HVC_MEM_HOST)
    raise TypeError(type(x))


def components(specs):
    """specs: list of dicts(blocks_w, blocks_h, qtab, coef_offset, plane_offset, stride)"""
    arr = (Component * len(specs))()
    for a, s in zip(arr, specs):
        a.blocks_w, a.blocks_h, a.qtab = s["blocks_w"], s["blocks_h"], s.get("qtab", 0)
        a.coef_offset, a.plane_offset = s.get("coef_offset", 0), s.get("plane_offset", 0)
        a.stride = s.get("stride", s["blocks_w"] * 8)
    return arr


def layout_alignment(planes_bw_bh_qtab):
    """Where a resident batch's planes and frames should start (bytes).  Measured on MI355X (profiles/r05l_alignment_sweep.txt,
    r05m / r05n_layout_ab.txt, profiles/ANALYSIS.md section 7): with every plane of every frame -- pixels and coefficients -- on a
    64 KiB boundary k_decode_packed gains 0.4 ... 1.0 points of the HBM peak on 1080p 4:2:0 frames, whose second chroma plane
    otherwise starts 2 KiB off a 4 KiB boundary; 2 MiB boundaries give k_encode 0.6 ... 1.2 points on 4K 4:2:0 frames -- and cost 1080p
    frames 1.3 points, because they nearly double that batch's footprint.  So: the larger of 2 MiB / 64 KiB whose padding stays
    under 3 % of the frame.  The padding is neither read nor written."""
    tight = sum(bw * bh * 64 for bw, bh, _ in planes_bw_bh_qtab)
    for a in (2 << 20, 65536):
        if sum((bw * bh * 64 + a - 1) // a * a for bw, bh, _ in planes_bw_bh_qtab) <= 1.03 * tight:
            return a
    return 65536


def frame_layout(planes_bw_bh_qtab, align=1):
    """Frame record: component planes back to back, each starting on an `align`-byte boundary (1 = tight, the default;
    "auto" = layout_alignment()), pixel planes and coefficient planes alike; the frame strides are rounded up the same way.
    Returns (specs, coef elements per frame, pixel bytes per frame)."""
    if align == "auto":
        align = layout_alignment(planes_bw_bh_qtab)
    up = lambda x: (x + align - 1) // align * align
    specs, co, po = [], 0, 0
    for bw, bh, qt in planes_bw_bh_qtab:
        co, po = up(co * 2) // 2, up(po)
        specs.append(dict(blocks_w=bw, blocks_h=bh, qtab=qt, coef_offset=co, plane_offset=po, stride=bw * 8))
        co += bw * bh * 64
        po += bw * bh * 64
    return specs, up(co * 2) // 2, up(po)


def tight_records(t, specs, which):
    """the planes of a batch laid out by frame_layout(..., align) gathered into tight records (torch tensor [n, stride] ->
    [n, sum of the planes]): which = "plane_offset" (pixel bytes) or "coef_offset" (int16 elements) -- what the K5 golden
    checksums are defined on, whatever the resident layout"""
    import torch
    return torch.cat([t[:, s[which]:s[which] + s["blocks_w"] * s["blocks_h"] * 64] for s in specs], dim=1).contiguous()


def spread_records(tight, specs_tight, specs, stride, which):
    """the inverse: tight records [n, ...] into a zeroed batch [n, stride] laid out by `specs`"""
    import torch
    out = torch.zeros((tight.shape[0], stride), dtype=tight.dtype, device=tight.device)
    for a, b in zip(specs_tight, specs):
        n = a["blocks_w"] * a["blocks_h"] * 64
        out[:, b[which]:b[which] + n] = tight[:, a[which]:a[which] + n]
    return out


# -- host front end / back end (no GPU needed) ------------------------------------------------
def jpeg_read_header(data: bytes):
    """Decoder.Header.decode + Decoder.init geometry -> JpegInfo"""
    info = JpegInfo()
    _chk(lib().hvc_jpeg_read_header(data, len(data), C.byref(info)), "hvc_jpeg_read_header")
    return info


def jpeg_entropy_decode(data: bytes, info=None, restart_markers=False):
    """Huffman + DC prediction of one frame -> (info, int16 coefficient record).  restart_markers: the extension
    hvc_jpeg_entropy_decode_restart (DRI / RSTn honoured; off = the model's behaviour)"""
    info = info or jpeg_read_header(data)
    coefs = np.empty(info.coef_count, dtype=np.int16)
    fn = "hvc_jpeg_entropy_decode_restart" if restart_markers else "hvc_jpeg_entropy_decode"
    _chk(getattr(lib(), fn)(data, len(data), C.byref(info), coefs.ctypes.data), fn)
    return info, coefs


def jpeg_entropy_decode2(data_a: bytes, data_b: bytes):
    """hvc_jpeg_entropy_decode2: two files decoded in turn on this thread -> ((status, info, record), (status, info, record))"""
    ia, ib = jpeg_read_header(data_a), jpeg_read_header(data_b)
    ca, cb = np.empty(ia.coef_count, dtype=np.int16), np.empty(ib.coef_count, dtype=np.int16)
    sa, sb = C.c_int(1), C.c_int(1)
    _chk(lib().hvc_jpeg_entropy_decode2(data_a, len(data_a), C.byref(ia), ca.ctypes.data, C.byref(sa),
                                        data_b, len(data_b), C.byref(ib), cb.ctypes.data, C.byref(sb)), "hvc_jpeg_entropy_decode2")
    return (sa.value, ia, ca), (sb.value, ib, cb)


def jpeg_get_yuv_frame(info, pixels):
    """Decoder.get_yuv_frame: the crops of components 0, 1, 2 back to back -- HvcError(HVC_E_BAD_JPEG) where the model's
    Frame.of_planes raises (fewer than three components, planes it cannot name as 4:2:0 / 4:2:2 / 4:4:4)."""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    need = sum(info.comp[i].actual_width * info.comp[i].actual_height for i in range(min(info.n_comp, 3)))
    out = np.empty(need, dtype=np.uint8)
    n = C.c_size_t()
    _chk(lib().hvc_jpeg_get_yuv_frame(C.byref(info), pixels.ctypes.data, out.ctypes.data, need, C.byref(n)))
    return out[:n.value]


def jpeg_get_cropped_planes(info, pixels):
    """Decoder.crop over every decoded plane, back to back in scan order, whatever the sampling."""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    need = sum(info.comp[i].actual_width * info.comp[i].actual_height for i in range(info.n_comp))
    out = np.empty(need, dtype=np.uint8)
    n = C.c_size_t()
    _chk(lib().hvc_jpeg_get_cropped_planes(C.byref(info), pixels.ctypes.data, out.ctypes.data, need, C.byref(n)))
    return out


YUV_FORMATS = {"420": 420, "422": 422, "444": 444, "YUY2": 1, "UYVY": 2, "YVYU": 3}   # Yuv_format.arg_type (yuv_format.ml:66-77)


def yuv_frame_bytes(fmt, width, height):
    n = C.c_size_t()
    _chk(lib().hvc_yuv_frame_bytes(fmt, width, height, C.byref(n)), "hvc_yuv_frame_bytes")
    return n.value


def quant_table(chroma_table, quality):
    out = np.empty(64, dtype=np.uint16)
    _chk(lib().hvc_quant_table(1 if chroma_table else 0, quality, out.ctypes.data))
    return out


def compare_planes(a, b):
    """Ocompare.{max_difference, total_difference, square_error} (tools/src/ocompare.ml:6-47)"""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    if a.shape != b.shape:
        raise ValueError("planes differ in size")  # the model asserts (ocompare.ml:9-10)
    mx, tot, se = C.c_int(), C.c_uint64(), C.c_uint64()
    _chk(lib().hvc_compare_planes(a.ctypes.data, b.ctypes.data, a.size, C.byref(mx), C.byref(tot), C.byref(se)))
    return mx.value, tot.value, se.value


def huffman_code_tables(table_set, ctx=None):
    """hvc_huffman_code_tables: the encoder back ends' code tables as {"dc": [[length, bits, category]], "ac": [[[length, bits,
    run, size]]]} -- the shape of tests/golden/g8_code_tables.json (Tables.Encoder.dc_table / ac_table).  ctx = None: the host
    coder's; a Context: the GPU coder's, read back from its device memory."""
    codes = np.zeros(16 + 256, dtype=np.uint32)
    _chk(lib().hvc_huffman_code_tables(ctx._h if ctx is not None else None, table_set, HVC_MEM_DEVICE if ctx is not None else HVC_MEM_HOST,
                                       codes.ctypes.data), "hvc_huffman_code_tables")
    dc = [[int(codes[i]) & 31, int(codes[i]) >> 5, i] for i in range(16) if codes[i]]
    ac = []
    for run in range(16):
        row = [[int(codes[16 + (run << 4 | size)]) & 31, int(codes[16 + (run << 4 | size)]) >> 5, run, size] for size in range(16)]
        while row and row[-1][0] == 0:
            row.pop()
        if row and row[0][0] == 0:     # no size-0 symbol for this run: the model's placeholder (tables.ml:536-543)
            row[0] = [0, 0, 0, 0]
        ac.append(row)
    return {"dc": dc, "ac": ac}


def jpeg_header(info):
    """SOI .. SOS of the file Encoder.write_headers produces for this geometry / quality"""
    n = C.c_size_t()
    buf = np.empty(2048, dtype=np.uint8)
    _chk(lib().hvc_jpeg_header(C.byref(info), buf.ctypes.data, buf.size, C.byref(n)), "hvc_jpeg_header")
    return buf[:n.value].tobytes()


def jpeg_encoder_layout(width, height, chroma, quality):
    info = JpegInfo()
    _chk(lib().hvc_jpeg_encoder_layout(width, height, chroma, quality, C.byref(info)), "hvc_jpeg_encoder_layout")
    return info


def jpeg_entropy_encode(info, coefs):
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    assert coefs.size == info.coef_count
    cap = 8 * coefs.size + 4096  # 26 bits per coefficient at worst, every byte stuffed
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t()
    _chk(lib().hvc_jpeg_entropy_encode(C.byref(info), coefs.ctypes.data, out.ctypes.data, cap, C.byref(n)),
         "hvc_jpeg_entropy_encode")
    return out[:n.value].tobytes()


class Context:
    """hvc_ctx: one per GPU / host thread."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _chk(lib().hvc_create(C.byref(self._h), device), "hvc_create(device=%d)" % device)

    def close(self):
        if self._h:
            lib().hvc_destroy(self._h)   # (drains what is in flight: only then may a submission's buffers go)
            self._h = C.c_void_p()
            getattr(self, "_slot_buffers", {}).clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, handle):
        """handle: a hipStream_t as int, e.g. torch.cuda.current_stream().cuda_stream; 0 = HIP's
        default (null) stream -- which is what PyTorch's default stream is."""
        _chk(lib().hvc_set_stream(self._h, C.c_void_p(handle)))

    def reset_stream(self):
        """back to the context's own (non-blocking) stream"""
        lib().hvc_reset_stream.argtypes = [C.c_void_p]
        _chk(lib().hvc_reset_stream(self._h))

    def synchronize(self):
        _chk(lib().hvc_synchronize(self._h))

    def set_host_cpus(self, cpulist):
        """which CPUs the batch pipelines' host threads may run on: "0-15,32-47", "auto" (the GPU's NUMA node), None"""
        _chk(lib().hvc_set_host_cpus(self._h, cpulist.encode() if cpulist else None), "hvc_set_host_cpus(%r)" % (cpulist,))

    def get_host_cpus(self):
        buf, n = C.create_string_buffer(256), C.c_int()
        _chk(lib().hvc_get_host_cpus(self._h, buf, 256, C.byref(n)))
        return buf.value.decode(), n.value

    def host_threads(self):
        """(threads the context's pool holds, threads it has ever started): the batch pipelines reuse them"""
        alive, ever = C.c_int(), C.c_uint64()
        _chk(lib().hvc_host_threads(self._h, C.byref(alive), C.byref(ever)))
        return alive.value, ever.value

    def timer_begin(self):
        _chk(lib().hvc_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float()
        _chk(lib().hvc_timer_end(self._h, C.byref(ms)))
        return ms.value

    def set_restart_markers(self, honour=True):
        """the extension: the context's file-level entry points honour DRI / RSTn (default off = the model's behaviour)"""
        _chk(lib().hvc_set_restart_markers(self._h, 1 if honour else 0), "hvc_set_restart_markers")

    def set_decode_kernel(self, which):
        """0 packed (default) | 1 unpacked int32 | 2 int64 for every block -- identical output"""
        lib().hvc_set_decode_kernel.argtypes = [C.c_void_p, C.c_int]
        _chk(lib().hvc_set_decode_kernel(self._h, which))

    def set_profiling(self, on=True):
        _chk(lib().hvc_set_profiling(self._h, 1 if on else 0))

    def last_kernel_ms(self):
        ms = C.c_float()
        _chk(lib().hvc_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def kernel_ms_history(self, n):
        arr = (C.c_float * n)()
        _chk(lib().hvc_kernel_ms_history(self._h, arr, n))
        return list(arr)

    def checksum_records(self, data, record_bytes, n_records, record_stride=None):
        """K5: position-weighted 64-bit checksum per record (numpy uint64 array); data: numpy (host) or a
        device tensor / address."""
        a, where = _addr(data)
        sums = np.zeros(max(n_records, 1), dtype=np.uint64)
        _chk(lib().hvc_checksum_records(self._h, a, record_bytes, record_bytes if record_stride is None else record_stride,
                                        n_records, sums.ctypes.data, where), "hvc_checksum_records")
        return sums[:n_records]

    def last_wide_blocks(self):
        n = C.c_uint64()
        _chk(lib().hvc_last_wide_blocks(self._h, C.byref(n)))
        return n.value

    # -- the asynchronous seam: pinned host memory + slots ---------------------
    def host_alloc(self, shape, dtype=np.uint8):
        """pinned host memory (hvc_host_alloc) as a numpy array; give it back with host_free(array)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _chk(lib().hvc_host_alloc(self._h, n, C.byref(p)), "hvc_host_alloc(%d)" % n)
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def host_free(self, arr):
        _chk(lib().hvc_host_free(self._h, arr.ctypes.data), "hvc_host_free")

    def host_register(self, arr):
        """pins memory the caller owns in place; host_unregister(arr) before it goes.  Whole pages only (include/hvc_jpeg.h):
        arr from page_aligned_empty(), whose allocation is registered to its last page"""
        size = (arr.nbytes + PAGE - 1) // PAGE * PAGE
        _chk(lib().hvc_host_register(self._h, arr.ctypes.data, size), "hvc_host_register")

    def host_unregister(self, arr):
        _chk(lib().hvc_host_unregister(self._h, arr.ctypes.data), "hvc_host_unregister")

    def decode_frames_submit(self, slot, coefs, coef_frame_stride, qtabs, comps, n_frames, pixels, pixel_frame_stride):
        """hvc_decode_frames on HOST coefficient records (numpy; pinned for the overlap), asynchronously in `slot`;
        pixels: numpy (host: downloaded) or a device tensor / address (written in place).  wait(slot) completes it."""
        ca, w1 = _addr(coefs)
        assert w1 == HVC_MEM_HOST, "the coefficient records of a submission are host memory"
        pa, w2 = _addr(pixels)
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(-1, 64)
        arr = comps if not isinstance(comps, list) else components(comps)
        _chk(lib().hvc_decode_frames_submit(self._h, slot, ca, coef_frame_stride, q.ctypes.data, q.shape[0], arr, len(arr),
                                            n_frames, pa, pixel_frame_stride, w2), "hvc_decode_frames_submit(slot %d)" % slot)
        self._keep(slot, coefs, pixels)

    def encode_frames_submit(self, slot, pixels, pixel_frame_stride, qtabs, comps, n_frames, coefs, coef_frame_stride):
        """the encoder mirror: HOST pixel records in, coefficient records to numpy (host) or a device tensor"""
        pa, w1 = _addr(pixels)
        assert w1 == HVC_MEM_HOST, "the pixel records of a submission are host memory"
        ca, w2 = _addr(coefs)
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(-1, 64)
        arr = comps if not isinstance(comps, list) else components(comps)
        _chk(lib().hvc_encode_frames_submit(self._h, slot, pa, pixel_frame_stride, q.ctypes.data, q.shape[0], arr, len(arr),
                                            n_frames, ca, coef_frame_stride, w2), "hvc_encode_frames_submit(slot %d)" % slot)
        self._keep(slot, pixels, coefs)

    def _keep(self, slot, *buffers):
        """the ABI's rule: a submission's buffers stay valid until its hvc_wait -- an upload from memory the interpreter has
        meanwhile freed is a GPU page fault (tools/stress_seam.py found it the hard way), so the binding holds on to them"""
        if not hasattr(self, "_slot_buffers"):
            self._slot_buffers = {}
        self._slot_buffers[slot] = buffers

    def wait(self, slot):
        try:
            _chk(lib().hvc_wait(self._h, slot), "hvc_wait(slot %d)" % slot)
        finally:
            getattr(self, "_slot_buffers", {}).pop(slot, None)

    def slot_done(self, slot):
        d = C.c_int()
        _chk(lib().hvc_slot_query(self._h, slot, C.byref(d)), "hvc_slot_query")
        return bool(d.value)

    def slot_last_stats(self, slot):
        st = SlotStats()
        _chk(lib().hvc_slot_last_stats(self._h, slot, C.byref(st)), "hvc_slot_last_stats")
        return st

    # -- decode -------------------------------------------------------------
    def dequant_idct_recon(self, coefs, qtab, blocks_w, blocks_h, n_planes, plane, stride=None,
                           coef_plane_stride=0, plane_stride=0):
        ca, w1 = _addr(coefs)
        pa, w2 = _addr(plane)
        assert w1 == w2, "coefs and plane must live in the same memory space"
        q = np.ascontiguousarray(qtab, dtype=np.uint16)
        assert q.size == 64
        _chk(lib().hvc_dequant_idct_recon(self._h, ca, coef_plane_stride, q.ctypes.data, blocks_w, blocks_h,
                                          n_planes, pa, stride or blocks_w * 8, plane_stride, w1))

    def decode_frames(self, coefs, coef_frame_stride, qtabs, comps, n_frames, pixels, pixel_frame_stride):
        ca, w1 = _addr(coefs)
        pa, w2 = _addr(pixels)
        assert w1 == w2
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(-1, 64)
        arr = comps if not isinstance(comps, list) else components(comps)
        _chk(lib().hvc_decode_frames(self._h, ca, coef_frame_stride, q.ctypes.data, q.shape[0], arr, len(arr),
                                     n_frames, pa, pixel_frame_stride, w1))

    def decode_frames_yuv444(self, coefs, coef_frame_stride, qtabs, comps, n_frames, width, height, frames,
                             frame_stride=None):
        """4:2:0 coefficient records -> tight 4:4:4 frames (block stage + crop + supersample_hv2 fused)."""
        ca, w1 = _addr(coefs)
        fa, w2 = _addr(frames)
        assert w1 == w2
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(-1, 64)
        arr = comps if not isinstance(comps, list) else components(comps)
        _chk(lib().hvc_decode_frames_yuv444(self._h, ca, coef_frame_stride, q.ctypes.data, q.shape[0], arr, len(arr),
                                            n_frames, width, height, fa,
                                            3 * width * height if frame_stride is None else frame_stride, w1),
             "hvc_decode_frames_yuv444")

    def jpeg_decode(self, data: bytes):
        """Decoder.decode_a_frame minus the crop: (info, padded pixel record as numpy uint8)"""
        info = jpeg_read_header(data)
        pixels = np.zeros(info.pixel_bytes, dtype=np.uint8)
        _chk(lib().hvc_jpeg_decode(self._h, data, len(data), C.byref(info), pixels.ctypes.data, pixels.size),
             "hvc_jpeg_decode")
        return info, pixels

    def jpeg_decode_yuv444(self, data: bytes):
        """decode_a_frame + Planar_444.of_420 for a 4:2:0 file: (info, uint8 [3][height][width])"""
        info = jpeg_read_header(data)
        frame = np.zeros(3 * info.width * info.height, dtype=np.uint8)
        _chk(lib().hvc_jpeg_decode_yuv444(self._h, data, len(data), C.byref(info), frame.ctypes.data, frame.size),
             "hvc_jpeg_decode_yuv444")
        return info, frame.reshape(3, info.height, info.width)

    def jpeg_decode_batch(self, jpegs, pixels, pixel_frame_stride, threads=8, frames_per_chunk=32, yuv444=False,
                          gpu_entropy=False):
        """config 3 pipeline.  jpegs: list of bytes; pixels: numpy (host) or torch cuda tensor.
        yuv444: 4:2:0 files straight to tight 4:4:4 frames (3 * width * height bytes each).
        gpu_entropy: the Huffman reader on the GPU as well (hvc_jpeg_decode_batch_gpu)."""
        n = len(jpegs)
        ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(j), C.c_void_p) for j in jpegs])
        sizes = (C.c_size_t * n)(*[len(j) for j in jpegs])
        pa, where = _addr(pixels)
        st = BatchStats()
        if gpu_entropy:
            _chk(lib().hvc_jpeg_decode_batch_gpu(self._h, ptrs, sizes, n, threads, frames_per_chunk, pa, pixel_frame_stride,
                                                 where, 1 if yuv444 else 0, C.byref(st)), "hvc_jpeg_decode_batch_gpu")
            return st
        fn = lib().hvc_jpeg_decode_batch_yuv444 if yuv444 else lib().hvc_jpeg_decode_batch
        _chk(fn(self._h, ptrs, sizes, n, threads, frames_per_chunk, pa, pixel_frame_stride, where, C.byref(st)),
             "hvc_jpeg_decode_batch_yuv444" if yuv444 else "hvc_jpeg_decode_batch")
        return st

    def jpeg_encode(self, y, u, v, width, height, chroma=420, quality=75):
        """Encoder.encode_420/422/444 ~frame ~quality -> jpeg bytes"""
        y, u, v = (np.ascontiguousarray(p, dtype=np.uint8) for p in (y, u, v))
        cap = 4 * width * height + 65536
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t()
        _chk(lib().hvc_jpeg_encode(self._h, y.ctypes.data, u.ctypes.data, v.ctypes.data, width, height, chroma, quality,
                                   out.ctypes.data, cap, C.byref(n)), "hvc_jpeg_encode")
        return out[:n.value].tobytes()

    def jpeg_entropy_decode_gpu(self, jpegs, device=False):
        """Huffman decoding of a batch of files on the GPU: (info, coefficient records [n][coef_count], used_gpu)"""
        n = len(jpegs)
        info = jpeg_read_header(jpegs[0])
        ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(j), C.c_void_p) for j in jpegs])
        sizes = (C.c_size_t * n)(*[len(j) for j in jpegs])
        used = C.c_int(-1)
        if device:
            import torch
            # canary elements behind the last record: the reader's write passes must never store past the records,
            # settled stream or not (this wrapper is the harness of the tests and of tools/stress_hdec.py)
            guard = 8192
            buf = torch.full((n * info.coef_count + guard,), 0x5A5A, dtype=torch.int16, device="cuda")
            out = buf[:n * info.coef_count].view(n, info.coef_count)
            torch.cuda.synchronize()
            _chk(lib().hvc_jpeg_entropy_decode_gpu(self._h, ptrs, sizes, n, out.data_ptr(), info.coef_count, 1,
                                                   C.byref(info), C.byref(used)), "hvc_jpeg_entropy_decode_gpu")
            torch.cuda.synchronize()
            if not bool((buf[n * info.coef_count:] == 0x5A5A).all()):
                raise RuntimeError("hvc_jpeg_entropy_decode_gpu wrote past the coefficient records")
            return info, out.cpu().numpy(), used.value
        out = np.empty((n, info.coef_count), dtype=np.int16)
        _chk(lib().hvc_jpeg_entropy_decode_gpu(self._h, ptrs, sizes, n, out.ctypes.data, info.coef_count, 0, C.byref(info),
                                               C.byref(used)), "hvc_jpeg_entropy_decode_gpu")
        return info, out, used.value

    def huffman_encode_frames(self, info, coefs, coef_frame_stride, n_frames, out_cap=None):
        """Encoder back end on the GPU: (list of per-frame entropy-coded segments as bytes).  coefs: host
        int16 array or device tensor holding n_frames records."""
        ca, where = _addr(coefs)
        cap = out_cap or (n_frames * (info.coef_count // 64) * 243 + 4096)
        if where == 1:
            import torch
            out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            offs = torch.zeros(n_frames + 1, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            _chk(lib().hvc_huffman_encode_frames(self._h, C.byref(info), ca, coef_frame_stride, n_frames, out.data_ptr(),
                                                 cap, offs.data_ptr(), 1), "hvc_huffman_encode_frames")
            o = offs.cpu().numpy()
            data = out[:int(o[-1])].cpu().numpy()
        else:
            out = np.empty(cap, dtype=np.uint8)
            offs = np.zeros(n_frames + 1, dtype=np.uint64)
            _chk(lib().hvc_huffman_encode_frames(self._h, C.byref(info), ca, coef_frame_stride, n_frames, out.ctypes.data,
                                                 cap, offs.ctypes.data, 0), "hvc_huffman_encode_frames")
            o, data = offs, out
        return [data[int(o[f]):int(o[f + 1])].tobytes() for f in range(n_frames)]

    def jpeg_encode_batch(self, frames, width, height, chroma=420, quality=75, threads=8, frames_per_chunk=16,
                          gpu_entropy=False):
        """Encoder.encode_4xx over a batch of raw planar frames (bytes / uint8 arrays in Frame.input layout).
        Returns (list of jpeg byte strings, BatchStats).  gpu_entropy: Huffman coding on the GPU as well."""
        n = len(frames)
        arrs = [np.frombuffer(f, dtype=np.uint8) if isinstance(f, (bytes, bytearray)) else
                np.ascontiguousarray(f, dtype=np.uint8).reshape(-1) for f in frames]
        cw = width if chroma == 444 else width // 2
        ch = height // 2 if chroma == 420 else height
        need = width * height + 2 * cw * ch
        for a in arrs:
            if a.size < need:
                raise ValueError("frame shorter than %d bytes" % need)
        cap = 4 * width * height + 65536
        outs = [np.empty(cap, dtype=np.uint8) for _ in range(n)]
        fp = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        op = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * max(n, 1))(*([cap] * n))
        sizes = (C.c_size_t * max(n, 1))()
        st = BatchStats()
        fn = lib().hvc_jpeg_encode_batch_gpu if gpu_entropy else lib().hvc_jpeg_encode_batch
        _chk(fn(self._h, fp, n, width, height, chroma, quality, threads, frames_per_chunk, op, caps, sizes, C.byref(st)),
             "hvc_jpeg_encode_batch_gpu" if gpu_entropy else "hvc_jpeg_encode_batch")
        return [outs[f][:sizes[f]].tobytes() for f in range(n)], st

    # -- encode -------------------------------------------------------------
    def fdct_quant(self, plane, qtab, blocks_w, blocks_h, n_planes, coefs, stride=None, plane_stride=0,
                   coef_plane_stride=0):
        pa, w1 = _addr(plane)
        ca, w2 = _addr(coefs)
        assert w1 == w2
        q = np.ascontiguousarray(qtab, dtype=np.uint16)
        _chk(lib().hvc_fdct_quant(self._h, pa, stride or blocks_w * 8, plane_stride, q.ctypes.data, blocks_w,
                                  blocks_h, n_planes, ca, coef_plane_stride, w1))

    def encode_frames(self, pixels, pixel_frame_stride, qtabs, comps, n_frames, coefs, coef_frame_stride):
        pa, w1 = _addr(pixels)
        ca, w2 = _addr(coefs)
        assert w1 == w2
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(-1, 64)
        arr = comps if not isinstance(comps, list) else components(comps)
        _chk(lib().hvc_encode_frames(self._h, pa, pixel_frame_stride, q.ctypes.data, q.shape[0], arr, len(arr),
                                     n_frames, ca, coef_frame_stride, w1))

    def encode_frames_recon(self, pixels, pixel_frame_stride, qtabs, comps, n_frames, coefs, coef_frame_stride, recon=None,
                            error=None):
        """Encoder.encode_block with ~compute_reconstruction_error:true: coefficients + recon / error records"""
        pa, w1 = _addr(pixels)
        ca, w2 = _addr(coefs)
        ra = _addr(recon)[0] if recon is not None else None
        ea = _addr(error)[0] if error is not None else None
        assert w1 == w2
        q = np.ascontiguousarray(qtabs, dtype=np.uint16).reshape(-1, 64)
        arr = comps if not isinstance(comps, list) else components(comps)
        _chk(lib().hvc_encode_frames_recon(self._h, pa, pixel_frame_stride, q.ctypes.data, q.shape[0], arr, len(arr), n_frames,
                                           ca, coef_frame_stride, ra, ea, w1), "hvc_encode_frames_recon")

    def upsample420(self, src, cw, ch, dst, n_planes=1, src_stride=None, dst_stride=None, src_plane_stride=0,
                    dst_plane_stride=0):
        sa, w1 = _addr(src)
        da, w2 = _addr(dst)
        assert w1 == w2
        _chk(lib().hvc_upsample420(self._h, sa, cw, ch, src_stride or cw, da, dst_stride or 2 * cw, n_planes,
                                   src_plane_stride, dst_plane_stride, w1))

    def _plane_op(self, name, src, sw, sh, dst, dw, n_planes, src_stride, dst_stride, src_plane_stride, dst_plane_stride):
        sa, w1 = _addr(src)
        da, w2 = _addr(dst)
        assert w1 == w2
        _chk(getattr(lib(), name)(self._h, sa, sw, sh, src_stride or sw, da, dst_stride or dw, n_planes, src_plane_stride,
                                  dst_plane_stride, w1), name)

    def subsample420(self, src, sw, sh, dst, n_planes=1, src_stride=None, dst_stride=None, src_plane_stride=0, dst_plane_stride=0):
        """Planar_444.subsample_hv2: sw x sh -> (sw // 2) x (sh // 2)"""
        self._plane_op("hvc_subsample420", src, sw, sh, dst, sw // 2, n_planes, src_stride, dst_stride, src_plane_stride, dst_plane_stride)

    def subsample422(self, src, sw, sh, dst, n_planes=1, src_stride=None, dst_stride=None, src_plane_stride=0, dst_plane_stride=0):
        """Planar_444.subsample_h2: sw x sh -> (sw // 2) x sh"""
        self._plane_op("hvc_subsample422", src, sw, sh, dst, sw // 2, n_planes, src_stride, dst_stride, src_plane_stride, dst_plane_stride)

    def upsample422(self, src, cw, h, dst, n_planes=1, src_stride=None, dst_stride=None, src_plane_stride=0, dst_plane_stride=0):
        """Planar_444.supersample_h2: cw x h -> 2cw x h"""
        self._plane_op("hvc_upsample422", src, cw, h, dst, 2 * cw, n_planes, src_stride, dst_stride, src_plane_stride, dst_plane_stride)

    def crop_planes(self, src, sw, sh, x_pos, y_pos, dst, dw, dh, n_planes=1, src_stride=None, dst_stride=None,
                    src_plane_stride=0, dst_plane_stride=0):
        """Yuv.crop of one plane (clamped source coordinates)"""
        sa, w1 = _addr(src)
        da, w2 = _addr(dst)
        assert w1 == w2
        _chk(lib().hvc_crop_planes(self._h, sa, sw, sh, src_stride or sw, x_pos, y_pos, da, dw, dh, dst_stride or dw, n_planes,
                                   src_plane_stride, dst_plane_stride, w1), "hvc_crop_planes")

    def yuv_convert(self, src, src_format, src_size, dst, dst_format, dst_size, offset=(0, 0), n_frames=1):
        """Oconv.main's loop body: n_frames raw frames of src_format / src_size -> dst_format / dst_size"""
        sa, w1 = _addr(src)
        da, w2 = _addr(dst)
        assert w1 == w2
        _chk(lib().hvc_yuv_convert(self._h, sa, src_format, src_size[0], src_size[1], offset[0], offset[1], da, dst_format,
                                   dst_size[0], dst_size[1], n_frames, w1), "hvc_yuv_convert")
