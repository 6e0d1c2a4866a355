"""Command line of the path, shaped after the two executables the reference's cram tests drive:

    python -m video_coding_amd model decode frame IN.jpg [OUT.yuv] [-yuv444]      jpeg/bin/model.ml:29-45
    python -m video_coding_amd model encode frame IN.yuv WxH OUT.jpg [-quality 75] [-chroma 420]
                                                                                  jpeg/bin/model.ml:86-109
    python -m video_coding_amd oyuv compare {max-difference,mean-difference,mean-square-error,psnr}
                                            {y,u,v,yuv} FILE-1 FILE-2 WxH [-format 420]
                                                                                  tools/src/ocompare.ml:83-135
    python -m video_coding_amd oyuv convert IN.yuv WxH OUT.yuv [W2xH2] [-frames A-B] [-format 420] [-out-format F]
                                            [-src-offset X,Y]    formats 420 422 444 YUY2 UYVY YVYU
                                                                                  tools/src/oconv.ml:58-133

Every pixel goes through libhvc_jpeg.so on the GPU (there is no CPU path); output text matches the
reference's (`print_s` of an int / a float), so jpeg/test/*.t expectations can be checked verbatim.
"""
import argparse
import sys

import numpy as np

from . import hvc, yuv


def size_arg(s):
    w, h = s.lower().split("x")
    return int(w), int(h)


def model_decode_frame(a):
    data = open(a.bits, "rb").read()
    ctx = hvc.Context(a.device)
    try:
        if a.yuv444:
            _, frame = ctx.jpeg_decode_yuv444(data)
            out = frame.reshape(-1)
        else:
            info, pixels = ctx.jpeg_decode(data)
            out = hvc.jpeg_get_yuv_frame(info, pixels)  # Decoder.get_yuv_frame: crop to the actual size
    finally:
        ctx.close()
    if a.yuv:
        with open(a.yuv, "wb") as f:
            f.write(out.tobytes())
    else:
        sys.stdout.buffer.write(out.tobytes())


def model_encode_frame(a):
    w, h = a.size
    y, u, v = yuv.read_frame(a.yuv, w, h, a.chroma)
    ctx = hvc.Context(a.device)
    try:
        jpg = ctx.jpeg_encode(y, u, v, w, h, a.chroma, a.quality)
    finally:
        ctx.close()
    with open(a.bits, "wb") as f:
        f.write(jpg)


METRICS = {"max-difference": (yuv.max_difference, str), "mean-difference": (yuv.mean_difference, yuv.float_to_string),
           "mean-square-error": (yuv.mean_square_error, yuv.float_to_string), "psnr": (yuv.psnr, yuv.float_to_string)}


def oyuv_compare(a):
    w, h = a.size
    f1 = yuv.read_frame(a.file1, w, h, a.format)
    f2 = yuv.read_frame(a.file2, w, h, a.format)
    fn, show = METRICS[a.metric]
    for i in {"y": (0,), "u": (1,), "v": (2,), "yuv": (0, 1, 2)}[a.plane]:
        print(show(fn(f1[i], f2[i])))


def format_arg(s):
    """Yuv_format.arg_type (tools/src/yuv_format.ml:66-77)"""
    try:
        return hvc.YUV_FORMATS[s.upper()]
    except KeyError:
        raise argparse.ArgumentTypeError("Invalid YUV format")


def _two(s):
    """Offset.arg_type / Range.arg_type split on 'x', ',' and '-' (common/src/offset.ml:10-17, range.ml:10-19)"""
    import re
    return re.split("[x,-]", s)


def range_arg(s):
    """Range.arg_type: N = that frame, -B = 0 .. B, A-B (also AxB, A,B)"""
    parts = _two(s)
    try:
        if len(parts) == 1:
            return int(parts[0]), int(parts[0])
        if len(parts) == 2:
            return (0 if parts[0] == "" else int(parts[0])), int(parts[1])
    except ValueError:
        pass
    raise argparse.ArgumentTypeError("Invalid frame size specified")   # (the reference's message, for both types)


def offset_arg(s):
    """Offset.arg_type: XxY, X,Y or X-Y (so no negative offsets on the command line, as in the reference)"""
    parts = _two(s)
    try:
        if len(parts) == 2:
            return int(parts[0]), int(parts[1])
    except ValueError:
        pass
    raise argparse.ArgumentTypeError("Invalid frame size specified")


def oyuv_convert(a):
    """Oconv.main (tools/src/oconv.ml:111-133): skip frames.start frames, then convert frames start .. end; a short read
    ends the run (Plane.End_of_image), whatever has been written stays"""
    size_out = a.out_size or a.size
    fmt_out = a.out_format if a.out_format is not None else a.format
    n_in = hvc.yuv_frame_bytes(a.format, *a.size)
    n_out = hvc.yuv_frame_bytes(fmt_out, *size_out)
    first, last = a.frames
    with (sys.stdin.buffer if a.infile == "-" else open(a.infile, "rb")) as f:
        if first:
            f.read(first * n_in) if a.infile == "-" else f.seek(first * n_in)
        raw = f.read(max(0, last - first + 1) * n_in)   # `for _ = 0 to end_ - start` (oconv.ml:120-131): a reversed range converts nothing
    n = len(raw) // n_in   # (whole frames only: Oconv.input returns false on a short one)
    out = np.zeros(n * n_out, dtype=np.uint8)
    if n:
        ctx = hvc.Context(a.device)
        try:
            ctx.yuv_convert(np.frombuffer(raw, dtype=np.uint8)[:n * n_in], a.format, a.size, out, fmt_out, size_out,
                            offset=a.src_offset, n_frames=n)
        finally:
            ctx.close()
    if a.outfile == "-":
        sys.stdout.buffer.write(out.tobytes())
    else:
        with open(a.outfile, "wb") as f:
            f.write(out.tobytes())


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m video_coding_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-device", type=int, default=0)
    top = ap.add_subparsers(dest="tool", required=True)

    model = top.add_parser("model").add_subparsers(dest="direction", required=True)
    dec = model.add_parser("decode").add_subparsers(dest="what", required=True)
    p = dec.add_parser("frame")
    p.add_argument("bits")
    p.add_argument("yuv", nargs="?")
    p.add_argument("-yuv444", action="store_true", help="4:2:0 file straight to a 4:4:4 frame (fused kernel)")
    p.set_defaults(fn=model_decode_frame)
    enc = model.add_parser("encode").add_subparsers(dest="what", required=True)
    p = enc.add_parser("frame")
    p.add_argument("yuv")
    p.add_argument("size", type=size_arg)
    p.add_argument("bits")
    p.add_argument("-quality", type=int, default=75)
    p.add_argument("-chroma", type=int, default=420, choices=[420, 422, 444])
    p.set_defaults(fn=model_encode_frame)

    oyuv = top.add_parser("oyuv").add_subparsers(dest="cmd", required=True)
    p = oyuv.add_parser("compare")
    p.add_argument("metric", choices=sorted(METRICS))
    p.add_argument("plane", choices=["y", "u", "v", "yuv"])
    p.add_argument("file1")
    p.add_argument("file2")
    p.add_argument("size", type=size_arg)
    p.add_argument("-format", type=int, default=420, choices=[420, 422, 444])
    p.set_defaults(fn=oyuv_compare)
    p = oyuv.add_parser("convert")
    p.add_argument("infile")
    p.add_argument("size", type=size_arg)
    p.add_argument("outfile")
    p.add_argument("out_size", type=size_arg, nargs="?")
    p.add_argument("-frames", type=range_arg, default=(0, 0))
    p.add_argument("-format", type=format_arg, default=420)
    p.add_argument("-out-format", dest="out_format", type=format_arg, default=None)
    p.add_argument("-src-offset", dest="src_offset", type=offset_arg, default=(0, 0))
    p.set_defaults(fn=oyuv_convert)

    a = ap.parse_args(argv)
    a.fn(a)


if __name__ == "__main__":
    main()
