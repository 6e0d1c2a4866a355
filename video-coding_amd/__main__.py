"""Command line of the path, shaped after the two executables the reference's cram tests drive:

    python -m video_coding_amd model decode frame IN.jpg [OUT.yuv] [-yuv444]      jpeg/bin/model.ml:29-45
    python -m video_coding_amd model encode frame IN.yuv WxH OUT.jpg [-quality 75] [-chroma 420]
                                                                                  jpeg/bin/model.ml:86-109
    python -m video_coding_amd oyuv compare {max-difference,mean-difference,mean-square-error,psnr}
                                            {y,u,v,yuv} FILE-1 FILE-2 WxH [-format 420]
                                                                                  tools/src/ocompare.ml:83-135
    python -m video_coding_amd oyuv convert IN.yuv WxH OUT.yuv [-format 420] [-out-format 444]
                                            (4:2:0 -> 4:4:4 at the same size only; tools/src/oconv.ml)

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


def oyuv_convert(a):
    w, h = a.size
    if (a.format, a.out_format) != (420, 444) or (w & 1) or (h & 1):
        raise SystemExit("only 4:2:0 -> 4:4:4 at an even size is implemented (Planar_444.convert_from_420)")
    y, u, v = yuv.read_frame(a.infile, w, h, 420)
    ctx = hvc.Context(a.device)
    try:
        out = np.zeros((2, h, w), dtype=np.uint8)
        ctx.upsample420(np.ascontiguousarray(np.stack([u, v])), w // 2, h // 2, out, n_planes=2)
    finally:
        ctx.close()
    yuv.write_frame(a.outfile, [y, out[0], out[1]])


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
    p.add_argument("-format", type=int, default=420, choices=[420, 422, 444])
    p.add_argument("-out-format", dest="out_format", type=int, default=444, choices=[420, 422, 444])
    p.set_defaults(fn=oyuv_convert)

    a = ap.parse_args(argv)
    a.fn(a)


if __name__ == "__main__":
    main()
