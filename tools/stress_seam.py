#!/usr/bin/env python3
"""Randomised sweep over the asynchronous seam (hvc_decode_frames_submit / hvc_encode_frames_submit / hvc_wait, pinned and
registered memory) against the BLOCKING entry points of a second context -- both product paths, the blocking ones pinned to
the model restatement by the test suite: random geometries (1 ... 4 components, up to 1080p), batch sizes, padded strides and
frame strides, 8- and 16-bit tables, adversarial coefficients (every block through the fix-up list), host and device
destinations, pinned / registered / pageable sources, all slots in flight in any order, file-level batch decodes (both readers) of the
same context in between, refill threads writing the NEXT
submission's buffers while earlier ones run, the context's stream switched under way -- and invalid arguments, which must come
back as the blocking call's status with the slot left free.  One summary line; exit code 1 on a mismatch.

    python tools/stress_seam.py [--cases 300] [--seed 7]
"""
import argparse
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_coding_amd as hvc  # noqa: E402


def random_case(rng):
    ncomp = int(rng.integers(1, 5))
    big = rng.random() < 0.1
    planes = []
    for i in range(ncomp):
        bw = int(rng.integers(1, 241 if big else 40))
        bh = int(rng.integers(1, 137 if big else 30))
        planes.append((bw, bh, int(rng.integers(0, 2))))
    n = int(rng.integers(1, 4 if big else 12))
    # the caller's layout: planes anywhere in the record (8-byte aligned), rows padded, frames padded
    specs, co, po = [], 0, 0
    for bw, bh, qt in planes:
        co += int(rng.integers(0, 3)) * 64
        po += int(rng.integers(0, 3)) * 64
        stride = bw * 8 + int(rng.integers(0, 3)) * 8
        specs.append(dict(blocks_w=bw, blocks_h=bh, qtab=qt, coef_offset=co, plane_offset=po, stride=stride))
        co += bw * bh * 64
        po += stride * bh * 8
    coef_span, pixel_span = co, max(s["plane_offset"] + (s["blocks_h"] * 8 - 1) * s["stride"] + s["blocks_w"] * 8 for s in specs)
    cfs = co + int(rng.integers(0, 3)) * 64
    pfs = po + int(rng.integers(0, 3)) * 64
    kind = rng.choice(["natural", "natural", "dense", "adversarial"])
    coefs = np.zeros((n, cfs), dtype=np.int16)
    for s in specs:
        m = s["blocks_w"] * s["blocks_h"] * 64
        if kind == "natural":
            v = np.zeros((n, m // 64, 64), dtype=np.int16)
            v[:, :, 0] = rng.integers(-300, 301, size=v.shape[:2])
            mask = rng.random((n, m // 64, 63)) < 0.15
            v[:, :, 1:][mask] = rng.integers(-40, 41, size=int(mask.sum()))
            v = v.reshape(n, m)
        elif kind == "dense":
            v = rng.integers(-255, 256, size=(n, m)).astype(np.int16)
        else:
            v = rng.integers(-2047, 2048, size=(n, m)).astype(np.int16)
        coefs[:, s["coef_offset"]:s["coef_offset"] + m] = v
    q = np.stack([hvc.hvc.quant_table(0, int(rng.choice([10, 50, 90]))), hvc.hvc.quant_table(1, int(rng.choice([10, 50, 90])))])
    if rng.random() < 0.1:
        q = q.copy()
        q[int(rng.integers(0, 2)), int(rng.integers(0, 64))] = int(rng.integers(256, 65536))     # a 16-bit table: the int64 kernel for the call
    return dict(n=n, specs=specs, comps=hvc.hvc.components(specs), cfs=cfs, pfs=pfs, q=q, coefs=coefs, kind=kind,
                coef_span=coef_span, pixel_span=pixel_span)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--log", default=None, help="a file that receives one line per case BEFORE the case runs (to name the case a crash happened in)")
    args = ap.parse_args()
    log = open(args.log, "w") if args.log else None

    def note(*a):
        if log:
            log.write(" ".join(str(x) for x in a) + "\n")
            log.flush()
            os.fsync(log.fileno())
    import torch
    rng = np.random.Generator(np.random.PCG64(args.seed))
    ctx, ref = hvc.Context(0), hvc.Context(0)
    S = hvc.hvc.HVC_SLOTS
    side = torch.cuda.Stream()
    stats = dict(cases=0, decode=0, encode=0, device_dst=0, pinned=0, registered=0, adversarial=0, wide_tables=0, refill_threads=0,
                 invalid_refused=0, stream_switches=0, mismatches=0)
    in_flight = {}

    def retire(slot):
        kind, want, dst, cleanup, _source = in_flight.pop(slot)   # (_source: the caller's buffers stay alive until hvc_wait -- the ABI's rule)
        ctx.wait(slot)
        got = dst.cpu().numpy() if hasattr(dst, "cpu") else dst
        if not np.array_equal(got, want):
            stats["mismatches"] += 1
            print("MISMATCH", kind, slot, file=sys.stderr)
        for fn in cleanup:
            fn()

    for case in range(args.cases):
        c = random_case(rng)
        note("case", case, "n", c["n"], c["kind"], [(x["blocks_w"], x["blocks_h"], x["qtab"], x["stride"]) for x in c["specs"]], "cfs", c["cfs"], "pfs", c["pfs"],
             "qmax", int(c["q"].max()))
        stats["cases"] += 1
        stats["adversarial"] += c["kind"] == "adversarial"
        stats["wide_tables"] += int((c["q"] > 255).any())
        # the blocking reference (pageable host memory both ways) on the other context
        pix_want = np.full((c["n"], c["pfs"]), 0x5A, dtype=np.uint8)
        ref.decode_frames(c["coefs"], c["cfs"], c["q"], c["comps"], c["n"], pix_want, c["pfs"])
        encode = bool((c["q"] <= 255).all()) and rng.random() < 0.3
        if encode:   # the encoder's direction on the decoded planes
            coef_want = np.full((c["n"], c["cfs"]), 0x1111, dtype=np.int16)
            ref.encode_frames(pix_want, c["pfs"], c["q"], c["comps"], c["n"], coef_want, c["cfs"])
        note("  reference done; encode", encode)
        slot = int(rng.integers(0, S))
        if slot in in_flight:
            note("  retire", slot, in_flight[slot][0])
            retire(slot)
        src = pix_want if encode else c["coefs"]
        cleanup = []
        how = rng.integers(0, 3)
        if how == 0:
            p = ctx.host_alloc(src.shape, src.dtype)
            cleanup.append(lambda p=p: ctx.host_free(p))
            stats["pinned"] += 1
        elif how == 1:
            p = hvc.hvc.page_aligned_empty(src.shape, src.dtype)      # whole pages of its own: what hvc_host_register takes
            ctx.host_register(p)
            cleanup.append(lambda p=p: ctx.host_unregister(p))
            stats["registered"] += 1
        else:
            p = np.empty_like(src)
        # the source buffer is filled by another thread while EARLIER submissions are in flight (the caller's reader)
        t = threading.Thread(target=lambda: np.copyto(p, src))
        t.start()
        stats["refill_threads"] += 1
        want = coef_want if encode else pix_want
        fill = 0x1111 if encode else 0x5A     # bytes outside the planes stay the caller's
        if rng.random() < 0.35:
            dst = torch.from_numpy(np.full(want.shape, fill, dtype=want.dtype)).cuda()
            torch.cuda.synchronize()
            stats["device_dst"] += 1
        else:
            dst = np.full(want.shape, fill, dtype=want.dtype)
        if rng.random() < 0.1:   # the context moves to another stream: what is in flight is drained first (hvc_set_stream)
            ctx.set_stream(side.cuda_stream if rng.random() < 0.5 else 0)
            stats["stream_switches"] += 1
        elif rng.random() < 0.1:
            ctx.reset_stream()
        note("  slot", slot, "source", ["pinned", "registered", "pageable"][int(how)], "at %#x + %d" % (p.ctypes.data, p.nbytes),
             "dst", "device" if hasattr(dst, "cpu") else "host at %#x + %d" % (dst.ctypes.data, dst.nbytes))
        # an invalid variation first: the blocking call's status, at once, and the slot stays free
        if rng.random() < 0.3:
            bad = int(rng.integers(0, 4))
            note("  invalid variation", bad)
            try:
                if encode:
                    ctx.encode_frames_submit(slot, p, c["pixel_span"] - 8 if bad == 0 else c["pfs"] + (4 if bad == 1 else 0), c["q"],
                                             c["comps"], -1 if bad == 2 else c["n"], dst, c["cfs"] + (3 if bad == 3 else 0))
                else:
                    ctx.decode_frames_submit(slot, p, c["coef_span"] - 8 if bad == 0 else c["cfs"] + (4 if bad == 1 else 0), c["q"],
                                             c["comps"], -1 if bad == 2 else c["n"], dst, c["pfs"] + (3 if bad == 3 else 0))
                if not (bad == 0 and c["n"] == 1):
                    stats["mismatches"] += 1
                    print("ACCEPTED an invalid submission", bad, file=sys.stderr)
                else:
                    ctx.wait(slot)
            except hvc.HvcError as e:
                stats["invalid_refused"] += 1
                if e.code not in (-1, -4) or not ctx.slot_done(slot):
                    stats["mismatches"] += 1
                    print("invalid submission: code", e.code, file=sys.stderr)
        t.join()
        if encode:
            ctx.encode_frames_submit(slot, p, c["pfs"], c["q"], c["comps"], c["n"], dst, c["cfs"])
            stats["encode"] += 1
        else:
            ctx.decode_frames_submit(slot, p, c["cfs"], c["q"], c["comps"], c["n"], dst, c["pfs"])
            stats["decode"] += 1
        in_flight[slot] = ("enc" if encode else "dec", want, dst, cleanup, p)
        note("  submitted")
        # another pipeline of the SAME context while the slots are in flight: the file-level batch decode (host or GPU Huffman
        # reader: its own pinned ring, the same copy stream and fix-up list) against the other context's single-file decode
        if rng.random() < 0.12:
            from video_coding_amd.synth import synth_pixels
            w, h = int(rng.integers(1, 20)) * 16, int(rng.integers(1, 12)) * 16
            sd, quality = int(rng.integers(0, 1 << 30)), int(rng.choice([30, 75, 95]))   # (a batch shares one geometry and one set of tables)
            files = [ref.jpeg_encode(synth_pixels(sd + 3 * k, h, w), synth_pixels(sd + 3 * k + 1, h // 2, w // 2), synth_pixels(sd + 3 * k + 2, h // 2, w // 2),
                                     w, h, 420, quality) for k in range(int(rng.integers(1, 4)))]
            batch = [files[k % len(files)] for k in range(int(rng.integers(1, 20)))]
            info = hvc.hvc.jpeg_read_header(batch[0])
            gpu_reader = bool(rng.integers(0, 2))
            note("  batch decode of", len(batch), "files", w, "x", h, "gpu reader" if gpu_reader else "host reader")
            out = np.zeros((len(batch), info.pixel_bytes), dtype=np.uint8)
            ctx.jpeg_decode_batch(batch, out, info.pixel_bytes, threads=int(rng.integers(1, 5)), frames_per_chunk=int(rng.integers(1, 9)),
                                  gpu_entropy=gpu_reader)
            singles = {f: ref.jpeg_decode(f)[1] for f in set(batch)}
            stats["batch_calls"] = stats.get("batch_calls", 0) + 1
            if not all(np.array_equal(out[k], singles[f]) for k, f in enumerate(batch)):
                stats["mismatches"] += 1
                print("MISMATCH batch decode", file=sys.stderr)
        if rng.random() < 0.25 and in_flight:
            early = list(in_flight)[int(rng.integers(0, len(in_flight)))]
            note("  early retire", early)
            retire(early)
    for slot in list(in_flight):
        retire(slot)
    ctx.close()
    ref.close()
    print(stats)
    return 1 if stats["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
