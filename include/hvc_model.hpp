/* hvc_model.hpp -- the reference's HOST interface for the block-transform path, in C++17 over the C ABI of hvc_jpeg.h.
 *
 * The reference (hardcamls/video-coding) is OCaml and this image has no OCaml toolchain, so the host side above the C ABI
 * exists twice: as the uncompiled OCaml binding a maintainer would add (integration/ocaml/) and, compiled and
 * tested, as this header-only mirror: the same module and function names, the same argument meaning and the same error
 * behaviour as the reference's interfaces, so that a test written against it reads like the reference's own
 * (tests/cpp/model_tests.cpp follows jpeg/model/test/test_chen_dct.ml and jpeg/test/model-encode-and-decode.t).
 *
 *   common/src/plane.mli:6-41      Plane   create, width, height, .![] (bounds-checked), blit_available, output, input
 *   common/src/frame.mli:10-36     Frame   create, of_planes (chroma mode inferred; raises as frame.ml:42-61), y / u / v, output, input
 *   jpeg/model/src/decoder.mli:8-59  Decoder::Header::decode, init, decode, get_decoded_planes, get_yuv_frame, decode_a_frame;
 *                                    decode_frames = the patch's decode_frames_gpu (the asynchronous seam, two slots)
 *   jpeg/model/src/encoder.mli:132-135  Encoder::encode_420 / encode_422 / encode_444; encode_frames = the patch's
 *                                    encode_frames_gpu (the asynchronous seam the encoder's way round)
 *   jpeg/model/src/quant_tables.mli  Quant_tables::scale
 *   jpeg/model/src/dct.mli:8-11    Dct::Chen through the block stage it lives in (Decoder::recon_of_coefs, Encoder::quant_of_pixels)
 *   tools/src/ocompare.ml:8-59     Ocompare::max_difference, psnr; Float.to_string as the cram tests print it
 *   tools/src/oconv.ml:111-133     Oconv::convert (one planar frame: size, offset, format)
 *
 * Everything computes on the GPU through libhvc_jpeg.so; where the model raises (`raise_s`), these throw hvc_model::Error
 * carrying the hvc_status.  There is no CPU fallback: Ctx's constructor throws without a gfx950 GPU. */
#ifndef HVC_MODEL_HPP
#define HVC_MODEL_HPP

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hvc_jpeg.h"

namespace hvc_model {

/* raise_s [%message "hvc" (code : int) (msg : string)] */
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what + ": " + hvc_strerror(c)), code(c) {}
};
inline void check(int rc, const char *what) {
    if (rc != HVC_OK) throw Error(rc, what);
}

/* one per host thread and GPU (include/hvc_jpeg.h conventions) */
class Ctx {
    hvc_ctx *h_ = nullptr;

  public:
    explicit Ctx(int device = 0) { check(hvc_create(&h_, device), "hvc_create"); }
    ~Ctx() { hvc_destroy(h_); }
    Ctx(const Ctx &) = delete;
    Ctx &operator=(const Ctx &) = delete;
    hvc_ctx *get() const { return h_; }
};

/* common/src/plane.ml:4-61 */
class Plane {
    int width_ = 0, height_ = 0;
    std::vector<uint8_t> plane_;

  public:
    struct End_of_image : std::exception {};
    static Plane create(int width, int height) { /* zero-filled (plane.ml:11-17) */
        Plane p;
        p.width_ = width;
        p.height_ = height;
        p.plane_.assign((size_t)width * (size_t)height, 0);
        return p;
    }
    int width() const { return width_; }
    int height() const { return height_; }
    uint8_t *data() { return plane_.data(); }
    const uint8_t *data() const { return plane_.data(); }
    size_t size() const { return plane_.size(); }
    uint8_t at(int x, int y) const { /* .![] (plane.ml:43-50) */
        if (x < 0 || x >= width_ || y < 0 || y >= height_) throw Error(HVC_E_INVALID_ARG, "[Plane.get] out of bounds");
        return plane_[(size_t)y * width_ + x];
    }
    void set(int x, int y, uint8_t v) { /* .![]<- (plane.ml:52-61) */
        if (x < 0 || x >= width_ || y < 0 || y >= height_) throw Error(HVC_E_INVALID_ARG, "[Plane.set] out of bounds");
        plane_[(size_t)y * width_ + x] = v;
    }
    static void blit_available(const Plane &src, Plane &dst) { /* the top-left part both hold (plane.ml:27-37) */
        const int w = src.width_ < dst.width_ ? src.width_ : dst.width_, h = src.height_ < dst.height_ ? src.height_ : dst.height_;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) dst.plane_[(size_t)y * dst.width_ + x] = src.plane_[(size_t)y * src.width_ + x];
    }
    void output(std::ostream &o) const { o.write(reinterpret_cast<const char *>(plane_.data()), (std::streamsize)plane_.size()); }
    void input(std::istream &i) {
        i.read(reinterpret_cast<char *>(plane_.data()), (std::streamsize)plane_.size());
        if ((size_t)i.gcount() != plane_.size()) throw End_of_image();
    }
};

/* common/src/frame.ml:3-76 */
class Frame {
  public:
    enum class Chroma_subsampling { C420 = 420, C422 = 422, C444 = 444 };

  private:
    Plane y_, u_, v_;
    Chroma_subsampling cs_ = Chroma_subsampling::C420;
    static int cwidth(Chroma_subsampling c, int w) { return c == Chroma_subsampling::C444 ? w : w / 2; }
    static int cheight(Chroma_subsampling c, int h) { return c == Chroma_subsampling::C420 ? h / 2 : h; }

  public:
    static Frame create(Chroma_subsampling c, int width, int height) {
        Frame f;
        f.cs_ = c;
        f.y_ = Plane::create(width, height);
        f.u_ = Plane::create(cwidth(c, width), cheight(c, height));
        f.v_ = Plane::create(cwidth(c, width), cheight(c, height));
        return f;
    }
    static Frame of_planes(Plane y, Plane u, Plane v) { /* infer_chroma_subsampling (frame.ml:42-56) */
        if (u.width() != v.width() || u.height() != v.height()) throw Error(HVC_E_BAD_JPEG, "Chroma planes must be same width and height");
        Frame f;
        bool found = false;
        for (Chroma_subsampling c : {Chroma_subsampling::C420, Chroma_subsampling::C422, Chroma_subsampling::C444})
            if (!found && cwidth(c, y.width()) == u.width() && cheight(c, y.height()) == u.height()) {
                f.cs_ = c;
                found = true;
            }
        if (!found) throw Error(HVC_E_BAD_JPEG, "Could not infer chroma subsampling");
        f.y_ = std::move(y);
        f.u_ = std::move(u);
        f.v_ = std::move(v);
        return f;
    }
    int width() const { return y_.width(); }
    int height() const { return y_.height(); }
    Chroma_subsampling chroma_subsampling() const { return cs_; }
    Plane &y() { return y_; }
    Plane &u() { return u_; }
    Plane &v() { return v_; }
    const Plane &y() const { return y_; }
    const Plane &u() const { return u_; }
    const Plane &v() const { return v_; }
    void output(std::ostream &o) const { /* Y then U then V, raw (frame.ml:66-70) */
        y_.output(o);
        u_.output(o);
        v_.output(o);
    }
    void input(std::istream &i) {
        y_.input(i);
        u_.input(i);
        v_.input(i);
    }
};

namespace Quant_tables { /* quant_tables.ml:139-147: scale luma / chroma quality (host, no GPU) */
enum Table { luma = 0, chroma = 1 };
inline std::array<uint16_t, 64> scale(Table table, int quality) {
    std::array<uint16_t, 64> t{};
    check(hvc_quant_table((int)table, quality, t.data()), "Quant_tables.scale");
    return t;
}
} // namespace Quant_tables

namespace Decoder {
/* Header.decode (decoder.ml:36-70) + the geometry of init (:294-345) */
struct Header {
    hvc_jpeg_info info{};
    static Header decode(const std::string &bits) {
        Header h;
        check(hvc_jpeg_read_header(reinterpret_cast<const uint8_t *>(bits.data()), bits.size(), &h.info), "Decoder.Header.decode");
        return h;
    }
    int width() const { return info.width; }
    int height() const { return info.height; }
    int components() const { return info.n_comp; }
};

/* decode_block minus the Huffman part for ONE block (decoder.ml:142-149 with dc_pred = 0, dct.ml:100-107, decoder.ml:213-224):
 * coefs in zig-zag order, DC absolute; returns Component.recon */
inline std::array<uint8_t, 64> recon_of_coefs(Ctx &ctx, const std::array<int16_t, 64> &coefs, const std::array<uint16_t, 64> &qtab) {
    alignas(16) int16_t c[64];
    alignas(16) uint8_t out[64];
    for (int i = 0; i < 64; i++) c[i] = coefs[i];
    check(hvc_dequant_idct_recon(ctx.get(), c, 0, qtab.data(), 1, 1, 1, out, 8, 0, HVC_MEM_HOST), "Decoder.decode_block");
    std::array<uint8_t, 64> r{};
    for (int i = 0; i < 64; i++) r[i] = out[i];
    return r;
}

class t {
    Ctx *ctx_;
    Header header_;
    std::string bits_;
    std::vector<uint8_t> pixels_;
    bool decoded_ = false;

  public:
    t(Ctx &ctx, Header header, std::string bits) : ctx_(&ctx), header_(std::move(header)), bits_(std::move(bits)) {}
    void decode() { /* Decoder.decode (decoder.ml:397): Huffman on the host, the block stage on the GPU */
        pixels_.assign(header_.info.pixel_bytes ? header_.info.pixel_bytes : 1, 0);
        check(hvc_jpeg_decode(ctx_->get(), reinterpret_cast<const uint8_t *>(bits_.data()), bits_.size(), &header_.info,
                              pixels_.data(), pixels_.size()),
              "Decoder.decode");
        decoded_ = true;
    }
    std::vector<Plane> get_decoded_planes() const { /* padded planes (decoder.ml:399-401) */
        if (!decoded_) throw Error(HVC_E_INVALID_ARG, "Decoder.get_decoded_planes before decode");
        std::vector<Plane> out;
        for (int i = 0; i < header_.info.n_comp; i++) {
            const hvc_component &L = header_.info.layout[i];
            Plane p = Plane::create(L.blocks_w * 8, L.blocks_h * 8);
            for (int y = 0; y < p.height(); y++)
                for (int x = 0; x < p.width(); x++) p.set(x, y, pixels_[L.plane_offset + (size_t)y * L.stride + x]);
            out.push_back(std::move(p));
        }
        return out;
    }
    Frame get_yuv_frame() const { /* crop + Frame.of_planes (decoder.ml:403-420): raises where of_planes does */
        if (!decoded_) throw Error(HVC_E_INVALID_ARG, "Decoder.get_yuv_frame before decode");
        size_t len = 0;
        std::vector<uint8_t> buf(pixels_.size() + 1);
        check(hvc_jpeg_get_yuv_frame(&header_.info, pixels_.data(), buf.data(), buf.size(), &len), "Decoder.get_yuv_frame");
        Plane pl[3];
        size_t off = 0;
        for (int i = 0; i < 3; i++) {
            const hvc_jpeg_component &c = header_.info.comp[i];
            pl[i] = Plane::create(c.actual_width, c.actual_height);
            for (size_t k = 0; k < pl[i].size(); k++) pl[i].data()[k] = buf[off + k];
            off += pl[i].size();
        }
        return Frame::of_planes(std::move(pl[0]), std::move(pl[1]), std::move(pl[2]));
    }
    const Header &header() const { return header_; }
};
inline t init(Ctx &ctx, const Header &header, const std::string &bits) { return t(ctx, header, bits); }
inline Frame decode_a_frame(Ctx &ctx, const std::string &bits) { /* decoder.ml:422-427 */
    t d = init(ctx, Header::decode(bits), bits);
    d.decode();
    return d.get_yuv_frame();
}

/* List.map ~f:decode_a_frame through the ABI's asynchronous seam -- the compiled twin of the OCaml patch's
 * Decoder.decode_frames_gpu (integration/ocaml/hvc_backend.patch): two slots, each with a pinned coefficient
 * record and a pinned pixel record that grow with the largest frame; while the GPU works on file k (upload, block stage,
 * download: hvc_decode_frames_submit returns at once), the calling thread's Huffman reader (here hvc_jpeg_entropy_decode, the
 * host half of Decoder.decode: decoder.ml:118-140, 143) is already filling file k + 1's record.  Same frames as decode_a_frame. */
inline std::vector<Frame> decode_frames(Ctx &ctx, const std::vector<std::string> &files) {
    struct Slot {
        int16_t *record = nullptr;
        uint8_t *pixels = nullptr;
        size_t coef_cap = 0, pixel_cap = 0;
        bool pending = false;
        hvc_jpeg_info info{};
    } slots[2];
    std::vector<Frame> frames;
    auto release = [&](Slot &s) {
        if (s.record) hvc_host_free(ctx.get(), s.record);
        if (s.pixels) hvc_host_free(ctx.get(), s.pixels);
        s.record = nullptr;
        s.pixels = nullptr;
        s.coef_cap = s.pixel_cap = 0;
    };
    auto retire = [&](int index) { /* the file whose submission sits in the slot: wait, crop, Frame.of_planes */
        Slot &s = slots[index];
        if (!s.pending) return;
        s.pending = false;
        check(hvc_wait(ctx.get(), index), "hvc_wait");
        size_t len = 0;
        std::vector<uint8_t> buf(s.info.pixel_bytes + 1);
        check(hvc_jpeg_get_yuv_frame(&s.info, s.pixels, buf.data(), buf.size(), &len), "Decoder.get_yuv_frame");
        Plane pl[3];
        size_t off = 0;
        for (int i = 0; i < 3; i++) {
            const hvc_jpeg_component &c = s.info.comp[i];
            pl[i] = Plane::create(c.actual_width, c.actual_height);
            for (size_t k = 0; k < pl[i].size(); k++) pl[i].data()[k] = buf[off + k];
            off += pl[i].size();
        }
        frames.push_back(Frame::of_planes(std::move(pl[0]), std::move(pl[1]), std::move(pl[2])));
    };
    struct Finally { /* a failure half way: nothing may stay in flight on pinned memory that is about to be freed */
        std::function<void()> f;
        ~Finally() { f(); }
    } finally{[&] {
        for (int i = 0; i < 2; i++) {
            if (slots[i].pending) (void)hvc_wait(ctx.get(), i);
            release(slots[i]);
        }
    }};
    for (size_t k = 0; k < files.size(); k++) {
        const int index = (int)(k & 1);
        retire(index); /* file k - 2 used this slot: its frame is due before the slot's records are written again */
        Slot &s = slots[index];
        const std::string &bits = files[k];
        const uint8_t *data = reinterpret_cast<const uint8_t *>(bits.data());
        check(hvc_jpeg_read_header(data, bits.size(), &s.info), "Decoder.Header.decode");
        if (s.info.coef_count > s.coef_cap || s.info.pixel_bytes > s.pixel_cap) {
            release(s);
            check(hvc_host_alloc(ctx.get(), s.info.coef_count * sizeof(int16_t), reinterpret_cast<void **>(&s.record)), "hvc_host_alloc");
            check(hvc_host_alloc(ctx.get(), s.info.pixel_bytes, reinterpret_cast<void **>(&s.pixels)), "hvc_host_alloc");
            s.coef_cap = s.info.coef_count;
            s.pixel_cap = s.info.pixel_bytes;
        }
        /* phase 1 on this thread -- while file k - 1 is in flight in the other slot */
        check(hvc_jpeg_entropy_decode(data, bits.size(), &s.info, s.record), "Decoder.huffman_decode");
        check(hvc_decode_frames_submit(ctx.get(), index, s.record, s.info.coef_count, &s.info.qtabs[0][0], s.info.n_qtabs, s.info.layout,
                                       s.info.n_comp, 1, s.pixels, s.info.pixel_bytes, HVC_MEM_HOST),
              "hvc_decode_frames_submit");
        s.pending = true;
    }
    retire((int)(files.size() & 1)); /* drain in submission order: the older of the two first */
    retire((int)((files.size() + 1) & 1));
    return frames;
}
} // namespace Decoder

namespace Encoder {
/* encode_block's front half for ONE block (encoder.ml:81-108, dct.ml:189-196): pixels in raster order; returns quant (zig-zag) */
inline std::array<int16_t, 64> quant_of_pixels(Ctx &ctx, const std::array<uint8_t, 64> &pixels, const std::array<uint16_t, 64> &qtab) {
    alignas(16) uint8_t p[64];
    alignas(16) int16_t c[64];
    for (int i = 0; i < 64; i++) p[i] = pixels[i];
    check(hvc_fdct_quant(ctx.get(), p, 8, 0, qtab.data(), 1, 1, 1, c, 0, HVC_MEM_HOST), "Encoder.encode_block");
    std::array<int16_t, 64> r{};
    for (int i = 0; i < 64; i++) r[i] = c[i];
    return r;
}
/* Encoder.encode_4xx ~frame ~quality ~writer (encoder.ml:512-541); returns the writer's buffer */
inline std::string encode(Ctx &ctx, const Frame &frame, int quality, Frame::Chroma_subsampling chroma) {
    if (frame.chroma_subsampling() != chroma) throw Error(HVC_E_INVALID_ARG, "Encoder.encode: the frame's chroma subsampling is another");
    std::vector<uint8_t> out(4 * (size_t)frame.width() * frame.height() + 65536);
    size_t len = 0;
    check(hvc_jpeg_encode(ctx.get(), frame.y().data(), frame.u().data(), frame.v().data(), frame.width(), frame.height(), (int)chroma,
                          quality, out.data(), out.size(), &len),
          "Encoder.encode");
    return std::string(reinterpret_cast<const char *>(out.data()), len);
}
inline std::string encode_420(Ctx &ctx, const Frame &f, int quality) { return encode(ctx, f, quality, Frame::Chroma_subsampling::C420); }
inline std::string encode_422(Ctx &ctx, const Frame &f, int quality) { return encode(ctx, f, quality, Frame::Chroma_subsampling::C422); }
inline std::string encode_444(Ctx &ctx, const Frame &f, int quality) { return encode(ctx, f, quality, Frame::Chroma_subsampling::C444); }

/* List.map ~f:(encode_4xx ~quality) through the ABI's asynchronous seam, the encoder's way round (the patch's
 * Encoder.encode_frames_gpu): the GPU does the FRONT half of encode_block for a whole frame -- level shift, forward DCT,
 * quantiser (hvc_encode_frames_submit: encoder.ml:81-108, dct.ml:109-196) -- and while it works on frame k + 1 the calling
 * thread runs the BACK half of frame k: rle + write_bits + headers (here hvc_jpeg_entropy_encode: encoder.ml:127-193, 371-418).
 * Two slots with pinned pixel and coefficient records.  Same files as encode_4xx, byte for byte. */
inline std::vector<std::string> encode_frames(Ctx &ctx, const std::vector<Frame> &frames, int quality) {
    struct Slot {
        uint8_t *pixels = nullptr;
        int16_t *record = nullptr;
        size_t pixel_cap = 0, coef_cap = 0;
        bool pending = false;
        hvc_jpeg_info info{};
    } slots[2];
    std::vector<std::string> files;
    auto release = [&](Slot &s) {
        if (s.pixels) hvc_host_free(ctx.get(), s.pixels);
        if (s.record) hvc_host_free(ctx.get(), s.record);
        s.pixels = nullptr;
        s.record = nullptr;
        s.pixel_cap = s.coef_cap = 0;
    };
    auto retire = [&](int index) { /* the frame whose coefficients the slot is waiting for: wait, then the host's half */
        Slot &s = slots[index];
        if (!s.pending) return;
        s.pending = false;
        check(hvc_wait(ctx.get(), index), "hvc_wait");
        std::vector<uint8_t> out(8 * s.info.coef_count + 4096);
        size_t len = 0;
        check(hvc_jpeg_entropy_encode(&s.info, s.record, out.data(), out.size(), &len), "Encoder.rle / write_bits");
        files.emplace_back(reinterpret_cast<const char *>(out.data()), len);
    };
    struct Finally {
        std::function<void()> f;
        ~Finally() { f(); }
    } finally{[&] {
        for (int i = 0; i < 2; i++) {
            if (slots[i].pending) (void)hvc_wait(ctx.get(), i);
            release(slots[i]);
        }
    }};
    for (size_t k = 0; k < frames.size(); k++) {
        const int index = (int)(k & 1);
        retire(index);
        Slot &s = slots[index];
        const Frame &f = frames[k];
        check(hvc_jpeg_encoder_layout(f.width(), f.height(), (int)f.chroma_subsampling(), quality, &s.info), "Encoder.create");
        check(hvc_jpeg_encoder_check(&s.info), "Encoder.encode_seq");
        if (s.info.pixel_bytes > s.pixel_cap || s.info.coef_count > s.coef_cap) {
            release(s);
            check(hvc_host_alloc(ctx.get(), s.info.pixel_bytes, reinterpret_cast<void **>(&s.pixels)), "hvc_host_alloc");
            check(hvc_host_alloc(ctx.get(), s.info.coef_count * sizeof(int16_t), reinterpret_cast<void **>(&s.record)), "hvc_host_alloc");
            s.pixel_cap = s.info.pixel_bytes;
            s.coef_cap = s.info.coef_count;
        }
        /* Plane.blit_available into the scans' zero-padded planes (encoder.ml:451-458, 512-520) */
        for (size_t b = 0; b < s.info.pixel_bytes; b++) s.pixels[b] = 0;
        const Plane *planes[3] = {&f.y(), &f.u(), &f.v()};
        for (int i = 0; i < 3; i++) {
            const hvc_component &L = s.info.layout[i];
            for (int y = 0; y < planes[i]->height(); y++)
                for (int x = 0; x < planes[i]->width(); x++)
                    s.pixels[L.plane_offset + (size_t)y * L.stride + x] = planes[i]->data()[(size_t)y * planes[i]->width() + x];
        }
        check(hvc_encode_frames_submit(ctx.get(), index, s.pixels, s.info.pixel_bytes, &s.info.qtabs[0][0], s.info.n_qtabs, s.info.layout,
                                       s.info.n_comp, 1, s.record, s.info.coef_count, HVC_MEM_HOST),
              "hvc_encode_frames_submit");
        s.pending = true;
    }
    retire((int)(frames.size() & 1));
    retire((int)((frames.size() + 1) & 1));
    return files;
}
} // namespace Encoder

namespace Oconv { /* `oyuv convert IN WxH OUT W2xH2` for one planar frame (tools/src/oconv.ml:111-133): the frame to 4:4:4
                     (Planar_444.convert_from_420 / _422), Yuv.crop at an offset into the new size, back to the output format */
inline Frame convert(Ctx &ctx, const Frame &in, int out_width, int out_height, Frame::Chroma_subsampling out_format, int x_off = 0,
                     int y_off = 0) {
    std::vector<uint8_t> src;
    for (const Plane *p : {&in.y(), &in.u(), &in.v()}) src.insert(src.end(), p->data(), p->data() + p->size());
    Frame out = Frame::create(out_format, out_width, out_height);
    std::vector<uint8_t> dst(out.y().size() + out.u().size() + out.v().size());
    check(hvc_yuv_convert(ctx.get(), src.data(), (int)in.chroma_subsampling(), in.width(), in.height(), x_off, y_off, dst.data(),
                          (int)out_format, out_width, out_height, 1, HVC_MEM_HOST),
          "Oconv.convert");
    size_t off = 0;
    for (Plane *p : {&out.y(), &out.u(), &out.v()}) {
        for (size_t k = 0; k < p->size(); k++) p->data()[k] = dst[off + k];
        off += p->size();
    }
    return out;
}
} // namespace Oconv

namespace Ocompare { /* tools/src/ocompare.ml:8-59 */
inline void same_size(const Plane &a, const Plane &b) {
    if (a.width() != b.width() || a.height() != b.height()) throw Error(HVC_E_INVALID_ARG, "Assert_failure ocompare.ml");
}
inline int max_difference(const Plane &a, const Plane &b) {
    same_size(a, b);
    int m = 0;
    check(hvc_compare_planes(a.data(), b.data(), a.size(), &m, nullptr, nullptr), "Ocompare.max_difference");
    return m;
}
inline double psnr(const Plane &a, const Plane &b, double r = 255.0) {
    same_size(a, b);
    uint64_t se = 0;
    check(hvc_compare_planes(a.data(), b.data(), a.size(), nullptr, nullptr, &se), "Ocompare.psnr");
    const double mse = (double)se / ((double)a.width() * (double)a.height());
    return 10.0 * std::log10(r * r / mse);
}
/* Float.to_string as `oyuv compare psnr` prints it: the shorter of %.15g / %.17g that reads back equal, a '.' behind an integer */
inline std::string float_to_string(double x) {
    if (std::isnan(x)) return "NAN";
    if (std::isinf(x)) return x > 0 ? "INF" : "-INF";
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.15g", x);
    if (std::strtod(buf, nullptr) != x) std::snprintf(buf, sizeof buf, "%.17g", x);
    std::string s(buf);
    if (s.find_first_not_of("-0123456789") == std::string::npos) s += ".";
    return s;
}
} // namespace Ocompare

} // namespace hvc_model
#endif /* HVC_MODEL_HPP */
