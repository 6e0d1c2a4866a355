// model_tests.cpp -- the reference's own tests for this path, written against include/hvc_model.hpp (the C++ mirror of the
// reference's host interface over the C ABI) so that they read like the originals:
//   test_chen_dct          jpeg/model/test/test_chen_dct.ml:47-87      (G1: the explicit block, its Chen fDCT / 4 rounded, the IDCT of that)
//   test_quant_tables      jpeg/model/test/test_quant_tables.ml:4-62   (G5: scale luma q)
//   model-encode-and-decode  jpeg/test/model-encode-and-decode.t:7-72  (the cram session, printed as `oyuv compare psnr` prints it)
//   test-nonstandard-sizes   jpeg/test/test-nonstandard-sizes.t:3-15     (52 x 44 through `oyuv convert`, padding and crop)
//   mini.jpg               jpeg/test_data: Encoder.encode_420 ~quality:75 of mini64x64.420 is that file (G3)
// plus the error behaviour of the interfaces (what raises in the model throws here).
//   model_tests host <golden dir> <fixtures.txt>      no GPU needed: Plane, Frame, Quant_tables, Header.decode
//   model_tests gpu  <golden dir> <fixtures.txt>      everything through the GPU path
// fixtures.txt (written by tests/test_cpp_model.py from tests/golden/*.json): "g1_input", "g1_fdct", "g1_idct" + 64 ints each,
// "luma <q>" + 64 ints per pinned quality.  Prints one line per check; the Python test compares the output.
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "hvc_model.hpp"

using namespace hvc_model;

static std::string read_all(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot read " + path);
    std::ostringstream s;
    s << f.rdbuf();
    return s.str();
}

static std::map<std::string, std::vector<int>> fixtures(const std::string &path) {
    std::map<std::string, std::vector<int>> m;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream s(line);
        std::string key;
        s >> key;
        if (key == "luma") {
            std::string q;
            s >> q;
            key += " " + q;
        }
        int v;
        while (s >> v) m[key].push_back(v);
    }
    return m;
}

static const int ZZ_FORWARD_OF_RASTER[64] = { // Zigzag.forward (zigzag.ml:71-137): raster position -> zig-zag position
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

template <class F>
static bool raises(F f, int code) {
    try {
        f();
    } catch (const Error &e) {
        return e.code == code;
    }
    return false;
}

static Frame input_yuv(const std::string &path, Frame::Chroma_subsampling c, int w, int h) { // jpeg/bin/model.ml:70-82
    Frame f = Frame::create(c, w, h);
    std::ifstream in(path, std::ios::binary);
    f.input(in);
    return f;
}

static int host_tests(const std::string &golden, std::map<std::string, std::vector<int>> &fx) {
    // test_quant_tables.ml: scale luma q for the pinned qualities
    for (auto &kv : fx)
        if (kv.first.rfind("luma ", 0) == 0) {
            const int q = std::atoi(kv.first.c_str() + 5);
            const auto t = Quant_tables::scale(Quant_tables::luma, q);
            bool same = kv.second.size() == 64;
            for (int i = 0; same && i < 64; i++) same = t[i] == kv.second[i];
            std::cout << "quant_tables scale luma " << q << (same ? " ok" : " MISMATCH") << "\n";
        }
    // Plane / Frame behave as plane.ml / frame.ml
    Plane p = Plane::create(5, 3);
    p.set(4, 2, 77);
    bool ok = p.at(4, 2) == 77 && p.at(0, 0) == 0 && raises([&] { p.at(5, 0); }, HVC_E_INVALID_ARG) && raises([&] { p.set(0, 3, 1); }, HVC_E_INVALID_ARG);
    Plane big = Plane::create(8, 8);
    Plane::blit_available(p, big);
    ok = ok && big.at(4, 2) == 77 && big.at(7, 7) == 0;
    std::cout << "plane " << (ok ? "ok" : "MISMATCH") << "\n";
    ok = Frame::of_planes(Plane::create(64, 64), Plane::create(32, 32), Plane::create(32, 32)).chroma_subsampling() == Frame::Chroma_subsampling::C420 &&
         Frame::of_planes(Plane::create(64, 64), Plane::create(32, 64), Plane::create(32, 64)).chroma_subsampling() == Frame::Chroma_subsampling::C422 &&
         Frame::of_planes(Plane::create(64, 64), Plane::create(64, 64), Plane::create(64, 64)).chroma_subsampling() == Frame::Chroma_subsampling::C444 &&
         Frame::of_planes(Plane::create(53, 45), Plane::create(26, 22), Plane::create(26, 22)).chroma_subsampling() == Frame::Chroma_subsampling::C420 &&
         raises([] { Frame::of_planes(Plane::create(64, 64), Plane::create(32, 32), Plane::create(32, 16)); }, HVC_E_BAD_JPEG) &&
         raises([] { Frame::of_planes(Plane::create(64, 64), Plane::create(16, 64), Plane::create(16, 64)); }, HVC_E_BAD_JPEG);
    std::cout << "frame of_planes " << (ok ? "ok" : "MISMATCH") << "\n";
    // Header.decode of the reference's files; what the model raises on
    const Decoder::Header m = Decoder::Header::decode(read_all(golden + "/mini.jpg")), mo = Decoder::Header::decode(read_all(golden + "/Mouse480.jpg"));
    ok = m.width() == 64 && m.height() == 64 && m.components() == 3 && mo.width() == 480 && mo.height() == 320 && mo.components() == 3 &&
         raises([] { Decoder::Header::decode(std::string("\xff\xd8\xff\xc2\x00\x04\x00\x00", 8)); }, HVC_E_UNSUPPORTED_MARKER) &&
         raises([] { Decoder::Header::decode("not a jpeg"); }, HVC_E_BAD_JPEG);
    std::cout << "header decode " << (ok ? "ok" : "MISMATCH") << "\n";
    ok = Ocompare::float_to_string(46.76864691904693) == "46.76864691904693" && Ocompare::float_to_string(46.760132097139362) == "46.760132097139362" &&
         Ocompare::float_to_string(3.0) == "3." && Ocompare::float_to_string(INFINITY) == "INF";
    std::cout << "float to_string " << (ok ? "ok" : "MISMATCH") << "\n";
    return 0;
}

static int gpu_tests(const std::string &golden, std::map<std::string, std::vector<int>> &fx) {
    Ctx ctx(0);
    // ---- test_chen_dct.ml:47-87
    std::array<uint16_t, 64> ones;
    ones.fill(1);
    const std::vector<int> &input = fx["g1_input"], &fdct = fx["g1_fdct"], &idct = fx["g1_idct"];
    std::array<uint8_t, 64> px;
    for (int i = 0; i < 64; i++) px[i] = (uint8_t)(input[i] + 128);   // level_shifted_input_block subtracts the 128 again
    const auto q = Encoder::quant_of_pixels(ctx, px, ones);           // table of ones: (x + 2) / 4 for x > 0, (x - 2) / 4 otherwise
    bool ok = true;
    for (int i = 0; i < 64; i++) ok = ok && q[ZZ_FORWARD_OF_RASTER[i]] == fdct[i];
    std::cout << "chen forward_8x8 " << (ok ? "ok" : "MISMATCH") << "\n";
    std::array<int16_t, 64> zz;
    for (int i = 0; i < 64; i++) zz[ZZ_FORWARD_OF_RASTER[i]] = (int16_t)fdct[i];
    const auto recon = Decoder::recon_of_coefs(ctx, zz, ones);         // = clip (inverse_8x8 fdct) + 128 (decoder.ml:213-224)
    ok = true;
    for (int i = 0; i < 64; i++) ok = ok && recon[i] == std::min(127, std::max(-128, idct[i])) + 128;
    std::cout << "chen inverse_8x8 " << (ok ? "ok" : "MISMATCH") << "\n";
    // ---- model-encode-and-decode.t: encode, decode, PSNR against the source as `oyuv compare psnr yuv` prints it
    struct Case { const char *file; Frame::Chroma_subsampling c; int quality; };
    const Case cases[] = {{"mini64x64.420", Frame::Chroma_subsampling::C420, 95}, {"mini64x64.420", Frame::Chroma_subsampling::C420, 50},
                          {"mini64x64.420", Frame::Chroma_subsampling::C420, 30}, {"mini64x64.422", Frame::Chroma_subsampling::C422, 75},
                          {"mini64x64.444", Frame::Chroma_subsampling::C444, 75}};
    for (const Case &k : cases) {
        const Frame src = input_yuv(golden + "/" + k.file, k.c, 64, 64);
        const std::string jpg = k.c == Frame::Chroma_subsampling::C420   ? Encoder::encode_420(ctx, src, k.quality)
                                : k.c == Frame::Chroma_subsampling::C422 ? Encoder::encode_422(ctx, src, k.quality)
                                                                         : Encoder::encode_444(ctx, src, k.quality);
        const Frame out = Decoder::decode_a_frame(ctx, jpg);
        std::cout << "$ model encode frame " << k.file << " 64x64 -quality " << k.quality << "; model decode frame; oyuv compare psnr\n";
        std::cout << Ocompare::float_to_string(Ocompare::psnr(src.y(), out.y())) << "\n"
                  << Ocompare::float_to_string(Ocompare::psnr(src.u(), out.u())) << "\n"
                  << Ocompare::float_to_string(Ocompare::psnr(src.v(), out.v())) << "\n";
    }
    // ---- test-nonstandard-sizes.t:3-15: 64x64 -> 52x44 by `oyuv convert` (twice, as the session does), encode q95, decode, PSNR
    {
        const Frame full = input_yuv(golden + "/mini64x64.420", Frame::Chroma_subsampling::C420, 64, 64);
        Frame src = Oconv::convert(ctx, full, 52, 44, Frame::Chroma_subsampling::C420);
        src = Oconv::convert(ctx, full, 52, 44, Frame::Chroma_subsampling::C420);
        const Frame out = Decoder::decode_a_frame(ctx, Encoder::encode_420(ctx, src, 95));
        std::cout << "$ oyuv convert 64x64 -> 52x44; model encode frame -quality 95; model decode frame; oyuv compare psnr\n";
        std::cout << Ocompare::float_to_string(Ocompare::psnr(src.y(), out.y())) << "\n"
                  << Ocompare::float_to_string(Ocompare::psnr(src.u(), out.u())) << "\n"
                  << Ocompare::float_to_string(Ocompare::psnr(src.v(), out.v())) << "\n";
    }
    // ---- mini.jpg is the model encoder's own output
    const std::string mini = read_all(golden + "/mini.jpg");
    ok = Encoder::encode_420(ctx, input_yuv(golden + "/mini64x64.420", Frame::Chroma_subsampling::C420, 64, 64), 75) == mini;
    std::cout << "encode_420 q75 = mini.jpg " << (ok ? "ok" : "MISMATCH") << "\n";
    // ---- init / decode / get_decoded_planes / get_yuv_frame step by step on Mouse480 (480 x 320, 4:2:0)
    const std::string mouse = read_all(golden + "/Mouse480.jpg");
    Decoder::t d = Decoder::init(ctx, Decoder::Header::decode(mouse), mouse);
    ok = raises([&] { d.get_yuv_frame(); }, HVC_E_INVALID_ARG);
    d.decode();
    const auto planes = d.get_decoded_planes();
    const Frame fr = d.get_yuv_frame();
    ok = ok && planes.size() == 3 && planes[0].width() == 480 && planes[0].height() == 320 && planes[1].width() == 240 && fr.width() == 480 &&
         fr.chroma_subsampling() == Frame::Chroma_subsampling::C420 && Ocompare::max_difference(fr.y(), planes[0]) == 0;
    std::cout << "decoder init / decode / get_yuv_frame " << (ok ? "ok" : "MISMATCH") << "\n";
    // ---- Decoder.decode_frames_gpu's twin: a list of files of several geometries through the asynchronous seam = List.map decode_a_frame
    {
        std::vector<std::string> files;
        for (int k = 0; k < 7; k++) files.push_back(k % 3 == 1 ? mouse : mini);   // 64 x 64 and 480 x 320 in turn: the slots' records grow
        files.push_back(Encoder::encode_444(ctx, input_yuv(golden + "/mini64x64.444", Frame::Chroma_subsampling::C444, 64, 64), 60));
        const std::vector<Frame> got = Decoder::decode_frames(ctx, files);
        ok = got.size() == files.size();
        for (size_t k = 0; ok && k < files.size(); k++) {
            const Frame want = Decoder::decode_a_frame(ctx, files[k]);
            ok = want.width() == got[k].width() && want.chroma_subsampling() == got[k].chroma_subsampling() &&
                 Ocompare::max_difference(want.y(), got[k].y()) == 0 && Ocompare::max_difference(want.u(), got[k].u()) == 0 &&
                 Ocompare::max_difference(want.v(), got[k].v()) == 0;
        }
        ok = ok && Decoder::decode_frames(ctx, {}).empty() && Decoder::decode_frames(ctx, {mini}).size() == 1;
        // a bad file in the middle: the exception leaves nothing in flight (the context is usable right after)
        std::string cut_mid = mini.substr(0, 300);   // (header cut inside a table segment)
        ok = ok && raises([&] { Decoder::decode_frames(ctx, {mini, cut_mid, mini}); }, HVC_E_BAD_JPEG) && Decoder::decode_frames(ctx, {mini, mini, mini}).size() == 3;
        std::cout << "decode_frames through the asynchronous seam " << (ok ? "ok" : "MISMATCH") << "\n";
    }
    // ---- Encoder.encode_frames_gpu's twin: frames of three samplings and two sizes through the seam = List.map encode_4xx, byte for byte
    {
        std::vector<Frame> fr;
        const Frame f420 = input_yuv(golden + "/mini64x64.420", Frame::Chroma_subsampling::C420, 64, 64);
        fr.push_back(f420);
        fr.push_back(input_yuv(golden + "/mini64x64.444", Frame::Chroma_subsampling::C444, 64, 64));
        fr.push_back(Oconv::convert(ctx, f420, 52, 44, Frame::Chroma_subsampling::C420));
        fr.push_back(input_yuv(golden + "/mini64x64.422", Frame::Chroma_subsampling::C422, 64, 64));
        fr.push_back(f420);
        const std::vector<std::string> got = Encoder::encode_frames(ctx, fr, 75);
        ok = got.size() == fr.size() && got[0] == mini && got[4] == mini;   // (mini.jpg: the model encoder's own file, G3)
        for (size_t k = 0; ok && k < fr.size(); k++) ok = got[k] == Encoder::encode(ctx, fr[k], 75, fr[k].chroma_subsampling());
        ok = ok && Encoder::encode_frames(ctx, {}, 75).empty();
        std::cout << "encode_frames through the asynchronous seam " << (ok ? "ok" : "MISMATCH") << "\n";
    }
    // ---- what the model raises on
    std::string cut = mini.substr(0, 300);   // header cut inside a table segment
    ok = raises([&] { Decoder::decode_a_frame(ctx, cut); }, HVC_E_BAD_JPEG) && raises([&] { Decoder::decode_a_frame(ctx, "garbage"); }, HVC_E_BAD_JPEG) &&
         raises([&] { Encoder::encode_422(ctx, input_yuv(golden + "/mini64x64.420", Frame::Chroma_subsampling::C420, 64, 64), 75); }, HVC_E_INVALID_ARG);
    std::cout << "raises " << (ok ? "ok" : "MISMATCH") << "\n";
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::cerr << "usage: model_tests host|gpu <golden dir> <fixtures.txt>\n";
        return 2;
    }
    try {
        auto fx = fixtures(argv[3]);
        return std::string(argv[1]) == "gpu" ? gpu_tests(argv[2], fx) : host_tests(argv[2], fx);
    } catch (const std::exception &e) {
        std::cout << "EXCEPTION " << e.what() << "\n";
        return 1;
    }
}
