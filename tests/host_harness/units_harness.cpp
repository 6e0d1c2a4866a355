// units_harness.cpp -- the host half of the GPU reader's restart-interval support, without a GPU: for every file named on
// the command line, hvc::prepare_gpu_decode_to() as hvc_capi_reader.hip calls it (restart intervals honoured), into buffers
// of exactly the size the callers give it (so that AddressSanitizer sees any byte written past them), then the layout
// checked -- slots inside the buffer, 16-byte aligned, in order, zeros behind every interval's bytes -- and printed:
//     <file> ERR <code> | PLAIN ok=<0|1> bytes=<n> | UNITS ok=<0|1> ri=<Ri> ipf=<n> bytes=<n> [<len> <fnv1a of the bytes>]...
// tests/test_host_units.py compares with its own cut of the file.  Test infrastructure, not part of the library.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hvc_hdec.h"
#include "hvc_jpeg.h"

static unsigned long long fnv(const uint8_t *p, size_t n) {
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) {
    const unsigned SB = HVC_HD_SUBSEQ_BITS / 8;
    hvc::tl_honour_restart = true;
    for (int a = 1; a < argc; a++) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) return 2;
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        uint8_t *jpg = (uint8_t *)std::malloc(n ? (size_t)n : 1); // (exactly n bytes: reads past the file show up too)
        if (std::fread(jpg, 1, (size_t)n, f) != (size_t)n) return 2;
        std::fclose(f);
        hvc_jpeg_info info;
        int r = hvc_jpeg_read_header(jpg, (size_t)n, &info);
        if (r) {
            std::printf("%s ERR %d\n", argv[a], r);
            std::free(jpg);
            continue;
        }
        const unsigned ri = hvc::restart_interval_of(jpg, (size_t)n);
        unsigned long long mcus = 0;
        if (info.n_comp > 0 && info.comp[0].hscale > 0 && info.comp[0].vscale > 0)
            mcus = (unsigned long long)(info.comp[0].decoded_width / (8 * info.comp[0].hscale)) *
                   (unsigned long long)(info.comp[0].decoded_height / (8 * info.comp[0].vscale));
        hvc::HdTables *t = new hvc::HdTables;
        bool ok = false;
        size_t got = 0;
        const size_t room = ((size_t)n + SB - 1) / SB * SB;
        if (!ri || mcus <= ri || (mcus + ri - 1) / ri > 4096) {
            uint8_t *dst = (uint8_t *)std::malloc(room ? room : 1);
            r = hvc::prepare_gpu_decode_to(jpg, (size_t)n, &info, *t, dst, room, &got, ok);
            if (r) std::printf("%s ERR %d\n", argv[a], r);
            else std::printf("%s PLAIN ok=%d bytes=%zu\n", argv[a], (int)ok, got);
            std::free(dst);
        } else {
            const unsigned ipf = (unsigned)((mcus + ri - 1) / ri);
            const size_t cap = room + (size_t)ipf * (2 * SB + 16);
            uint8_t *dst = (uint8_t *)std::malloc(cap);
            std::memset(dst, 0xA5, cap);
            std::vector<unsigned> off(ipf, 0xffffffffu), len(ipf, 0xffffffffu);
            hvc::RstUnits ru{ri, ipf, off.data(), len.data()};
            r = hvc::prepare_gpu_decode_to(jpg, (size_t)n, &info, *t, dst, cap, &got, ok, &ru);
            if (r) {
                std::printf("%s ERR %d\n", argv[a], r);
            } else {
                std::printf("%s UNITS ok=%d ri=%u ipf=%u bytes=%zu", argv[a], (int)ok, ri, ipf, got);
                if (ok) {
                    size_t at = 0, sum = 0;
                    for (unsigned k = 0; k < ipf; k++) {
                        const size_t slot = hvc::hd_unit_slot(len[k]);
                        if (off[k] != at || off[k] % 16 || at + slot > cap) { std::printf(" BAD-LAYOUT\n"); return 1; }
                        for (size_t i = len[k]; i < slot; i++)
                            if (dst[at + i]) { std::printf(" NOT-ZERO\n"); return 1; }
                        std::printf(" %u %016llx", len[k], fnv(dst + at, len[k]));
                        sum += len[k];
                        at += slot;
                    }
                    if (sum != got) { std::printf(" BAD-SUM\n"); return 1; }
                }
                std::printf("\n");
            }
            std::free(dst);
        }
        delete t;
        std::free(jpg);
    }
    return 0;
}
