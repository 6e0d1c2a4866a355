// hvc_huff.h -- parameter block of the GPU Huffman coder (internal).
#ifndef HVC_HUFF_H
#define HVC_HUFF_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace hvc {

struct HuffComp {
    int bw, bh, nblk;  // the component's coefficient plane in blocks
    int tile0;         // first workgroup tile (256 blocks) of this component inside a frame
    int h, v;          // sampling factors = blocks per MCU in x / y
    int mcu_base;      // index of this component's first block inside an MCU
    int table;         // 0 = luma tables, 1 = chroma tables (Parameters.c4xx, encoder.ml:306-349)
    size_t coef_off;   // int16 elements from the frame's coefficient record
};

struct HuffParams {
    const int16_t *coefs;
    size_t coef_fs;        // int16 elements between frames
    int n_frames, tiles_per_frame;
    int mbs_wide, mbs_high, blocks_per_mcu;
    unsigned blocks_per_frame;   // coded blocks = mbs_wide * mbs_high * blocks_per_mcu
    HuffComp comp[3];
    const unsigned *tables;      // device: [2][16 dc + 256 ac], (code << 5) | length
    unsigned *lens;              // [n_frames][blocks_per_frame]: bit lengths in scan order, then offsets
    unsigned *frame_bits, *frame_bytes, *frame_pieces, *frame_ff;   // [n_frames]
    unsigned *bitbuf;            // [n_frames][bitbuf_words]: unstuffed segments, big-endian bit order
    size_t bitbuf_words;         // per frame, a multiple of 16 (64-byte pieces)
    unsigned *ff;                // [n_frames][ff_stride]: 0xFF bytes per piece, then offsets
    size_t ff_stride;
    unsigned long long *out_offsets;  // [n_frames + 1]
    uint8_t *out;
    unsigned long long out_cap;
    unsigned *status;            // bit 0: value without a code (HVC_E_RANGE); bit 2: out too small
};

hipError_t launch_huffman_encode(const HuffParams &P, hipStream_t s);

// hvc_entropy.cpp
void default_enc_tables(uint32_t (*out)[16 + 256]);

} // namespace hvc

struct hvc_jpeg_info;
namespace hvc {
void jpeg_header_bytes(const ::hvc_jpeg_info *info, std::vector<uint8_t> &o);

}
#endif
