// Minimal BGZF (blocked gzip) codec on zlib, multi-threaded per block.
// Replaces the htslib BGZF reader/writer the reference reaches through rust-htslib
// (src/deduplicate_sam.rs:78-93, 357-365, 402-404); "next" row N1 of SURVEY.md 8f.
// Parity is defined on the DEcompressed stream: compressed bytes depend on the deflate
// backend and level of htslib, which are not reproducible here.
#pragma once

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace umi {
namespace bgzf {

struct IoError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline std::vector<uint8_t> read_file(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw IoError("Invalid input path: " + path); // deduplicate_sam.rs:78
    std::fseek(f, 0, SEEK_END);
    long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)sz);
    if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) {
        std::fclose(f);
        throw IoError("short read on " + path);
    }
    std::fclose(f);
    return buf;
}

struct BlockRef {
    size_t in_off;   // start of the deflate payload
    uint32_t in_len; // payload bytes
    uint32_t out_len;
    size_t out_off;
};

inline void parallel_for(size_t n, unsigned threads, const std::function<void(size_t)> &fn)
{
    if (threads <= 1 || n <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    std::atomic<bool> failed{false};
    std::string err;
    for (unsigned t = 0; t < threads; t++)
        pool.emplace_back([&] {
            try {
                for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
            } catch (const std::exception &e) {
                if (!failed.exchange(true)) err = e.what();
            }
        });
    for (auto &th : pool) th.join();
    if (failed) throw IoError(err);
}

// Whole-file BGZF decompression: scan the block headers (BSIZE in the BC subfield),
// read ISIZE from each trailer, inflate all blocks in parallel into one buffer.
inline std::vector<uint8_t> decompress(const std::vector<uint8_t> &in, unsigned threads)
{
    std::vector<BlockRef> blocks;
    size_t off = 0, total = 0;
    while (off < in.size()) {
        if (in.size() - off < 18 || in[off] != 0x1f || in[off + 1] != 0x8b || in[off + 2] != 8 || !(in[off + 3] & 4))
            throw IoError("not a BGZF block at offset " + std::to_string(off));
        const uint32_t xlen = in[off + 10] | (in[off + 11] << 8);
        size_t x = off + 12;
        const size_t xend = x + xlen;
        if (xend > in.size()) throw IoError("truncated BGZF block");
        uint32_t bsize = 0;
        while (x + 4 <= xend) {
            const uint32_t slen = in[x + 2] | (in[x + 3] << 8);
            if (in[x] == 'B' && in[x + 1] == 'C' && slen == 2 && x + 6 <= xend)
                bsize = (in[x + 4] | (in[x + 5] << 8)) + 1u;
            x += 4 + (size_t)slen;
        }
        // a block holds its header, the extra field and the 8-byte trailer (CRC32, ISIZE) at least
        if (!bsize || bsize > in.size() - off || (size_t)bsize < (xend - off) + 8)
            throw IoError("truncated BGZF block");
        const size_t tail = off + bsize - 8;
        const uint32_t isize = in[tail + 4] | (in[tail + 5] << 8) | (in[tail + 6] << 16) | ((uint32_t)in[tail + 7] << 24);
        // (a BGZF block inflates to at most 64 KiB; a larger ISIZE is a corrupt trailer, and the
        // sum would size the output buffer)
        if (isize > 0x10000u) throw IoError("Failed to parse record: corrupt BGZF block");
        blocks.push_back({xend, (uint32_t)(tail - xend), isize, total});
        total += isize;
        off += bsize;
    }
    std::vector<uint8_t> out(total);
    parallel_for(blocks.size(), threads, [&](size_t i) {
        const BlockRef &b = blocks[i];
        if (b.out_len == 0) return;
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) throw IoError("inflateInit2 failed");
        zs.next_in = const_cast<Bytef *>(in.data() + b.in_off);
        zs.avail_in = b.in_len;
        zs.next_out = out.data() + b.out_off;
        zs.avail_out = b.out_len;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.avail_out != 0) throw IoError("Failed to parse record: corrupt BGZF block");
    });
    return out;
}

// BGZF compression of a whole buffer: 0xff00-byte payloads, parallel deflate, EOF block.
inline void compress_to_file(const std::string &path, const uint8_t *data, size_t len, unsigned threads, int level = 6)
{
    constexpr size_t PAYLOAD = 0xff00;
    const size_t nblk = (len + PAYLOAD - 1) / PAYLOAD;
    std::vector<std::vector<uint8_t>> out(nblk);
    parallel_for(nblk, threads, [&](size_t i) {
        const size_t o = i * PAYLOAD, n = std::min(PAYLOAD, len - o);
        std::vector<uint8_t> &b = out[i];
        b.resize(18 + compressBound((uLong)n) + 8);
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw IoError("deflateInit2 failed");
        zs.next_in = const_cast<Bytef *>(data + o);
        zs.avail_in = (uInt)n;
        zs.next_out = b.data() + 18;
        zs.avail_out = (uInt)(b.size() - 18 - 8);
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) throw IoError("deflate failed");
        const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
        if (bsize > 0xffff) throw IoError("BGZF block overflow");
        const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                 (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
        std::memcpy(b.data(), hdr, 18);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data + o, (uInt)n);
        uint8_t *t = b.data() + 18 + clen;
        for (int k = 0; k < 4; k++) t[k] = (uint8_t)(crc >> (8 * k));
        for (int k = 0; k < 4; k++) t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k));
        b.resize(18 + clen + 8);
    });
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw IoError("cannot open output " + path);
    for (auto &b : out)
        if (std::fwrite(b.data(), 1, b.size(), f) != b.size()) {
            std::fclose(f);
            throw IoError("Failed to write the record");
        }
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0};
    std::fwrite(eof, 1, sizeof(eof), f);
    std::fclose(f);
}

} // namespace bgzf
} // namespace umi
