// BAM container view over a decompressed BGZF stream: header passthrough and zero-copy
// record accessors.  Stands in for the rust-htslib Reader/Record calls of
// src/deduplicate_sam.rs:78-144 and src/utils/read.rs:56-111 ("next" row N1/N2).
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bgzf.hpp"

namespace umi {
namespace bam {

struct FormatError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline int32_t rd_i32(const uint8_t *p)
{
    int32_t v;
    std::memcpy(&v, p, 4);
    return v;
}
inline uint16_t rd_u16(const uint8_t *p)
{
    uint16_t v;
    std::memcpy(&v, p, 2);
    return v;
}

// One alignment record inside the decompressed buffer (block_size prefix included in
// [begin, end) so that a survivor can be copied out verbatim).
struct Record {
    const uint8_t *begin = nullptr;
    const uint8_t *end = nullptr;
    const uint8_t *body() const { return begin + 4; }
    int32_t tid() const { return rd_i32(body()); }
    int32_t pos() const { return rd_i32(body() + 4); }
    uint8_t l_read_name() const { return body()[8]; }
    uint8_t mapq() const { return body()[9]; }
    uint16_t n_cigar() const { return rd_u16(body() + 12); }
    uint16_t flag() const { return rd_u16(body() + 14); }
    int32_t l_seq() const { return rd_i32(body() + 16); }
    int32_t mtid() const { return rd_i32(body() + 20); }
    int32_t mpos() const { return rd_i32(body() + 24); }
    int32_t tlen() const { return rd_i32(body() + 28); }
    const uint8_t *qname() const { return body() + 32; }
    size_t qname_len() const { return l_read_name() ? (size_t)l_read_name() - 1 : 0; } // without the NUL
    const uint8_t *cigar() const { return qname() + l_read_name(); }
    const uint8_t *seq() const { return cigar() + 4 * (size_t)n_cigar(); }
    const uint8_t *qual() const { return seq() + ((size_t)l_seq() + 1) / 2; }
    bool is_unmapped() const { return flag() & 0x4; }
    bool is_reverse() const { return flag() & 0x10; }
    bool is_paired() const { return flag() & 0x1; }
    bool is_last_in_template() const { return flag() & 0x80; }
    bool is_mate_unmapped() const { return flag() & 0x8; }

    // utils::get_unclipped_pos (src/utils/mod.rs:96-104) over rust-htslib 0.49's
    // CigarStringView::{pos,end_pos,leading_*,trailing_*}.  That crate's source is not in the
    // reference tree: restated from its documented behaviour (clips counted only at the very
    // ends: H, or S next to a terminal H) -- PARITY UNPINNED.
    int64_t unclipped_pos() const
    {
        const uint8_t *c = cigar();
        const int n = n_cigar();
        auto op = [&](int i) { uint32_t v; std::memcpy(&v, c + 4 * i, 4); return v & 0xf; };
        auto len = [&](int i) { uint32_t v; std::memcpy(&v, c + 4 * i, 4); return (int64_t)(v >> 4); };
        if (is_reverse()) {
            int64_t end_pos = pos();
            for (int i = 0; i < n; i++) {
                const uint32_t o = op(i);
                if (o == 0 || o == 2 || o == 3 || o == 7 || o == 8) end_pos += len(i); // M D N = X
            }
            int64_t soft = 0, hard = 0;
            if (n > 0) {
                if (op(n - 1) == 4) soft = len(n - 1);
                else if (op(n - 1) == 5) {
                    hard = len(n - 1);
                    if (n > 1 && op(n - 2) == 4) soft = len(n - 2);
                }
            }
            return end_pos - 1 + soft + hard;
        }
        int64_t soft = 0, hard = 0;
        if (n > 0) {
            if (op(0) == 4) soft = len(0);
            else if (op(0) == 5) {
                hard = len(0);
                if (n > 1 && op(1) == 4) soft = len(1);
            }
        }
        return (int64_t)pos() - soft - hard;
    }

    // UcSAMRead::new (src/utils/read.rs:56-63): (sum(qual as f32) / seq_len as f32) as i32
    int32_t avg_qual() const
    {
        volatile float sum = 0.0f;
        const uint8_t *q = qual();
        const int32_t n = l_seq();
        if (n < 65536) {
            // every partial sum is an integer below 2^24, which f32 holds exactly: the running f32
            // sum of the reference equals the integer sum (and this loop vectorises)
            uint32_t isum = 0;
            for (int32_t i = 0; i < n; i++) isum += q[i];
            sum = (float)isum;
        } else {
            for (int32_t i = 0; i < n; i++) sum = sum + (float)q[i];
        }
        volatile float avg = sum / (float)n;
        float v = avg;
        if (v != v) return 0;
        if (v >= 2147483648.0f) return INT32_MAX;
        if (v <= -2147483648.0f) return INT32_MIN;
        return (int32_t)v;
    }
};

struct File {
    umi::bgzf::Bytes data;     // whole decompressed stream
    size_t header_len = 0;     // magic .. end of the reference list: copied verbatim to the output
                               // (Header::from_template, deduplicate_sam.rs:357-362)
    int32_t n_ref = 0;
    std::vector<Record> records;

    // Every length field of the file is checked against the bytes that are there before anything
    // is indexed by it: a truncated or corrupt file ends in FormatError ("Failed to parse record",
    // the message of the reference's htslib read loop, deduplicate_sam.rs:93-95), never in a read
    // outside the buffer.
    void parse()
    {
        parse_behind([](size_t) {});
    }
    // need(upto): called before any byte below `upto` is looked at (the inflate may still be running
    // ahead of the parse: umi::bgzf::Inflater::wait)
    template <class Need> void parse_behind(Need need)
    {
        const uint8_t *p = data.data();
        const size_t size = data.size();
        need(12);
        if (size < 12 || std::memcmp(p, "BAM\1", 4) != 0) throw FormatError("Invalid input path: not a BAM file");
        const int32_t l_text = rd_i32(p + 4);
        if (l_text < 0 || (size_t)l_text > size - 12) throw FormatError("truncated BAM header");
        size_t q = 8 + (size_t)l_text; // offsets, not pointers: nothing is formed past the buffer
        need(q + 4);
        n_ref = rd_i32(p + q);
        q += 4;
        if (n_ref < 0 || (size_t)n_ref > (size - q) / 8) throw FormatError("truncated BAM header");
        for (int32_t r = 0; r < n_ref; r++) {
            if (size - q < 4) throw FormatError("truncated BAM header");
            need(q + 4);
            const int32_t l_name = rd_i32(p + q);
            q += 4;
            if (l_name < 0 || size - q < 4 || (size_t)l_name > size - q - 4) throw FormatError("truncated BAM header");
            q += (size_t)l_name + 4;
        }
        header_len = q;
        records.reserve(size / 96); // (a guess; it grows if the records are shorter)
        size_t have = 0; // bytes known to be there
        while (q < size) {
            if (size - q < 4) throw FormatError("Failed to parse record");
            if (q + 36 > have) { // (a stretch at a time: need() may have to look at the inflate's flags)
                have = std::min(size, q + (1u << 20));
                need(have);
            }
            const int32_t bs = rd_i32(p + q);
            if (bs < 32 || (size_t)bs > size - q - 4) throw FormatError("Failed to parse record");
            const Record rec{p + q, p + q + 4 + (size_t)bs};
            // the variable-length fields must fit the record: qname, cigar, packed seq, qual
            const int32_t l_seq = rec.l_seq();
            if (l_seq < 0) throw FormatError("Failed to parse record");
            const uint64_t need_bytes = 32ull + rec.l_read_name() + 4ull * rec.n_cigar() +
                                        ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
            if (need_bytes > (uint64_t)bs) throw FormatError("Failed to parse record");
            records.push_back(rec);
            q += 4 + (size_t)bs;
        }
        need(size);
    }
};

} // namespace bam
} // namespace umi
