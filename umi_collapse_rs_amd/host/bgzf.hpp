// Minimal BGZF (blocked gzip) codec on zlib, multi-threaded per block.
// Replaces the htslib BGZF reader/writer the reference reaches through rust-htslib
// (src/deduplicate_sam.rs:78-93, 357-365, 402-404); "next" row N1 of SURVEY.md 8f.
// Parity is defined on the DEcompressed stream: compressed bytes depend on the deflate
// backend and level of htslib, which are not reproducible here.
#pragma once

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace umi {
namespace bgzf {

struct IoError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Bytes whose storage is not zeroed when it is sized: a file of 2 M reads is 0.4 GB inflated, and
// std::vector<uint8_t>(n) would write all of it once, on one thread, before the inflate threads
// write it again (0.08 s of a run of 0.4 s).
// Large blocks are 2 MB-aligned and marked for transparent huge pages: first touch and the unmapping
// at exit are per page, and 0.4 GB is 100,000 small ones.
template <class T> struct default_init_allocator {
    using value_type = T;
    default_init_allocator() = default;
    template <class U> default_init_allocator(const default_init_allocator<U> &) noexcept {}
    template <class U> struct rebind {
        using other = default_init_allocator<U>;
    };
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T), huge = 2u << 20;
        void *p = nullptr;
        if (bytes >= 2 * huge) {
            if (posix_memalign(&p, huge, (bytes + huge - 1) / huge * huge) != 0) throw std::bad_alloc();
            (void)madvise(p, (bytes + huge - 1) / huge * huge, MADV_HUGEPAGE);
        } else if (!(p = std::malloc(bytes ? bytes : 1))) {
            throw std::bad_alloc();
        }
        return (T *)p;
    }
    void deallocate(T *p, size_t) noexcept { std::free(p); }
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const default_init_allocator<U> &) const noexcept { return true; }
    template <class U> bool operator!=(const default_init_allocator<U> &) const noexcept { return false; }
};
using Bytes = std::vector<uint8_t, default_init_allocator<uint8_t>>;

struct BlockRef {
    size_t in_off;   // start of the deflate payload
    uint32_t in_len; // payload bytes
    uint32_t out_len;
    size_t out_off;
};

inline void parallel_for(size_t n, unsigned threads, const std::function<void(size_t)> &fn);

// the whole file, read by `threads` readers at once (pread on slices)
inline Bytes read_file(const std::string &path, unsigned threads = 1)
{
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw IoError("Invalid input path: " + path); // deduplicate_sam.rs:78
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        throw IoError("Invalid input path: " + path);
    }
    const size_t sz = (size_t)st.st_size, slice = 8u << 20;
    Bytes buf(sz);
    try {
        parallel_for((sz + slice - 1) / slice, threads, [&](size_t i) {
            size_t off = i * slice;
            const size_t end = std::min(sz, off + slice);
            while (off < end) {
                const ssize_t got = ::pread(fd, buf.data() + off, end - off, (off_t)off);
                if (got <= 0) throw IoError("short read on " + path);
                off += (size_t)got;
            }
        });
    } catch (...) {
        ::close(fd);
        throw;
    }
    ::close(fd);
    return buf;
}

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

// Whole-file BGZF decompression: scan the block headers (BSIZE in the BC subfield), read ISIZE from
// each trailer, inflate all blocks in parallel into one buffer -- a run of blocks per task, one
// inflate state per run.  The caller may walk the buffer behind the inflate threads: wait(upto)
// returns once every byte before `upto` is there (the BAM parse, sequential by nature, then costs
// no time of its own).
class Inflater
{
  public:
    Bytes out;

    Inflater(const Bytes &in, unsigned threads) : in_(in)
    {
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
            blocks_.push_back({xend, (uint32_t)(tail - xend), isize, total});
            total += isize;
            off += bsize;
        }
        out.resize(total); // (not zeroed: the inflate threads are the first to touch it)
        base_ = out.data(); // (the caller may swap `out` away while the threads run: they write through this)
        total_ = total;
        n_runs_ = (blocks_.size() + RUN - 1) / RUN;
        done_.reset(new std::atomic<int>[n_runs_ + 1]);
        for (size_t r = 0; r <= n_runs_; r++) done_[r].store(0);
        driver_ = std::thread([this, threads] {
            try {
                parallel_for(n_runs_, threads, [this](size_t r) { inflate_run(r); });
            } catch (const std::exception &e) {
                error_ = e.what();
                failed_.store(true, std::memory_order_release);
            }
        });
    }
    ~Inflater()
    {
        if (driver_.joinable()) driver_.join();
    }
    Inflater(const Inflater &) = delete;
    Inflater &operator=(const Inflater &) = delete;

    // every byte before `upto` (clamped to the size) is inflated when this returns
    void wait(size_t upto)
    {
        upto = std::min(upto, total_);
        while (ready_ < upto) {
            if (failed_.load(std::memory_order_acquire)) finish();
            if (cursor_ < n_runs_ && done_[cursor_].load(std::memory_order_acquire)) {
                cursor_++;
                const size_t next_block = std::min(blocks_.size(), cursor_ * RUN);
                ready_ = next_block < blocks_.size() ? blocks_[next_block].out_off : total_;
            } else {
                std::this_thread::yield();
            }
        }
    }
    // all of it; rethrows what an inflate thread ran into
    void finish()
    {
        if (driver_.joinable()) driver_.join();
        if (failed_) throw IoError(error_);
        ready_ = total_;
    }

  private:
    static constexpr size_t RUN = 16;
    void inflate_run(size_t r)
    {
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) throw IoError("inflateInit2 failed");
        for (size_t i = r * RUN; i < std::min(blocks_.size(), (r + 1) * RUN); i++) {
            const BlockRef &b = blocks_[i];
            if (b.out_len == 0) continue;
            if (i != r * RUN) inflateReset2(&zs, -15);
            zs.next_in = const_cast<Bytef *>(in_.data() + b.in_off);
            zs.avail_in = b.in_len;
            zs.next_out = base_ + b.out_off;
            zs.avail_out = b.out_len;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) {
                inflateEnd(&zs);
                throw IoError("Failed to parse record: corrupt BGZF block");
            }
        }
        inflateEnd(&zs);
        done_[r].store(1, std::memory_order_release);
    }

    const Bytes &in_;
    std::vector<BlockRef> blocks_;
    size_t n_runs_ = 0, cursor_ = 0, ready_ = 0, total_ = 0;
    uint8_t *base_ = nullptr;
    std::unique_ptr<std::atomic<int>[]> done_;
    std::atomic<bool> failed_{false};
    std::string error_;
    std::thread driver_;
};

inline Bytes decompress(const Bytes &in, unsigned threads)
{
    Inflater inf(in, threads);
    inf.finish();
    return std::move(inf.out);
}

// BGZF compression of a stream given as pieces (the header, runs of surviving records: nothing is
// copied together first): 0xff00-byte payloads, parallel deflate -- a run of blocks per task, one
// deflate state per run, a block's payload gathered into a buffer that stays in the core's cache --
// while one thread writes the finished runs to the file in order; EOF block.
struct Piece {
    const uint8_t *p;
    size_t len;
};
inline void compress_pieces_to_file(const std::string &path, const std::vector<Piece> &pieces, unsigned threads, int level = 6)
{
    constexpr size_t PAYLOAD = 0xff00, RUN = 16, SLOT = 18 + 0x10000 + 1024 + 8;
    std::vector<size_t> start(pieces.size() + 1, 0); // offset of every piece in the stream
    for (size_t i = 0; i < pieces.size(); i++) start[i + 1] = start[i] + pieces[i].len;
    const size_t len = start.back(), nblk = (len + PAYLOAD - 1) / PAYLOAD, n_runs = (nblk + RUN - 1) / RUN;
    const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw IoError("cannot open output " + path);
    // (compressBound of a 0xff00 payload is below 0x10000 + 1024; address space only: pages are touched as written)
    Bytes slot_bytes(std::max<size_t>(nblk, 1) * SLOT);
    uint8_t *const slots = slot_bytes.data();
    std::vector<uint32_t> size_of(nblk, 0);
    std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[n_runs + 1]);
    for (size_t r = 0; r <= n_runs; r++) done[r].store(0);
    std::atomic<bool> write_failed{false};
    auto write_all = [&](const uint8_t *p, size_t n) {
        while (n) {
            const ssize_t w = ::write(fd, p, n);
            if (w <= 0) {
                write_failed = true;
                return;
            }
            p += w;
            n -= (size_t)w;
        }
    };
    std::thread writer([&] {
        for (size_t r = 0; r < n_runs && !write_failed; r++) {
            while (done[r].load(std::memory_order_acquire) == 0) std::this_thread::yield();
            if (done[r].load() < 0) return; // (a compressor failed)
            for (size_t i = r * RUN; i < std::min(nblk, (r + 1) * RUN); i++) write_all(slots + i * SLOT, size_of[i]);
        }
    });
    std::string err;
    try {
        parallel_for(n_runs, threads, [&](size_t r) {
            struct Flag { // (the writer is never left waiting for a run that will not come)
                std::atomic<int> &f;
                int v = -1;
                ~Flag() { f.store(v, std::memory_order_release); }
            } flag{done[r]};
            z_stream zs;
            std::memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw IoError("deflateInit2 failed");
            std::unique_ptr<uint8_t[]> payload(new uint8_t[PAYLOAD]);
            size_t piece = std::upper_bound(start.begin(), start.end(), r * RUN * PAYLOAD) - start.begin() - 1;
            for (size_t i = r * RUN; i < std::min(nblk, (r + 1) * RUN); i++) {
                const size_t o = i * PAYLOAD, n = std::min(PAYLOAD, len - o);
                for (size_t got = 0; got < n;) { // the block's bytes out of the pieces
                    while (start[piece + 1] <= o + got) piece++;
                    const size_t at = o + got - start[piece], take = std::min(n - got, pieces[piece].len - at);
                    std::memcpy(payload.get() + got, pieces[piece].p + at, take);
                    got += take;
                }
                uint8_t *b = slots + i * SLOT;
                if (i != r * RUN) deflateReset(&zs);
                zs.next_in = payload.get();
                zs.avail_in = (uInt)n;
                zs.next_out = b + 18;
                zs.avail_out = (uInt)(SLOT - 18 - 8);
                const int rc = deflate(&zs, Z_FINISH);
                const size_t clen = zs.total_out;
                if (rc != Z_STREAM_END) {
                    deflateEnd(&zs);
                    throw IoError("deflate failed");
                }
                const uint32_t bsize = (uint32_t)(18 + clen + 8 - 1);
                if (bsize > 0xffff) {
                    deflateEnd(&zs);
                    throw IoError("BGZF block overflow");
                }
                const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                         (uint8_t)(bsize & 0xff), (uint8_t)(bsize >> 8)};
                std::memcpy(b, hdr, 18);
                const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), payload.get(), (uInt)n);
                uint8_t *t = b + 18 + clen;
                for (int k = 0; k < 4; k++) t[k] = (uint8_t)(crc >> (8 * k));
                for (int k = 0; k < 4; k++) t[4 + k] = (uint8_t)((uint32_t)n >> (8 * k));
                size_of[i] = (uint32_t)(18 + clen + 8);
            }
            deflateEnd(&zs);
            flag.v = 1;
        });
    } catch (const std::exception &e) {
        err = e.what();
    }
    writer.join();
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0,
                                    0, 0, 0, 0, 0, 0, 0, 0};
    if (err.empty() && !write_failed) write_all(eof, sizeof(eof));
    ::close(fd);
    if (!err.empty()) throw IoError(err);
    if (write_failed) throw IoError("Failed to write the record");
}

// BGZF compression of a whole buffer
inline void compress_to_file(const std::string &path, const uint8_t *data, size_t len, unsigned threads, int level = 6)
{
    compress_pieces_to_file(path, std::vector<Piece>{{data, len}}, threads, level);
}

} // namespace bgzf
} // namespace umi
