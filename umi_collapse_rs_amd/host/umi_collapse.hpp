// C++ host-side mirror of the reference's plugin interface for the hot path, on top of
// the C ABI (include/umihip.h).  Header-only; link with -lumihip.
//
//   BitSet               src/utils/bitset.rs:9-14 (one word, umi_len <= 21) + to_bitset
//                        (src/utils/mod.rs:63-83)
//   ReadFreq<R>          src/utils/read_freq.rs:4-13
//   DataStruct contract  src/data/mod.rs:11-17   -> HipNaive (GPU-backed Naive, naive.rs)
//   Algorithm contract   src/algo/mod.rs:13-20   -> Directional (directional.rs:15-91),
//                                                    Adjacency  (adjacency.rs:15-63)
//   dedup_buckets()      the bucket loop src/deduplicate_sam.rs:207-233 through one
//                        batched call
//
// Error behaviour: where the reference panics (panic = "abort"), these throw
// umi::Error (std::runtime_error); nothing falls back to a CPU computation.
#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "../../include/umihip.h"

namespace umi {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("umihip error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc)
{
    if (rc != UMI_OK) throw Error(rc, umi_last_error());
}

// src/utils/bitset.rs:9-14, single word
struct BitSet {
    uint64_t bits = 0;
    uint64_t n_bits = 0;
    bool operator==(const BitSet &o) const { return bits == o.bits; }  // bitset.rs:94-101
    bool operator<(const BitSet &o) const { return bits < o.bits; }    // bitset.rs:105-112
    int32_t hash() const                                               // bitset.rs:130-147
    {
        int64_t h = 1234 ^ (int64_t)bits;
        return (int32_t)(uint32_t)(uint64_t)(h ^ (h >> 32));
    }
};

// utils::to_bitset (src/utils/mod.rs:63-83); throws where the reference panics (:77-79)
inline BitSet to_bitset(const std::string &s)
{
    BitSet b;
    check(umi_encode_umis(reinterpret_cast<const uint8_t *>(s.data()), 1, (int)s.size(), &b.bits, &b.n_bits));
    return b;
}

template <class R> struct ReadFreq { // src/utils/read_freq.rs:4-13
    R read;
    int32_t freq;
};

// The reference's HashMap<&BitSet, &ReadFreq<R>> of one bucket.  Iteration order here is
// insertion (first-appearance) order: the canonical determinisation of SURVEY.md 8c.
template <class R> using UmiReads = std::vector<std::pair<const BitSet *, const ReadFreq<R> *>>;
using UmiFreqMap = std::vector<std::pair<const BitSet *, int32_t>>;

class Context {
  public:
    explicit Context(int device_id = 0) { check(umi_ctx_create(device_id, &h_)); }
    // several GPUs of the node: dedup_batch shards its buckets over them (umi_ctx_create_multi)
    explicit Context(const std::vector<int> &device_ids)
    {
        check(umi_ctx_create_multi(device_ids.data(), (int)device_ids.size(), &h_));
    }
    ~Context() { umi_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    umi_ctx *get() const { return h_; }
    void set_option(const char *name, int64_t v) { check(umi_ctx_set_option(h_, name, v)); }
    static Context &global()
    {
        static Context c(0);
        return c;
    }

  private:
    umi_ctx *h_ = nullptr;
};

// trait DataStruct (src/data/mod.rs:11-17) with Naive's semantics, neighbour lists built on
// the GPU in the constructor (= DataStruct::new).
class HipNaive {
  public:
    HipNaive() = default; // Default
    HipNaive(const UmiFreqMap &umi_freq, size_t umi_length, int32_t max_edits, Context &ctx = Context::global())
    {
        std::vector<uint64_t> keys, nmask;
        std::vector<int32_t> freq;
        bool any_n = false;
        for (auto &kv : umi_freq) {
            index_[kv.first->bits] = (uint32_t)umis_.size();
            umis_.push_back(kv.first);
            keys.push_back(kv.first->bits);
            nmask.push_back(kv.first->n_bits);
            any_n |= kv.first->n_bits != 0;
            freq.push_back(kv.second);
        }
        check(umi_data_new(ctx.get(), keys.data(), any_n ? nmask.data() : nullptr, freq.data(),
                           (uint32_t)umis_.size(), (int)umi_length, max_edits, &h_));
    }
    HipNaive(HipNaive &&o) noexcept { *this = std::move(o); }
    HipNaive &operator=(HipNaive &&o) noexcept
    {
        std::swap(h_, o.h_);
        umis_.swap(o.umis_);
        index_.swap(o.index_);
        return *this;
    }
    ~HipNaive() { umi_data_free(h_); }

    // DataStruct::remove_near (naive.rs:26-40)
    std::unordered_set<const BitSet *> remove_near(const BitSet &umi, int32_t k, int32_t max_freq)
    {
        std::vector<uint32_t> out(umis_.size() + 1);
        uint32_t n = 0;
        check(umi_data_remove_near(h_, index_.at(umi.bits), k, max_freq, out.data(), &n));
        std::unordered_set<const BitSet *> res;
        for (uint32_t i = 0; i < n; i++) res.insert(umis_[out[i]]);
        return res;
    }
    // DataStruct::contains (naive.rs:42-44)
    bool contains(const BitSet &umi) const
    {
        auto it = index_.find(umi.bits);
        if (it == index_.end()) return false;
        int rc = umi_data_contains(h_, it->second);
        if (rc < 0) check(rc);
        return rc == 1;
    }
    std::unordered_map<std::string, float> stats() const { return {}; }

  private:
    umi_data *h_ = nullptr;
    std::vector<const BitSet *> umis_;
    std::unordered_map<uint64_t, uint32_t> index_;
};

// ClusterTracker stand-in (src/utils/cluster_tracker.rs): root -> members, first pass only
// (the reference's second pass is unfinished, deduplicate_sam.rs:236-239)
using ClusterTracker = std::unordered_map<const BitSet *, std::vector<const BitSet *>>;

inline int32_t threshold(float percentage, int32_t freq)
{ // directional.rs:38, Rust `as i32` semantics
    /* freq + 1 wraps in a release build of the reference (Cargo.toml:16-19) */
    volatile float prod = percentage * (float)(int32_t)((uint32_t)freq + 1u);
    float v = prod;
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

// src/algo/directional.rs:15-91
class Directional {
  public:
    Directional(int32_t k, float percentage, bool track_cluster = false) : k_(k), percentage_(percentage), track_(track_cluster) {}

    template <class R, class D = HipNaive>
    std::vector<const R *> apply(const UmiReads<R> &reads, ClusterTracker *tracker, size_t umi_length)
    {
        UmiFreqMap data_member; // :64-65
        std::unordered_map<uint64_t, const ReadFreq<R> *> by_key;
        for (auto &kv : reads) {
            data_member.emplace_back(kv.first, kv.second->freq);
            by_key[kv.first->bits] = kv.second;
        }
        UmiReads<R> umi_freqs(reads); // :67-70
        std::stable_sort(umi_freqs.begin(), umi_freqs.end(),
                         [](const auto &a, const auto &b) { return b.second->freq < a.second->freq; }); // :72
        D data(data_member, umi_length, k_); // :74
        std::vector<const R *> res;
        for (auto &entry : umi_freqs) { // :78-88
            if (!data.contains(*entry.first)) continue;
            // visit_and_remove (:30-54), explicit stack instead of recursion
            std::vector<const BitSet *> stack{entry.first};
            while (!stack.empty()) {
                const BitSet *start = stack.back();
                stack.pop_back();
                const int32_t thr = threshold(percentage_, by_key.at(start->bits)->freq); // :38
                auto near = data.remove_near(*start, k_, thr);                            // :39
                for (const BitSet *v : near) {
                    if (track_ && tracker) (*tracker)[entry.first].push_back(v); // :42-44
                    if (v == start) continue;                                    // :48-50
                    stack.push_back(v);
                }
            }
            res.push_back(&entry.second->read); // :86
        }
        return res;
    }

  private:
    int32_t k_;
    float percentage_;
    bool track_;
};

// src/algo/adjacency.rs:15-63 (remove_near(umi, k, 0): the reference's behaviour)
class Adjacency {
  public:
    Adjacency(int32_t k, float percentage = 0.5f, bool track_cluster = false, int32_t max_freq = 0)
        : k_(k), max_freq_(max_freq), track_(track_cluster)
    {
        (void)percentage;
    }
    template <class R, class D = HipNaive>
    std::vector<const R *> apply(const UmiReads<R> &reads, ClusterTracker *tracker, size_t umi_length)
    {
        UmiReads<R> freq(reads); // :40-43
        std::stable_sort(freq.begin(), freq.end(),
                         [](const auto &a, const auto &b) { return b.second->freq < a.second->freq; }); // :45
        UmiFreqMap m;
        for (auto &kv : reads) m.emplace_back(kv.first, kv.second->freq); // :47
        D data(m, umi_length, k_);                                         // :49
        std::vector<const R *> res;
        for (auto &entry : freq) { // :52-60
            if (!data.contains(*entry.first)) continue;
            auto near = data.remove_near(*entry.first, k_, max_freq_); // :56
            if (track_ && tracker)
                for (const BitSet *v : near) (*tracker)[entry.first].push_back(v);
            res.push_back(&entry.second->read);
        }
        return res;
    }

  private:
    int32_t k_, max_freq_;
    bool track_;
};

// The bucket loop src/deduplicate_sam.rs:207-233 as ONE batched call: `buckets` in the
// order the loop would visit them, each in first-appearance order.  Returns the surviving
// reads in the reference's output order and fills the loop's counters.
struct DedupCounters {
    size_t total_umi_count = 0, max_umi_count = 0, deduped_count = 0; // :217-219
    umi_stats stats{};
};
template <class R>
std::vector<const R *> dedup_buckets(const std::vector<UmiReads<R>> &buckets, size_t umi_length, int32_t k,
                                     float percentage, int algo, DedupCounters *counters = nullptr,
                                     Context &ctx = Context::global(), int32_t adj_max_freq = 0)
{
    std::vector<uint64_t> keys, nmask, off{0};
    std::vector<int32_t> freq;
    std::vector<const R *> reads;
    bool any_n = false;
    DedupCounters c;
    for (auto &b : buckets) {
        UmiReads<R> v(b);
        std::stable_sort(v.begin(), v.end(),
                         [](const auto &a, const auto &b2) { return b2.second->freq < a.second->freq; });
        for (auto &kv : v) {
            keys.push_back(kv.first->bits);
            nmask.push_back(kv.first->n_bits);
            any_n |= kv.first->n_bits != 0;
            freq.push_back(kv.second->freq);
            reads.push_back(&kv.second->read);
        }
        off.push_back(keys.size());
        c.total_umi_count += b.size();
        c.max_umi_count = std::max(c.max_umi_count, b.size());
    }
    std::vector<uint8_t> kept(keys.size() + 1);
    check(umi_dedup_batch(ctx.get(), keys.data(), any_n ? nmask.data() : nullptr, freq.data(), off.data(),
                          buckets.size(), (int)umi_length, k, percentage, algo, adj_max_freq, kept.data(),
                          nullptr, &c.stats));
    std::vector<const R *> out;
    for (size_t i = 0; i < keys.size(); i++)
        if (kept[i]) out.push_back(reads[i]);
    c.deduped_count = out.size();
    if (counters) *counters = c;
    return out;
}

} // namespace umi
