// C++ host mirror (umi_collapse.hpp) against the oracle, on a real GPU.
// Built by `make cpptest`; run by tests/test_gpu_cpp_host.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../oracle/umi_oracle.h"
#include "../../umi_collapse_rs_amd/host/umi_collapse.hpp"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

struct Read { int id; };

static std::vector<uint32_t> oracle_apply(const std::vector<std::string> &umis, const std::vector<int32_t> &freq,
                                          int k, float p, bool dir, int max_freq)
{
    std::vector<orc_bitset> bs(umis.size());
    for (size_t i = 0; i < umis.size(); i++) orc_to_bitset((const uint8_t *)umis[i].data(), (int)umis[i].size(), &bs[i]);
    std::vector<uint32_t> out(umis.size() + 1);
    uint32_t n = dir ? orc_directional_apply(bs.data(), freq.data(), (uint32_t)umis.size(), k, p, out.data(), nullptr, nullptr)
                     : orc_adjacency_apply(bs.data(), freq.data(), (uint32_t)umis.size(), k, max_freq, out.data(), nullptr, nullptr);
    out.resize(n);
    return out;
}

int main()
{
    using namespace umi;
    // KAT G8
    std::vector<std::string> umis = {"AAAAAAAAAAAA", "AAAAAAAAAAAT", "AAAAAAAAAATT", "AAAAAAAAATTT", "CCCCCCCCCCCC", "AAAAAAAAAAAC"};
    std::vector<int32_t> freq = {5, 2, 1, 1, 1, 3};
    std::vector<BitSet> keys;
    std::vector<ReadFreq<Read>> rfs;
    for (size_t i = 0; i < umis.size(); i++) { keys.push_back(to_bitset(umis[i])); rfs.push_back({{(int)i}, freq[i]}); }
    UmiReads<Read> reads;
    for (size_t i = 0; i < umis.size(); i++) reads.emplace_back(&keys[i], &rfs[i]);
    ClusterTracker tracker;
    auto out = Directional(1, 0.5f, true).apply<Read>(reads, &tracker, 12);
    REQUIRE(out.size() == 2 && out[0]->id == 0 && out[1]->id == 4);
    REQUIRE(tracker[&keys[0]].size() == 5 && tracker[&keys[4]].size() == 1);
    auto adj = Adjacency(1).apply<Read>(reads, nullptr, 12);
    const int want_adj[] = {0, 5, 1, 2, 3, 4};
    REQUIRE(adj.size() == 6);
    for (int i = 0; i < 6; i++) REQUIRE(adj[i]->id == want_adj[i]);
    REQUIRE(to_bitset("ACGT").bits == 0xaf0 && to_bitset("ACGT").hash() == 3618);
    bool threw = false;
    try { to_bitset("ACGX"); } catch (const Error &e) { threw = e.code == UMI_ERR_CHAR; }
    REQUIRE(threw);

    // random buckets: trait path and batched path against the oracle
    std::mt19937 rng(7);
    const char A[] = "ACGTN";
    std::vector<UmiReads<Read>> buckets;
    std::vector<std::vector<BitSet>> all_keys(30);
    std::vector<std::vector<ReadFreq<Read>>> all_rf(30);
    std::vector<int> expect_ids;
    int next_id = 0;
    for (int b = 0; b < 30; b++) {
        int L = 10, n_mol = 1 + rng() % 25;
        std::vector<std::string> u;
        std::vector<int32_t> f;
        for (int m = 0; m < n_mol; m++) {
            std::string t(L, 'A');
            for (auto &c : t) c = A[rng() % 4];
            int copies = 1 + rng() % 5;
            for (int c = 0; c < copies; c++) {
                std::string s = t;
                for (auto &ch : s) { if (rng() % 100 < 6) ch = A[rng() % 4]; if (rng() % 100 < 1) ch = 'N'; }
                size_t at = 0;
                for (; at < u.size(); at++) if (u[at] == s) break;
                if (at == u.size()) { u.push_back(s); f.push_back(1); } else f[at]++;
            }
        }
        for (size_t i = 0; i < u.size(); i++) { all_keys[b].push_back(to_bitset(u[i])); all_rf[b].push_back({{next_id + (int)i}, f[i]}); }
        UmiReads<Read> r;
        for (size_t i = 0; i < u.size(); i++) r.emplace_back(&all_keys[b][i], &all_rf[b][i]);
        for (float p : {0.5f, 1.0f}) {
            auto got = Directional(1, p).apply<Read>(r, nullptr, L);
            auto want = oracle_apply(u, f, 1, p, true, 0);
            REQUIRE(got.size() == want.size());
            for (size_t i = 0; i < got.size(); i++) REQUIRE(got[i]->id == next_id + (int)want[i]);
        }
        auto got = Adjacency(2, 0.5f, false, 2).apply<Read>(r, nullptr, L);
        auto want = oracle_apply(u, f, 2, 0.5f, false, 2);
        REQUIRE(got.size() == want.size());
        for (size_t i = 0; i < got.size(); i++) REQUIRE(got[i]->id == next_id + (int)want[i]);
        for (uint32_t w : oracle_apply(u, f, 1, 0.5f, true, 0)) expect_ids.push_back(next_id + (int)w);
        buckets.push_back(r);
        next_id += (int)u.size();
    }
    DedupCounters cnt;
    auto survivors = dedup_buckets<Read>(buckets, 10, 1, 0.5f, UMI_ALGO_DIRECTIONAL, &cnt);
    REQUIRE(survivors.size() == expect_ids.size());
    for (size_t i = 0; i < survivors.size(); i++) REQUIRE(survivors[i]->id == expect_ids[i]);
    REQUIRE(cnt.deduped_count == expect_ids.size() && cnt.total_umi_count == (size_t)next_id);
    std::printf("cpp host mirror ok: %d UMIs in 30 buckets, %zu survivors\n", next_id, survivors.size());
    return 0;
}
