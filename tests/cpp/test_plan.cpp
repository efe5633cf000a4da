// CPU-only test of the host planner (umi_collapse_rs_amd/csrc/umihip_plan.hpp), built with
// -fsanitize=address,undefined by tests/test_plan_cpu.py: random bucket tables through build_plan,
// gen_bs_tasks and partition_buckets_lpt, checking what the kernels rely on -- every pair of every
// bucket is covered exactly once by exactly one kernel's tasks, the entry ranges cover exactly the
// entries the fused kernel leaves, segment descriptors, scan chunks and capacities are consistent --
// and that nothing reads or writes outside its vectors while doing so.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "../../umi_collapse_rs_amd/csrc/umihip_plan.hpp"

using namespace umihip;

static int failures = 0;
#define CHECK(cond, ...)                                                         \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAIL %s:%d %s -- ", __FILE__, __LINE__, #cond); \
            std::fprintf(stderr, __VA_ARGS__);                                   \
            std::fprintf(stderr, "\n");                                          \
            if (++failures > 20) std::exit(1);                                   \
        }                                                                        \
    } while (0)

static uint64_t pairs_of(uint64_t n) { return n * (n - (n ? 1 : 0)) / 2; }

// pairs (row < col) of rows [r0, r1) x cols [c0, c1)
static uint64_t pairs_in(uint64_t r0, uint64_t r1, uint64_t c0, uint64_t c1)
{
    uint64_t t = 0;
    for (uint64_t r = r0; r < r1; r++) {
        const uint64_t lo = std::max(c0, r + 1);
        if (c1 > lo) t += c1 - lo;
    }
    return t;
}

static void check_plan(const std::vector<uint64_t> &off, int umi_len, int k, uint32_t small_max, bool use_bs,
                       uint32_t fused_max, bool prune, bool cache_prefix, bool tables, uint32_t min_run,
                       uint32_t seg_min, uint32_t col_chunk)
{
    Plan pl;
    const uint64_t nb = off.size() - 1;
    const bool key32 = umi_len <= 16;
    build_plan(off.data(), nb, small_max, use_bs, umi_len, fused_max, prune, cache_prefix, tables, min_run, seg_min, k,
               key32, pl);
    { // the same plan through the table pre-pass (what the library does): copy, counters, tables
        TablePass tp;
        std::vector<uint64_t> copy(nb + 1, 0xDEADull);
        scan_table(off.data(), nb, fused_max, copy.data(), tp);
        CHECK(tp.bad_at == ~0ull && copy == off, "table pass: copy / monotone");
        Plan p2;
        build_plan(off.data(), nb, small_max, use_bs, umi_len, fused_max, prune, cache_prefix, tables, min_run, seg_min,
                   k, key32, p2, &tp);
        CHECK(p2.n_fused == pl.n_fused && p2.n_pairs == pl.n_pairs && p2.n_pairs_eval == pl.n_pairs_eval &&
                  p2.max_bucket == pl.max_bucket,
              "table pass: counters %" PRIu64 " %" PRIu64 " %" PRIu64 " %" PRIu64 " vs %" PRIu64 " %" PRIu64 " %" PRIu64 " %" PRIu64,
              p2.n_fused, p2.n_pairs, p2.n_pairs_eval, p2.max_bucket, pl.n_fused, pl.n_pairs, pl.n_pairs_eval, pl.max_bucket);
        CHECK(p2.ranges.size() == pl.ranges.size() && p2.segs.size() == pl.segs.size() &&
                  p2.small_tasks.size() == pl.small_tasks.size() && p2.big_tasks.size() == pl.big_tasks.size() &&
                  p2.bs_buckets.size() == pl.bs_buckets.size() && p2.seg_blocks.size() == pl.seg_blocks.size() &&
                  p2.seg_task_cap == pl.seg_task_cap && p2.plane_words == pl.plane_words,
              "table pass: tables");
        for (size_t i = 0; i < pl.ranges.size(); i++)
            CHECK(p2.ranges[i].start == pl.ranges[i].start && p2.ranges[i].end == pl.ranges[i].end &&
                      p2.ranges[i].seg == pl.ranges[i].seg, "table pass: range %zu", i);
        if (nb > 1) { // a step backwards is reported at its bucket
            std::vector<uint64_t> broken = off;
            const uint64_t at = nb / 2;
            if (broken[at + 1] > 0) {
                broken[at + 1] = broken[at] > 0 ? broken[at] - 1 : 0;
                if (broken[at + 1] < broken[at]) {
                    scan_table(broken.data(), nb, fused_max, nullptr, tp);
                    CHECK(tp.bad_at == at, "table pass: step backwards at %" PRIu64 " reported at %" PRIu64, at, tp.bad_at);
                }
            }
        }
    }
    gen_bs_tasks(pl, umi_len, col_chunk, k, nullptr, key32);
    // per bucket: pairs covered by each family of tasks
    std::map<uint64_t, uint64_t> bucket_of_start; // start -> index
    for (uint64_t b = 0; b < nb; b++)
        if (off[b + 1] > off[b]) bucket_of_start[off[b]] = b;
    auto bucket_at = [&](uint64_t pos) {
        auto it = bucket_of_start.upper_bound(pos);
        --it;
        return it->second;
    };
    std::vector<uint64_t> covered(nb, 0);
    std::vector<int> families(nb, 0); // bit per family
    for (auto &t : pl.small_tasks) {
        const uint64_t b = bucket_at(t.row0);
        CHECK(t.row_end == off[b + 1] && t.col1 == off[b + 1] && t.col0 == t.row0, "small task of bucket %" PRIu64, b);
        covered[b] += pairs_in(t.row0, std::min<uint64_t>(t.row0 + SMALL_ROWS, t.row_end), t.col0, t.col1);
        families[b] |= 1;
    }
    for (auto &t : pl.big_tasks) {
        const uint64_t b = bucket_at(t.row0);
        CHECK(t.row_end == off[b + 1] && t.col1 <= off[b + 1] && t.col0 >= t.row0, "big task of bucket %" PRIu64, b);
        covered[b] += pairs_in(t.row0, std::min<uint64_t>(t.row0 + BIG_ROWS, t.row_end), t.col0, t.col1);
        families[b] |= 2;
    }
    const uint32_t gpl = (uint32_t)bs_groups_per_lane(umi_len);
    for (int li = 0; li < 4; li++)
        for (auto &t : pl.bs_tasks[li]) {
            const uint64_t b = bucket_at(t.bucket_start);
            const uint64_t s = off[b], e = off[b + 1];
            CHECK(t.bucket_start == s && t.bucket_end == e && t.ngroups == (e - s + 31) / 32, "bs task geometry");
            const uint64_t rows = (uint64_t)(li == 0 ? 64u : 256u) * gpl * 32;
            const uint64_t r0 = s + (uint64_t)t.group0 * 32, r1 = std::min(e, r0 + rows);
            CHECK(t.col0 >= r0 && t.col1 <= e && t.col0 < t.col1, "bs task columns");
            CHECK((t.col0 < r1) == (t.diag != 0), "diag flag");
            covered[b] += pairs_in(r0, r1, t.col0, t.col1);
            families[b] |= 4;
        }
    for (auto &rt : pl.tab_rows) { // the scan finds the column tiles on the device: all of [row tile start, e)
        const uint64_t b = bucket_at(rt.bucket_start);
        const uint64_t s = off[b], e = off[b + 1];
        const uint64_t r0 = s + (uint64_t)rt.group0 * 32, r1 = std::min<uint64_t>(e, r0 + (uint64_t)64 * BS_TAB_G * 32);
        covered[b] += pairs_in(r0, r1, r0, e);
        families[b] |= 8;
    }
    uint64_t n_fused = 0, seg_entries = 0;
    std::vector<int> seg_of_bucket(nb, -1);
    for (size_t si = 0; si < pl.segs.size(); si++) {
        const SegDesc &sd = pl.segs[si];
        const uint64_t b = bucket_at(sd.start);
        CHECK(sd.start == off[b] && sd.end == off[b + 1], "segment extent");
        seg_of_bucket[b] = (int)si;
        covered[b] += pairs_of(sd.end - sd.start); // (pigeonhole: every pair within k shares a part)
        families[b] |= 16;
        seg_entries += sd.end - sd.start;
        CHECK(pl.seg_parts == k + 1, "parts");
        uint64_t base_cover = 0;
        for (int j = 0; j < pl.seg_parts; j++) {
            const int b0 = j * umi_len / (k + 1), b1 = (j + 1) * umi_len / (k + 1);
            CHECK(sd.b0[j] == b0 && sd.nb[j] >= 1 && sd.b0[j] + sd.nb[j] <= b1, "part %d bases", j);
            const int bpb = key32 ? 2 : 3;
            CHECK(sd.mask[j] == ((((uint64_t)1 << (bpb * sd.nb[j])) - 1) << (bpb * sd.b0[j])), "part %d mask", j);
            base_cover |= sd.mask[j];
        }
        (void)base_cover;
    }
    CHECK(seg_entries == pl.seg_entries, "segment entries");
    // scan chunks: partition of [0, seg_bins), each inside one (segment, part)
    uint64_t next_bin = 0;
    for (auto &ch : pl.seg_chunks) {
        CHECK(ch.bin0 == next_bin && ch.nbins >= 1 && ch.nbins <= SEG_SCAN_CHUNK, "chunk order");
        const SegDesc &sd = pl.segs[ch.seg];
        const uint64_t first = sd.bin_off[ch.part], bins = 1ull << (2 * sd.nb[ch.part]);
        CHECK(ch.bin0 >= first && ch.bin0 + ch.nbins <= first + bins, "chunk inside its part");
        next_bin += ch.nbins;
    }
    CHECK(next_bin == pl.seg_bins, "chunks cover the bins");
    // task capacity: every entry of a part in one bin; all bins of 2 entries; bins of the size
    // with the most tasks per entry (32 chunks of 64 rows at two tiles per task)
    uint64_t worst = 0;
    for (auto &sd : pl.segs) {
        const uint64_t n = sd.end - sd.start;
        for (int j = 0; j < pl.seg_parts; j++) {
            const uint64_t bins = 1ull << (2 * sd.nb[j]);
            uint64_t w = seg_tasks_of_bin((uint32_t)n);
            w = std::max<uint64_t>(w, std::min<uint64_t>(bins, n / 2) * seg_tasks_of_bin(2));
            const uint64_t c = 2049;
            w = std::max<uint64_t>(w, std::min<uint64_t>(bins, n / c) * seg_tasks_of_bin((uint32_t)c) + seg_tasks_of_bin((uint32_t)(n % c)));
            CHECK(seg_task_bound(n, bins) >= w, "task bound %" PRIu64 " < %" PRIu64, seg_task_bound(n, bins), w);
            worst += w;
        }
    }
    CHECK(pl.seg_task_cap >= worst, "task capacity %" PRIu64 " < %" PRIu64, pl.seg_task_cap, worst);
    for (uint32_t c = 0; c < 70000; c++) // the bound per sub-bucket the capacity is built on
        CHECK(seg_tasks_of_bin(c) <= 1 + c / 7, "tasks of a sub-bucket of %u entries: %u", c, seg_tasks_of_bin(c));
    uint64_t total_pairs = 0, max_bucket = 0;
    for (uint64_t b = 0; b < nb; b++) {
        const uint64_t n = off[b + 1] - off[b];
        max_bucket = std::max(max_bucket, n);
        total_pairs += pairs_of(n);
        if (n >= 1 && n <= fused_max) n_fused++;
        if (n < 2 || n <= fused_max) {
            CHECK(families[b] == 0, "bucket %" PRIu64 " (n=%" PRIu64 ") has tasks", b, n);
            continue;
        }
        CHECK(families[b] != 0 && (families[b] & (families[b] - 1)) == 0, "bucket %" PRIu64 ": families %d", b, families[b]);
        CHECK(covered[b] == pairs_of(n), "bucket %" PRIu64 " n=%" PRIu64 ": %" PRIu64 " of %" PRIu64 " pairs", b, n,
              covered[b], pairs_of(n));
    }
    CHECK(pl.n_pairs == total_pairs && pl.max_bucket == max_bucket && pl.n_fused == n_fused, "counters");
    // ranges: exactly the entries of the buckets the fused kernel leaves, ascending, disjoint; an
    // entry of a segment lies in a range that carries that segment, any other entry in one that
    // carries none
    std::vector<uint32_t> want(off.back(), SEG_NONE - 1); // SEG_NONE - 1: not covered by any range
    for (uint64_t b = 0; b < nb; b++)
        if (off[b + 1] - off[b] > fused_max)
            for (uint64_t i = off[b]; i < off[b + 1]; i++)
                want[i] = seg_of_bucket[b] >= 0 ? (uint32_t)seg_of_bucket[b] : SEG_NONE;
    uint64_t prev_end = 0;
    std::vector<uint8_t> seen(off.back(), 0);
    for (auto &r : pl.ranges) {
        CHECK(r.start >= prev_end && r.end > r.start && r.end <= off.back() && r.end - r.start <= RANGE_CHUNK,
              "range [%u, %u)", r.start, r.end);
        if (r.end > off.back()) return;
        prev_end = r.end;
        for (uint64_t i = r.start; i < r.end; i++) {
            CHECK(want[i] == r.seg, "entry %" PRIu64 ": range segment %u, wanted %u", i, r.seg, want[i]);
            seen[i] = 1;
        }
    }
    for (uint64_t i = 0; i < off.back(); i++)
        CHECK((seen[i] != 0) == (want[i] != SEG_NONE - 1), "entry %" PRIu64 " coverage", i);
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 300;
    std::mt19937_64 rng(12345);
    const uint64_t sizes[] = {0, 1, 2, 31, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 4097,
                              9000, 32767, 32768, 40000, 65536, 70001};
    for (int it = 0; it < iters; it++) {
        std::vector<uint64_t> off{0};
        const int nb = 1 + (int)(rng() % 12);
        for (int b = 0; b < nb; b++) {
            uint64_t n = sizes[rng() % (sizeof(sizes) / sizeof(sizes[0]))];
            if (rng() % 4 == 0) n = rng() % 3000;
            off.push_back(off.back() + n);
        }
        const int umi_len = 1 + (int)(rng() % 21), k = (int)(rng() % 5);
        const uint32_t fused_max = (rng() % 3) ? 128 : (uint32_t)(rng() % 129);
        const uint32_t small_max = (rng() % 3) ? 1024 : (uint32_t)(rng() % 3000);
        const uint32_t seg_min = (rng() % 4) ? 512 : ((rng() % 2) ? 0 : 2 + (uint32_t)(rng() % 5000));
        const bool use_bs = k <= BS_MAX_K && (rng() % 5) != 0;
        const uint32_t col_chunk = BS_COL_TILE * (1 + (uint32_t)(rng() % 32));
        check_plan(off, umi_len, k, small_max, use_bs, fused_max, (rng() % 6) == 0, (rng() % 2) != 0, (rng() % 2) != 0,
                   (uint32_t)(rng() % 8), seg_min, col_chunk);
    }
    // partition: complete, deterministic, balanced to within the largest cost
    for (int it = 0; it < 200; it++) {
        std::vector<uint64_t> off{0};
        const int nb = (int)(rng() % 300);
        for (int b = 0; b < nb; b++) off.push_back(off.back() + ((rng() % 7 == 0) ? rng() % 100000 : rng() % 200));
        for (uint32_t world : {1u, 2u, 3u, 8u, 64u}) {
            std::vector<uint32_t> owner, again;
            partition_buckets_lpt(off.data(), (uint64_t)nb, world, owner);
            partition_buckets_lpt(off.data(), (uint64_t)nb, world, again);
            CHECK(owner == again && owner.size() == (size_t)nb, "partition");
            std::vector<long double> load(world, 0);
            long double biggest = 0;
            for (int b = 0; b < nb; b++) {
                CHECK(owner[b] < world, "owner range");
                const long double n = (long double)(off[b + 1] - off[b]), c = n * n + n;
                load[owner[b]] += c;
                biggest = std::max(biggest, c);
            }
            long double lo = load[0], hi = load[0];
            for (auto l : load) {
                lo = std::min(lo, l);
                hi = std::max(hi, l);
            }
            CHECK(hi - lo <= biggest + 1, "balance: %Lg vs %Lg (largest %Lg)", hi, lo, biggest);
        }
    }
    if (failures) {
        std::fprintf(stderr, "%d failures\n", failures);
        return 1;
    }
    std::puts("plan ok");
    return 0;
}
