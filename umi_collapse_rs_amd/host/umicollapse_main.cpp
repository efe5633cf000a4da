// `umicollapse`: the reference's command-line surface (src/cli.rs:7-77, src/main.rs:17-103)
// over the MI355X hot path.  BAM in -> read staging (src/deduplicate_sam.rs:93-177) ->
// ONE batched GPU call for every alignment position (replaces the loop :207-233) -> BAM out.
//
// "Next" rows N1/N2 of SURVEY.md 8f.  Deterministic where the reference is not: buckets and
// freq ties follow first appearance in the input (canonical determinisation).
// --paired (N4): template length joins the alignment key, second mates are skipped while
// staging and follow their surviving first mates into the output (UcWriter,
// deduplicate_sam.rs:339-459).
// --tag (N3): the reference stops after collecting its ClusterTrackers (the second pass is a
// TODO, deduplicate_sam.rs:236-239, so its output holds no deduplicated records at all).  Here
// the pass is finished from what the trackers hold (cluster_tracker.rs:76-103): every staged
// read is written, in file order, with MI:i = cluster id (offset + index of the cluster's root
// among the survivors), cs:i = reads in the cluster, su:i = reads with the same UMI at the same
// position.  Nothing in the reference to be in parity with: tests/bamio.py defines it.
// Not implemented, as in the reference: --mode fastq (main.rs:49-50), --two-pass, --algo cc.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <future>
#include <string>
#include <unistd.h>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/umihip.h"
#include "bam.hpp"
#include "bgzf.hpp"

namespace {

struct Cli { // src/cli.rs:7-77 (same flags, same defaults)
    std::string mode = "bam", input, output, algo = "dir", merge, data = "ngrambktree";
    int k = 1;
    size_t umi_length = 0;
    float percentage = 0.5f;
    unsigned num_threads = 1;
    uint8_t umi_sep = '_';
    bool two_pass = false, paired = false, remove_unpaired = false, remove_chimeric = false,
         keep_unmapped = false, track_clusters = false;
    // development switches (not in the reference)
    std::string dump_staging; // write the staged hot-path input here and stop before the GPU
    bool passthrough = false; // write every mapped record back (codec round trip), no dedup
    std::vector<int> devices{0}; // --device <ID> or --devices <ID,ID,...>
    int compress_level = 1;      // --compress-level 0..9: deflate level of the output's BGZF blocks.  Parity is
                                 // defined on the decompressed stream (htslib's own level and backend are
                                 // not reproducible here), and level 1 deflates a third of level 6's time
    std::string stage = "auto";  // --stage gpu|host|auto: where the reads are merged per (position, UMI) and
                                 // put in rank order (auto: on the GPU unless --paired or --tag need the
                                 // host's per-read bookkeeping)
};

[[noreturn]] void die(const std::string &msg)
{ // the reference panics (panic = "abort")
    std::fprintf(stderr, "umicollapse: %s\n", msg.c_str());
    std::fflush(stderr);
    std::_Exit(101); // (no static destructors: the GPU's start-up thread may still be running)
}

// libumihip.so is opened by hand, on the side thread that wakes the GPU: mapping the library and
// the HIP runtime it brings along (static initialisers, the code objects' registration) takes
// ~0.15 s before main() of a process that otherwise lives 0.45 s, and the first thing the
// program does -- reading and inflating the input -- needs none of it.
struct HipLib {
    void *handle = nullptr;
    int (*ctx_create_multi)(const int *, int, umi_ctx **) = nullptr;
    int (*ctx_set_option)(umi_ctx *, const char *, int64_t) = nullptr;
    const char *(*last_error)(void) = nullptr;
    // (the forms for keys of any number of words: one word is the ordinary call behind them)
    int (*stage_reads)(umi_ctx *, const uint64_t *, int, const uint8_t *, const int32_t *, uint64_t, int, int, int,
                       uint64_t *, uint64_t *, int32_t *, uint64_t *, uint64_t *, uint64_t *, uint64_t *) = nullptr;
    int (*dedup_batch)(umi_ctx *, const uint64_t *, const uint64_t *, int, const int32_t *, const uint64_t *, uint64_t,
                       int, int, float, int, int32_t, uint8_t *, uint32_t *, umi_stats *) = nullptr;
    std::string error;
    bool load()
    {
        if (handle) return true;
        char exe[4096];
        const ssize_t n = readlink("/proc/self/exe", exe, sizeof(exe) - 1);
        std::string dir = n > 0 ? std::string(exe, (size_t)n) : std::string(".");
        dir = dir.substr(0, dir.find_last_of('/'));
        const std::string path = dir + "/../libumihip.so"; // bin/umicollapse beside the package's library
        handle = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!handle) {
            error = std::string("cannot load ") + path + ": " + dlerror() + " (there is no CPU path)";
            return false;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(handle, name);
            if (!p && error.empty()) error = std::string("libumihip.so lacks ") + name;
            return p;
        };
        ctx_create_multi = (decltype(ctx_create_multi))sym("umi_ctx_create_multi");
        ctx_set_option = (decltype(ctx_set_option))sym("umi_ctx_set_option");
        last_error = (decltype(last_error))sym("umi_last_error");
        stage_reads = (decltype(stage_reads))sym("umi_stage_reads_wide");
        dedup_batch = (decltype(dedup_batch))sym("umi_dedup_batch_wide");
        return error.empty();
    }
};

// A UMI key: BitSet.bits of the reference (src/utils/bitset.rs:9-27), up to 85 bases in four words
constexpr int MAX_WORDS = 4;
struct UmiKey {
    uint64_t w[MAX_WORDS];
    bool operator==(const UmiKey &o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
};
struct UmiKeyHash {
    size_t operator()(const UmiKey &k) const
    {
        uint64_t x = k.w[0] * 0x9E3779B97F4A7C15ull ^ (k.w[1] + 0x7F4A7C15u) * 0xD6E8FEB86659FD93ull ^ (k.w[2] << 7) ^ (k.w[3] >> 3);
        x ^= x >> 31;
        x *= 0xBF58476D1CE4E5B9ull;
        return (size_t)(x ^ (x >> 29));
    }
};

// src/utils/mod.rs:63-83 with the codes of src/utils/read.rs:23-31 (the library's umi_encode_umis[_wide],
// restated here so that the staging of the host path needs no library call); base b at bits
// 3b .. 3b+2 of the word string, bit by bit: a base may sit across two words (bitset.rs:52-75)
bool encode_umi(const uint8_t *u, size_t len, UmiKey *key, UmiKey *nmask)
{
    UmiKey k{{0, 0, 0, 0}}, nm{{0, 0, 0, 0}};
    for (size_t b = 0; b < len; b++) {
        uint64_t c;
        switch (u[b]) {
        case 'A': c = 0; break;
        case 'T': c = 5; break;
        case 'C': c = 6; break;
        case 'G': c = 3; break;
        case 'N': c = 4; break;
        default: return false;
        }
        for (int j = 0; j < 3; j++) {
            const size_t bit = 3 * b + j;
            if ((c >> j) & 1) k.w[bit >> 6] |= 1ull << (bit & 63);
            if (c == 4) nm.w[bit >> 6] |= 1ull << (bit & 63);
        }
    }
    *key = k;
    *nmask = nm;
    return true;
}

void usage()
{
    std::puts("Usage: umicollapse [OPTIONS] -i <INPUT_FILE> -o <OUITPUT_FILE>\n"
              "  -m, --mode <MODE>        Either fastq or SAM/BAM mode [default: bam]\n"
              "  -k <K>                   Number of substitution edits to allow [default: 1]\n"
              "  -u <UMI_LENGTH>          The UMI length [default: 0 = autodetect]\n"
              "  -p <PERCENTAGE>          Directional threshold percentage [default: 0.5]\n"
              "      --num-threads <N>    Threads used in reader/writer [default: 1]\n"
              "      --umi_sep <BYTE>     Separator byte value between UMI and read name [default: 95]\n"
              "      --algo <ALGO>        adj or dir [default: dir]\n"
              "      --merge <MERGE>      any, avgqual or mapqual [default: mapqual in bam mode]\n"
              "      --data <DATA>        accepted; every value gives Naive's result (as in the reference)\n"
              "      --keep-unmapped      Keep unmapped reads\n"
              "      --paired             Paired-end mode: template length joins the alignment key,\n"
              "                           second mates follow their surviving first mates\n"
              "      --remove-unpaired    Remove unpaired reads (paired-end mode)\n"
              "      --remove-chimeric    Remove chimeric pairs (paired-end mode)\n"
              "      --tag                Write every read tagged with its cluster (MI, cs, su) instead of\n"
              "                           removing duplicates\n"
              "      --two-pass           accepted and rejected (see header)\n"
              "      --compress-level <N> deflate level of the output BAM, 0..9 [default: 1]\n"
              "      --stage <WHERE>      gpu, host or auto: where reads are merged per (position, UMI) [default: auto]\n"
              "      --device <ID>        GPU to use [default: 0]\n"
              "      --devices <ID,..>    several GPUs of the node: alignment positions are sharded over them");
}

// a GPU id: decimal digits only (atoi would take "x" for device 0)
int device_id(const char *text)
{
    char *end = nullptr;
    const long v = std::strtol(text, &end, 10);
    if (end == text || *end != '\0' || v < 0 || v > 1023) die(std::string("not a GPU id: '") + text + "'");
    return (int)v;
}

Cli parse(int argc, char **argv)
{
    Cli c;
    auto need = [&](int &i) -> const char * {
        if (i + 1 >= argc) die(std::string("a value is required for '") + argv[i] + "'");
        return argv[++i];
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-m" || a == "--mode") c.mode = need(i);
        else if (a == "-i") c.input = need(i);
        else if (a == "-o") c.output = need(i);
        else if (a == "-k") c.k = std::atoi(need(i));
        else if (a == "-u") c.umi_length = (size_t)std::atol(need(i));
        else if (a == "-p") c.percentage = std::strtof(need(i), nullptr);
        else if (a == "--num-threads") c.num_threads = (unsigned)std::atoi(need(i));
        else if (a == "--umi_sep") c.umi_sep = (uint8_t)std::atoi(need(i)); // a number, cli.rs:31-32
        else if (a == "--algo") c.algo = need(i);
        else if (a == "--merge") c.merge = need(i);
        else if (a == "--data") c.data = need(i);
        else if (a == "--two-pass") c.two_pass = true;
        else if (a == "--paired") c.paired = true;
        else if (a == "--remove-unpaired") c.remove_unpaired = true;
        else if (a == "--remove-chimeric") c.remove_chimeric = true;
        else if (a == "--keep-unmapped") c.keep_unmapped = true;
        else if (a == "--tag") c.track_clusters = true;
        else if (a == "--dump-staging") c.dump_staging = need(i);
        else if (a == "--passthrough") c.passthrough = true;
        else if (a == "--stage") c.stage = need(i);
        else if (a == "--compress-level") {
            c.compress_level = std::atoi(need(i));
            if (c.compress_level < 0 || c.compress_level > 9) die("--compress-level wants 0..9");
        }
        else if (a == "--device") c.devices.assign(1, device_id(need(i)));
        else if (a == "--devices") { // the GPUs of the node the position buckets are sharded over
            c.devices.clear();
            std::string list = need(i);
            for (size_t p = 0; p <= list.size();) {
                const size_t q = std::min(list.find(',', p), list.size());
                if (q == p) die("--devices wants a comma separated list of GPU ids");
                c.devices.push_back(device_id(list.substr(p, q - p).c_str()));
                p = q + 1;
            }
        }
        else if (a == "-h" || a == "--help") { usage(); std::exit(0); }
        else die("unexpected argument '" + a + "'");
    }
    if (c.input.empty() || c.output.empty()) { usage(); die("-i and -o are required"); }
    return c;
}

struct Entry { // one (alignment key, UMI): ReadFreq of src/utils/read_freq.rs + its key
    UmiKey key, nmask;
    int32_t freq;
    int32_t score;  // avg qual or mapq of the representative
    uint32_t rep;   // record index of the representative read
    uint32_t bucket;
};

// Align (deduplicate_sam.rs:478-481): Alignment{strand, coord, ref} or, with --paired,
// PairedAlignment{strand, coord, ref, tlen} (:547-553); ref as tid (equal names <=> equal tid)
struct AlignKey {
    uint64_t coord, ref_strand, tlen;
    bool operator==(const AlignKey &o) const { return coord == o.coord && ref_strand == o.ref_strand && tlen == o.tlen; }
};

struct KeyHash {
    size_t operator()(const AlignKey &k) const
    {
        uint64_t x = k.coord * 0x9E3779B97F4A7C15ull ^ (k.ref_strand + 0x7F4A7C15u) ^ (k.tlen * 0xD6E8FEB86659FD93ull);
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        return (size_t)(x ^ (x >> 32));
    }
};

// ReverseRead (deduplicate_sam.rs:272-286): the mate a written paired record is waiting for
std::string mate_key(const uint8_t *qname, size_t n, int32_t tid, int32_t pos)
{
    std::string s((const char *)qname, n);
    s.append((const char *)&tid, 4);
    s.append((const char *)&pos, 4);
    return s;
}

// UcSAMRead::get_umi_length (read.rs:65-75,87-94): first separator followed by a base
// (caseless [ATCGN]), length of that run.
size_t detect_umi_length(const uint8_t *q, size_t n, uint8_t sep)
{
    auto is_base = [](uint8_t ch) {
        switch (ch | 0x20) { case 'a': case 't': case 'c': case 'g': case 'n': return true; default: return false; }
    };
    for (size_t i = 0; i + 1 < n; i++)
        if (q[i] == sep && is_base(q[i + 1])) {
            size_t j = i + 1;
            while (j < n && is_base(q[j])) j++;
            return j - i - 1;
        }
    die("No UMI group found in pattern match");
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

int main(int argc, char **argv)
{
    const double t_main_realtime = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    Cli args = parse(argc, argv);
    const double t_start = now_s();
    if (args.merge.empty()) args.merge = args.mode == "fastq" ? "avgqual" : "mapqual"; // main.rs:33-39
    if (args.track_clusters && args.two_pass) die("Cannot track clusters with the two pass algorithm!");
    if (args.paired && args.keep_unmapped) die("Cannot keep unmapped reads with paired-end reads!");
    if (args.mode == "fastq") die("fastq mode is not implemented (nor in the reference: main.rs:49-50)");
    if (args.mode != "bam" && args.mode != "sam") return 0; // main.rs:49-95: nothing happens
    if (args.track_clusters && args.paired) die("--tag with --paired is not implemented (the reference never reaches its tagging pass)");
    int algo, merge;
    if (args.algo == "dir") algo = UMI_ALGO_DIRECTIONAL;
    else if (args.algo == "adj") algo = UMI_ALGO_ADJACENCY;
    else die("Invalid algorithm combination: " + args.algo + " , " + args.merge + " and " + args.data); // main.rs:86-91
    if (args.merge == "any") merge = 0;
    else if (args.merge == "avgqual") merge = 1;
    else if (args.merge == "mapqual") merge = 2;
    else die("Invalid algorithm combination: " + args.algo + " , " + args.merge + " and " + args.data);

    // The GPU is woken while the file is read: context creation and the first launch of the
    // library's kernels (their code objects are loaded then) take ~0.1 s of a process that lives
    // half a second, none of it on the device.  A tiny staging call and a tiny batch go through;
    // whoever needs the context first waits for this thread.
    HipLib lib;
    std::future<umi_ctx *> warm;
    std::string warm_error;
    if (!args.passthrough && args.dump_staging.empty())
        warm = std::async(std::launch::async, [&]() -> umi_ctx * {
            umi_ctx *c = nullptr;
            if (!lib.load()) {
                warm_error = lib.error;
                return nullptr;
            }
            if (lib.ctx_create_multi(args.devices.data(), (int)args.devices.size(), &c) != UMI_OK) {
                warm_error = lib.last_error();
                return nullptr;
            }
            const uint64_t akey[2] = {0, 0};
            const uint8_t umis[8] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'A'};
            uint64_t k[2], nm[2], rp[2], off[3], ne = 0, nbk = 0;
            int32_t fr[2];
            uint8_t kept[2];
            umi_stats wst;
            if (lib.stage_reads(c, akey, 1, umis, nullptr, 2, 4, 1, 0, k, nm, fr, rp, off, &ne, &nbk) != UMI_OK ||
                lib.dedup_batch(c, k, nullptr, 1, fr, off, nbk, 4, 1, 0.5f, UMI_ALGO_DIRECTIONAL, 0, kept, nullptr, &wst) != UMI_OK)
                warm_error = lib.last_error(); // (reported when the real call fails the same way)
            return c;
        });

    // finer split of the program's time, printed with UMICOLLAPSE_CLOCK (tools/e2e_probe.py)
    std::vector<std::pair<const char *, double>> laps;
    double t_lap = now_s();
    auto lap = [&](const char *what) {
        const double t = now_s();
        laps.emplace_back(what, t - t_lap);
        t_lap = t;
    };
    auto leave = [&]() {
        if (warm.valid()) warm.wait(); // (a file without staged reads: the start-up thread may still be at it)
        if (std::getenv("UMICOLLAPSE_CLOCK")) { // (for tools/e2e_probe.py: what lies before main and after _Exit)
            std::fprintf(stderr, "laps:");
            for (const auto &l : laps) std::fprintf(stderr, " %s %.3f", l.first, l.second);
            std::fprintf(stderr, "\nclock: main at %.6f, exit at %.6f (realtime)\n", t_main_realtime,
                         std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
        }
        std::fflush(nullptr);
        std::_Exit(0); // the output file is closed; device memory and the runtime go with the process
    };
    try {
        // ---- read: BGZF inflate (threaded) + BAM parse
        umi::bam::File in;
        // (the compressed bytes are not given back before the process ends: unmapping 0.1 GB takes 6 ms)
        static umi::bgzf::Bytes raw;
        raw = umi::bgzf::read_file(args.input, args.num_threads);
        lap("read");
        {
            umi::bgzf::Inflater inflater(raw, args.num_threads); // (the parse walks behind the inflate threads)
            in.data.swap(inflater.out);                          // (same storage: the threads write through their pointer)
            in.parse_behind([&](size_t upto) { inflater.wait(upto); });
            inflater.finish();
        }
        lap("inflate+parse");
        const double t_read = now_s();

        // ---- staging: deduplicate_sam.rs:93-177, in three passes so that --num-threads helps:
        //  A (parallel over records)  alignment key, UMI key, merge score of every read
        //  B (parallel over shards of the alignment-key hash; every shard walks the reads in
        //     file order)              per-bucket UMI maps with the reference's merge rule
        //  C (sequential)             buckets in order of first appearance, entries in rank order
        const uint32_t n_rec = (uint32_t)in.records.size();
        const unsigned T = std::max(1u, args.num_threads);
        size_t umi_length = args.umi_length;
        // the filters of the read loop (:95-129); returns ReadInfo::state
        auto classify = [&](const umi::bam::Record &r, uint8_t &is_unpaired, uint8_t &is_chimeric) -> uint8_t {
            is_unpaired = is_chimeric = 0;
            if (args.paired && r.is_paired() && r.is_last_in_template()) return 3; // :95-97
            if (r.is_unmapped()) return 1;                                         // :102-108
            if (args.paired && !args.passthrough) {                                // :110-129
                if (!r.is_paired()) {
                    is_unpaired = 1;
                    if (args.remove_unpaired) return 5;
                }
                if (r.is_paired() && r.is_mate_unmapped()) return 4;
                if (r.is_paired() && r.tid() != r.mtid()) {
                    is_chimeric = 1;
                    if (args.remove_chimeric) return 5;
                }
            }
            return 0;
        };
        if (umi_length == 0 && !args.passthrough) // autodetect on the first staged read (:154-156)
            for (uint32_t ri = 0; ri < n_rec; ri++) {
                uint8_t u, c;
                if (classify(in.records[ri], u, c) == 0) {
                    umi_length = detect_umi_length(in.records[ri].qname(), in.records[ri].qname_len(), args.umi_sep);
                    break;
                }
            }
        // (where the reads are merged per (position, UMI): on the GPU unless something needs the host's
        // per-read bookkeeping -- decided here because the per-read pass only encodes UMIs for the host path)
        if (args.stage != "auto" && args.stage != "gpu" && args.stage != "host") die("--stage wants gpu, host or auto");
        bool gpu_stage = args.stage != "host" && !args.passthrough && !args.paired && !args.track_clusters &&
                         args.dump_staging.empty() && umi_length >= 1;
        if (args.stage == "gpu" && !gpu_stage) die("--stage gpu does not go with --paired, --tag or --dump-staging");
        struct ReadInfo {
            uint64_t coord, ref_strand, tlen;
            int32_t score;
            uint8_t state; // 0 staged, 1 unmapped, 2 error, 3 second mate (not counted),
                           // 4 mate unmapped, 5 filtered (--remove-unpaired / --remove-chimeric)
            uint8_t unpaired, chimeric;
            uint32_t umi_at; // offset of the UMI in the read name
        };
        std::vector<ReadInfo> info(n_rec);
        std::vector<UmiKey> rkey, rnm; // per read: its UMI key and N mask (host staging only: the device encodes its own)
        auto encode_all = [&]() {       // utils/mod.rs:63-83 for every staged read; the first bad character ends the run
            rkey.resize(n_rec);
            rnm.resize(n_rec);
            std::vector<uint32_t> bad(T, UINT32_MAX);
            const uint32_t per = (n_rec + T - 1) / T;
            umi::bgzf::parallel_for(T, T, [&](size_t t) {
                for (uint32_t ri = (uint32_t)t * per; ri < std::min(n_rec, ((uint32_t)t + 1) * per); ri++)
                    if (info[ri].state == 0 && !args.passthrough &&
                        !encode_umi(in.records[ri].qname() + info[ri].umi_at, umi_length, &rkey[ri], &rnm[ri]) &&
                        bad[t] == UINT32_MAX)
                        bad[t] = ri;
            });
            for (unsigned t = 0; t < T; t++)
                if (bad[t] != UINT32_MAX) die("Unknown character in UMI sequence");
        };
        // GPU staging: what the device wants of a read -- alignment key, UMI text, score -- is written by the
        // per-read pass itself, at the read's own index (closed up afterwards if some reads are not staged)
        using U64s = std::vector<uint64_t, umi::bgzf::default_init_allocator<uint64_t>>;
        using I32s = std::vector<int32_t, umi::bgzf::default_init_allocator<int32_t>>;
        U64s akey, rep64;
        umi::bgzf::Bytes umis;
        I32s sc;
        std::vector<uint8_t> fits(T, 1);
        std::vector<int64_t> c_min(T, INT64_MAX), c_max(T, INT64_MIN); // coordinates and (ref, strand) codes seen, per thread
        std::vector<uint64_t> rs_max(T, 0);
        if (gpu_stage) {
            akey.resize(n_rec);
            umis.resize((size_t)n_rec * umi_length);
            sc.resize(n_rec);
        }
        std::vector<std::string> errors(T);
        std::vector<uint32_t> first_error(T, UINT32_MAX);
        const uint32_t chunk = (n_rec + T - 1) / T;
        umi::bgzf::parallel_for(T, T, [&](size_t t) {
            const uint32_t lo = (uint32_t)t * chunk, hi = std::min(n_rec, lo + chunk);
            // (the thread's extremes in locals: sixteen threads updating neighbours of one cache line
            // per read made this pass 0.27 s instead of 0.02)
            int64_t my_c_min = INT64_MAX, my_c_max = INT64_MIN;
            uint64_t my_rs_max = 0;
            for (uint32_t ri = lo; ri < hi; ri++) {
                const umi::bam::Record &r = in.records[ri];
                ReadInfo &ii = info[ri];
                ii.tlen = 0;
                ii.state = classify(r, ii.unpaired, ii.chimeric);
                if (ii.state != 0 || args.passthrough) continue;
                if (args.paired) ii.tlen = (uint64_t)(int64_t)r.tlen(); // record.insert_size(), :138
                // Alignment{strand, coord, ref} (:141-145); equality on tid == equality on the name
                ii.coord = (uint64_t)r.unclipped_pos();
                ii.ref_strand = ((uint64_t)(uint32_t)r.tid() << 1) | (r.is_reverse() ? 1u : 0u);
                const uint8_t *q = r.qname();
                const size_t qn = r.qname_len();
                const uint8_t *sp = (const uint8_t *)std::memchr(q, args.umi_sep, qn); // read.rs:100
                const char *err = nullptr;
                const size_t at = sp ? (size_t)(sp - q) + 1 : 0;
                if (!sp) err = "failed to get the umi";
                else if (umi_length == 0) err = "Empty UMI sequence extracted";
                else if (umi_length > UMI_MAX_WIDE_UMI_LEN) err = "UMIs of more than 85 bases are not handled";
                else if (at + umi_length > qn) err = "UMI runs past the end of the read name";
                if (err) {
                    ii.state = 2;
                    if (first_error[t] == UINT32_MAX) { first_error[t] = ri; errors[t] = err; }
                    continue;
                }
                ii.score = merge == 2 ? (int32_t)r.mapq() : r.avg_qual();
                ii.umi_at = (uint32_t)at;
                if (gpu_stage) {
                    // Alignment{strand, coord, ref} in 64 bits: ref id (31) | strand (1) | coordinate (32)
                    const int64_t c = (int64_t)ii.coord;
                    if (c < INT32_MIN || c > INT32_MAX) fits[t] = 0;
                    my_c_min = std::min(my_c_min, c);
                    my_c_max = std::max(my_c_max, c);
                    my_rs_max = std::max(my_rs_max, ii.ref_strand);
                    akey[ri] = (ii.ref_strand << 32) | (uint64_t)(uint32_t)(int32_t)c; // (packed tighter below)
                    std::memcpy(&umis[(size_t)ri * umi_length], q + at, umi_length);
                    sc[ri] = ii.score;
                }
            }
            c_min[t] = my_c_min;
            c_max[t] = my_c_max;
            rs_max[t] = my_rs_max;
        });
        for (unsigned t = 0; t < T; t++) // the reference panics at the first offending read
            if (first_error[t] != UINT32_MAX) die(errors[t]);
        lap("per-read");
        if (!gpu_stage && !args.passthrough) encode_all();

        size_t total_read_count = 0, unmapped = 0, unpaired = 0, chimeric = 0;
        std::vector<uint32_t> out_records; // records written before dedup (--keep-unmapped, :104-106)
        for (uint32_t ri = 0; ri < n_rec; ri++) {
            if (info[ri].state != 3) total_read_count++; // :99
            unpaired += info[ri].unpaired;
            chimeric += info[ri].chimeric;
            if (info[ri].state == 4) unmapped++; // :118-121
            if (info[ri].state == 1) {
                unmapped++;
                if (args.keep_unmapped || args.passthrough) out_records.push_back(ri);
            } else if (args.passthrough) {
                out_records.push_back(ri);
            }
        }

        // ---- staging: reads -> unique (position, UMI) entries in canonical order (:148-176 and the
        // rank order of directional.rs:67-72).  On the GPU (umi_stage_reads: sorts and a segmented
        // merge) where the alignment key packs into 64 bits and nothing needs the per-read
        // bookkeeping of the host version below; both give the same arrays.
        size_t n = 0, nb = 0, max_umi = 0;
        bool any_n = false;
        const int n_words = umi_length ? (int)((3 * umi_length + 63) / 64) : 1; // words per key (bitset.rs:17-18)
        // (not zeroed when sized: the staging call writes them)
        std::vector<uint64_t, umi::bgzf::default_init_allocator<uint64_t>> keys, nmask, off; // keys / nmask: n_words words per entry
        std::vector<int32_t, umi::bgzf::default_init_allocator<int32_t>> freq;
        std::vector<uint32_t> rep;
        std::vector<std::vector<uint32_t>> global_of;
        std::vector<uint32_t> entry_of;
        KeyHash hasher;
        umi_ctx *ctx = nullptr;
        double t_init = 0.0;
        auto need_ctx = [&]() { // (t_init: what of the GPU's start-up was left to wait for)
            if (ctx) return;
            const double t0 = now_s();
            if (warm.valid()) {
                ctx = warm.get();
                if (!ctx) die(warm_error);
            } else {
                if (!lib.load()) die(lib.error);
                if (lib.ctx_create_multi(args.devices.data(), (int)args.devices.size(), &ctx) != UMI_OK) die(lib.last_error());
            }
            t_init += now_s() - t0;
        };
        if (gpu_stage) {
            // the staged reads closed up (nothing moves while every read so far is staged)
            std::vector<uint32_t> staged; // staged[j] = record of the j-th staged read, once a read has been left out
            bool moved = false;
            size_t ns = 0;
            for (uint32_t ri = 0; ri < n_rec; ri++) {
                if (info[ri].state != 0) {
                    if (!moved) {
                        moved = true;
                        staged.reserve(n_rec);
                        for (uint32_t j = 0; j < ri; j++) staged.push_back(j);
                    }
                    continue;
                }
                if (moved) {
                    akey[ns] = akey[ri];
                    std::memmove(&umis[ns * umi_length], &umis[(size_t)ri * umi_length], umi_length);
                    sc[ns] = sc[ri];
                    staged.push_back(ri);
                }
                ns++;
            }
            rep64.resize(ns);
            for (uint8_t f : fits) gpu_stage = gpu_stage && f;
            // The alignment key in as few bits as the file needs -- (ref, strand) code above the coordinate
            // counted from the smallest one -- so that with the UMI it fits the device sort's one 64-bit key
            // (a human genome: 6 + 28 bits, and 28 more for 12 bases)
            int akey_bits = 64;
            if (gpu_stage && ns) {
                const int64_t lo = *std::min_element(c_min.begin(), c_min.end()), hi = *std::max_element(c_max.begin(), c_max.end());
                const uint64_t rs_hi = *std::max_element(rs_max.begin(), rs_max.end());
                auto bits_of = [](uint64_t v) { int b = 1; while (b < 64 && (v >> b)) b++; return b; };
                const int cbits = bits_of((uint64_t)(hi - lo)), rbits = bits_of(rs_hi);
                if (cbits + rbits < 64) {
                    akey_bits = cbits + rbits;
                    const size_t per = (ns + T - 1) / T;
                    umi::bgzf::parallel_for(T, T, [&](size_t t) {
                        for (size_t j = t * per; j < std::min(ns, (t + 1) * per); j++) {
                            const uint64_t a = akey[j];
                            akey[j] = ((a >> 32) << cbits) | (uint64_t)((int64_t)(int32_t)(uint32_t)a - lo);
                        }
                    });
                }
            }
            lap("fill");
            if (!gpu_stage) encode_all(); // (a coordinate beyond 32 bits: the host staging takes the file)
            if (gpu_stage) {
                need_ctx();
                lap("wait-gpu");
                keys.resize(ns * n_words); nmask.resize(ns * n_words); freq.resize(ns); off.resize(ns + 1);
                uint64_t ne = 0, nbk = 0;
                if (lib.stage_reads(ctx, akey.data(), akey_bits, umis.data(), sc.data(), ns, (int)umi_length, n_words, merge != 0 ? 1 : 0,
                                    keys.data(), nmask.data(), freq.data(), rep64.data(), off.data(), &ne, &nbk) != UMI_OK)
                    die(lib.last_error());
                lap("stage-call");
                n = (size_t)ne;
                nb = (size_t)nbk;
                keys.resize(n * n_words); nmask.resize(n * n_words); freq.resize(n); off.resize(nb + 1);
                rep.resize(n);
                for (size_t i = 0; i < n; i++) rep[i] = moved ? staged[rep64[i]] : (uint32_t)rep64[i];
                for (uint64_t m : nmask) any_n |= m != 0;
                for (size_t b = 0; b < nb; b++) max_umi = std::max<size_t>(max_umi, off[b + 1] - off[b]);
                lap("after-stage");
            }
        }
        if (!gpu_stage) {
        struct Shard {
            std::unordered_map<AlignKey, uint32_t, KeyHash> bucket_of; // Align -> local bucket
            std::vector<std::unordered_map<UmiKey, uint32_t, UmiKeyHash>> umi_index;          // key -> local entry
            std::vector<std::vector<uint32_t>> bucket_entries;
            std::vector<uint32_t> bucket_first; // first read of the bucket
            std::vector<Entry> entries;
        };
        std::vector<Shard> shards(args.passthrough ? 0 : T);
        entry_of.assign(args.track_clusters ? n_rec : 0, 0); // read -> entry of its shard (--tag)
        umi::bgzf::parallel_for(shards.size(), T, [&](size_t t) {
            Shard &sh = shards[t];
            for (uint32_t ri = 0; ri < n_rec; ri++) {
                const ReadInfo &ii = info[ri];
                if (ii.state != 0) continue;
                const AlignKey akey{ii.coord, ii.ref_strand, ii.tlen};
                if (hasher(akey) % T != t) continue;
                auto it = sh.bucket_of.find(akey);
                uint32_t b;
                if (it == sh.bucket_of.end()) {
                    b = (uint32_t)sh.bucket_entries.size();
                    sh.bucket_of.emplace(akey, b);
                    sh.bucket_entries.emplace_back();
                    sh.umi_index.emplace_back();
                    sh.bucket_first.push_back(ri);
                } else {
                    b = it->second;
                }
                auto &idx = sh.umi_index[b];
                auto e = idx.find(rkey[ri]);
                if (args.track_clusters) entry_of[ri] = e == idx.end() ? (uint32_t)sh.entries.size() : e->second;
                if (e == idx.end()) { // Vacant :161-163
                    idx.emplace(rkey[ri], (uint32_t)sh.entries.size());
                    sh.bucket_entries[b].push_back((uint32_t)sh.entries.size());
                    sh.entries.push_back({rkey[ri], rnm[ri], 1, ii.score, ri, b});
                } else { // Occupied :164-175
                    Entry &en = sh.entries[e->second];
                    const bool keep_existing = merge == 0 ? true : en.score >= ii.score; // merge/mod.rs:21,35,49
                    en.freq += 1;
                    if (!keep_existing) { en.rep = ri; en.score = ii.score; }
                }
            }
        });

        // buckets in order of first appearance; inside a bucket the stable freq-descending order
        // of directional.rs:67-72 (creation order of a bucket's entries = first appearance)
        struct BucketRef { uint32_t first, shard, local; };
        std::vector<BucketRef> order;
        for (uint32_t t = 0; t < shards.size(); t++) {
            n += shards[t].entries.size();
            for (uint32_t b = 0; b < shards[t].bucket_entries.size(); b++)
                order.push_back({shards[t].bucket_first[b], t, b});
        }
        std::sort(order.begin(), order.end(), [](const BucketRef &x, const BucketRef &y) { return x.first < y.first; });
        nb = order.size();
        keys.assign(n * n_words, 0); nmask.assign(n * n_words, 0); off.assign(nb + 1, 0);
        freq.assign(n, 0);
        rep.assign(n, 0);
        size_t w = 0;
        global_of.assign(args.track_clusters ? shards.size() : 0, {}); // (shard, entry) -> index
        for (size_t t = 0; t < global_of.size(); t++) global_of[t].resize(shards[t].entries.size());
        for (size_t b = 0; b < nb; b++) {
            Shard &sh = shards[order[b].shard];
            auto &v = sh.bucket_entries[order[b].local];
            std::stable_sort(v.begin(), v.end(), [&](uint32_t x, uint32_t y) { return sh.entries[y].freq < sh.entries[x].freq; });
            for (uint32_t ei : v) {
                const Entry &en = sh.entries[ei];
                if (args.track_clusters) global_of[order[b].shard][ei] = (uint32_t)w;
                for (int q = 0; q < n_words; q++) {
                    keys[w * n_words + q] = en.key.w[q];
                    nmask[w * n_words + q] = en.nmask.w[q];
                    any_n |= en.nmask.w[q] != 0;
                }
                freq[w] = en.freq; rep[w] = en.rep;
                w++;
            }
            off[b + 1] = w;
            max_umi = std::max(max_umi, v.size());
        }
        }
        const double t_stage0 = now_s();
        std::fprintf(stderr, "UMI collapsing reading finished in %.3f seconds\n", t_stage0 - t_start); // :178-183
        if (!args.dump_staging.empty()) { // test hook: staged hot-path input, no GPU touched
            FILE *f = std::fopen(args.dump_staging.c_str(), "wb");
            if (!f) die("cannot open " + args.dump_staging);
            const uint64_t hdr[4] = {n, nb, umi_length, (uint64_t)n_words};
            std::fwrite(hdr, 8, 4, f);
            std::fwrite(keys.data(), 8, n * n_words, f); std::fwrite(nmask.data(), 8, n * n_words, f);
            std::fwrite(freq.data(), 4, n, f); std::fwrite(rep.data(), 4, n, f);
            std::fwrite(off.data(), 8, nb + 1, f);
            std::fclose(f);
            return 0;
        }

        // ---- the hot path: one batched call replaces the bucket loop :207-233
        lap("to-hot-path");
        std::vector<uint8_t> kept(n + 1, 0);
        std::vector<uint32_t> root(args.track_clusters ? n + 1 : 0);
        umi_stats st;
        std::memset(&st, 0, sizeof(st));
        double t_gpu0 = now_s(), t_gpu1 = t_gpu0;
        if (!args.passthrough && n) {
            need_ctx();
            // The reference accepts every --data value and always runs Naive
            // (deduplicate_sam.rs:210-213): the result -- and here the path -- is the same for all of them.
            t_gpu0 = now_s();
            if (lib.dedup_batch(ctx, keys.data(), any_n ? nmask.data() : nullptr, n_words, freq.data(), off.data(), nb,
                                (int)umi_length, args.k, args.percentage, algo, 0 /* adjacency.rs:56 */,
                                kept.data(), args.track_clusters ? root.data() : nullptr, &st) != UMI_OK)
                die(lib.last_error());
            t_gpu1 = now_s();
        }
        // (the context is not put away: tearing the HIP runtime down costs a process that lives half
        // a second another 0.1 s, and the process ends below without running destructors)
        // --tag: cluster id / size per entry from the root of every entry.  Survivors in index
        // order are the roots in the order ClusterTracker::track sees them (bucket by bucket,
        // rank order inside), so offset + idx (cluster_tracker.rs:88-100, deduplicate_sam.rs:215)
        // is the running survivor count.
        std::vector<uint32_t> cluster_id, cluster_reads;
        std::vector<uint32_t> tagged; // staged reads in file order
        if (args.track_clusters) {
            cluster_id.assign(n, 0);
            cluster_reads.assign(n, 0);
            uint32_t next = 0;
            for (size_t i = 0; i < n; i++)
                if (kept[i]) cluster_id[i] = next++;
            for (size_t i = 0; i < n; i++) cluster_reads[root[i]] += (uint32_t)freq[i]; // temp_freq, :83-85
            for (uint32_t ri = 0; ri < n_rec; ri++)
                if (info[ri].state == 0) tagged.push_back(ri);
        } else if (!args.paired) {
            for (size_t i = 0; i < n; i++)
                if (kept[i]) out_records.push_back(rep[i]); // :227-231, in rank order per bucket
        } else {
            // UcWriter (:382-459): every written paired record leaves (qname, mate ref, mate pos)
            // in a set; when the reference name of the written records changes, and once at the
            // end, the input is scanned again in file order and the second mates found in the set
            // are written.  The file is in memory here, so the scans walk record indices (per
            // reference for the partial passes).  The reference's set hashes the coordinate but
            // compares only names (:288-296); here the coordinate is part of the identity.
            std::unordered_map<int32_t, std::vector<uint32_t>> mates_on; // tid -> second mates, file order
            std::vector<uint32_t> mates_all;
            for (uint32_t ri = 0; ri < n_rec; ri++) {
                const umi::bam::Record &r = in.records[ri];
                if (!r.is_unmapped() && r.is_paired() && r.is_last_in_template() && !r.is_mate_unmapped()) { // :425-429
                    mates_on[r.tid()].push_back(ri);
                    mates_all.push_back(ri);
                }
            }
            std::unordered_set<std::string> waiting;
            auto write_reversed = [&](const std::vector<uint32_t> &cands) {
                for (uint32_t ri : cands) {
                    if (waiting.empty()) break;
                    const umi::bam::Record &r = in.records[ri];
                    auto it = waiting.find(mate_key(r.qname(), r.qname_len(), r.tid(), r.pos()));
                    if (it != waiting.end()) {
                        out_records.push_back(ri);
                        waiting.erase(it);
                    }
                }
            };
            bool have_ref = false;
            int32_t cur_ref = 0;
            for (size_t i = 0; i < n; i++) {
                if (!kept[i]) continue;
                const umi::bam::Record &r = in.records[rep[i]];
                if (!have_ref) {
                    have_ref = true;
                } else if (cur_ref != r.tid()) {
                    auto m = mates_on.find(cur_ref);
                    if (m != mates_on.end()) write_reversed(m->second); // write_reversed(false), :390-393
                }
                cur_ref = r.tid();
                if (r.is_paired()) waiting.insert(mate_key(r.qname(), r.qname_len(), r.mtid(), r.mpos())); // :395-401
                out_records.push_back(rep[i]);
            }
            if (have_ref) write_reversed(mates_all); // close(), :411-415
        }

        // ---- write: header verbatim (Header::from_template :357-362) + surviving records verbatim
        constexpr size_t TAG_BYTES = 3 * 7; // three int32 aux fields
        lap("hot-path+select");
        if (tagged.empty()) {
            // the stream as pieces of the input (neighbouring survivors are one piece): the compressor
            // gathers each block's 64 KB itself, nothing is copied together first
            std::vector<umi::bgzf::Piece> pieces;
            pieces.reserve(out_records.size() / 2 + 2);
            pieces.push_back({in.data.data(), in.header_len});
            for (uint32_t ri : out_records) {
                const uint8_t *rb = in.records[ri].begin;
                const size_t len = (size_t)(in.records[ri].end - rb);
                if (pieces.back().p + pieces.back().len == rb) pieces.back().len += len;
                else pieces.push_back({rb, len});
            }
            umi::bgzf::compress_pieces_to_file(args.output, pieces, args.num_threads, args.compress_level);
        } else {
        size_t out_len = in.header_len;
        for (uint32_t ri : out_records) out_len += (size_t)(in.records[ri].end - in.records[ri].begin);
        for (uint32_t ri : tagged) out_len += (size_t)(in.records[ri].end - in.records[ri].begin) + TAG_BYTES;
        umi::bgzf::Bytes out(out_len);
        std::memcpy(out.data(), in.data.data(), in.header_len);
        size_t o = in.header_len;
        for (uint32_t ri : out_records) {
            const size_t len = (size_t)(in.records[ri].end - in.records[ri].begin);
            std::memcpy(out.data() + o, in.records[ri].begin, len);
            o += len;
        }
        for (uint32_t ri : tagged) {
            const size_t len = (size_t)(in.records[ri].end - in.records[ri].begin);
            std::memcpy(out.data() + o, in.records[ri].begin, len);
            const int32_t block_size = (int32_t)(len - 4 + TAG_BYTES);
            std::memcpy(out.data() + o, &block_size, 4);
            o += len;
            const AlignKey akey{info[ri].coord, info[ri].ref_strand, info[ri].tlen};
            const uint32_t e = global_of[hasher(akey) % T][entry_of[ri]];
            const uint32_t r = root[e];
            const struct { const char *tag; int32_t v; } aux[3] = {
                {"MI", (int32_t)cluster_id[r]}, {"cs", (int32_t)cluster_reads[r]}, {"su", freq[e]}};
            for (const auto &a : aux) {
                out[o++] = (uint8_t)a.tag[0];
                out[o++] = (uint8_t)a.tag[1];
                out[o++] = 'i';
                std::memcpy(out.data() + o, &a.v, 4);
                o += 4;
            }
        }
        umi::bgzf::compress_to_file(args.output, out.data(), out.size(), args.num_threads, args.compress_level);
        }
        lap("write");
        const double t_end = now_s();

        // counters of deduplicate_sam.rs:243-268
        std::fprintf(stderr, "Number of input reads: %zu\n", total_read_count);
        std::fprintf(stderr, "Number of removed unmapped reads: %zu\n", unmapped);
        if (args.paired) {
            std::fprintf(stderr, "Number of unpaired reads: %zu\n", unpaired);
            std::fprintf(stderr, "Number of chimeric reads: %zu\n", chimeric);
        }
        std::fprintf(stderr, "Number of unique alignment positions: %zu\n", nb);
        std::fprintf(stderr, "Number of UMIs: %zu\n", n);
        std::fprintf(stderr, "Average number of UMIs per alignment position: %g\n", nb ? (double)n / (double)nb : 0.0);
        std::fprintf(stderr, "Max number of UMIs over all alignment positions: %zu\n", max_umi);
        std::fprintf(stderr, args.track_clusters ? "Number of groups of reads: %llu\n" : "Number of reads after deduplicating: %llu\n",
                     (unsigned long long)st.n_kept); // :259-266
        std::fprintf(stderr,
                     "phases: read+inflate %.3f s, staging (%s) %.3f s, gpu init %.3f s, hot path (H2D+GPU+D2H) %.3f s [%llu pairs], write %.3f s\n",
                     t_read - t_start, gpu_stage ? "gpu" : "host", t_stage0 - t_read - (gpu_stage ? t_init : 0.0), t_init,
                     t_gpu1 - t_gpu0, (unsigned long long)st.n_pairs, t_end - t_gpu1);
        std::fprintf(stderr, "UMI collapsing finished in %.3f seconds\n", t_end - t_start); // main.rs:97-102
        leave(); // (from inside the scope of the file's buffers: they go with the process, unmapped by nobody)
    } catch (const std::exception &e) {
        die(e.what());
    }
    leave();
}
