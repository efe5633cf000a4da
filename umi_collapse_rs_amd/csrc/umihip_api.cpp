// Host side of libumihip.so: context, work planning, the C ABI of include/umihip.h.
// The product path has no CPU fallback: every entry point that computes needs the
// gfx950 kernels in this same library and a live HIP device.
#include "../../include/umihip.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "umihip_internal.h"
#include "umihip_plan.hpp"

#include <rccl/rccl.h> // (types only: the library itself is opened when the first collective is asked for)

using namespace umihip;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(UMI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return UMI_OK;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return fail(UMI_ERR_HIP, "hipFree: %s", hipGetErrorString(e));
        }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(e == hipErrorOutOfMemory ? UMI_ERR_NOMEM : UMI_ERR_HIP,
                        "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        }
        cap = want;
        return UMI_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T> T *as() const { return (T *)p; }
};

constexpr int MAX_ROUNDS = 1 << 20;
constexpr int DAG_ROUNDS = 3; // one-way rounds enqueued before the host first looks (freq at least
                              // halves along a one-way pair at p <= 0.5: depth 2 at config 2)

// grow-only pinned host buffer; every write goes through put(), which checks the extent
struct PinnedBuf {
    char *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return UMI_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc((void **)&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(UMI_ERR_NOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
        }
        cap = want;
        return UMI_OK;
    }
    int put(size_t off, const void *src, size_t bytes)
    {
        if (off > cap || bytes > cap - off)
            return fail(UMI_ERR_HIP, "internal: staging write of %zu bytes at %zu beyond %zu", bytes, off, cap);
        if (bytes) memcpy(p + off, src, bytes);
        return UMI_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace

// RCCL, opened on first use (dlopen: a process that never asks for a collective -- the umicollapse
// program, a single-GPU host -- does not map it; one that already holds a copy, PyTorch's, gets that
// copy instead of a second one beside it)
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool load()
    {
        if (handle) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) {
            error = std::string("cannot open librccl.so: ") + dlerror();
            return false;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(handle, name);
            if (!p && error.empty()) error = std::string("librccl.so lacks ") + name;
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!error.empty()) {
            handle = nullptr;
            return false;
        }
        return true;
    }
};

struct umi_ctx {
    // the communicators of a multi-device context's devices (umi_dedup_batch_device_multi), made on first use
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    // a multi-device context (umi_ctx_create_multi) owns one ordinary context per device and has
    // no device state of its own; an ordinary one has no subs
    std::vector<umi_ctx *> subs;
    // per-device staging of the sharded host-buffer call (pinned): this rank's entries gathered
    // from the caller's arrays, its share of the bucket table, its results
    PinnedBuf sh_in, sh_out;
    std::vector<uint64_t> sh_boff;
    int device = 0;
    hipStream_t own_stream = nullptr;
    bool profile = false;
    uint64_t edge_capacity = 1u << 20;
    uint64_t ovf_capacity = 1u << 18; // filter hits beyond the blocks' LDS queues (grows like the edge list)
    uint32_t small_max = 1024;
#ifdef UMIHIP_DEV
    static constexpr bool LEGACY_DEFAULT = true;
#else
    static constexpr bool LEGACY_DEFAULT = false; // the round-1 tile kernels are not in this build
#endif
    bool use_bitslice = LEGACY_DEFAULT;
    uint32_t bs_col_chunk = BS_COL_CHUNK;
    uint32_t bs_tab_min_run = 4; // table variant only where a run of equal high bases is about this long
    bool bs_transposed = true; // table variant: walk items with the columns of a run across the lanes
    uint32_t bs_tab_waves = 0; // one-wave blocks of the item walk (0: 128 per CU; items are dealt
                               // statically over them, the dispatcher evens out the rest)
    uint32_t fused_max = FUSED_MAX;
    uint32_t fused_blocks = 20; // 256-thread blocks per CU of the fused kernel's persistent grid
    bool fused_sliced = true;
    int bs_unit = 2;
    bool bs_sorted = LEGACY_DEFAULT; // sort large buckets by key and reuse prefix state along column runs
    bool bs_tables = LEGACY_DEFAULT; // ... and look the low units up in per-lane register tables (32-bit keys)
    int two_phase = 2; // directional collapse: 0 plain label propagation; 1 components of the symmetric
                       // pairs by hook/jump rounds, then the DAG; 2 the components by union-find
    bool seg_index = true;   // large buckets through the n-gram partition (umihip_seg.hip)
    uint32_t seg_min = 512;  // ... from this many entries up
    uint64_t split_min = 200000; // multi-device: a bucket at least this large that dominates the call
                                 // has its pairs split over the devices instead of the buckets
    uint32_t table_pieces = 1; // many buckets: the table is walked, uploaded and handed to the fused kernel in
                               // this many pieces (measured on 10^5 positions: one launch 0.13 ms, four 0.24)
    bool seg_ckey = true;    // its pair kernel compares 3-bit-per-base compare keys where they fit 32 bits
    bool seg_sliced = true;  // ... 64 columns at a time from ballots of the columns' code bits (k <= 3)
    bool seg_unite = true;   // its pair kernel unites symmetric pairs on the spot (batched directional path)
    bool seg_lds = true;     // counting sort of the partition through per-block LDS histograms (where
                             // every part has at most SEG_LDS_BINS bins), else one atomic per entry
    int seg_occ[2][2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};
    uint32_t seg_blocks = 0; // one-wave blocks of its pair kernel (0: all resident at once -- 24 per CU by
                             // registers, 28 by LDS with the compare keys' leaner loop)
    // workspace
    DevBuf fkey, thr, label, lab, edges, edge_dist, ovf, counters, boff, status, blocked;
    DevBuf plan_tables; // ranges, segment descriptors, scan chunks, popcount tile tasks: one upload
    DevBuf bs_tasks, plane_tasks, planes, tab_rows, tab_items;
    DevBuf seg_bin_cnt, seg_bin_start, seg_tasks, seg_sub_rec, seg_priv_edges, seg_priv_dist, seg_priv_cnt;
    PinnedBuf h_plan_alt[2]; // staging of the plan's tables, in turn; [plan_flip ^ 1] = what plan_tables holds
    int plan_flip = 0;
    size_t plan_uploaded = 0;
    const void *plan_uploaded_to = nullptr;
    int n_cus = 256;
    DevBuf fkey_sorted, perm, iota, sort_tmp, sample_pos, sample_out; // prune mode
    bool prune = false;
    Plan plan;
    TablePass table_pass;
    // staging for the host-buffer entry point
    DevBuf in_keys, in_nmask, in_freq, out_kept, out_root;
    // read staging (umi_stage_reads*): workspace; device copies of the host-buffer form
    DevBuf stage_ws, st_align, st_umi, st_score, st_keys, st_nmask, st_freq, st_rep, st_boff;
    uint64_t *h_boff = nullptr;               // pinned staging of the bucket table
    size_t h_boff_cap = 0;
    PinnedBuf h_tasks;                        // pinned staging of the bit-sliced tile-task lists
    unsigned long long *h_counters = nullptr; // pinned mirror of the control block (CTRL_BYTES), then, in a line of
                                              // its own, the sequence number of the last copy that has arrived
    unsigned long long ctrl_seq = 0;          // ... and of the last copy asked for
    int dag_rounds_ahead = DAG_ROUNDS;        // one-way rounds enqueued before the host first looks: as many as the
                                              // call before needed (a quiet round returns at once)
    // umi_dedup_batch_device_begin / umi_dedup_batch_end: a call whose work is all on the stream and whose
    // control block has not been looked at yet (deferred), or whose result waits to be handed out
    struct PendingCall {
        bool deferred = false, have_result = false;
        unsigned long long seq = 0;
        hipStream_t stream = nullptr;
        int rc = UMI_OK;
        umi_stats st;
    } pending;
    bool spin_wait = true;                    // watch that number instead of hipStreamSynchronize (option "spin_wait")
    unsigned long long *h_seq() const { return h_counters + CTRL_BYTES / sizeof(unsigned long long); }
    uint32_t *h_changed() const { return (uint32_t *)(h_counters + CNT_COUNT); }
    uint32_t *d_changed() const { return (uint32_t *)(counters.as<unsigned long long>() + CNT_COUNT); }
    static constexpr int N_EVENTS = 10;
    hipEvent_t ev[N_EVENTS] = {};
};

namespace {

int check_common(umi_ctx *ctx, const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int k,
                 int algo, uint64_t *n_out, int max_len = UMI_MAX_UMI_LEN)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!bucket_off) return fail(UMI_ERR_ARG, "bucket_off is NULL");
    if (umi_len < 1 || umi_len > max_len)
        return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, max_len);
    if (k < 0) return fail(UMI_ERR_ARG, "k must be >= 0 (got %d)", k);
    if (algo != UMI_ALGO_DIRECTIONAL && algo != UMI_ALGO_ADJACENCY)
        return fail(UMI_ERR_ARG, "unknown algo %d", algo);
    // (that the table is monotone is checked while it is copied to its pinned staging buffer,
    // before anything is launched: Pipeline::upload_and_prep)
    const uint64_t n = n_buckets ? bucket_off[n_buckets] : 0;
    if (n_buckets && bucket_off[0] != 0) return fail(UMI_ERR_ARG, "bucket_off[0] must be 0");
    if (n >= 0x7FFFFFF0ull) // bit 31 of an edge endpoint is a flag
        return fail(UMI_ERR_ARG, "%llu entries exceed the 31-bit index space of one call",
                    (unsigned long long)n);
    *n_out = n;
    return UMI_OK;
}

// rounds of one propagation phase until a whole round changes nothing; the per-round flags
// are checked on the device (a round after a quiet one returns at once), the host looks
// every `batch` rounds
template <class LaunchRound>
int run_rounds(umi_ctx *ctx, hipStream_t s, LaunchRound launch_round, int &rounds, int batch = 4)
{
    uint32_t *d_changed = ctx->d_changed();
    for (;;) {
        HIP_TRY(hipMemsetAsync(d_changed, 0, sizeof(uint32_t) * MAX_ROUNDS_PER_SYNC, s));
        for (int r = 0; r < batch; r++) HIP_TRY(launch_round(d_changed, r));
        HIP_TRY(hipMemcpyAsync(ctx->h_changed(), d_changed, sizeof(uint32_t) * batch, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        bool done = false;
        for (int r = 0; r < batch && !done; r++) {
            rounds++;
            done = ctx->h_changed()[r] == 0;
        }
        if (done) return UMI_OK;
        if (rounds > MAX_ROUNDS) return fail(UMI_ERR_HIP, "label propagation diverged");
        batch = std::min(2 * batch, MAX_ROUNDS_PER_SYNC);
    }
}

// label[v] = smallest rank that reaches v over the permitted-edge list (directional.rs:30-54,
// 78-88); ctx->label holds the start labels (v itself, or what the fused kernel left)
int directional_labels(umi_ctx *ctx, const uint2 *d_edges, unsigned long long *d_cnt, uint32_t edge_cap,
                       uint64_t n_edges, uint32_t n, hipStream_t s, int &rounds)
{
    int rc;
    uint32_t *d_label = ctx->label.as<uint32_t>();
#ifdef UMIHIP_DEV
    if (!ctx->two_phase)
        return run_rounds(ctx, s, [&](uint32_t *d_changed, int r) {
            return launch_prop_round(d_edges, d_cnt, edge_cap, d_label, n, d_changed, r, (uint32_t)n_edges, s);
        }, rounds);
#endif
    if ((rc = ctx->lab.reserve((size_t)n * 4))) return rc;
    uint32_t *d_lab = ctx->lab.as<uint32_t>();
#ifdef UMIHIP_DEV
    if (ctx->two_phase != 2) {
        HIP_TRY(launch_iota(d_lab, n, s));
        if ((rc = run_rounds(ctx, s, [&](uint32_t *d_changed, int r) {
                 return launch_cc_round(d_edges, d_cnt, edge_cap, d_label, n, d_changed, r, (uint32_t)n_edges, s);
             }, rounds, 6))) // a giant component of 10^6 entries settles in 5
            return rc;
    } else
#endif
    {
        HIP_TRY(launch_uf_components(d_edges, d_cnt, edge_cap, d_label, d_lab, n, (uint32_t)n_edges, s));
        rounds++;
    }
    if ((rc = run_rounds(ctx, s, [&](uint32_t *d_changed, int r) {
             return launch_dag_round(d_edges, d_cnt, edge_cap, d_label, d_lab, n, d_changed, r,
                                     (uint32_t)n_edges, s);
         }, rounds, 3))) // freq at least halves along a one-way pair at p <= 0.5
        return rc;
    HIP_TRY(launch_map_labels(d_label, d_lab, n, s));
    return UMI_OK;
}

// The device pipeline shared by both batched entry points and by umi_data_new, as a
// sequence of stages over one stream.  mode MODE_NEIGHBOURS stops after the pair kernels
// (the edge list then holds the neighbour pairs).
class Pipeline {
  public:
    Pipeline(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask, const int32_t *d_freq,
             const uint64_t *bucket_off, uint64_t n_buckets, uint32_t n, int umi_len, int k,
             float percentage, int mode, int32_t adj_max_freq, uint8_t *d_kept, uint32_t *d_root,
             hipStream_t s, uint32_t part = 0, uint32_t n_parts = 1)
        : ctx(ctx), d_keys(d_keys), d_nmask(d_nmask), d_freq(d_freq), bucket_off(bucket_off),
          n_buckets(n_buckets), n(n), umi_len(umi_len), k(k), percentage(percentage), mode(mode),
          adj_max_freq(adj_max_freq), d_kept(d_kept), d_root(d_root), s(s), pl(ctx->plan),
          key32(umi_len <= 16),
          need_pairs(!(mode == MODE_ADJACENCY && adj_max_freq < 1)), // reference adj: only the query goes
          // the fused kernel collapses whole buckets itself: not when only an edge list is wanted
          fused_max((mode == MODE_NEIGHBOURS || !need_pairs || n_parts > 1)
                        ? 0u
                        : std::min<uint32_t>(ctx->fused_max, FUSED_MAX)),
          prof(ctx->profile), part(part), n_parts(n_parts)
    {
        memset(&st, 0, sizeof(st));
        st.n_umis = n;
        st.n_buckets = n_buckets;
    }

    int run(umi_stats *stats)
    {
        HIP_TRY(hipSetDevice(ctx->device));
        int rc = run_stages();
        // Leave with the stream drained on every path: enqueued work reads caller memory
        // (bucket_off) and workspace buffers that the next call may reallocate.
        if (!drained) (void)hipStreamSynchronize(s);
        if (rc == UMI_OK && stats) *stats = st;
        return rc;
    }

    uint64_t edge_count() const { return n_edges; }
    // the caller keeps a copy of bucket_off in device memory (resident like keys and freq): no
    // staging copy, no upload
    void use_device_table(const uint64_t *d_table) { d_boff_caller = d_table; }
    // keys of n_words > 1 words per entry (umi_len > 21).  Filter keys and the segment index work on
    // the first word's 21 bases (two UMIs within k overall are within k there), every hit is decided
    // on all words; the fused kernel slices all words' bases; buckets in between go to the exact
    // all-pairs kernel of umihip_wide.hip.
    void use_wide_keys(int words) { n_words = words; }
    void allow_deferred_end() { may_defer = true; }

  private:
    umi_ctx *ctx;
    const uint64_t *d_keys, *d_nmask;
    const int32_t *d_freq;
    const uint64_t *bucket_off;
    const uint64_t *d_boff_caller = nullptr;
    int n_words = 1;
    bool wide() const { return n_words > 1; }
    int plan_umi_len() const { return wide() ? 21 : umi_len; } // bases the filter keys hold
    const uint64_t *d_boff() const { return d_boff_caller ? d_boff_caller : ctx->boff.as<uint64_t>(); }
    uint64_t n_buckets;
    uint32_t n;
    int umi_len, k;
    float percentage;
    int mode;
    int32_t adj_max_freq;
    uint8_t *d_kept;
    uint32_t *d_root;
    hipStream_t s;
    Plan &pl; // lives in the context: its vectors keep their capacity between calls
    const bool key32, need_pairs;
    uint32_t fused_max; // (0 for multi-word keys)
    const bool prof;
    const uint32_t part, n_parts; // n_parts > 1: evaluate only every n_parts-th tile task, stop
                                  // after the pair kernels (multi-GPU split of one call's pairs)
    uint64_t task_counter = 0;    // running index over all tile tasks, for that split
    bool prune = false, drained = false, fused_ran = false, seg_timed = false;
    bool may_defer = false; // umi_dedup_batch_device_begin: the end of the call may be left on the stream
    size_t zero_behind_control = 0; // bytes of the segment index's counters that sit behind the control block
    uint32_t priv_blocks_for_collapse = 0; // blocks of the segment index's pair kernel whose private edge slots the
                                           // collapse's flatten launch appends to the list (0: appended already)
    umi_stats st;
    unsigned long long *d_cnt = nullptr;
    size_t n_tasks = 0;              // tile tasks of the pair kernels (fused buckets excluded)
    uint64_t n_edges = 0;  // entries the pair kernels appended to the edge list
    uint64_t n_direct = 0; // symmetric pairs united where they were found
    uint64_t seg_tasks_made = 0;
    uint32_t cap_used = 0;
    const void *bs_fkey = nullptr;   // filter keys the bit-sliced tiles are cut from
    [[maybe_unused]] const uint32_t *bs_perm = nullptr;
    // device tables of the plan (inside ctx->plan_tables)
    const RangeTask *d_ranges = nullptr;
    const SegDesc *d_segs = nullptr;
    const SegScanChunk *d_chunks = nullptr;
    const PairTask *d_small = nullptr, *d_big = nullptr;
    const SegBlock *d_seg_blocks = nullptr;
    SegArgs seg;

    // The bit-sliced tile kernels of the earlier versions (options bs_sorted / bs_tables / prune,
    // k > the segment index's reach, seg_index = 0) keep their own overflow list, which the host
    // has to look at between the pair stage and the collapse.
#ifdef UMIHIP_DEV
    bool legacy_tiles() const { return pl.n_bs() || !pl.tab_rows.empty(); }
#else
    bool legacy_tiles() const { return false; }
#endif
    // Everything of a call enqueued back to back, one synchronisation at the end: the batched
    // directional path (and the reference's adjacency, which needs no pairs) without those tiles.
    bool one_sync() const
    {
        return (mode == MODE_DIRECTIONAL || !need_pairs) && n_parts == 1 && ctx->two_phase == 2 && !legacy_tiles();
    }

    int run_stages()
    {
        int rc;
        if (wide() && (mode != MODE_DIRECTIONAL || k > 3)) fused_max = 0; // (the wide fused kernel: directional, sliced)
        // The fused kernel needs nothing from the plan (it walks the bucket table itself and does
        // everything for its buckets): with many buckets it is enqueued first, and the host walks
        // the table -- tile tasks, pair counts, the entry ranges left for prep and finalize --
        // while it runs.  With few buckets the plan is there at once and says whether any bucket
        // is the fused kernel's at all.
        const bool plan_first = n_buckets <= 4096;
        if (plan_first) {
            // (a table that steps backwards must not reach the planner)
            for (uint64_t b = 0; b < n_buckets; b++)
                if (bucket_off[b + 1] < bucket_off[b])
                    return fail(UMI_ERR_ARG, "bucket_off not monotone at bucket %llu", (unsigned long long)b);
            plan_host();
            // the segment index's bin counters and scan words go behind the control block: one fill
            // clears both
            if (pl.seg_parts && need_pairs) zero_behind_control = seg_zero_bytes();
        }
        if ((rc = reserve_core())) return rc;
        if ((rc = upload_table(!plan_first))) return rc;
        if (plan_first && pl.n_fused && (rc = fused_stage(0, n_buckets))) return rc;
        if (!plan_first) plan_host();
        if (ctx->table_pass.bad_at != ~0ull)
            return fail(UMI_ERR_ARG, "bucket_off not monotone at bucket %llu", (unsigned long long)ctx->table_pass.bad_at);
        if ((rc = upload_plan())) return rc;
        if ((rc = prep_stage())) return rc;
#ifdef UMIHIP_DEV
        if (prune) {
            if ((rc = prune_stage())) return rc;
        } else {
            // the sorts are enqueued first: the host cuts the tile tasks (tens of thousands for a
            // deep position) while they run
            if (need_pairs && pl.any_sorted() && (rc = sort_stage())) return rc;
            gen_bs_tasks(pl, umi_len, ctx->bs_col_chunk, k, nullptr, key32);
            for (auto &v : pl.bs_tasks) keep_my_share(v);
        }
#endif
        if (need_pairs && pl.seg_parts)
            HIP_TRY(launch_seg_build(seg, ctx->fkey.p, d_freq, key32, d_cnt, s));
        if ((rc = upload_bitsliced())) return rc;
        if (one_sync()) return run_one_sync();
        if ((rc = pair_stage())) return rc;
        if (mode == MODE_NEIGHBOURS || n_parts > 1) return finish_neighbours();
        if (mode == MODE_DIRECTIONAL || !need_pairs)
            rc = collapse_directional();
        else
            rc = collapse_adjacency();
        if (rc) return rc;
        return finish();
    }

    // significant bits of a filter key: 2 per base in 32-bit keys, 3 in 64-bit ones
    int key_bits() const { return key32 ? 2 * umi_len : std::min(64, 3 * umi_len); }

    // keep this rank's share of a task list (round-robin over one running index: the tasks
    // of a list are similar in size and adjacent ones touch the same tiles)
    template <class T> void keep_my_share(std::vector<T> &v)
    {
        if (n_parts <= 1) return;
        size_t w = 0;
        for (size_t i = 0; i < v.size(); i++)
            if ((task_counter++ % n_parts) == part) v[w++] = v[i];
        v.resize(w);
    }

    // the tail of a call that may be left on the stream: the control copy with its sequence number, no wait
    // (umi_dedup_batch_device_begin; the caller's umi_dedup_batch_end looks)
    int defer_control()
    {
        ctx->pending.seq = ++ctx->ctrl_seq;
        ctx->pending.stream = s;
        HIP_TRY(launch_control_to_host(d_cnt, ctx->h_counters, CTRL_BYTES, s, ctx->h_seq(), ctx->pending.seq));
        ctx->pending.st = st;
        ctx->pending.deferred = true;
        drained = true; // (run() does not wait either)
        return UMI_OK;
    }
    int read_control()
    {
        if (ctx->spin_wait && !prof) { // (the phase profile reads events afterwards: they want the runtime's wait)
            // The control block's arrival is the end of the stream's work: watched for here (bounded: a
            // stream in error never delivers, and the runtime's wait below reports why)
            const unsigned long long seq = ++ctx->ctrl_seq;
            HIP_TRY(launch_control_to_host(d_cnt, ctx->h_counters, CTRL_BYTES, s, ctx->h_seq(), seq));
            const auto t0 = std::chrono::steady_clock::now();
            bool there = false;
            for (unsigned spins = 0; !there; spins++) {
                there = __atomic_load_n(ctx->h_seq(), __ATOMIC_ACQUIRE) == seq;
                if (!there && (spins & 1023u) == 1023u &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                    break;
            }
            if (!there) HIP_TRY(hipStreamSynchronize(s));
        } else {
            HIP_TRY(launch_control_to_host(d_cnt, ctx->h_counters, CTRL_BYTES, s));
            HIP_TRY(hipStreamSynchronize(s));
        }
        const unsigned long long bad =
            ctx->h_counters[CNT_ERROR] + (ctx->h_counters[CNT_RISES] - ctx->h_counters[CNT_START_RISES]);
        if (bad)
            return fail(UMI_ERR_ORDER,
                        "%llu entries break the input contract (freq < 1, not in freq-descending rank "
                        "order inside a bucket, or an N base without nmask)",
                        bad);
        return UMI_OK;
    }
    int sync_counters() { return read_control(); }

    // workspace whose size follows from n and n_buckets alone
    int reserve_core()
    {
        int rc;
        if ((rc = ctx->fkey.reserve((size_t)n * 8)) || (rc = ctx->thr.reserve((size_t)n * 4)) ||
            (rc = ctx->label.reserve((size_t)n * 4)) || (rc = ctx->lab.reserve((size_t)n * 4)) ||
            (rc = ctx->counters.reserve(CTRL_BYTES + zero_behind_control)) || (rc = ctx->boff.reserve((n_buckets + 1) * 8)))
            return rc;
        if (mode == MODE_ADJACENCY && need_pairs)
            if ((rc = ctx->status.reserve(n)) || (rc = ctx->blocked.reserve(n))) return rc;
        d_cnt = ctx->counters.as<unsigned long long>();
        bs_fkey = ctx->fkey.p;
        return UMI_OK;
    }

    // host planning: which kernel takes which bucket
    void plan_host()
    {
        // (the table is monotone: upload_table has looked)
        scan_table_range(bucket_off, 0, n_buckets, fused_max, nullptr, ctx->table_pass);
        const bool seg_on = ctx->seg_index && need_pairs && !ctx->prune;
        build_plan(bucket_off, n_buckets, wide() ? 0x7FFFFFFFu : ctx->small_max, ctx->use_bitslice && k <= BS_MAX_K && !wide(),
                   plan_umi_len(), fused_max, ctx->prune, ctx->bs_sorted && ctx->bs_unit == 2 && need_pairs,
                   ctx->bs_tables && key32, ctx->bs_tab_min_run, seg_on ? std::max(ctx->seg_min, 1u) : 0u, k, key32, pl,
                   &ctx->table_pass);
        prune = ctx->prune && need_pairs && !pl.bs_buckets.empty() && !wide();
        keep_my_share(pl.small_tasks);
        keep_my_share(pl.big_tasks);
        st.max_bucket = pl.max_bucket;
        st.n_pairs = pl.n_pairs;
    }

    size_t seg_bins_bytes() const { return (pl.seg_bins * 4 + 15) & ~(size_t)15; }
    size_t seg_zero_bytes() const
    {
        return (seg_bins_bytes() + (1 + pl.seg_chunks.size()) * sizeof(unsigned long long) + 15) & ~(size_t)15;
    }

    // the plan's tables -- entry ranges, segment descriptors and scan chunks, popcount tile tasks --
    // through one pinned block and one copy; workspace whose size the plan gives
    int upload_plan()
    {
        int rc;
        size_t off = 0;
        auto place = [&](size_t bytes) {
            const size_t at = off;
            off = (off + bytes + 63) & ~(size_t)63;
            return at;
        };
        const size_t o_ranges = place(pl.ranges.size() * sizeof(RangeTask));
        const size_t o_segs = place(pl.segs.size() * sizeof(SegDesc));
        const size_t o_chunks = place(pl.seg_chunks.size() * sizeof(SegScanChunk));
        const size_t o_blocks = place(pl.seg_blocks.size() * sizeof(SegBlock));
        const size_t o_small = place(pl.small_tasks.size() * sizeof(PairTask));
        const size_t o_big = place(pl.big_tasks.size() * sizeof(PairTask));
        const size_t total = std::max<size_t>(off, 64);
        // (two staging blocks in turn: what the previous call uploaded stays comparable -- a host that
        // calls again with the same bucket table, a resident job's next step, uploads nothing)
        PinnedBuf &hp = ctx->h_plan_alt[ctx->plan_flip];
        if ((rc = hp.reserve(total)) || (rc = ctx->plan_tables.reserve(total))) return rc;
        if (off) memset(hp.p, 0, off); // (padding between the tables: compared too)
        if ((rc = hp.put(o_ranges, pl.ranges.data(), pl.ranges.size() * sizeof(RangeTask))) ||
            (rc = hp.put(o_segs, pl.segs.data(), pl.segs.size() * sizeof(SegDesc))) ||
            (rc = hp.put(o_chunks, pl.seg_chunks.data(), pl.seg_chunks.size() * sizeof(SegScanChunk))) ||
            (rc = hp.put(o_blocks, pl.seg_blocks.data(), pl.seg_blocks.size() * sizeof(SegBlock))) ||
            (rc = hp.put(o_small, pl.small_tasks.data(), pl.small_tasks.size() * sizeof(PairTask))) ||
            (rc = hp.put(o_big, pl.big_tasks.data(), pl.big_tasks.size() * sizeof(PairTask))))
            return rc;
        const PinnedBuf &prev = ctx->h_plan_alt[ctx->plan_flip ^ 1];
        const bool same = off && ctx->plan_uploaded == off && ctx->plan_uploaded_to == ctx->plan_tables.p &&
                          prev.p && prev.cap >= off && memcmp(prev.p, hp.p, off) == 0;
        if (off && !same) {
            HIP_TRY(hipMemcpyAsync(ctx->plan_tables.p, hp.p, off, hipMemcpyHostToDevice, s));
            ctx->plan_uploaded = off;
            ctx->plan_uploaded_to = ctx->plan_tables.p;
            ctx->plan_flip ^= 1; // (this block now holds what the device holds)
        }
        const char *d = (const char *)ctx->plan_tables.p;
        d_ranges = (const RangeTask *)(d + o_ranges);
        d_segs = (const SegDesc *)(d + o_segs);
        d_chunks = (const SegScanChunk *)(d + o_chunks);
        d_seg_blocks = (const SegBlock *)(d + o_blocks);
        d_small = (const PairTask *)(d + o_small);
        d_big = (const PairTask *)(d + o_big);

        if ((rc = ctx->plane_tasks.reserve(std::max<size_t>(1, pl.plane_tasks.size()) * sizeof(PlaneTask))) ||
            (rc = ctx->planes.reserve(std::max<uint64_t>(1, pl.plane_words) * sizeof(uint32_t))))
            return rc;
        memset(&seg, 0, sizeof(seg));
        if (pl.seg_parts && need_pairs) {
            if (pl.seg_task_cap > 0x7FFFFFF0ull / sizeof(SegTask) || pl.seg_bins > 0x7FFFFFF0ull ||
                pl.seg_entries * (uint64_t)pl.seg_parts > 0x7FFFFFF0ull)
                return fail(UMI_ERR_NOMEM, "segment index of this call too large (set seg_index=0)");
            const size_t m = (size_t)pl.seg_entries * (size_t)pl.seg_parts;
            if (pl.segs.size() >= (1u << 24)) return fail(UMI_ERR_NOMEM, "too many segments in one call (set seg_index=0)");
            // (the bin counters and, behind them, the scan's ticket and status words: one block, one memset)
            const size_t bins_bytes = seg_bins_bytes(), zero_bytes = seg_zero_bytes();
            if ((!zero_behind_control && (rc = ctx->seg_bin_cnt.reserve(zero_bytes))) ||
                (rc = ctx->seg_bin_start.reserve(pl.seg_bins * 4)) ||
                (rc = ctx->seg_tasks.reserve(pl.seg_task_cap * sizeof(SegTask))) ||
                (rc = ctx->seg_sub_rec.reserve(m * sizeof(SegRec32))))
                return rc;
            static_assert(sizeof(SegRec32) == 16 && sizeof(SegRec64) == 16, "one 16-byte store per record");
            seg.segs = d_segs;
            seg.chunks = d_chunks;
            seg.n_chunks = (uint32_t)pl.seg_chunks.size();
            seg.n_parts = pl.seg_parts;
            seg.bin_cnt = zero_behind_control ? (uint32_t *)((char *)ctx->counters.p + CTRL_BYTES) : ctx->seg_bin_cnt.as<uint32_t>();
            seg.bin_start = ctx->seg_bin_start.as<uint32_t>();
            seg.scan_state = (unsigned long long *)((char *)seg.bin_cnt + bins_bytes);
            seg.tasks = ctx->seg_tasks.as<SegTask>();
            seg.task_cap = (uint32_t)pl.seg_task_cap;
            seg.sub_rec = ctx->seg_sub_rec.p;
            seg.ranges = d_ranges;
            seg.n_ranges = (uint32_t)pl.ranges.size();
            seg.umi_len = plan_umi_len();
            seg.key_words = n_words;
            seg.full_umi_len = umi_len;
            seg.use_ckey = key32 && ctx->seg_ckey && pl.seg_max_rest <= 10 ? 1u : 0u;
            seg.col_sliced = ctx->seg_sliced ? 1u : 0u;
            if (ctx->seg_lds && pl.seg_max_bins <= SEG_LDS_BINS && !pl.seg_blocks.empty()) {
                seg.blocks = d_seg_blocks;
                seg.n_blocks = (uint32_t)pl.seg_blocks.size();
                seg.lds_bins = pl.seg_max_bins;
                seg.parts_per_pass = std::max(1u, std::min({4u, (uint32_t)pl.seg_parts, SEG_LDS_BINS / std::max(1u, pl.seg_max_bins)}));
            }
            if (!zero_behind_control) HIP_TRY(hipMemsetAsync(seg.bin_cnt, 0, zero_bytes, s));
        }
        return UMI_OK;
    }

    // control block, bucket table
    // fuse: the table is walked in a few pieces, each uploaded and handed to the fused kernel while
    // the host walks the next (a batch of 10^5 small positions: the walk and the 0.8 MB copy are
    // what the kernel would otherwise wait for)
    int upload_table(bool fuse)
    {
        if (prof) HIP_TRY(hipEventRecord(ctx->ev[0], s));
        HIP_TRY(hipMemsetAsync(d_cnt, 0, CTRL_BYTES + zero_behind_control, s));
        // the bucket table goes through a pinned buffer: a true async DMA instead of the
        // runtime's staged copy of pageable memory (it is on the critical path of prep)
        if (ctx->h_boff_cap < n_buckets + 1) {
            if (ctx->h_boff) (void)hipHostFree(ctx->h_boff);
            ctx->h_boff = nullptr;
            ctx->h_boff_cap = 0;
            const size_t want = (n_buckets + 1) + (n_buckets + 1) / 4 + 64;
            HIP_TRY(hipHostMalloc((void **)&ctx->h_boff, want * 8));
            ctx->h_boff_cap = want;
        }
        if (mode == MODE_ADJACENCY && need_pairs) {
            HIP_TRY(hipMemsetAsync(ctx->status.p, 0, n, s));
            HIP_TRY(hipMemsetAsync(ctx->blocked.p, 0, n, s));
        }
        // What stands between the call and the first kernel is kept short: copy and monotonicity
        // (a table that steps backwards must not reach the device).  The counters and the list of
        // buckets beyond the fused kernel's reach are gathered by plan_host, behind the launch.
        TablePass &tp = ctx->table_pass;
        scan_table_reset(tp);
        const uint64_t pieces = fuse && fused_max >= 1 ? ctx->table_pieces : 1;
        ctx->h_boff[0] = bucket_off[0];
        for (uint64_t c = 0; c < pieces; c++) {
            const uint64_t b0 = n_buckets * c / pieces, b1 = n_buckets * (c + 1) / pieces;
            uint64_t prev = bucket_off[b0], back = 0;
            if (d_boff_caller) {
                // The table is on the device already: nothing to copy, and nothing to look at before
                // the launch either -- the fused kernel refuses entries that step backwards or lead
                // outside the arrays by itself, and the host's walk behind the launch (plan_host)
                // notes the first such bucket: run_stages fails the call on it.
            } else {
                for (uint64_t b = b0; b < b1; b++) {
                    const uint64_t v = bucket_off[b + 1];
                    back |= v < prev;
                    ctx->h_boff[b + 1] = prev = v;
                }
            }
            if (back) {
                for (uint64_t b = b0; b < b1; b++)
                    if (bucket_off[b + 1] < bucket_off[b])
                        return fail(UMI_ERR_ARG, "bucket_off not monotone at bucket %llu", (unsigned long long)b);
            }
            if (!d_boff_caller)
                HIP_TRY(hipMemcpyAsync(ctx->boff.as<uint64_t>() + b0, ctx->h_boff + b0, (b1 - b0 + 1) * 8,
                                       hipMemcpyHostToDevice, s));
            int rc;
            if (fuse && b1 > b0 && (rc = fused_stage(b0, b1))) return rc;
        }
        return UMI_OK;
    }

    // buckets [b0, b1) of the table
    int fused_stage(uint64_t b0, uint64_t b1)
    {
        if (fused_max < 1 || b1 <= b0) return UMI_OK;
        if (prof && !fused_ran) HIP_TRY(hipEventRecord(ctx->ev[5], s));
        // (label[] of a bucket the fused kernel finishes is read by nobody on the one-synchronisation
        // path -- its collapse covers the other buckets' ranges only: the store is left out there)
#ifdef UMIHIP_DEV
        const bool need_label = true; // (the tile kernels' collapse variants walk label[] of every entry)
#else
        const bool need_label = !((mode == MODE_DIRECTIONAL || !need_pairs) && n_parts == 1);
#endif
        if (wide())
            HIP_TRY(launch_small_buckets_wide(d_keys, d_nmask, n_words, d_freq, percentage, d_boff() + b0, (uint32_t)(b1 - b0),
                                              fused_max, n, d_kept, d_root, k, umi_len, d_cnt,
                                              (uint32_t)ctx->n_cus * ctx->fused_blocks, s));
        else
        HIP_TRY(launch_small_buckets(d_keys, d_nmask, d_freq, percentage, d_boff() + b0,
                                     (uint32_t)(b1 - b0), fused_max, n, need_label ? ctx->label.as<uint32_t>() : nullptr, d_kept,
                                     d_root, k, umi_len, ctx->fused_sliced, mode, adj_max_freq, d_cnt,
                                     (uint32_t)ctx->n_cus * ctx->fused_blocks, s));
        if (prof) HIP_TRY(hipEventRecord(ctx->ev[6], s));
        if (!fused_ran) st.n_pair_launches += 1;
        fused_ran = true;
        return UMI_OK;
    }

    // filter keys / thresholds / labels / contract check of the entries the fused kernel left;
    // the entries of a segment also count themselves into the bins of its parts
    int prep_stage()
    {
        // With the LDS counting sort the segments' entries are prepared by its count kernel; the
        // entry kernel keeps the other buckets' ranges (none at all for a call of deep positions:
        // only the rises at bucket starts are left to count)
        const bool fold = seg.blocks != nullptr;
        if (fold) {
            seg.prep_keys = d_keys;
            seg.prep_nmask = d_nmask;
            seg.prep_freq = d_freq;
            seg.prep_thr = ctx->thr.as<int32_t>();
            seg.prep_label = ctx->label.as<uint32_t>();
            seg.prep_percentage = percentage;
            seg.prep_counters = d_cnt;
        }
        bool other_ranges = false;
        for (const RangeTask &r : pl.ranges) other_ranges = other_ranges || r.seg == SEG_NONE;
        HIP_TRY(launch_prep(d_keys, d_nmask, d_freq, d_boff(), n_buckets, d_ranges,
                            (uint32_t)pl.ranges.size(), n, fused_max, plan_umi_len(), percentage, key32, ctx->fkey.p,
                            ctx->thr.as<int32_t>(), ctx->label.as<uint32_t>(), nullptr, d_cnt,
                            seg.n_chunks && !seg.blocks ? d_segs : nullptr, pl.seg_parts, seg.bin_cnt, s, fold,
                            !fold || other_ranges, std::max({ctx->seg_min, fused_max + 1, 1u}), n_words, umi_len));
        return UMI_OK;
    }

#ifdef UMIHIP_DEV
    // optional: sort every large bucket by filter key, read back the keys at the tile
    // boundaries, keep only the tile tasks whose key ranges can still hold a pair within k
    int prune_stage()
    {
        const size_t ksz = key32 ? 4 : 8;
        size_t tmp_bytes = 0, n_pos = 0;
        for (auto &bb : pl.bs_buckets) {
            tmp_bytes = std::max(tmp_bytes, sort_temp_bytes(key32, (uint32_t)(bb.e - bb.s), key_bits()));
            n_pos += (bb.e - bb.s + BS_COL_TILE - 1) / BS_COL_TILE + 1;
        }
        int rc;
        if ((rc = ctx->fkey_sorted.reserve((size_t)n * ksz)) || (rc = ctx->perm.reserve((size_t)n * 4)) ||
            (rc = ctx->iota.reserve((size_t)n * 4)) || (rc = ctx->sort_tmp.reserve(tmp_bytes)) ||
            (rc = ctx->sample_pos.reserve(n_pos * 4)) || (rc = ctx->sample_out.reserve(n_pos * 8)))
            return rc;
        std::vector<uint32_t> pos;
        pos.reserve(n_pos);
        for (auto &bb : pl.bs_buckets) {
            HIP_TRY(sort_bucket(ctx->fkey.p, key32, 0, key_bits(), (uint32_t)bb.s, (uint32_t)(bb.e - bb.s),
                                ctx->fkey_sorted.p, ctx->perm.as<uint32_t>(),
                                ctx->iota.as<uint32_t>(), ctx->sort_tmp.p, tmp_bytes, s));
            for (uint64_t q = bb.s; q < bb.e; q += BS_COL_TILE) pos.push_back((uint32_t)q);
            pos.push_back((uint32_t)(bb.e - 1));
        }
        std::vector<uint64_t> flat(pos.size());
        HIP_TRY(hipMemcpyAsync(ctx->sample_pos.p, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(gather_keys(ctx->fkey_sorted.p, key32, ctx->sample_pos.as<uint32_t>(),
                            (uint32_t)pos.size(), ctx->sample_out.as<uint64_t>(), s));
        HIP_TRY(hipMemcpyAsync(flat.data(), ctx->sample_out.p, pos.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<std::vector<uint64_t>> samples(pl.bs_buckets.size());
        size_t o = 0;
        for (size_t bi = 0; bi < pl.bs_buckets.size(); bi++) {
            const size_t cnt =
                (pl.bs_buckets[bi].e - pl.bs_buckets[bi].s + BS_COL_TILE - 1) / BS_COL_TILE + 1;
            samples[bi].assign(flat.begin() + o, flat.begin() + o + cnt);
            o += cnt;
        }
        gen_bs_tasks(pl, umi_len, ctx->bs_col_chunk, k, &samples, key32);
        for (auto &v : pl.bs_tasks) keep_my_share(v);
        bs_fkey = ctx->fkey_sorted.p;
        bs_perm = ctx->perm.as<uint32_t>();
        return UMI_OK;
    }

    // Buckets whose tiles reuse the counter state of the high units along runs of columns are
    // sorted by filter key (the runs come from the order); the other large buckets keep their
    // order behind an identity permutation, so that one pair of arrays serves every tile.
    int sort_stage()
    {
        const size_t ksz = key32 ? 4 : 8;
        size_t tmp_bytes = 0;
        bool all = true;
        for (auto &bb : pl.bs_buckets) {
            if (bb.pu || bb.live) tmp_bytes = std::max(tmp_bytes, sort_temp_bytes(key32, (uint32_t)(bb.e - bb.s), key_bits()));
            else all = false;
        }
        int rc;
        // + 64 B: the table kernel reads column keys a group ahead of the chunk it works on
        if ((rc = ctx->fkey_sorted.reserve((size_t)n * ksz + 64)) || (rc = ctx->perm.reserve((size_t)n * 4)) ||
            (rc = ctx->iota.reserve((size_t)n * 4)) || (rc = ctx->sort_tmp.reserve(tmp_bytes)))
            return rc;
        if (!all) {
            HIP_TRY(hipMemcpyAsync(ctx->fkey_sorted.p, ctx->fkey.p, (size_t)n * ksz, hipMemcpyDeviceToDevice, s));
            HIP_TRY(launch_iota(ctx->perm.as<uint32_t>(), n, s));
        }
        for (auto &bb : pl.bs_buckets)
            if (bb.pu || bb.live) {
                // The scan + item walk never look at the order inside a column run: the two live
                // units (8 bits) can stay unsorted, one radix pass less.  Only on the onesweep
                // path (the library's merge path for smaller inputs gets the full key).
                const int begin_bit = bb.live && sort_is_onesweep((uint32_t)(bb.e - bb.s)) ? 4 * bb.live : 0;
                HIP_TRY(sort_bucket(ctx->fkey.p, key32, begin_bit, key_bits(), (uint32_t)bb.s, (uint32_t)(bb.e - bb.s),
                                    ctx->fkey_sorted.p, ctx->perm.as<uint32_t>(),
                                    ctx->iota.as<uint32_t>(), ctx->sort_tmp.p, tmp_bytes, s));
            }
        bs_fkey = ctx->fkey_sorted.p;
        bs_perm = ctx->perm.as<uint32_t>();
        return UMI_OK;
    }

#endif // UMIHIP_DEV

    // bit-sliced tile tasks + bit planes of the large buckets the segment index does not take
    int upload_bitsliced()
    {
        n_tasks = pl.small_tasks.size() + pl.big_tasks.size() + pl.n_bs() + pl.tab_rows.size() +
                  (pl.seg_parts ? 1 : 0);
        if (need_pairs) st.n_pairs_evaluated = pl.n_pairs_eval;
#ifdef UMIHIP_DEV
        if (need_pairs && legacy_tiles()) {
            int rc;
            if (pl.tab_items_max > 0x7FFFFFF0ull / sizeof(TabItem))
                return fail(UMI_ERR_NOMEM, "item list of the table kernel too large (set bs_tables=0)");
            if ((rc = ctx->bs_tasks.reserve(pl.n_bs() * sizeof(BsTask))) ||
                (rc = ctx->tab_rows.reserve(pl.tab_rows.size() * sizeof(TabRowTile))) ||
                (rc = ctx->tab_items.reserve(pl.tab_items_max * sizeof(TabItem))))
                return rc;
            // through a pinned buffer (the runtime's staged copy of pageable memory runs at a
            // fifth of the DMA rate and holds the stream meanwhile)
            const size_t bs_bytes = pl.n_bs() * sizeof(BsTask);
            const size_t plane_bytes = pl.plane_tasks.size() * sizeof(PlaneTask);
            const size_t row_bytes = pl.tab_rows.size() * sizeof(TabRowTile);
            if ((rc = ctx->h_tasks.reserve(bs_bytes + plane_bytes + row_bytes))) return rc;
            const char *h = ctx->h_tasks.p;
            size_t off = 0;
            for (auto &v : pl.bs_tasks) {
                if ((rc = ctx->h_tasks.put(off, v.data(), v.size() * sizeof(BsTask)))) return rc;
                off += v.size() * sizeof(BsTask);
            }
            if (row_bytes) {
                if ((rc = ctx->h_tasks.put(bs_bytes + plane_bytes, pl.tab_rows.data(), row_bytes))) return rc;
                HIP_TRY(hipMemcpyAsync(ctx->tab_rows.p, h + bs_bytes + plane_bytes, row_bytes, hipMemcpyHostToDevice, s));
            }
            if ((rc = ctx->h_tasks.put(bs_bytes, pl.plane_tasks.data(), plane_bytes))) return rc;
            if (bs_bytes) HIP_TRY(hipMemcpyAsync(ctx->bs_tasks.p, h, bs_bytes, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(ctx->plane_tasks.p, h + bs_bytes, plane_bytes, hipMemcpyHostToDevice, s));
            HIP_TRY(launch_build_planes(bs_fkey, key32, ctx->plane_tasks.as<PlaneTask>(),
                                        (uint32_t)pl.plane_tasks.size(), ctx->planes.as<uint32_t>(),
                                        umi_len, s));
        }
#endif
        if (prof) HIP_TRY(hipEventRecord(ctx->ev[1], s));
        return UMI_OK;
    }

    // lists the pair kernels append to
    int reserve_lists(uint64_t cap, uint64_t &ovf_cap)
    {
        int rc;
        if (cap > 0x7FFFFFF0ull) return fail(UMI_ERR_NOMEM, "edge list too large");
        if ((rc = ctx->edges.reserve(cap * sizeof(uint2)))) return rc;
        if (mode == MODE_NEIGHBOURS && (rc = ctx->edge_dist.reserve(cap))) return rc;
        cap_used = (uint32_t)cap;
        ovf_cap = std::min<uint64_t>(std::max<uint64_t>(ctx->ovf_capacity, 1024), 0x7FFFFFF0ull);
        if (legacy_tiles() && (rc = ctx->ovf.reserve(ovf_cap * sizeof(uint2)))) return rc;
        return UMI_OK;
    }

    // resident one-wave blocks per CU of the segment index's pair kernel variant in use (asked of the
    // runtime once per variant and context; capped at 16, four waves per SIMD: the kernel is bound by
    // VALU issue and by the unions' trips to memory, and config 2 takes the same time with 16, 22 or
    // 24 blocks per CU.  The runtime's answer is an upper bound only: it says 28 for the 39-register
    // variant, but of 26 blocks per CU the last start only when the first have left -- s_memrealtime
    // stamps; 24 per CU are all running 1.5 us after the first -- and a late block of a persistent
    // grid does its whole share alone)
    uint32_t seg_occupancy()
    {
        const bool has_n = d_nmask != nullptr, ck = seg.use_ckey != 0;
        int &slot = ctx->seg_occ[key32 ? 1 : 0][has_n ? 1 : 0][ck ? 1 : 0];
        if (!slot) slot = seg_pair_blocks_per_cu(key32, has_n, ck);
        return (uint32_t)slot;
    }

    // all pair kernels of the call, largest work first (nothing waits on the host in here)
    int enqueue_pairs(uint64_t ovf_cap)
    {
        PairArgs a;
        a.keys = d_keys;
        a.nmask = d_nmask;
        a.freq = d_freq;
        a.thr = ctx->thr.as<int32_t>();
        a.fkey = ctx->fkey.p;
        a.tasks = d_small;
        a.bs_tasks = ctx->bs_tasks.as<BsTask>();
        a.planes = ctx->planes.as<uint32_t>();
        a.perm = nullptr;
        a.edges = ctx->edges.as<uint2>();
        a.edge_dist = ctx->edge_dist.as<uint8_t>();
        a.counters = d_cnt;
        a.ovf = ctx->ovf.as<uint2>();
        a.ovf_cap = (uint32_t)ovf_cap;
        a.edge_cap = cap_used;
        a.k = k;
        a.mode = mode;
        a.adj_max_freq = adj_max_freq;
        a.n_entries = n;
        if (pl.seg_parts) { // the large buckets' sub-buckets: persistent one-wave blocks
            const uint32_t blocks = std::max(1u, (uint32_t)std::min<uint64_t>(
                pl.seg_task_cap, ctx->seg_blocks ? ctx->seg_blocks : (uint64_t)ctx->n_cus * std::min(16u, seg_occupancy())));
            int rc;
            if ((rc = ctx->seg_priv_edges.reserve((size_t)blocks * SEG_PRIV_CAP * sizeof(uint2))) ||
                (rc = ctx->seg_priv_cnt.reserve((size_t)blocks * 4)) ||
                (mode == MODE_NEIGHBOURS && (rc = ctx->seg_priv_dist.reserve((size_t)blocks * SEG_PRIV_CAP))))
                return rc;
            seg.priv_edges = ctx->seg_priv_edges.as<uint2>();
            seg.priv_dist = ctx->seg_priv_dist.as<uint8_t>();
            seg.priv_cnt = ctx->seg_priv_cnt.as<uint32_t>();
            seg.uf_parent = one_sync() && ctx->seg_unite ? ctx->label.as<uint32_t>() : nullptr;

            if (prof) HIP_TRY(hipEventRecord(ctx->ev[7], s));
            HIP_TRY(launch_seg_pairs(a, seg, key32, percentage, part, n_parts, blocks, s));
            if (prof) HIP_TRY(hipEventRecord(ctx->ev[8], s));
            seg_timed = true;
            // what the blocks still hold in their private slots: one-way pairs only when the symmetric
            // ones were united where they were found, and then the collapse's flatten launch moves them
            // to the list (collapse_desc); else two small launches here, ahead of the list's unions
            priv_blocks_for_collapse = seg.uf_parent && mode == MODE_DIRECTIONAL ? blocks : 0u;
            if (!priv_blocks_for_collapse) HIP_TRY(launch_seg_edge_append(a, seg, blocks, s));
            st.n_pair_launches += 1;
        }
#ifdef UMIHIP_DEV
        PairArgs b = a; // bit-sliced tiles are cut from the key-sorted arrays
        b.fkey = bs_fkey;
        b.perm = bs_perm;
        if (!pl.tab_rows.empty()) // the table variant first (the largest buckets)
            HIP_TRY(launch_bs_tab(b, ctx->tab_rows.as<TabRowTile>(), (uint32_t)pl.tab_rows.size(),
                                  ctx->tab_items.as<TabItem>(), (uint32_t)pl.tab_items_max, umi_len, part,
                                  n_parts,
                                  ctx->bs_tab_waves ? ctx->bs_tab_waves
                                                    : (ctx->bs_transposed ? 128u : 16u) * (uint32_t)ctx->n_cus,
                                  ctx->bs_transposed, s));
        size_t first = pl.n_bs();
        for (int li = 3; li >= 0; li--) { // lists sit in the device array in index order
            first -= pl.bs_tasks[li].size();
            PairArgs w = b;
            w.bs_tasks = b.bs_tasks + first;
            HIP_TRY(launch_bs_pairs(w, (uint32_t)pl.bs_tasks[li].size(), li != 0, key32, umi_len,
                                    ctx->bs_unit, li == 2 ? 3 : (li == 3 ? 4 : 0), s));
        }
#endif
        if (wide()) { // every bucket as 64-row chunks against its later entries
            HIP_TRY(launch_wide_pairs(a, (uint32_t)pl.small_tasks.size(), n_words, s));
            st.n_pair_launches += pl.small_tasks.empty() ? 0 : 1;
            return UMI_OK;
        }
        PairArgs big = a;
        big.tasks = d_big;
        HIP_TRY(launch_pairs(big, (uint32_t)pl.big_tasks.size(), true, key32, s));
        HIP_TRY(launch_pairs(a, (uint32_t)pl.small_tasks.size(), false, key32, s));
        for (auto &v : pl.bs_tasks) st.n_pair_launches += v.empty() ? 0 : 1;
        st.n_pair_launches += pl.tab_rows.empty() ? 0 : 2;
        st.n_pair_launches += (pl.small_tasks.empty() ? 0 : 1) + (pl.big_tasks.empty() ? 0 : 1);
        return UMI_OK;
    }

    void note_pair_counters()
    {
        n_edges = ctx->h_counters[CNT_EDGES] + ctx->h_counters[CNT_EDGES_MOVED];
        n_direct = ctx->h_counters[CNT_UF_DIRECT];
        seg_tasks_made = ctx->h_counters[CNT_SEG_TASKS];
        st.n_candidates = ctx->h_counters[CNT_CANDIDATES];
        st.n_pairs_evaluated = pl.n_pairs_eval + ctx->h_counters[CNT_SEG_PAIRS];
        if (!pl.tab_rows.empty()) // the table kernel walks only the column tiles its scan kept
            st.n_pairs_evaluated = st.n_pairs_evaluated - pl.n_pairs_eval_tab +
                                   (ctx->h_counters[CNT_ITEMS] + ctx->h_counters[CNT_DIAG_ITEMS]) *
                                       (64ull * BS_TAB_G * 32) * BS_TAB_TILE;
    }

    // all-pairs: fused small buckets straight to label[]/status[]; tile kernels to the edge list
    // (redone once with a larger list if it overflowed: the exact count is known by then)
    int pair_stage()
    {
        if (!(need_pairs && n_tasks)) {
            if (prof) HIP_TRY(hipEventRecord(ctx->ev[2], s));
            return UMI_OK;
        }
        int rc;
        uint64_t cap = std::max<uint64_t>(ctx->edge_capacity, 1024);
        for (int attempt = 0;; attempt++) {
            uint64_t ovf_cap = 0;
            if ((rc = reserve_lists(cap, ovf_cap))) return rc;
            if ((rc = enqueue_pairs(ovf_cap))) return rc;
            if ((rc = sync_counters())) return rc;
            // filter hits that did not fit the blocks' LDS queues (very dense tiles): checked now,
            // or -- if their list ran over as well -- everything again with a longer list
            const uint64_t n_ovf = ctx->h_counters[CNT_OVF];
            bool redo = false;
            if (n_ovf > ovf_cap) {
                ctx->ovf_capacity = n_ovf + n_ovf / 8 + 1024;
                redo = true;
            }
#ifdef UMIHIP_DEV
            else if (n_ovf) {
                PairArgs b;
                memset(&b, 0, sizeof(b));
                b.keys = d_keys;
                b.nmask = d_nmask;
                b.freq = d_freq;
                b.thr = ctx->thr.as<int32_t>();
                b.fkey = bs_fkey;
                b.perm = bs_perm;
                b.edges = ctx->edges.as<uint2>();
                b.edge_dist = ctx->edge_dist.as<uint8_t>();
                b.counters = d_cnt;
                b.ovf = ctx->ovf.as<uint2>();
                b.ovf_cap = (uint32_t)ovf_cap;
                b.edge_cap = cap_used;
                b.k = k;
                b.mode = mode;
                b.adj_max_freq = adj_max_freq;
                b.n_entries = n;
                HIP_TRY(launch_verify_list(b, key32, (uint32_t)n_ovf, s));
                st.n_pair_launches += 1;
                if ((rc = sync_counters())) return rc;
            }
#endif
            if (prof && !redo) HIP_TRY(hipEventRecord(ctx->ev[2], s));
            note_pair_counters();
            if (pl.seg_parts && seg_tasks_made > seg.task_cap)
                return fail(UMI_ERR_HIP, "internal: %llu segment tasks for a list of %u", (unsigned long long)seg_tasks_made, seg.task_cap);
            if (!redo && n_edges <= cap) break;
            if (attempt >= 3) return fail(UMI_ERR_HIP, "edge list overflow persists");
            if (n_edges > cap) {
                cap = n_edges + n_edges / 16 + 1024;
                ctx->edge_capacity = cap;
            }
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_EDGES], 0, 2 * sizeof(unsigned long long), s));
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_OVF], 0, 5 * sizeof(unsigned long long), s)); // and the item counters
        }
        st.n_edges = n_edges;
        return UMI_OK;
    }

    // The batched directional path with one synchronisation: pair kernels, union-find over the
    // symmetric pairs, forest flattened, DAG_ROUNDS rounds along the one-way pairs, kept mask -- all
    // enqueued behind one another (the kernels read the edge count on the device; a round is a no-op
    // once the one before it was quiet), the control block read once.  What the host may find then:
    // the edge list ran over (longer list, pairs and collapse again), or the last round still moved a
    // label (more rounds, finalize again).
    CollapseDesc collapse_desc() const
    {
        CollapseDesc d;
        d.parent = ctx->label.as<uint32_t>();
        d.lab = ctx->lab.as<uint32_t>();
        d.edges = ctx->edges.as<uint2>();
        d.edge_cap = cap_used;
        d.ranges = d_ranges;
        d.n_ranges = (uint32_t)pl.ranges.size();
        d.n = n;
        d.kept = d_kept;
        d.root = d_root;
        d.counters = d_cnt;
        d.changed = ctx->d_changed();
        if (priv_blocks_for_collapse) {
            d.priv_edges = seg.priv_edges;
            d.priv_cnt = seg.priv_cnt;
            d.priv_blocks = priv_blocks_for_collapse;
        }
        return d;
    }
    int run_one_sync()
    {
        int rc;
        uint32_t *d_label = ctx->label.as<uint32_t>();
        uint32_t *d_changed = ctx->d_changed();
        const bool have_pairs = need_pairs && n_tasks;
        uint64_t cap = std::max<uint64_t>(ctx->edge_capacity, 1024);
        // rounds along the one-way pairs enqueued ahead of the host's look (the last as the check beside the
        // finalize pass): what the call before on this context needed -- a position of clusters (chains of
        // one-way pairs down a freq ladder) takes five or six where a uniform one takes three, and each look
        // of the host in between costs a wait and a second finalize pass
        const int ahead = std::min(std::max(ctx->dag_rounds_ahead, DAG_ROUNDS), MAX_ROUNDS_PER_SYNC);
        for (int attempt = 0;; attempt++) {
            if (have_pairs) {
                uint64_t ovf_cap = 0;
                if ((rc = reserve_lists(cap, ovf_cap))) return rc;
                if ((rc = enqueue_pairs(ovf_cap))) return rc;
            }
            if (prof) HIP_TRY(hipEventRecord(ctx->ev[2], s));
            if (have_pairs) {
                // symmetric pairs in the list: those of the tile kernels, and the segment index's
                // when it does not unite them itself
                if (!seg.uf_parent || legacy_tiles() || !pl.small_tasks.empty() || !pl.big_tasks.empty())
                    HIP_TRY(launch_uf_union_list(ctx->edges.as<uint2>(), d_cnt, cap_used, d_label, cap_used, s));
                const CollapseDesc cd = collapse_desc();
                HIP_TRY(launch_collapse_flatten(cd, s));
                // (the last of them as a check beside the finalize pass: in the common case it would change
                // nothing, and the kept mask stands)
                for (int r = 0; r < ahead - 1; r++) HIP_TRY(launch_collapse_round(cd, r, s));
                HIP_TRY(launch_collapse_finalize(cd, s, ahead - 1));
            } else { // no pair of this call reaches the edge list: every entry outside the fused buckets survives
                HIP_TRY(launch_finalize(d_label, d_ranges, (uint32_t)pl.ranges.size(), n, d_kept, d_root, d_cnt, s));
            }
            if (prof) {
                HIP_TRY(hipEventRecord(ctx->ev[3], s));
                HIP_TRY(hipEventRecord(ctx->ev[4], s));
            }
            // (a call without pair work -- every position the fused kernel's -- has nothing left to decide
            // on the host: its end may be left to the caller)
            if (may_defer && !have_pairs && !prof && ctx->spin_wait) {
                st.n_pairs_evaluated = pl.n_pairs_eval;
                return defer_control();
            }
            if ((rc = read_control())) return rc;
            note_pair_counters();
            if (pl.seg_parts && seg_tasks_made > seg.task_cap)
                return fail(UMI_ERR_HIP, "internal: %llu segment tasks for a list of %u", (unsigned long long)seg_tasks_made, seg.task_cap);
            if (n_edges <= cap || !have_pairs) break;
            if (attempt >= 3) return fail(UMI_ERR_HIP, "edge list overflow persists");
            // the list was too short: the exact count is known now
            cap = n_edges + n_edges / 16 + 1024;
            ctx->edge_capacity = cap;
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_EDGES], 0, 2 * sizeof(unsigned long long), s));
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_KEPT], 0, sizeof(unsigned long long), s));
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_UF_DIRECT], 0, 2 * sizeof(unsigned long long), s));
            HIP_TRY(hipMemsetAsync(d_changed, 0, CTRL_BYTES - CTRL_FLAGS_OFF, s)); // flags and sync words
            HIP_TRY(launch_iota(d_label, n, s)); // (the fused buckets' entries are finished: their labels are free)
        }
        st.n_edges = n_edges + n_direct;
        int rounds = have_pairs && st.n_edges ? 1 : 0;
        for (int r = 0; have_pairs && r < ahead; r++) rounds += (r == 0 || ctx->h_changed()[r - 1]) ? 1 : 0;
        const int rounds_first = rounds;
        if (have_pairs && ctx->h_changed()[ahead - 1]) { // a deeper chain of one-way pairs than that
            // (comp[] is flat and lab[] only ever falls: the rounds go on where the first ones stopped)
            const CollapseDesc cd = collapse_desc();
            if ((rc = run_rounds(ctx, s, [&](uint32_t *, int r) { return launch_collapse_round(cd, r, s); }, rounds, 4)))
                return rc;
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_KEPT], 0, sizeof(unsigned long long), s));
            HIP_TRY(launch_collapse_finalize(cd, s));
            if ((rc = read_control())) return rc;
        }
        // (rounds = 1 for the union-find pass + the rounds that ran, the last of them quiet)
        if (have_pairs) ctx->dag_rounds_ahead = std::max(DAG_ROUNDS, (rounds > rounds_first ? rounds : rounds_first) - 1);
        st.n_rounds = (uint32_t)rounds;
        return finish_stats();
    }

    int finish_neighbours()
    {
        if (!(need_pairs && n_tasks)) { // pair_stage did not read the counters back
            int rc = sync_counters();
            if (rc) return rc;
        }
        drained = true;
        if (prof) {
            // (the pair stage records its end marker after its last synchronisation)
            HIP_TRY(hipEventSynchronize(ctx->ev[2]));
            HIP_TRY(hipEventElapsedTime(&st.ms_prep, ctx->ev[0], ctx->ev[1]));
            HIP_TRY(hipEventElapsedTime(&st.ms_pairs, ctx->ev[1], ctx->ev[2]));
            st.ms_total = st.ms_prep + st.ms_pairs;
        }
        return UMI_OK;
    }

    // min-rank label propagation to the fixed point: rounds are enqueued in growing batches,
    // each round skips itself on the device once the previous one changed nothing
    int collapse_directional()
    {
        if (n_edges) {
            int rounds = 0;
            const int rc = directional_labels(ctx, ctx->edges.as<uint2>(), d_cnt, cap_used, n_edges, n, s, rounds);
            if (rc) return rc;
            st.n_rounds = (uint32_t)rounds;
        }
        if (prof) HIP_TRY(hipEventRecord(ctx->ev[3], s));
        HIP_TRY(launch_finalize(ctx->label.as<uint32_t>(), d_ranges, (uint32_t)pl.ranges.size(), n,
                                d_kept, d_root, d_cnt, s));
        return UMI_OK;
    }

    // adjacency with max_freq >= 1: greedy root loop, one decision level per iteration
    int collapse_adjacency()
    {
        uint8_t *d_status = ctx->status.as<uint8_t>();
        uint8_t *d_blocked = ctx->blocked.as<uint8_t>();
        int iters = 0;
        for (;;) {
            HIP_TRY(hipMemsetAsync(&d_cnt[CNT_UNKNOWN], 0, sizeof(unsigned long long), s));
            HIP_TRY(launch_adj_iter(ctx->edges.as<uint2>(), d_cnt, cap_used, d_status, d_blocked,
                                    ctx->label.as<uint32_t>(), n, d_cnt, (uint32_t)n_edges, s));
            HIP_TRY(hipMemcpyAsync(&ctx->h_counters[CNT_UNKNOWN], &d_cnt[CNT_UNKNOWN],
                                   sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            iters++;
            if (ctx->h_counters[CNT_UNKNOWN] == 0) break;
            if (iters > MAX_ROUNDS) return fail(UMI_ERR_HIP, "adjacency collapse diverged");
        }
        st.n_rounds = (uint32_t)iters;
        if (prof) HIP_TRY(hipEventRecord(ctx->ev[3], s));
        HIP_TRY(launch_adj_finalize(d_status, ctx->label.as<uint32_t>(), d_ranges,
                                    (uint32_t)pl.ranges.size(), n, d_kept, d_root, d_cnt, s));
        return UMI_OK;
    }

    int finish()
    {
        if (prof) HIP_TRY(hipEventRecord(ctx->ev[4], s));
        int rc = sync_counters();
        if (rc) return rc;
        return finish_stats();
    }

    // (the control block has just been read and the stream is idle)
    int finish_stats()
    {
        drained = true;
        st.n_kept = ctx->h_counters[CNT_KEPT] + ctx->h_counters[CNT_KEPT_FUSED];
        if (prof) {
            hipEvent_t *ev = ctx->ev;
            HIP_TRY(hipEventElapsedTime(&st.ms_prep, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&st.ms_pairs, ev[1], ev[2]));
            if (fused_ran) { // the fused kernel ran inside the prep window
                float ms_fused = 0.0f;
                HIP_TRY(hipEventElapsedTime(&ms_fused, ev[5], ev[6]));
                st.ms_prep -= ms_fused;
                st.ms_pairs += ms_fused;
            }
            HIP_TRY(hipEventElapsedTime(&st.ms_collapse, ev[2], ev[3]));
            HIP_TRY(hipEventElapsedTime(&st.ms_finalize, ev[3], ev[4]));
            HIP_TRY(hipEventElapsedTime(&st.ms_total, ev[0], ev[4]));
            // the kernel that does the call's pair work, by itself: the segment index's pair kernel
            // where a deep position is in the call, else the fused small-bucket kernel
            if (seg_timed) {
                HIP_TRY(hipEventElapsedTime(&st.ms_kernel, ev[7], ev[8]));
                st.kernel_id = UMI_KERNEL_SEG_PAIRS;
            } else if (fused_ran) {
                HIP_TRY(hipEventElapsedTime(&st.ms_kernel, ev[5], ev[6]));
                st.kernel_id = UMI_KERNEL_FUSED;
            }
        }
        return UMI_OK;
    }
};

// Collapse of an edge list that is already on the device (the multi-GPU split gathers the
// ranks' partial lists into one): labels start as the identity, then the same propagation /
// greedy passes as the single-call pipeline.
class EdgeCollapse {
  public:
    EdgeCollapse(umi_ctx *ctx, uint32_t n, const uint2 *d_edges, uint32_t n_edges, int mode,
                 uint8_t *d_kept, uint32_t *d_root, hipStream_t s)
        : ctx(ctx), n(n), d_edges(d_edges), n_edges(n_edges), mode(mode), d_kept(d_kept),
          d_root(d_root), s(s)
    {
    }
    int run(umi_stats *stats)
    {
        HIP_TRY(hipSetDevice(ctx->device));
        int rc = stages(stats);
        (void)hipStreamSynchronize(s);
        return rc;
    }

  private:
    umi_ctx *ctx;
    uint32_t n;
    const uint2 *d_edges;
    uint32_t n_edges;
    int mode;
    uint8_t *d_kept;
    uint32_t *d_root;
    hipStream_t s;

    int stages(umi_stats *stats)
    {
        int rc;
        if ((rc = ctx->label.reserve((size_t)n * 4)) || (rc = ctx->counters.reserve(CTRL_BYTES)))
            return rc;
        if (mode == MODE_ADJACENCY && ((rc = ctx->status.reserve(n)) || (rc = ctx->blocked.reserve(n))))
            return rc;
        unsigned long long *d_cnt = ctx->counters.as<unsigned long long>();
        memset(ctx->h_counters, 0, CNT_COUNT * sizeof(unsigned long long));
        ctx->h_counters[CNT_EDGES] = n_edges;
        HIP_TRY(hipMemcpyAsync(d_cnt, ctx->h_counters, CNT_COUNT * sizeof(unsigned long long),
                               hipMemcpyHostToDevice, s));
        HIP_TRY(launch_iota(ctx->label.as<uint32_t>(), n, s));
        umi_stats st;
        memset(&st, 0, sizeof(st));
        st.n_umis = n;
        st.n_edges = n_edges;
        if (mode == MODE_DIRECTIONAL) {
            if (n_edges) {
                int rounds = 0;
                if ((rc = directional_labels(ctx, d_edges, d_cnt, n_edges, n_edges, n, s, rounds))) return rc;
                st.n_rounds = (uint32_t)rounds;
            }
            HIP_TRY(launch_finalize(ctx->label.as<uint32_t>(), nullptr, 0, n, d_kept, d_root, d_cnt, s));
        } else {
            uint8_t *d_status = ctx->status.as<uint8_t>(), *d_blocked = ctx->blocked.as<uint8_t>();
            HIP_TRY(hipMemsetAsync(d_status, 0, n, s));
            HIP_TRY(hipMemsetAsync(d_blocked, 0, n, s));
            int iters = 0;
            for (;;) {
                HIP_TRY(hipMemsetAsync(&d_cnt[CNT_UNKNOWN], 0, sizeof(unsigned long long), s));
                HIP_TRY(launch_adj_iter(d_edges, d_cnt, n_edges, d_status, d_blocked,
                                        ctx->label.as<uint32_t>(), n, d_cnt, n_edges, s));
                HIP_TRY(hipMemcpyAsync(&ctx->h_counters[CNT_UNKNOWN], &d_cnt[CNT_UNKNOWN],
                                       sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                iters++;
                if (ctx->h_counters[CNT_UNKNOWN] == 0) break;
                if (iters > MAX_ROUNDS) return fail(UMI_ERR_HIP, "adjacency collapse diverged");
            }
            st.n_rounds = (uint32_t)iters;
            HIP_TRY(launch_adj_finalize(d_status, ctx->label.as<uint32_t>(), nullptr, 0, n, d_kept, d_root, d_cnt, s));
        }
        HIP_TRY(hipMemcpyAsync(ctx->h_counters, d_cnt, CNT_COUNT * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        st.n_kept = ctx->h_counters[CNT_KEPT];
        if (stats) *stats = st;
        return UMI_OK;
    }
};

void settle(umi_ctx *ctx);
int run_pipeline(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                 const int32_t *d_freq, const uint64_t *bucket_off, uint64_t n_buckets, uint32_t n,
                 int umi_len, int k, float percentage, int mode, int32_t adj_max_freq,
                 uint8_t *d_kept, uint32_t *d_root, hipStream_t s, umi_stats *stats,
                 const uint64_t *d_bucket_off = nullptr, int n_words = 1, bool may_defer = false)
{
    settle(ctx); // (a deferred call owns the workspace until its end has been seen)
    Pipeline p(ctx, d_keys, d_nmask, d_freq, bucket_off, n_buckets, n, umi_len, k, percentage, mode, adj_max_freq,
               d_kept, d_root, s);
    p.use_device_table(d_bucket_off);
    p.use_wide_keys(n_words);
    if (may_defer) p.allow_deferred_end();
    return p.run(stats);
}

// the end of a deferred call: its control block has arrived (watched for, as in Pipeline::read_control),
// the contract verdict, the counts
int finish_pending(umi_ctx *ctx, umi_stats *stats)
{
    umi_ctx::PendingCall &pc = ctx->pending;
    if (pc.deferred) {
        pc.deferred = false;
        HIP_TRY(hipSetDevice(ctx->device));
        const auto t0 = std::chrono::steady_clock::now();
        bool there = false;
        for (unsigned spins = 0; !there; spins++) {
            there = __atomic_load_n(ctx->h_seq(), __ATOMIC_ACQUIRE) == pc.seq;
            if (!there && (spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
        if (!there) HIP_TRY(hipStreamSynchronize(pc.stream));
        const unsigned long long bad =
            ctx->h_counters[CNT_ERROR] + (ctx->h_counters[CNT_RISES] - ctx->h_counters[CNT_START_RISES]);
        pc.rc = UMI_OK;
        if (bad)
            pc.rc = fail(UMI_ERR_ORDER,
                         "%llu entries break the input contract (freq < 1, not in freq-descending rank "
                         "order inside a bucket, or an N base without nmask)",
                         bad);
        pc.st.n_candidates = ctx->h_counters[CNT_CANDIDATES];
        pc.st.n_kept = ctx->h_counters[CNT_KEPT] + ctx->h_counters[CNT_KEPT_FUSED];
        pc.have_result = true;
    }
    if (!pc.have_result) return fail(UMI_ERR_ARG, "umi_dedup_batch_end without umi_dedup_batch_device_begin");
    pc.have_result = false;
    if (pc.rc == UMI_OK && stats) *stats = pc.st;
    return pc.rc;
}
// every other entry point that touches the context's workspace first lets a deferred call end (its
// result keeps waiting for umi_dedup_batch_end)
void settle(umi_ctx *ctx)
{
    if (ctx && ctx->pending.deferred) {
        umi_stats st;
        (void)finish_pending(ctx, &st);
        ctx->pending.have_result = true;
    }
}


// ---- multi-device host-buffer call ------------------------------------------------------------
// Buckets are independent (src/deduplicate_sam.rs:207-233 shares nothing between iterations but
// additive counters), so a call shards with no exchange between the devices: every bucket goes to
// one device (longest processing time first on n_b^2, partition_buckets_lpt), a host thread per
// device gathers its buckets' entries into pinned staging, runs the ordinary pipeline on its
// device's own stream and workspace, and scatters kept / root back to the caller's arrays at the
// buckets' own offsets.  One giant bucket does not shard that way: then every device evaluates its
// share of the bucket's sub-bucket tasks (umi_pairs_partial_device's path), the edge lists are
// copied to the first device and collapsed there.
struct ShardResult {
    int rc = UMI_OK;
    std::string err;
    umi_stats st;
};

void merge_stats(umi_stats &a, const umi_stats &b)
{
    a.n_kept += b.n_kept;
    a.n_pairs_evaluated += b.n_pairs_evaluated;
    a.n_candidates += b.n_candidates;
    a.n_edges += b.n_edges;
    a.n_rounds = std::max(a.n_rounds, b.n_rounds);
    a.n_pair_launches += b.n_pair_launches;
    a.ms_total = std::max(a.ms_total, b.ms_total);
    a.ms_prep = std::max(a.ms_prep, b.ms_prep);
    a.ms_pairs = std::max(a.ms_pairs, b.ms_pairs);
    a.ms_collapse = std::max(a.ms_collapse, b.ms_collapse);
    a.ms_finalize = std::max(a.ms_finalize, b.ms_finalize);
}

int dedup_batch_single(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                       const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int k, float percentage,
                       int algo, int32_t adj_max_freq, uint8_t *kept, uint32_t *root, umi_stats *stats, int n_words = 1);

// one device's share of a bucket-sharded call (runs on its own host thread)
void run_shard(umi_ctx *sub, const std::vector<uint64_t> &mine, const uint64_t *keys, const uint64_t *nmask,
               const int32_t *freq, const uint64_t *bucket_off, int umi_len, int k, float percentage, int algo,
               int32_t adj_max_freq, uint8_t *kept, uint32_t *root, ShardResult &res, int n_words)
{
    const size_t kw = 8 * (size_t)n_words; // bytes of a key (entry-major: the words of an entry are adjacent)
    memset(&res.st, 0, sizeof(res.st));
    uint64_t n_local = 0;
    sub->sh_boff.assign(1, 0);
    for (uint64_t b : mine) {
        n_local += bucket_off[b + 1] - bucket_off[b];
        sub->sh_boff.push_back(n_local);
    }
    if (n_local == 0) return;
    // pinned staging: keys | nmask | freq in, kept | root out
    const size_t o_keys = 0, o_nmask = o_keys + n_local * kw, o_freq = o_nmask + (nmask ? n_local * kw : 0);
    const size_t in_bytes = o_freq + n_local * 4;
    const size_t o_kept = 0, o_root = (n_local + 15) & ~(size_t)15, out_bytes = o_root + (root ? n_local * 4 : 0);
    auto fail_here = [&](int rc) {
        res.rc = rc;
        res.err = umi_last_error(); // (this thread's message)
    };
    if (hipSetDevice(sub->device) != hipSuccess) {
        res.rc = UMI_ERR_HIP;
        res.err = "hipSetDevice failed on a worker thread";
        return;
    }
    int rc;
    // UMIHIP_TIMING=1: where the wall time of a sharded host-buffer call goes, per device, on stderr
    const bool timing = getenv("UMIHIP_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if ((rc = sub->sh_in.reserve(in_bytes)) || (rc = sub->sh_out.reserve(out_bytes))) return fail_here(rc);
    uint64_t at = 0;
    for (uint64_t b : mine) {
        const uint64_t s0 = bucket_off[b], len = bucket_off[b + 1] - s0;
        memcpy(sub->sh_in.p + o_keys + at * kw, keys + s0 * (size_t)n_words, len * kw);
        if (nmask) memcpy(sub->sh_in.p + o_nmask + at * kw, nmask + s0 * (size_t)n_words, len * kw);
        memcpy(sub->sh_in.p + o_freq + at * 4, freq + s0, len * 4);
        at += len;
    }
    umi_stats st;
    const double t1 = now();
    rc = dedup_batch_single(sub, (const uint64_t *)(sub->sh_in.p + o_keys),
                            nmask ? (const uint64_t *)(sub->sh_in.p + o_nmask) : nullptr,
                            (const int32_t *)(sub->sh_in.p + o_freq), sub->sh_boff.data(), mine.size(), umi_len, k,
                            percentage, algo, adj_max_freq, (uint8_t *)(sub->sh_out.p + o_kept),
                            root ? (uint32_t *)(sub->sh_out.p + o_root) : nullptr, &st, n_words);
    if (rc) return fail_here(rc);
    res.st = st;
    const double t2 = now();
    // back to the caller's index space: a bucket's entries keep their order, a root index moves
    // with its bucket
    at = 0;
    for (uint64_t b : mine) {
        const uint64_t s0 = bucket_off[b], len = bucket_off[b + 1] - s0;
        memcpy(kept + s0, sub->sh_out.p + o_kept + at, len);
        if (root) {
            const uint32_t *src = (const uint32_t *)(sub->sh_out.p + o_root) + at;
            const uint32_t shift = (uint32_t)(s0 - at); // (mod 2^32: local index + shift = global index)
            for (uint64_t i = 0; i < len; i++) root[s0 + i] = src[i] + shift;
        }
        at += len;
    }
    if (timing)
        fprintf(stderr, "umihip multi: device %d: %llu entries in %zu buckets: host gather %.4f s, call (H2D + GPU + D2H) "
                        "%.4f s, host scatter %.4f s\n",
                sub->device, (unsigned long long)n_local, mine.size(), t1 - t0, t2 - t1, now() - t2);
}

// the multi-device split of a call whose work is one giant bucket
int dedup_batch_split(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                      const uint64_t *bucket_off, uint64_t n_buckets, uint64_t n, int umi_len, int k,
                      float percentage, int algo, int32_t adj_max_freq, uint8_t *kept, uint32_t *root,
                      umi_stats *stats)
{
    const uint32_t n_dev = (uint32_t)ctx->subs.size();
    const int mode = algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY;
    std::vector<ShardResult> res(n_dev);
    std::vector<uint64_t> n_edges(n_dev, 0);
    std::vector<std::thread> pool;
    for (uint32_t r = 0; r < n_dev; r++)
        pool.emplace_back([&, r] {
            umi_ctx *sub = ctx->subs[r];
            ShardResult &out = res[r];
            auto fail_here = [&](int rc) {
                out.rc = rc;
                out.err = umi_last_error();
            };
            memset(&out.st, 0, sizeof(out.st));
            if (hipSetDevice(sub->device) != hipSuccess) {
                out.rc = UMI_ERR_HIP;
                out.err = "hipSetDevice failed on a worker thread";
                return;
            }
            int rc;
            if ((rc = sub->in_keys.reserve(n * 8)) || (rc = sub->in_freq.reserve(n * 4)) ||
                (nmask && (rc = sub->in_nmask.reserve(n * 8))))
                return fail_here(rc);
            hipStream_t s = sub->own_stream;
            hipError_t e = hipMemcpyAsync(sub->in_keys.p, keys, n * 8, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(sub->in_freq.p, freq, n * 4, hipMemcpyHostToDevice, s);
            if (e == hipSuccess && nmask) e = hipMemcpyAsync(sub->in_nmask.p, nmask, n * 8, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) {
                out.rc = UMI_ERR_HIP;
                out.err = std::string("upload failed: ") + hipGetErrorString(e);
                return;
            }
            Pipeline p(sub, sub->in_keys.as<uint64_t>(), nmask ? sub->in_nmask.as<uint64_t>() : nullptr,
                       sub->in_freq.as<int32_t>(), bucket_off, n_buckets, (uint32_t)n, umi_len, k, percentage, mode,
                       adj_max_freq, nullptr, nullptr, s, r, n_dev);
            if ((rc = p.run(&out.st))) return fail_here(rc);
            n_edges[r] = p.edge_count();
        });
    for (auto &t : pool) t.join();
    for (uint32_t r = 0; r < n_dev; r++)
        if (res[r].rc) return fail(res[r].rc, "device %d: %s", ctx->subs[r]->device, res[r].err.c_str());
    // the edge lists to the first device (peer copy, or through the host where peers cannot see
    // each other), the collapse there
    umi_ctx *c0 = ctx->subs[0];
    uint64_t total = 0;
    for (uint64_t e : n_edges) total += e;
    if (total >= 0x7FFFFFF0ull) return fail(UMI_ERR_NOMEM, "gathered edge list too large");
    HIP_TRY(hipSetDevice(c0->device));
    int rc;
    if ((rc = c0->out_kept.reserve(n)) || (rc = c0->out_root.reserve(n * 4))) return rc;
    DevBuf gathered;
    if ((rc = gathered.reserve(std::max<uint64_t>(total, 1) * sizeof(uint2)))) return rc;
    uint64_t at = 0;
    hipError_t e = hipSuccess;
    for (uint32_t r = 0; r < n_dev && e == hipSuccess; r++) {
        if (!n_edges[r]) continue;
        e = hipMemcpy((char *)gathered.p + at * sizeof(uint2), ctx->subs[r]->edges.p, n_edges[r] * sizeof(uint2),
                      hipMemcpyDefault);
        at += n_edges[r];
    }
    if (e != hipSuccess) {
        gathered.release();
        return fail(UMI_ERR_HIP, "edge gather failed: %s", hipGetErrorString(e));
    }
    umi_stats st;
    memset(&st, 0, sizeof(st));
    rc = EdgeCollapse(c0, (uint32_t)n, gathered.as<uint2>(), (uint32_t)total, mode, c0->out_kept.as<uint8_t>(),
                      root ? c0->out_root.as<uint32_t>() : nullptr, c0->own_stream)
             .run(&st);
    if (rc == UMI_OK) {
        e = hipMemcpy(kept, c0->out_kept.p, n, hipMemcpyDeviceToHost);
        if (e == hipSuccess && root) e = hipMemcpy(root, c0->out_root.p, n * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(UMI_ERR_HIP, "download failed: %s", hipGetErrorString(e));
    }
    gathered.release();
    if (rc) return rc;
    if (stats) {
        umi_stats total_st = res[0].st;
        for (uint32_t r = 1; r < n_dev; r++) merge_stats(total_st, res[r].st);
        total_st.n_umis = n;
        total_st.n_buckets = n_buckets;
        total_st.max_bucket = res[0].st.max_bucket;
        total_st.n_pairs = res[0].st.n_pairs;
        total_st.n_kept = st.n_kept;
        total_st.n_edges = total;
        total_st.n_rounds = st.n_rounds;
        *stats = total_st;
    }
    return UMI_OK;
}

int dedup_batch_multi(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                      const uint64_t *bucket_off, uint64_t n_buckets, uint64_t n, int umi_len, int k,
                      float percentage, int algo, int32_t adj_max_freq, uint8_t *kept, uint32_t *root,
                      umi_stats *stats, int n_words = 1)
{
    const uint32_t n_dev = (uint32_t)ctx->subs.size();
    // one bucket with more than half of the call's n_b^2: its pairs are split, not the buckets
    // (only where pairs are evaluated at all: the reference's adjacency needs none)
    long double cost = 0, top = 0;
    uint64_t max_bucket = 0;
    for (uint64_t b = 0; b < n_buckets; b++) {
        const long double sz = (long double)(bucket_off[b + 1] - bucket_off[b]);
        cost += sz * sz;
        top = std::max(top, sz * sz);
        max_bucket = std::max<uint64_t>(max_bucket, (uint64_t)sz);
    }
    const bool need_pairs = !(algo == UMI_ALGO_ADJACENCY && adj_max_freq < 1);
    if (n_dev > 1 && need_pairs && max_bucket >= ctx->split_min && top * 2 > cost && n_words == 1)
        return dedup_batch_split(ctx, keys, nmask, freq, bucket_off, n_buckets, n, umi_len, k, percentage, algo,
                                 adj_max_freq, kept, root, stats);
    std::vector<uint32_t> owner;
    partition_buckets_lpt(bucket_off, n_buckets, n_dev, owner);
    std::vector<std::vector<uint64_t>> mine(n_dev);
    for (uint64_t b = 0; b < n_buckets; b++) mine[owner[b]].push_back(b);
    std::vector<ShardResult> res(n_dev);
    std::vector<std::thread> pool;
    for (uint32_t r = 0; r < n_dev; r++)
        pool.emplace_back([&, r] {
            run_shard(ctx->subs[r], mine[r], keys, nmask, freq, bucket_off, umi_len, k, percentage, algo, adj_max_freq,
                      kept, root, res[r], n_words);
        });
    for (auto &t : pool) t.join();
    for (uint32_t r = 0; r < n_dev; r++)
        if (res[r].rc) return fail(res[r].rc, "device %d: %s", ctx->subs[r]->device, res[r].err.c_str());
    if (stats) {
        umi_stats total = res[0].st;
        for (uint32_t r = 1; r < n_dev; r++) {
            merge_stats(total, res[r].st);
            total.max_bucket = std::max(total.max_bucket, res[r].st.max_bucket);
            total.n_pairs += res[r].st.n_pairs;
        }
        total.n_umis = n;
        total.n_buckets = n_buckets;
        *stats = total;
    }
    return UMI_OK;
}

int dedup_batch_single(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                    const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int k,
                    float percentage, int algo, int32_t adj_max_freq, uint8_t *kept, uint32_t *root,
                    umi_stats *stats, int n_words)
{
    uint64_t n = 0;
    int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n, n_words > 1 ? UMI_MAX_WIDE_UMI_LEN : UMI_MAX_UMI_LEN);
    if (rc) return rc;
    if (n && (!keys || !freq || !kept)) return fail(UMI_ERR_ARG, "keys/freq/kept is NULL");
    if (n == 0) {
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->n_buckets = n_buckets;
        }
        return UMI_OK;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t kb = (size_t)n * 8 * (size_t)n_words;
    if ((rc = ctx->in_keys.reserve(kb)) || (rc = ctx->in_freq.reserve(n * 4)) ||
        (rc = ctx->out_kept.reserve(n)) || (rc = ctx->out_root.reserve(n * 4)))
        return rc;
    if (nmask && (rc = ctx->in_nmask.reserve(kb))) return rc;
    hipStream_t s = ctx->own_stream;
    HIP_TRY(hipMemcpyAsync(ctx->in_keys.p, keys, kb, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->in_freq.p, freq, n * 4, hipMemcpyHostToDevice, s));
    if (nmask) HIP_TRY(hipMemcpyAsync(ctx->in_nmask.p, nmask, kb, hipMemcpyHostToDevice, s));
    rc = run_pipeline(ctx, ctx->in_keys.as<uint64_t>(),
                      nmask ? ctx->in_nmask.as<uint64_t>() : nullptr, ctx->in_freq.as<int32_t>(),
                      bucket_off, n_buckets, (uint32_t)n, umi_len, k, percentage,
                      algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY,
                      adj_max_freq, ctx->out_kept.as<uint8_t>(),
                      root ? ctx->out_root.as<uint32_t>() : nullptr, s, stats, nullptr, n_words);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(kept, ctx->out_kept.p, n, hipMemcpyDeviceToHost, s));
    if (root) HIP_TRY(hipMemcpyAsync(root, ctx->out_root.p, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return UMI_OK;
}


} // namespace

extern "C" {

const char *umi_last_error(void) { return g_err.c_str(); }

int umi_abi_version(void) { return UMI_ABI_VERSION; }

int umi_ctx_create(int device_id, umi_ctx **out)
{
    if (!out) return fail(UMI_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(UMI_ERR_NODEV, "no HIP device visible (%s)",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n_dev)
        return fail(UMI_ERR_ARG, "device_id %d outside 0..%d", device_id, n_dev - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(UMI_ERR_NODEV, "device %d is %s; this library carries gfx950 code only",
                    device_id, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));
    umi_ctx *ctx = new (std::nothrow) umi_ctx();
    if (!ctx) return fail(UMI_ERR_NOMEM, "out of host memory");
    ctx->device = device_id;
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t err = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (err == hipSuccess)
        err = hipHostMalloc((void **)&ctx->h_counters, CTRL_BYTES + 64);
    if (err == hipSuccess) *ctx->h_seq() = 0;
    for (int i = 0; i < umi_ctx::N_EVENTS && err == hipSuccess; i++) err = hipEventCreate(&ctx->ev[i]);

    if (err != hipSuccess) {
        umi_ctx_destroy(ctx);
        return fail(UMI_ERR_HIP, "context setup failed: %s", hipGetErrorString(err));
    }
    *out = ctx;
    return UMI_OK;
}

void umi_ctx_destroy(umi_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->subs.empty()) settle(ctx);
    if (!ctx->subs.empty()) { // a multi-device context: nothing of its own on a device
        for (ncclComm_t c : ctx->comms)
            if (c && ctx->rccl.CommDestroy) (void)ctx->rccl.CommDestroy(c);
        for (umi_ctx *sub : ctx->subs) umi_ctx_destroy(sub);
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    ctx->sh_in.release();
    ctx->sh_out.release();
    DevBuf *bufs[] = {&ctx->tab_rows, &ctx->tab_items, &ctx->bs_tasks, &ctx->plane_tasks, &ctx->planes, &ctx->plan_tables, &ctx->fkey_sorted, &ctx->perm,
                      &ctx->seg_bin_cnt, &ctx->seg_bin_start, &ctx->seg_tasks,
                      &ctx->seg_sub_rec, &ctx->seg_priv_edges, &ctx->seg_priv_dist, &ctx->seg_priv_cnt,
                      &ctx->iota, &ctx->sort_tmp, &ctx->sample_pos, &ctx->sample_out,
                      &ctx->fkey,    &ctx->thr,      &ctx->label,    &ctx->lab,      &ctx->edges,    &ctx->ovf,
                      &ctx->edge_dist, &ctx->counters,
                      &ctx->boff,    &ctx->status,   &ctx->blocked,  &ctx->in_keys,
                      &ctx->in_nmask, &ctx->in_freq, &ctx->out_kept, &ctx->out_root,
                      &ctx->stage_ws, &ctx->st_align, &ctx->st_umi, &ctx->st_score, &ctx->st_keys, &ctx->st_nmask,
                      &ctx->st_freq, &ctx->st_rep, &ctx->st_boff};
    for (DevBuf *b : bufs) b->release();
    if (ctx->h_boff) (void)hipHostFree(ctx->h_boff);
    ctx->h_tasks.release();
    ctx->h_plan_alt[0].release();
    ctx->h_plan_alt[1].release();
    if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
    for (int i = 0; i < umi_ctx::N_EVENTS; i++)
        if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int umi_ctx_set_option(umi_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name) return fail(UMI_ERR_ARG, "ctx/name is NULL");
    if (!strcmp(name, "split_min")) {
        if (value < 2) return fail(UMI_ERR_ARG, "split_min must be >= 2");
        ctx->split_min = (uint64_t)value;
        return UMI_OK;
    }
    if (!ctx->subs.empty()) { // every device of a multi-device context
        for (umi_ctx *sub : ctx->subs) {
            const int rc = umi_ctx_set_option(sub, name, value);
            if (rc) return rc;
        }
        return UMI_OK;
    }
    if (!strcmp(name, "profile")) {
        ctx->profile = value != 0;
    } else if (!strcmp(name, "edge_capacity")) {
        if (value < 1) return fail(UMI_ERR_ARG, "edge_capacity must be >= 1");
        ctx->edge_capacity = (uint64_t)value;
#ifdef UMIHIP_DEV
    } else if (!strcmp(name, "prune")) {
        ctx->prune = value != 0;
    } else if (!strcmp(name, "bs_unit")) {
        if (value < 1 || value > 3) return fail(UMI_ERR_ARG, "bs_unit must be 1, 2 or 3");
        ctx->bs_unit = (int)value;
    } else if (!strcmp(name, "bs_sorted")) {
        ctx->bs_sorted = value != 0;
    } else if (!strcmp(name, "bs_tables")) {
        ctx->bs_tables = value != 0;
    } else if (!strcmp(name, "two_phase")) {
        if (value < 0 || value > 2) return fail(UMI_ERR_ARG, "two_phase must be 0, 1 or 2");
        ctx->two_phase = (int)value;
    } else if (!strcmp(name, "bs_col_chunk")) {
        if (value < BS_COL_TILE || value > (1 << 24) || value % BS_COL_TILE)
            return fail(UMI_ERR_ARG, "bs_col_chunk must be a multiple of %d in %d..%d", BS_COL_TILE,
                        BS_COL_TILE, 1 << 24);
        ctx->bs_col_chunk = (uint32_t)value;
    } else if (!strcmp(name, "bs_tab_min_run")) {
        if (value < 0 || value > (1 << 30)) return fail(UMI_ERR_ARG, "bs_tab_min_run must be in 0..2^30");
        ctx->bs_tab_min_run = (uint32_t)value;
    } else if (!strcmp(name, "bs_transposed")) {
        ctx->bs_transposed = value != 0;
    } else if (!strcmp(name, "bs_tab_waves")) {
        if (value < 0 || value > (1 << 20)) return fail(UMI_ERR_ARG, "bs_tab_waves must be in 0..%d", 1 << 20);
        ctx->bs_tab_waves = (uint32_t)value;
    } else if (!strcmp(name, "bitslice")) {
        ctx->use_bitslice = value != 0;
    } else if (!strcmp(name, "ovf_capacity")) {
        if (value < 1) return fail(UMI_ERR_ARG, "ovf_capacity must be >= 1");
        ctx->ovf_capacity = (uint64_t)value;
#endif
    } else if (!strcmp(name, "seg_index")) {
        ctx->seg_index = value != 0;
    } else if (!strcmp(name, "table_pieces")) {
        if (value < 1 || value > 64) return fail(UMI_ERR_ARG, "table_pieces must be in 1..64");
        ctx->table_pieces = (uint32_t)value;
    } else if (!strcmp(name, "seg_ckey")) {
        ctx->seg_ckey = value != 0;
    } else if (!strcmp(name, "spin_wait")) {
        ctx->spin_wait = value != 0;
    } else if (!strcmp(name, "seg_sliced")) {
        ctx->seg_sliced = value != 0;
    } else if (!strcmp(name, "seg_unite")) {
        ctx->seg_unite = value != 0;
    } else if (!strcmp(name, "seg_lds")) {
        ctx->seg_lds = value != 0;
    } else if (!strcmp(name, "seg_blocks")) {
        if (value < 0 || value > (1 << 22)) return fail(UMI_ERR_ARG, "seg_blocks must be in 0..2^22");
        ctx->seg_blocks = (uint32_t)value;
    } else if (!strcmp(name, "seg_min")) {
        if (value < 2 || value > (1ll << 31)) return fail(UMI_ERR_ARG, "seg_min must be in 2..2^31");
        ctx->seg_min = (uint32_t)value;
    } else if (!strcmp(name, "fused_sliced")) {
        ctx->fused_sliced = value != 0;
    } else if (!strcmp(name, "fused_blocks")) {
        if (value < 1 || value > 64) return fail(UMI_ERR_ARG, "fused_blocks must be in 1..64");
        ctx->fused_blocks = (uint32_t)value;
    } else if (!strcmp(name, "fused_max")) {
        if (value < 0) return fail(UMI_ERR_ARG, "fused_max must be >= 0");
        ctx->fused_max = (uint32_t)std::min<int64_t>(value, FUSED_MAX);
    } else if (!strcmp(name, "small_max")) {
        if (value < 0) return fail(UMI_ERR_ARG, "small_max must be >= 0");
        ctx->small_max = (uint32_t)std::min<int64_t>(value, 1 << 30);
    } else {
        return fail(UMI_ERR_ARG, "unknown option '%s'", name);
    }
    return UMI_OK;
}

int umi_ctx_create_multi(const int *device_ids, int n_devices, umi_ctx **out)
{
    if (!out) return fail(UMI_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return fail(UMI_ERR_ARG, "device_ids is NULL or n_devices outside 1..64");
    umi_ctx *ctx = new (std::nothrow) umi_ctx();
    if (!ctx) return fail(UMI_ERR_NOMEM, "out of host memory");
    for (int i = 0; i < n_devices; i++) {
        umi_ctx *sub = nullptr;
        const int rc = umi_ctx_create(device_ids[i], &sub);
        if (rc) {
            const std::string msg = umi_last_error();
            umi_ctx_destroy(ctx);
            return fail(rc, "%s", msg.c_str());
        }
        ctx->subs.push_back(sub);
    }
    ctx->device = device_ids[0];
    *out = ctx;
    return UMI_OK;
}

int umi_ctx_device_count(const umi_ctx *ctx) { return !ctx ? 0 : (ctx->subs.empty() ? 1 : (int)ctx->subs.size()); }

int umi_partition_buckets(const uint64_t *bucket_off, uint64_t n_buckets, uint32_t n_ranks, uint32_t *owner)
{
    if (!bucket_off || (!owner && n_buckets)) return fail(UMI_ERR_ARG, "bucket_off/owner is NULL");
    if (n_ranks < 1) return fail(UMI_ERR_ARG, "n_ranks must be >= 1");
    for (uint64_t b = 0; b < n_buckets; b++) {
        if (bucket_off[b + 1] < bucket_off[b]) return fail(UMI_ERR_ARG, "bucket_off not monotone at bucket %llu", (unsigned long long)b);
        // (the cost n_b^2 + n_b is kept in 64 bits; the batched calls take no larger bucket either)
        if (bucket_off[b + 1] - bucket_off[b] >= 0x7FFFFFF0ull)
            return fail(UMI_ERR_ARG, "bucket %llu holds %llu entries: beyond the 31-bit index space of one call",
                        (unsigned long long)b, (unsigned long long)(bucket_off[b + 1] - bucket_off[b]));
    }
    std::vector<uint32_t> o;
    partition_buckets_lpt(bucket_off, n_buckets, n_ranks, o);
    for (uint64_t b = 0; b < n_buckets; b++) owner[b] = o[b];
    return UMI_OK;
}

int umi_dedup_batch(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                    const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int k,
                    float percentage, int algo, int32_t adj_max_freq, uint8_t *kept, uint32_t *root,
                    umi_stats *stats)
{
    if (ctx && !ctx->subs.empty()) {
        uint64_t n = 0;
        int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n);
        if (rc) return rc;
        if (n && (!keys || !freq || !kept)) return fail(UMI_ERR_ARG, "keys/freq/kept is NULL");
        for (uint64_t b = 0; b < n_buckets; b++)
            if (bucket_off[b + 1] < bucket_off[b])
                return fail(UMI_ERR_ARG, "bucket_off not monotone at bucket %llu", (unsigned long long)b);
        if (n == 0) {
            if (stats) {
                memset(stats, 0, sizeof(*stats));
                stats->n_buckets = n_buckets;
            }
            return UMI_OK;
        }
        if (ctx->subs.size() == 1)
            return dedup_batch_single(ctx->subs[0], keys, nmask, freq, bucket_off, n_buckets, umi_len, k, percentage,
                                      algo, adj_max_freq, kept, root, stats);
        return dedup_batch_multi(ctx, keys, nmask, freq, bucket_off, n_buckets, n, umi_len, k, percentage, algo,
                                 adj_max_freq, kept, root, stats);
    }
    return dedup_batch_single(ctx, keys, nmask, freq, bucket_off, n_buckets, umi_len, k, percentage, algo,
                              adj_max_freq, kept, root, stats);
}

int umi_pack_mask_device(umi_ctx *ctx, const uint8_t *d_kept, uint64_t n, uint8_t *d_bits, void *hip_stream)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!ctx->subs.empty()) {
        if (ctx->subs.size() > 1)
            return fail(UMI_ERR_ARG, "device pointers belong to one device: use a single-device context");
        ctx = ctx->subs[0];
    }
    if (n && (!d_kept || !d_bits)) return fail(UMI_ERR_ARG, "kept/bits is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(launch_pack_mask(d_kept, n, d_bits, (hipStream_t)hip_stream));
    return UMI_OK;
}

// ---- multi-word keys: umi_len 22..UMI_MAX_WIDE_UMI_LEN -------------------------------------------
namespace {
int wide_words(int umi_len) { return (3 * umi_len + 63) / 64; } // bitset.rs:17-18
}

int umi_encode_umis_wide(const uint8_t *ascii, uint64_t n, int umi_len, int n_words, uint64_t *keys, uint64_t *nmask)
{ // src/utils/mod.rs:63-83 for keys of any length: base i at bits 3i .. 3i+2 of the word string
    if (!ascii || !keys) return fail(UMI_ERR_ARG, "ascii/keys is NULL");
    if (umi_len < 1 || umi_len > UMI_MAX_WIDE_UMI_LEN) return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_WIDE_UMI_LEN);
    if (n_words != wide_words(umi_len)) return fail(UMI_ERR_ARG, "n_words must be %d for umi_len %d", wide_words(umi_len), umi_len);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t *k = keys + i * (uint64_t)n_words, *m = nmask ? nmask + i * (uint64_t)n_words : nullptr;
        for (int w = 0; w < n_words; w++) {
            k[w] = 0;
            if (m) m[w] = 0;
        }
        for (int b = 0; b < umi_len; b++) {
            uint64_t c;
            switch (ascii[i * (uint64_t)umi_len + b]) { // read.rs:23-31
            case 'A': c = 0; break;
            case 'T': c = 5; break;
            case 'C': c = 6; break;
            case 'G': c = 3; break;
            case 'N': c = 4; break;
            default: return fail(UMI_ERR_CHAR, "Unknown character in UMI sequence: %u (UMI %llu)",
                                 (unsigned)ascii[i * (uint64_t)umi_len + b], (unsigned long long)i);
            }
            for (int j = 0; j < 3; j++) { // bit by bit: a base may straddle two words (bitset.rs:52-61)
                const int bit = 3 * b + j;
                if ((c >> j) & 1) k[bit >> 6] |= 1ull << (bit & 63);
                if (c == 4 && m) m[bit >> 6] |= 1ull << (bit & 63); // set_n_bit, bitset.rs:63-75
            }
        }
    }
    return UMI_OK;
}

int umi_dedup_batch_wide_device(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask, int n_words,
                                const int32_t *d_freq, const uint64_t *bucket_off, uint64_t n_buckets, int umi_len,
                                int k, float percentage, int algo, int32_t adj_max_freq, uint8_t *d_kept,
                                uint32_t *d_root, void *hip_stream, umi_stats *stats)
{
    if (ctx && !ctx->subs.empty()) {
        if (ctx->subs.size() > 1)
            return fail(UMI_ERR_ARG, "device pointers belong to one device: use a single-device context");
        ctx = ctx->subs[0];
    }
    uint64_t n = 0;
    int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n, UMI_MAX_WIDE_UMI_LEN);
    if (rc) return rc;
    if (n_words != wide_words(umi_len)) return fail(UMI_ERR_ARG, "n_words must be %d for umi_len %d", wide_words(umi_len), umi_len);
    if (n && (!d_keys || !d_freq || !d_kept)) return fail(UMI_ERR_ARG, "keys/freq/kept is NULL");
    if (n == 0) {
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->n_buckets = n_buckets;
        }
        return UMI_OK;
    }
    return run_pipeline(ctx, d_keys, d_nmask, d_freq, bucket_off, n_buckets, (uint32_t)n, umi_len, k, percentage,
                        algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY, adj_max_freq, d_kept, d_root,
                        (hipStream_t)hip_stream, stats, nullptr, n_words);
}

int umi_dedup_batch_wide(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                         const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int k, float percentage,
                         int algo, int32_t adj_max_freq, uint8_t *kept, uint32_t *root, umi_stats *stats)
{
    if (n_words == 1) // one word: the ordinary call
        return umi_dedup_batch(ctx, keys, nmask, freq, bucket_off, n_buckets, umi_len, k, percentage, algo,
                               adj_max_freq, kept, root, stats);
    uint64_t n = 0;
    int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n, UMI_MAX_WIDE_UMI_LEN);
    if (rc) return rc;
    if (n_words != wide_words(umi_len)) return fail(UMI_ERR_ARG, "n_words must be %d for umi_len %d", wide_words(umi_len), umi_len);
    if (n && (!keys || !freq || !kept)) return fail(UMI_ERR_ARG, "keys/freq/kept is NULL");
    for (uint64_t b = 0; b < n_buckets; b++)
        if (bucket_off[b + 1] < bucket_off[b])
            return fail(UMI_ERR_ARG, "bucket_off not monotone at bucket %llu", (unsigned long long)b);
    if (n == 0) {
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->n_buckets = n_buckets;
        }
        return UMI_OK;
    }
    // a multi-device context shards the buckets over its devices exactly as for one-word keys
    if (!ctx->subs.empty() && ctx->subs.size() > 1)
        return dedup_batch_multi(ctx, keys, nmask, freq, bucket_off, n_buckets, n, umi_len, k, percentage, algo,
                                 adj_max_freq, kept, root, stats, n_words);
    return dedup_batch_single(ctx->subs.empty() ? ctx : ctx->subs[0], keys, nmask, freq, bucket_off, n_buckets, umi_len, k,
                              percentage, algo, adj_max_freq, kept, root, stats, n_words);
}

int umi_stage_reads_wide_device(umi_ctx *ctx, const uint64_t *d_align_key, int align_key_bits, const uint8_t *d_umi_ascii,
                                const int32_t *d_score, uint64_t n_reads, int umi_len, int n_words, int merge,
                                uint64_t *d_keys, uint64_t *d_nmask, int32_t *d_freq, uint64_t *d_rep,
                                uint64_t *d_bucket_off, uint64_t *n_entries, uint64_t *n_buckets, void *hip_stream)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!ctx->subs.empty()) ctx = ctx->subs[0]; // (staging runs on the first device of a multi-device context)
    if (!n_entries || !n_buckets || !d_bucket_off) return fail(UMI_ERR_ARG, "n_entries / n_buckets / d_bucket_off is NULL");
    if (n_reads && (!d_align_key || !d_umi_ascii || !d_keys || !d_freq || !d_rep))
        return fail(UMI_ERR_ARG, "a required device pointer is NULL");
    if (umi_len < 1 || umi_len > UMI_MAX_WIDE_UMI_LEN) return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_WIDE_UMI_LEN);
    if (n_words != wide_words(umi_len)) return fail(UMI_ERR_ARG, "n_words must be %d for umi_len %d", wide_words(umi_len), umi_len);
    if (align_key_bits < 1 || align_key_bits > 64) return fail(UMI_ERR_ARG, "align_key_bits must be in 1..64");
    if (merge != 0 && merge != 1) return fail(UMI_ERR_ARG, "merge must be 0 (any) or 1 (highest score, first on ties)");
    if (n_reads >= (1ull << 30)) return fail(UMI_ERR_ARG, "%llu reads exceed the 30-bit index space of one staging call", (unsigned long long)n_reads);
    HIP_TRY(hipSetDevice(ctx->device));
    settle(ctx);
    int rc;
    if ((rc = ctx->stage_ws.reserve(stage_workspace_bytes((uint32_t)n_reads, n_words)))) return rc;
    hipStream_t s = (hipStream_t)hip_stream; // (NULL = the default stream, as in every device-pointer call)
    const int r = stage_reads_on_device(ctx->stage_ws.p, d_align_key, align_key_bits, d_umi_ascii, d_score,
                                        (uint32_t)n_reads, umi_len, n_words, merge, d_keys, d_nmask, d_freq, d_rep,
                                        d_bucket_off, n_entries, n_buckets, ctx->h_counters, s);
    if (r == 1) return fail(UMI_ERR_CHAR, "Unknown character in UMI sequence");
    if (r < 0) return fail(UMI_ERR_HIP, "staging: %s", hipGetErrorString((hipError_t)(-r)));
    return UMI_OK;
}

int umi_stage_reads_device(umi_ctx *ctx, const uint64_t *d_align_key, int align_key_bits, const uint8_t *d_umi_ascii,
                           const int32_t *d_score, uint64_t n_reads, int umi_len, int merge, uint64_t *d_keys,
                           uint64_t *d_nmask, int32_t *d_freq, uint64_t *d_rep, uint64_t *d_bucket_off,
                           uint64_t *n_entries, uint64_t *n_buckets, void *hip_stream)
{
    if (umi_len < 1 || umi_len > UMI_MAX_UMI_LEN) return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_UMI_LEN);
    return umi_stage_reads_wide_device(ctx, d_align_key, align_key_bits, d_umi_ascii, d_score, n_reads, umi_len, 1, merge,
                                       d_keys, d_nmask, d_freq, d_rep, d_bucket_off, n_entries, n_buckets, hip_stream);
}

int umi_stage_reads_wide(umi_ctx *ctx, const uint64_t *align_key, int align_key_bits, const uint8_t *umi_ascii,
                         const int32_t *score, uint64_t n_reads, int umi_len, int n_words, int merge, uint64_t *keys,
                         uint64_t *nmask, int32_t *freq, uint64_t *rep, uint64_t *bucket_off, uint64_t *n_entries,
                         uint64_t *n_buckets)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!ctx->subs.empty()) ctx = ctx->subs[0];
    if (!n_entries || !n_buckets || !bucket_off) return fail(UMI_ERR_ARG, "n_entries / n_buckets / bucket_off is NULL");
    if (n_reads && (!align_key || !umi_ascii || !keys || !freq || !rep)) return fail(UMI_ERR_ARG, "a required pointer is NULL");
    if (umi_len < 1 || umi_len > UMI_MAX_WIDE_UMI_LEN) return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_WIDE_UMI_LEN);
    if (n_words != wide_words(umi_len)) return fail(UMI_ERR_ARG, "n_words must be %d for umi_len %d", wide_words(umi_len), umi_len);
    if (n_reads >= (1ull << 30)) return fail(UMI_ERR_ARG, "%llu reads exceed the 30-bit index space of one staging call", (unsigned long long)n_reads);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    const size_t n = (size_t)n_reads, m = std::max<size_t>(n, 1), kw = 8 * (size_t)n_words;
    if ((rc = ctx->st_align.reserve(m * 8)) || (rc = ctx->st_umi.reserve(m * (size_t)umi_len)) ||
        (rc = ctx->st_score.reserve(m * 4)) || (rc = ctx->st_keys.reserve(m * kw)) || (rc = ctx->st_nmask.reserve(m * kw)) ||
        (rc = ctx->st_freq.reserve(m * 4)) || (rc = ctx->st_rep.reserve(m * 8)) || (rc = ctx->st_boff.reserve((m + 1) * 8)))
        return rc;
    hipStream_t s = ctx->own_stream;
    if (n) {
        HIP_TRY(hipMemcpyAsync(ctx->st_align.p, align_key, n * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(ctx->st_umi.p, umi_ascii, n * (size_t)umi_len, hipMemcpyHostToDevice, s));
        if (score) HIP_TRY(hipMemcpyAsync(ctx->st_score.p, score, n * 4, hipMemcpyHostToDevice, s));
    }
    if ((rc = umi_stage_reads_wide_device(ctx, ctx->st_align.as<uint64_t>(), align_key_bits, ctx->st_umi.as<uint8_t>(),
                                          score ? ctx->st_score.as<int32_t>() : nullptr, n_reads, umi_len, n_words, merge,
                                          ctx->st_keys.as<uint64_t>(), ctx->st_nmask.as<uint64_t>(),
                                          ctx->st_freq.as<int32_t>(), ctx->st_rep.as<uint64_t>(),
                                          ctx->st_boff.as<uint64_t>(), n_entries, n_buckets, s)))
        return rc;
    const size_t e = (size_t)*n_entries, b = (size_t)*n_buckets;
    if (e) {
        HIP_TRY(hipMemcpyAsync(keys, ctx->st_keys.p, e * kw, hipMemcpyDeviceToHost, s));
        if (nmask) HIP_TRY(hipMemcpyAsync(nmask, ctx->st_nmask.p, e * kw, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(freq, ctx->st_freq.p, e * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(rep, ctx->st_rep.p, e * 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(bucket_off, ctx->st_boff.p, (b + 1) * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return UMI_OK;
}

int umi_stage_reads(umi_ctx *ctx, const uint64_t *align_key, int align_key_bits, const uint8_t *umi_ascii,
                    const int32_t *score, uint64_t n_reads, int umi_len, int merge, uint64_t *keys, uint64_t *nmask,
                    int32_t *freq, uint64_t *rep, uint64_t *bucket_off, uint64_t *n_entries, uint64_t *n_buckets)
{
    if (umi_len < 1 || umi_len > UMI_MAX_UMI_LEN) return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_UMI_LEN);
    return umi_stage_reads_wide(ctx, align_key, align_key_bits, umi_ascii, score, n_reads, umi_len, 1, merge, keys, nmask,
                                freq, rep, bucket_off, n_entries, n_buckets);
}

int umi_encode_umis(const uint8_t *ascii, uint64_t n, int umi_len, uint64_t *keys, uint64_t *nmask)
{
    if ((!ascii && n) || !keys) return fail(UMI_ERR_ARG, "ascii/keys is NULL");
    if (umi_len < 1 || umi_len > UMI_MAX_UMI_LEN)
        return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_UMI_LEN);
    // code table of src/utils/read.rs:23-31; 0xFF = the reference panics
    uint8_t code[256];
    memset(code, 0xFF, sizeof(code));
    code['A'] = 0x0; code['T'] = 0x5; code['C'] = 0x6; code['G'] = 0x3; code['N'] = 0x4;
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *s = ascii + i * (uint64_t)umi_len;
        uint64_t key = 0, nm = 0;
        for (int b = 0; b < umi_len; b++) {
            const uint8_t c = code[s[b]];
            if (c == 0xFF)
                return fail(UMI_ERR_CHAR, "Unknown character in UMI sequence: %u (UMI %llu)",
                            (unsigned)s[b], (unsigned long long)i);
            key |= (uint64_t)c << (3 * b);
            if (c == 0x4) nm |= (uint64_t)0x7 << (3 * b);
        }
        keys[i] = key;
        if (nmask) nmask[i] = nm;
    }
    return UMI_OK;
}

int umi_dedup_batch_device(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                           const int32_t *d_freq, const uint64_t *bucket_off, uint64_t n_buckets,
                           int umi_len, int k, float percentage, int algo, int32_t adj_max_freq,
                           uint8_t *d_kept, uint32_t *d_root, void *hip_stream, umi_stats *stats)
{
    return umi_dedup_batch_device_table(ctx, d_keys, d_nmask, d_freq, bucket_off, nullptr, n_buckets, umi_len, k,
                                        percentage, algo, adj_max_freq, d_kept, d_root, hip_stream, stats);
}

int umi_dedup_batch_device_table(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                                 const int32_t *d_freq, const uint64_t *bucket_off, const uint64_t *d_bucket_off,
                                 uint64_t n_buckets, int umi_len, int k, float percentage, int algo,
                                 int32_t adj_max_freq, uint8_t *d_kept, uint32_t *d_root, void *hip_stream,
                                 umi_stats *stats)
{
    if (ctx && !ctx->subs.empty()) {
        if (ctx->subs.size() > 1)
            return fail(UMI_ERR_ARG, "device pointers belong to one device: use a single-device context");
        ctx = ctx->subs[0];
    }
    uint64_t n = 0;
    int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n);
    if (rc) return rc;
    if (n && (!d_keys || !d_freq || !d_kept)) return fail(UMI_ERR_ARG, "keys/freq/kept is NULL");
    if (n == 0) {
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->n_buckets = n_buckets;
        }
        return UMI_OK;
    }
    return run_pipeline(ctx, d_keys, d_nmask, d_freq, bucket_off, n_buckets, (uint32_t)n, umi_len,
                        k, percentage,
                        algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY,
                        adj_max_freq, d_kept, d_root, (hipStream_t)hip_stream, stats, d_bucket_off);
}

int umi_dedup_batch_device_begin(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask, const int32_t *d_freq,
                                 const uint64_t *bucket_off, const uint64_t *d_bucket_off, uint64_t n_buckets,
                                 int umi_len, int k, float percentage, int algo, int32_t adj_max_freq, uint8_t *d_kept,
                                 uint32_t *d_root, void *hip_stream)
{
    if (ctx && !ctx->subs.empty()) {
        if (ctx->subs.size() > 1)
            return fail(UMI_ERR_ARG, "device pointers belong to one device: use a single-device context");
        ctx = ctx->subs[0];
    }
    uint64_t n = 0;
    int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n);
    if (rc) return rc;
    if (n && (!d_keys || !d_freq || !d_kept)) return fail(UMI_ERR_ARG, "keys/freq/kept is NULL");
    settle(ctx);
    umi_ctx::PendingCall &pc = ctx->pending;
    pc.have_result = false;
    memset(&pc.st, 0, sizeof(pc.st));
    if (n == 0) {
        pc.st.n_buckets = n_buckets;
        pc.rc = UMI_OK;
        pc.have_result = true;
        return UMI_OK;
    }
    umi_stats st;
    memset(&st, 0, sizeof(st));
    rc = run_pipeline(ctx, d_keys, d_nmask, d_freq, bucket_off, n_buckets, (uint32_t)n, umi_len, k, percentage,
                      algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY, adj_max_freq, d_kept, d_root,
                      (hipStream_t)hip_stream, &st, d_bucket_off, 1, true);
    if (rc) return rc; // (nothing is pending: the call failed where a plain call would have)
    if (!pc.deferred) { // the call had decisions to take on the host and has run to its end
        pc.st = st;
        pc.rc = UMI_OK;
        pc.have_result = true;
    }
    return UMI_OK;
}

int umi_dedup_batch_end(umi_ctx *ctx, umi_stats *stats)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!ctx->subs.empty()) ctx = ctx->subs[0];
    return finish_pending(ctx, stats);
}

int umi_dedup_batch_device_multi(umi_ctx *ctx, const uint64_t *const *d_keys, const uint64_t *const *d_nmask,
                                 const int32_t *const *d_freq, const uint64_t *const *bucket_off,
                                 const uint64_t *n_buckets, int umi_len, int k, float percentage, int algo,
                                 int32_t adj_max_freq, uint8_t *const *d_kept, uint32_t *const *d_root,
                                 uint8_t *const *d_mask_bits_all, uint64_t slice_bytes, umi_stats *stats)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (ctx->subs.empty()) return fail(UMI_ERR_ARG, "a multi-device context is needed (umi_ctx_create_multi; one device will do)");
    const uint32_t n_dev = (uint32_t)ctx->subs.size();
    if (!d_keys || !d_freq || !bucket_off || !n_buckets || !d_kept) return fail(UMI_ERR_ARG, "a required array of pointers is NULL");
    for (uint32_t r = 0; r < n_dev; r++)
        for (uint32_t q = 0; q < r; q++)
            if (d_mask_bits_all && ctx->subs[r]->device == ctx->subs[q]->device)
                return fail(UMI_ERR_ARG, "device %d is named twice: RCCL wants one rank per device", ctx->subs[r]->device);
    std::vector<uint64_t> n(n_dev, 0);
    for (uint32_t r = 0; r < n_dev; r++) {
        int rc = check_common(ctx->subs[r], bucket_off[r], n_buckets[r], umi_len, k, algo, &n[r]);
        if (rc) return rc;
        if (n[r] && (!d_keys[r] || !d_freq[r] || !d_kept[r])) return fail(UMI_ERR_ARG, "device %u: keys/freq/kept is NULL", r);
        if (d_mask_bits_all && (!d_mask_bits_all[r] || slice_bytes < (n[r] + 7) / 8))
            return fail(UMI_ERR_ARG, "device %u: the gathered mask needs %u slices of at least %llu bytes", r, n_dev,
                        (unsigned long long)((n[r] + 7) / 8));
    }
    // the communicators, once per context
    if (d_mask_bits_all && ctx->comms.empty()) {
        if (!ctx->rccl.load()) return fail(UMI_ERR_HIP, "%s", ctx->rccl.error.c_str());
        std::vector<int> ids;
        for (umi_ctx *sub : ctx->subs) ids.push_back(sub->device);
        ctx->comms.assign(n_dev, nullptr);
        const ncclResult_t e = ctx->rccl.CommInitAll(ctx->comms.data(), (int)n_dev, ids.data());
        if (e != ncclSuccess) {
            ctx->comms.clear();
            return fail(UMI_ERR_HIP, "ncclCommInitAll: %s", ctx->rccl.GetErrorString(e));
        }
    }
    // every device's shard through the ordinary pipeline on its own stream (a host thread each: the
    // pipeline plans and synchronises), its mask packed to bits into its slot of its gather buffer
    const int mode = algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY;
    std::vector<ShardResult> res(n_dev);
    std::vector<std::thread> pool;
    for (uint32_t r = 0; r < n_dev; r++)
        pool.emplace_back([&, r] {
            umi_ctx *sub = ctx->subs[r];
            ShardResult &out = res[r];
            memset(&out.st, 0, sizeof(out.st));
            out.st.n_buckets = n_buckets[r];
            hipError_t e = hipSetDevice(sub->device);
            int rc = UMI_OK;
            if (e == hipSuccess && n[r])
                rc = run_pipeline(sub, d_keys[r], d_nmask ? d_nmask[r] : nullptr, d_freq[r], bucket_off[r], n_buckets[r],
                                  (uint32_t)n[r], umi_len, k, percentage, mode, adj_max_freq, d_kept[r],
                                  d_root ? d_root[r] : nullptr, sub->own_stream, &out.st);
            if (rc == UMI_OK && e == hipSuccess && d_mask_bits_all) {
                uint8_t *slot = d_mask_bits_all[r] + (size_t)r * slice_bytes;
                e = hipMemsetAsync(slot, 0, slice_bytes, sub->own_stream);
                if (e == hipSuccess && n[r]) e = launch_pack_mask(d_kept[r], n[r], slot, sub->own_stream);
            }
            if (rc) {
                out.rc = rc;
                out.err = umi_last_error();
            } else if (e != hipSuccess) {
                out.rc = UMI_ERR_HIP;
                out.err = hipGetErrorString(e);
            }
        });
    for (auto &t : pool) t.join();
    for (uint32_t r = 0; r < n_dev; r++)
        if (res[r].rc) return fail(res[r].rc, "device %d: %s", ctx->subs[r]->device, res[r].err.c_str());
    // the all-gatherv of the kept mask: every device's slice to every device over RCCL / xGMI, as
    // padded slices in place (a device's send buffer is its own slot of its receive buffer) --
    // <= 1 bit per unique UMI, latency-bound; one group, so the ranks of this one process progress together
    if (d_mask_bits_all) {
        ncclResult_t e = ctx->rccl.GroupStart();
        for (uint32_t r = 0; r < n_dev && e == ncclSuccess; r++)
            e = ctx->rccl.AllGather(d_mask_bits_all[r] + (size_t)r * slice_bytes, d_mask_bits_all[r], slice_bytes, ncclUint8,
                                    ctx->comms[r], ctx->subs[r]->own_stream);
        const ncclResult_t e2 = ctx->rccl.GroupEnd();
        if (e != ncclSuccess || e2 != ncclSuccess)
            return fail(UMI_ERR_HIP, "ncclAllGather: %s", ctx->rccl.GetErrorString(e != ncclSuccess ? e : e2));
        for (uint32_t r = 0; r < n_dev; r++) {
            HIP_TRY(hipSetDevice(ctx->subs[r]->device));
            HIP_TRY(hipStreamSynchronize(ctx->subs[r]->own_stream));
        }
    }
    if (stats) {
        umi_stats total = res[0].st;
        total.n_umis = n[0];
        for (uint32_t r = 1; r < n_dev; r++) {
            merge_stats(total, res[r].st);
            total.max_bucket = std::max(total.max_bucket, res[r].st.max_bucket);
            total.n_pairs += res[r].st.n_pairs;
            total.n_umis += n[r];
            total.n_buckets += n_buckets[r];
        }
        *stats = total;
    }
    return UMI_OK;
}

int umi_pairs_partial_device(umi_ctx *ctx, const uint64_t *d_keys, const uint64_t *d_nmask,
                             const int32_t *d_freq, const uint64_t *bucket_off, uint64_t n_buckets,
                             int umi_len, int k, float percentage, int algo, int32_t adj_max_freq,
                             uint32_t part, uint32_t n_parts, uint64_t *d_edges,
                             uint64_t edge_capacity, uint64_t *n_edges_out, void *hip_stream,
                             umi_stats *stats)
{
    if (ctx && !ctx->subs.empty()) {
        if (ctx->subs.size() > 1)
            return fail(UMI_ERR_ARG, "device pointers belong to one device: use a single-device context");
        ctx = ctx->subs[0];
    }
    uint64_t n = 0;
    int rc = check_common(ctx, bucket_off, n_buckets, umi_len, k, algo, &n);
    if (rc) return rc;
    if (!n_edges_out) return fail(UMI_ERR_ARG, "n_edges_out is NULL");
    if (n_parts < 2) return fail(UMI_ERR_ARG, "n_parts must be >= 2 (use umi_dedup_batch_device)");
    if (part >= n_parts) return fail(UMI_ERR_ARG, "part %u outside 0..%u", part, n_parts - 1);
    *n_edges_out = 0;
    if (n == 0) return UMI_OK;
    if (!d_keys || !d_freq) return fail(UMI_ERR_ARG, "keys/freq is NULL");
    const int mode = algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY;
    hipStream_t s = (hipStream_t)hip_stream;
    Pipeline p(ctx, d_keys, d_nmask, d_freq, bucket_off, n_buckets, (uint32_t)n, umi_len, k,
               percentage, mode, adj_max_freq, nullptr, nullptr, s, part, n_parts);
    if ((rc = p.run(stats))) return rc;
    *n_edges_out = p.edge_count();
    if (p.edge_count() > edge_capacity)
        return fail(UMI_ERR_NOMEM, "edge buffer holds %llu entries, %llu needed",
                    (unsigned long long)edge_capacity, (unsigned long long)p.edge_count());
    if (p.edge_count()) {
        if (!d_edges) return fail(UMI_ERR_ARG, "d_edges is NULL");
        HIP_TRY(hipMemcpyAsync(d_edges, ctx->edges.p, p.edge_count() * sizeof(uint2),
                               hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    return UMI_OK;
}

int umi_collapse_edges_device(umi_ctx *ctx, uint64_t n, const uint64_t *d_edges, uint64_t n_edges,
                              int algo, uint8_t *d_kept, uint32_t *d_root, void *hip_stream,
                              umi_stats *stats)
{
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!ctx->subs.empty()) {
        if (ctx->subs.size() > 1)
            return fail(UMI_ERR_ARG, "device pointers belong to one device: use a single-device context");
        ctx = ctx->subs[0];
    }
    if (algo != UMI_ALGO_DIRECTIONAL && algo != UMI_ALGO_ADJACENCY)
        return fail(UMI_ERR_ARG, "unknown algo %d", algo);
    if (n >= 0x7FFFFFF0ull || n_edges >= 0x7FFFFFF0ull) return fail(UMI_ERR_ARG, "too many entries/edges");
    settle(ctx);
    if (n == 0) return UMI_OK;
    if (!d_kept || (n_edges && !d_edges)) return fail(UMI_ERR_ARG, "kept/edges is NULL");
    return EdgeCollapse(ctx, (uint32_t)n, (const uint2 *)d_edges, (uint32_t)n_edges,
                        algo == UMI_ALGO_DIRECTIONAL ? MODE_DIRECTIONAL : MODE_ADJACENCY, d_kept,
                        d_root, (hipStream_t)hip_stream)
        .run(stats);
}

} // extern "C"

// ---- per-bucket DataStruct path ------------------------------------------------
struct umi_data {
    umi_ctx *ctx = nullptr;
    uint32_t n = 0;
    int max_edits = 0;
    std::vector<int32_t> freq;
    std::vector<uint8_t> present;
    // CSR of neighbours with dist <= max_edits (both directions), sorted by index
    std::vector<uint64_t> off;
    std::vector<uint32_t> nbr;
    std::vector<uint8_t> nbr_dist;
};

extern "C" {

int umi_data_new(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                 uint32_t n, int umi_len, int max_edits, umi_data **out)
{
    return umi_data_new_wide(ctx, keys, nmask, 1, freq, n, umi_len, max_edits, out);
}

int umi_data_new_wide(umi_ctx *ctx, const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                      uint32_t n, int umi_len, int max_edits, umi_data **out)
{
    if (!out) return fail(UMI_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!ctx) return fail(UMI_ERR_ARG, "ctx is NULL");
    if (!ctx->subs.empty()) ctx = ctx->subs[0]; // one bucket's store lives on one device
    if (n && (!keys || !freq)) return fail(UMI_ERR_ARG, "keys/freq is NULL");
    if (umi_len < 1 || umi_len > UMI_MAX_WIDE_UMI_LEN)
        return fail(UMI_ERR_ARG, "umi_len %d outside 1..%d", umi_len, UMI_MAX_WIDE_UMI_LEN);
    if (n_words != (3 * umi_len + 63) / 64) return fail(UMI_ERR_ARG, "n_words must be %d for umi_len %d", (3 * umi_len + 63) / 64, umi_len);
    if (max_edits < 0) return fail(UMI_ERR_ARG, "max_edits must be >= 0");
    if (n >= 0x7FFFFFF0u) // same index space as the batched calls (bit 31 of an edge endpoint is a flag)
        return fail(UMI_ERR_ARG, "%u entries exceed the 31-bit index space of one store", n);
    std::unique_ptr<umi_data> d(new (std::nothrow) umi_data());
    if (!d) return fail(UMI_ERR_NOMEM, "out of host memory");
    d->ctx = ctx;
    d->n = n;
    d->max_edits = max_edits;
    d->freq.assign(freq, freq + n);
    d->present.assign(n, 1);
    d->off.assign((size_t)n + 1, 0);
    if (n >= 2) {
        int rc;
        HIP_TRY(hipSetDevice(ctx->device));
        const size_t kb = (size_t)n * 8 * (size_t)n_words;
        if ((rc = ctx->in_keys.reserve(kb)) || (rc = ctx->in_freq.reserve((size_t)n * 4)) ||
            (nmask && (rc = ctx->in_nmask.reserve(kb))))
            return rc;
        hipStream_t s = ctx->own_stream;
        // the neighbour build ignores freq and order; feed ones so that prep's
        // contract check (rank order) does not apply to an unordered map
        std::vector<int32_t> ones(n, 1);
        hipError_t e = hipMemcpyAsync(ctx->in_keys.p, keys, kb, hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(ctx->in_freq.p, ones.data(), (size_t)n * 4, hipMemcpyHostToDevice, s);
        if (e == hipSuccess && nmask)
            e = hipMemcpyAsync(ctx->in_nmask.p, nmask, kb, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s); // `ones` must outlive the copy
        if (e != hipSuccess) return fail(UMI_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
        const uint64_t boff[2] = {0, n};
        umi_stats st;
        rc = run_pipeline(ctx, ctx->in_keys.as<uint64_t>(),
                          nmask ? ctx->in_nmask.as<uint64_t>() : nullptr,
                          ctx->in_freq.as<int32_t>(), boff, 1, n, umi_len, max_edits, 0.0f,
                          MODE_NEIGHBOURS, 0, nullptr, nullptr, s, &st, nullptr, n_words);
        if (rc) return rc;
        const size_t E = (size_t)st.n_edges;
        std::vector<uint2> pairs(E);
        std::vector<uint8_t> pd(E);
        if (E) {
            e = hipMemcpy(pairs.data(), ctx->edges.p, E * sizeof(uint2), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(pd.data(), ctx->edge_dist.p, E, hipMemcpyDeviceToHost);
            if (e != hipSuccess)
                return fail(UMI_ERR_HIP, "download failed: %s", hipGetErrorString(e));
        }
        for (size_t i = 0; i < E; i++) {
            d->off[pairs[i].x + 1]++;
            d->off[pairs[i].y + 1]++;
        }
        for (uint32_t i = 0; i < n; i++) d->off[i + 1] += d->off[i];
        d->nbr.resize(2 * E);
        d->nbr_dist.resize(2 * E);
        std::vector<uint64_t> fill(d->off.begin(), d->off.end() - 1);
        for (size_t i = 0; i < E; i++) {
            uint64_t pa = fill[pairs[i].x]++, pb = fill[pairs[i].y]++;
            d->nbr[pa] = pairs[i].y; d->nbr_dist[pa] = pd[i];
            d->nbr[pb] = pairs[i].x; d->nbr_dist[pb] = pd[i];
        }
        // ascending neighbour index inside each row (the device appends in arrival order)
        std::vector<std::pair<uint32_t, uint8_t>> tmp;
        for (uint32_t i = 0; i < n; i++) {
            const uint64_t a = d->off[i], b = d->off[i + 1];
            tmp.clear();
            for (uint64_t t = a; t < b; t++) tmp.emplace_back(d->nbr[t], d->nbr_dist[t]);
            std::sort(tmp.begin(), tmp.end());
            for (uint64_t t = a; t < b; t++) {
                d->nbr[t] = tmp[t - a].first;
                d->nbr_dist[t] = tmp[t - a].second;
            }
        }
    }
    *out = d.release();
    return UMI_OK;
}

int umi_data_remove_near(umi_data *d, uint32_t query, int k, int32_t max_freq, uint32_t *out_idx,
                         uint32_t *out_n)
{
    if (!d || !out_n || (!out_idx && d->n)) return fail(UMI_ERR_ARG, "NULL argument");
    if (query >= d->n) return fail(UMI_ERR_ARG, "query %u outside 0..%u", query, d->n);
    if (k > d->max_edits)
        return fail(UMI_ERR_ARG, "k %d exceeds max_edits %d given to umi_data_new", k, d->max_edits);
    // naive.rs:29-37 over the precomputed neighbour row; merge the query itself
    // (dist 0, removed whatever its freq) in index order
    uint32_t cnt = 0;
    bool self_done = false;
    auto emit_self = [&]() {
        if (!self_done && k >= 0 && d->present[query]) {
            d->present[query] = 0;
            out_idx[cnt++] = query;
        }
        self_done = true;
    };
    for (uint64_t t = d->off[query]; t < d->off[query + 1]; t++) {
        const uint32_t o = d->nbr[t];
        if (o > query) emit_self();
        if (d->present[o] && (int)d->nbr_dist[t] <= k &&
            (d->nbr_dist[t] == 0 || d->freq[o] <= max_freq)) {
            d->present[o] = 0;
            out_idx[cnt++] = o;
        }
    }
    emit_self();
    *out_n = cnt;
    return UMI_OK;
}

int umi_data_contains(const umi_data *d, uint32_t idx)
{
    if (!d) return fail(UMI_ERR_ARG, "d is NULL");
    if (idx >= d->n) return fail(UMI_ERR_ARG, "idx %u outside 0..%u", idx, d->n);
    return d->present[idx] ? 1 : 0;
}

void umi_data_free(umi_data *d) { delete d; }

} // extern "C"
