// knobs.h -- every AFESP_* environment variable of the library, read in ONE place.
//
// Until round 5 the variables were read where they were used: some per call, some once per process through function-local statics
// (a test that needed the other value had to start a process of its own).  Now: knobs_refresh() -- called at the top of every C-ABI
// entry point that takes a context (capi.hip, guarded) and lazily by the first knobs() of a process -- looks at the environment's
// AFESP_ entries, and when they differ from what it saw last it parses ALL of them into a fresh table; knobs() hands out that table.
// So every knob, whatever it selects, follows the environment at the granularity of one C-ABI call, and none changes in the middle of
// one.  A table that has been handed out is never modified (two tables, swapped), so a thread of another context reading beside a
// refresh sees the old or the new table, whole.
//
// Three kinds (include/afesp.h):
//   PATH      selects which kernels evaluate a quantity (different summation orders, results equal to ~1e-13): what tests use to send
//             a small system down a large system's path, or to compare two forms of one product.  Not for production use.
//   TUNING    tile / slice / pool sizes and scheduling; results identical up to summation order.
//   DIAGNOSTIC printing and measurement builds; no effect on results.
#pragma once
#include <stdint.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include <unistd.h>   // environ

namespace afesp {

struct Knobs {
    // ---------------------------------------------------------------------------------------------------------------- PATH
    int64_t small_max = (int64_t)1 << 20;   // AFESP_SMALL_MAX   largest o^2 v^2 on the small-system paths (launch-fused / lanes); 0: every size large
    bool no_lanes = false;                  // AFESP_NO_LANES=1  small systems on one stream
    bool fused = true;                      // AFESP_FUSED=0     call-by-call iteration instead of the launch-fused one
    bool fused_lanes = true;                // AFESP_FUSED_LANES=0  big products of a levelled sequence on the main stream only
    int pp_sym = -1;                        // AFESP_PP_SYM=0/1  plain a <= b ladder / pair form (default: pp_sym_pays decides)
    bool ring_tg = true;                    // AFESP_RING_TG=0   ring products on the gather kernel at every size
    int64_t ring_tg_min = 3584;             // AFESP_RING_TG_MIN smallest o v on the grouped ring launches
    bool ring_pack = true;                  // AFESP_RING_PACK=0 k_asym_c + permuting copies instead of ring_pack_kernel
    bool large_tail = true;                 // AFESP_LARGE_TAIL=0  large systems: update, energy, DIIS push as three kernels
    bool tall = true;                       // AFESP_TALL=0      no streamed tall x skinny kernel
    int64_t tall_min = (int64_t)1 << 17;    // AFESP_TALL_MIN    its smallest tall extent
    bool tall_dual = true;                  // AFESP_TALL_DUAL=0 y and x_voov as two launches
    bool gett_sk = true;                    // AFESP_GETT_SK=0   whole (tile, K slice) items instead of stream-K
    bool t_gemm_gett = false;               // AFESP_T_GEMM=gett (T) products on the grouped gather kernel
    bool t_one_pool = false;                // AFESP_T_ONE_POOL  the (T) block pool as one allocation
    bool cc_reinit = true;                  // AFESP_CC_REINIT=0 afesp_ccsd_init always builds a fresh state
    int cc_shard = -1;                      // AFESP_CC_SHARD=0/1  rank split of the iteration off / on (default: afesp_ccsd_set_split)
    bool cc_time_slice = false;             // AFESP_CC_TIME_SLICE="rank,world"  measurement: one rank's share without a communicator
    int cc_time_rank = 0, cc_time_world = 1;
    int ao2mo_tg = -1;                      // AFESP_AO2MO_TG=0/1   quarter transforms never / always on the LDS-DMA GEMM (default: even n >= 96)
    bool ao2mo_pair = true;                 // AFESP_AO2MO_PAIR=0   n <= 64: gather-GEMM form instead of the LDS-resident pair transform
    bool ao2mo_mixed = true;                // AFESP_AO2MO_MIXED=0  128-row tiles only
    bool ao2mo_pad = true;                  // AFESP_AO2MO_PAD=0    LDS-DMA transforms: temporaries with columns of n doubles instead of 16 ceil(n / 16)
    int ao2mo_blocked = -1;                 // AFESP_AO2MO_BLOCKED=0/1  whole tensor / slab by slab (default: by size)
    bool mp2_packed = true;                 // AFESP_MP2_PACKED=0   the five-launch MP2 energy at every size
    bool no_graph = false;                  // AFESP_NO_GRAPH=1     call-by-call small path: never capture a graph
    int graph_after = 40;                   // AFESP_GRAPH_AFTER    ... capture after this many iterations
    bool no_preload = false;                // AFESP_NO_PRELOAD=1   no start-up thread
    bool preload_lanes = false;             // AFESP_PRELOAD_LANES=1  streams of the call-by-call path made ahead of time
    bool preload_gett = false;              // AFESP_PRELOAD_GETT=1   the gather kernel's module loaded ahead of time
    // -------------------------------------------------------------------------------------------------------------- TUNING
    int pp_split = 0;                       // AFESP_PP_SPLIT    K slices of the ladder's two pair products (0: the launcher's choice)
    int pp_tiles[6] = {0, 0, 0, 0, 0, 0};   // AFESP_PP_TILES="tm,tn,split,tm,tn,split"
    bool pp_tiles_set = false;
    int64_t repack_min = 256;               // AFESP_REPACK_MIN
    int64_t plan_device_from = 32768;       // AFESP_PLAN_DEVICE_FROM
    double fused_big_flop = 4e8;            // AFESP_FUSED_BIG_FLOP
    int64_t fused_items = 0;                // AFESP_FUSED_ITEMS    target wave items per product stage (0: 8 per CU)
    int64_t fused_min_steps = 8;            // AFESP_FUSED_MIN_STEPS
    int64_t fused_max_mfma = 128;           // AFESP_FUSED_MAX_MFMA
    int64_t fused_nb = 4;                   // AFESP_FUSED_NB       pipeline depth code of the product stages (2: shallower)
    int split_below = 192;                  // AFESP_SPLIT_BELOW
    int split_min_steps = 4;                // AFESP_SPLIT_MIN_STEPS
    int tg_patch = 64;                      // AFESP_TG_PATCH       tiles per XCD patch of the LDS-DMA GEMM
    int tg_grid = 0;                        // AFESP_TG_GRID        diagnostic: fewer workgroups
    int tg_prio_shift = 11;                 // AFESP_TG_PRIO_SHIFT  priority time slice, 2^x * 10 ns
    int tg_dynamic = 1;                     // AFESP_TG_DYNAMIC=0/1/2  tickets never / for long launches / also for short ones (tests)
    int t_block = 0;                        // AFESP_T_BLOCK        occupied block size of the (T) enumeration (0: chosen)
    int64_t t_pool_gib = -1;                // AFESP_T_POOL_GIB     (T) block pool budget (-1: a quarter of the device, <= 64 GiB)
    int64_t t_split_tiles = 1024;           // AFESP_T_SPLIT_TILES
    // ---------------------------------------------------------------------------------------------------------- DIAGNOSTIC
    int tg_dbg = 0;                         // AFESP_TG_DBG=1       measurement: no C stores
    bool graph_debug = false, preload_debug = false, fused_debug = false, fused_per_op = false, gett_debug = false, t_debug = false;
    bool contract_trace = false, plan_verify = false, stamps_grouped = false;
};

namespace knobs_detail {
inline const char* get(const char* name) { return getenv(name); }
inline bool is(const char* name, char c) { const char* e = get(name); return e && e[0] == c; }
inline void parse(Knobs& k)
{
    k = Knobs();
    const char* e;
    if ((e = get("AFESP_SMALL_MAX"))) k.small_max = (int64_t)atof(e);
    k.no_lanes = is("AFESP_NO_LANES", '1');
    k.fused = !is("AFESP_FUSED", '0');
    k.fused_lanes = !is("AFESP_FUSED_LANES", '0');
    if ((e = get("AFESP_PP_SYM"))) k.pp_sym = e[0] == '1' ? 1 : 0;
    k.ring_tg = !is("AFESP_RING_TG", '0');
    if ((e = get("AFESP_RING_TG_MIN"))) k.ring_tg_min = (int64_t)atoll(e);
    k.ring_pack = !is("AFESP_RING_PACK", '0');
    k.large_tail = !is("AFESP_LARGE_TAIL", '0');
    k.tall = !is("AFESP_TALL", '0');
    if ((e = get("AFESP_TALL_MIN"))) k.tall_min = (int64_t)atof(e);
    k.tall_dual = !is("AFESP_TALL_DUAL", '0');
    k.gett_sk = !is("AFESP_GETT_SK", '0');
    k.t_gemm_gett = (e = get("AFESP_T_GEMM")) && !strcmp(e, "gett");
    k.t_one_pool = get("AFESP_T_ONE_POOL") != nullptr;
    k.cc_reinit = !is("AFESP_CC_REINIT", '0');
    if ((e = get("AFESP_CC_SHARD"))) k.cc_shard = e[0] == '1' ? 1 : 0;
    if ((e = get("AFESP_CC_TIME_SLICE"))) {
        int r = 0, w = 1;
        if (sscanf(e, "%d,%d", &r, &w) == 2 && w > 1 && r >= 0 && r < w) { k.cc_time_slice = true; k.cc_time_rank = r; k.cc_time_world = w; }
    }
    if ((e = get("AFESP_AO2MO_TG"))) k.ao2mo_tg = e[0] == '1' ? 1 : 0;
    k.ao2mo_pair = !is("AFESP_AO2MO_PAIR", '0');
    k.ao2mo_mixed = !is("AFESP_AO2MO_MIXED", '0');
    k.ao2mo_pad = !is("AFESP_AO2MO_PAD", '0');
    if ((e = get("AFESP_AO2MO_BLOCKED"))) k.ao2mo_blocked = e[0] == '1' ? 1 : 0;
    k.mp2_packed = !is("AFESP_MP2_PACKED", '0');
    k.no_graph = is("AFESP_NO_GRAPH", '1');
    if ((e = get("AFESP_GRAPH_AFTER"))) k.graph_after = atoi(e);
    k.no_preload = is("AFESP_NO_PRELOAD", '1');
    k.preload_lanes = is("AFESP_PRELOAD_LANES", '1');
    k.preload_gett = is("AFESP_PRELOAD_GETT", '1');
    if ((e = get("AFESP_PP_SPLIT"))) k.pp_split = atoi(e);
    k.pp_tiles[2] = k.pp_tiles[5] = k.pp_split;
    if ((e = get("AFESP_PP_TILES"))) {
        k.pp_tiles_set = true;
        sscanf(e, "%d,%d,%d,%d,%d,%d", &k.pp_tiles[0], &k.pp_tiles[1], &k.pp_tiles[2], &k.pp_tiles[3], &k.pp_tiles[4], &k.pp_tiles[5]);
    }
    if ((e = get("AFESP_REPACK_MIN"))) k.repack_min = (int64_t)atoll(e);
    if ((e = get("AFESP_PLAN_DEVICE_FROM"))) k.plan_device_from = (int64_t)atoll(e);
    if ((e = get("AFESP_FUSED_BIG_FLOP"))) k.fused_big_flop = atof(e);
    if ((e = get("AFESP_FUSED_ITEMS"))) k.fused_items = (int64_t)atoll(e);
    if ((e = get("AFESP_FUSED_MIN_STEPS"))) k.fused_min_steps = (int64_t)atoll(e);
    if ((e = get("AFESP_FUSED_MAX_MFMA"))) k.fused_max_mfma = (int64_t)atoll(e);
    if ((e = get("AFESP_FUSED_NB"))) k.fused_nb = (int64_t)atoll(e);
    if ((e = get("AFESP_SPLIT_BELOW"))) k.split_below = atoi(e);
    if ((e = get("AFESP_SPLIT_MIN_STEPS"))) k.split_min_steps = atoi(e) < 1 ? 1 : atoi(e);
    if ((e = get("AFESP_TG_PATCH"))) k.tg_patch = atoi(e);
    if ((e = get("AFESP_TG_GRID"))) k.tg_grid = atoi(e);
    if ((e = get("AFESP_TG_PRIO_SHIFT"))) k.tg_prio_shift = atoi(e);
    if ((e = get("AFESP_TG_DYNAMIC"))) k.tg_dynamic = e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1;
    if ((e = get("AFESP_T_BLOCK"))) k.t_block = atoi(e);
    if ((e = get("AFESP_T_POOL_GIB"))) k.t_pool_gib = (int64_t)atoll(e);
    if ((e = get("AFESP_T_SPLIT_TILES"))) k.t_split_tiles = (int64_t)atoll(e);
    if ((e = get("AFESP_TG_DBG"))) k.tg_dbg = atoi(e);
    k.graph_debug = get("AFESP_GRAPH_DEBUG") != nullptr;
    k.preload_debug = get("AFESP_PRELOAD_DEBUG") != nullptr;
    k.fused_debug = get("AFESP_FUSED_DEBUG") != nullptr;
    k.fused_per_op = get("AFESP_FUSED_PER_OP") != nullptr;
    k.gett_debug = get("AFESP_GETT_DEBUG") != nullptr;
    k.t_debug = get("AFESP_T_DEBUG") != nullptr;
    k.contract_trace = get("AFESP_CONTRACT_TRACE") != nullptr;
    k.plan_verify = get("AFESP_PLAN_VERIFY") != nullptr;
    k.stamps_grouped = get("AFESP_STAMPS_GROUPED") != nullptr;
}
// FNV-1a over the environment's AFESP_ entries (name and value); 1 when there is none
inline uint64_t fingerprint()
{
    uint64_t h = 1469598103934665603ull;
    for (char** p = environ; p && *p; ++p) {
        const char* s = *p;
        if (s[0] != 'A' || strncmp(s, "AFESP_", 6) != 0) continue;
        for (; *s; ++s) h = (h ^ (unsigned char)*s) * 1099511628211ull;
        h = (h ^ 0xffu) * 1099511628211ull;
    }
    return h | 1ull;
}
struct Store {
    Knobs tab[2];
    std::atomic<int> cur{0};
    std::atomic<uint64_t> seen{0};   // fingerprint the current table was parsed from (0: never)
    std::mutex mu;
};
inline Store& store()
{
    static Store s;
    return s;
}
}  // namespace knobs_detail

// looks at the environment; re-parses every knob when an AFESP_ entry has changed since the last look
inline void knobs_refresh()
{
    knobs_detail::Store& s = knobs_detail::store();
    const uint64_t f = knobs_detail::fingerprint();
    if (s.seen.load(std::memory_order_acquire) == f) return;
    std::lock_guard<std::mutex> lk(s.mu);
    if (s.seen.load(std::memory_order_relaxed) == f) return;
    const int next = s.cur.load(std::memory_order_relaxed) ^ 1;
    knobs_detail::parse(s.tab[next]);
    s.cur.store(next, std::memory_order_release);
    s.seen.store(f, std::memory_order_release);
}
inline const Knobs& knobs()
{
    knobs_detail::Store& s = knobs_detail::store();
    if (s.seen.load(std::memory_order_acquire) == 0) knobs_refresh();
    return s.tab[s.cur.load(std::memory_order_acquire)];
}

}  // namespace afesp
