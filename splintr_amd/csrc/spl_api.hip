// spl_api.hip -- the C ABI (include/splintr_hip.h): handle, device tables, workspace, launch order.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/splintr_hip.h"
#include "spl_kernels.hip"
#include "spl_tables.h"

using namespace spl;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(SPL_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

constexpr size_t QCOUNT_WORDS = 16;      // Batch::qcount
enum { KI_MARK = 0, KI_SPECIAL, KI_PRETOK, KI_DEFER, KI_BPELANES, KI_BPELONG, KI_COUNT, KI_SCAN, KI_COMPACT, KI_N };
const char* const k_names[KI_N] = {"memset+k_mark_docs", "k_special_scan", "k_pretok", "k_deferred_wave", "k_bpe_lanes64|k_bpe_segments",
                                   "k_bpe_long", "k_count", "k_scan", "k_compact_docs|k_tile_out"};

template <class T> int dev_upload(const std::vector<T>& v, const T** out) {
    void* p = nullptr;
    const size_t bytes = v.size() * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    if (bytes) HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *out = (const T*)p;
    return SPL_OK;
}

struct Special { std::string lit; uint32_t id; };

}  // namespace

struct spl_tokenizer {
    int device = 0;
    HostTables ht;
    DeviceTables dt{};
    const uint32_t* d_tok_off = nullptr;
    const uint8_t* d_tok_bytes = nullptr;
    std::vector<Special> specials;
    uint32_t max_special_id = 0;
    uint8_t* d_sp_lits = nullptr;      // uploaded lazily; invalidated by spl_add_special
    bool sp_uploaded = false;
    // workspace
    uint64_t cap_bytes = 0, cap_docs = 0;
    uint32_t* d_zero = nullptr;    // [tbits | tstart | skip | qcount]
    size_t zero_words = 0, bitmap_words = 0;
    uint32_t* d_stage = nullptr;
    uint32_t* d_rank = nullptr;
    uint32_t* d_aux = nullptr;    // inside d_rank's allocation
    uint2* d_q64 = nullptr; uint2* d_qlong = nullptr; uint32_t* d_qdefer = nullptr;
    uint32_t qcap64 = 0, qcaplong = 0, qcapdefer = 0;
    unsigned long long* d_dbg = nullptr;
    uint32_t* d_blk = nullptr;
    // tile-owned mode: tile records, the tiles' token slots, group sums; and whether the token bitmap may hold
    // stale bits (after hipMalloc or a multi-pass call) -- the single-pass kernel needs it all-zero
    TileDesc* d_tdesc = nullptr;
    uint32_t* d_tile_bits = nullptr; uint32_t* d_tcnt = nullptr;     // queue mode
    uint32_t* d_tile_ids = nullptr;
    uint32_t* d_tctl = nullptr;
    uint32_t tgroups = 0, tpar = 0;
    bool bitmap_dirty = true;
    // host-path staging
    uint8_t* d_in_text = nullptr; uint64_t* d_in_off = nullptr; uint32_t* d_out_ids = nullptr; uint64_t* d_out_off = nullptr;
    uint64_t in_cap_bytes = 0, in_cap_docs = 0;
    // profiling
    bool prof = false;
    hipEvent_t ev[KI_N + 1]{};
    bool ev_ready = false;
    double prof_ms[SPL_MAX_KERNELS]{};
    uint64_t prof_n[SPL_MAX_KERNELS]{};
    uint32_t* last_qcount = nullptr;
    bool dbg_on = false;
    int stop_phase = 0;     // spl_debug_phases bits 3..5 (profiling builds of the instruction mix per phase)
    int force_tile = 0;     // 0 auto, 1 small tiles, 2 large tiles (spl_debug_phases bit 1/2)
};

struct spl_result {
    std::vector<uint32_t> ids;
    std::vector<uint64_t> off;
};

namespace {

void free_workspace(spl_tokenizer* t) {
    hipFree(t->d_zero); hipFree(t->d_stage); hipFree(t->d_rank);
    hipFree(t->d_q64); hipFree(t->d_qlong); hipFree(t->d_qdefer); hipFree(t->d_blk); hipFree(t->d_dbg);
    hipFree(t->d_tdesc); hipFree(t->d_tile_ids); hipFree(t->d_tctl); hipFree(t->d_tile_bits); hipFree(t->d_tcnt);
    t->d_tdesc = nullptr; t->d_tile_ids = nullptr; t->d_tctl = nullptr; t->d_tile_bits = nullptr; t->d_tcnt = nullptr;
    t->d_zero = nullptr; t->d_stage = nullptr; t->d_rank = nullptr;
    t->d_q64 = nullptr; t->d_qlong = nullptr; t->d_qdefer = nullptr; t->d_blk = nullptr; t->d_dbg = nullptr;
    t->cap_bytes = t->cap_docs = 0;
}

int reserve(spl_tokenizer* t, uint64_t max_bytes, uint64_t max_docs) {
    if (max_bytes <= t->cap_bytes && max_docs <= t->cap_docs) return SPL_OK;
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    const uint64_t nb = std::max<uint64_t>(max_bytes, t->cap_bytes), nd = std::max<uint64_t>(max_docs, t->cap_docs);
    free_workspace(t);
    const size_t nblk = (size_t)(nb / RANK_BLK) + 2;
    t->bitmap_words = nblk * 32 + 64;
    t->zero_words = 3 * t->bitmap_words + QCOUNT_WORDS;
    HIP_TRY(hipMalloc((void**)&t->d_zero, t->zero_words * 4));
    HIP_TRY(hipMalloc((void**)&t->d_stage, (nb + 8192) * 4));
    HIP_TRY(hipMalloc((void**)&t->d_rank, (nb + 8192) * 12));   // ranks + two words of aux per byte
    t->d_aux = t->d_rank + (nb + 8192);
    const size_t tiles_s = (size_t)(nb / TileGeom<SPL_TILE_SMALL>::TBv) + 2;
    t->qcaplong = (uint32_t)(nb / 2 + 64);          // long chunks AND every miss of a deferred segment
    t->qcapdefer = (uint32_t)(2 * tiles_s + 64);
    t->qcap64 = (uint32_t)(nb / 17 + 64);
    HIP_TRY(hipMalloc((void**)&t->d_q64, (size_t)t->qcap64 * 8));
    HIP_TRY(hipMalloc((void**)&t->d_dbg, (16 + 4 * SPL_DEBUG_BLOCKS) * 8));
    HIP_TRY(hipMalloc((void**)&t->d_qlong, (size_t)t->qcaplong * 8));
    HIP_TRY(hipMalloc((void**)&t->d_qdefer, (size_t)t->qcapdefer * 4));
    HIP_TRY(hipMalloc((void**)&t->d_blk, (nblk + 2) * 4));
    {
        const size_t dbytes = (size_t)std::min<uint64_t>(nb, std::max<uint64_t>(SPL_DIRECT_MAX_BYTES, SPL_QUEUE_MAX_BYTES));
        const size_t tiles = dbytes / TileGeom<SPL_TILE_SMALL>::TBv + 2;
        t->tgroups = (uint32_t)(tiles / 64 + 2);
        HIP_TRY(hipMalloc((void**)&t->d_tdesc, tiles * sizeof(TileDesc)));
        HIP_TRY(hipMalloc((void**)&t->d_tile_ids, tiles * (size_t)(TileGeom<SPL_TILE_SMALL>::Wv + 1) * 4));
        HIP_TRY(hipMalloc((void**)&t->d_tile_bits, tiles * (size_t)TILE_BITS_W * 4));
        HIP_TRY(hipMalloc((void**)&t->d_tcnt, tiles * 4));
        HIP_TRY(hipMalloc((void**)&t->d_tctl, (16 + 2 * (size_t)t->tgroups) * 4));
        HIP_TRY(hipMemset(t->d_tctl, 0, (16 + 2 * (size_t)t->tgroups) * 4));
        t->tpar = 0;
    }
    t->bitmap_dirty = true;
    t->cap_bytes = nb;
    t->cap_docs = nd;
    return SPL_OK;
}

int upload_specials(spl_tokenizer* t) {
    if (t->sp_uploaded) return SPL_OK;
    // 32-byte header: the set of first bytes (256 bits); then one record per literal
    std::vector<uint8_t> recs(SP_HDR + t->specials.size() * SP_REC + 16, 0);
    for (size_t k = 0; k < t->specials.size(); k++) {
        const uint8_t c0 = (uint8_t)t->specials[k].lit[0];
        recs[c0 >> 3] |= (uint8_t)(1u << (c0 & 7));
        uint8_t* r = recs.data() + SP_HDR + k * SP_REC;
        r[0] = (uint8_t)t->specials[k].lit.size();
        memcpy(r + 4, &t->specials[k].id, 4);
        memcpy(r + 8, t->specials[k].lit.data(), t->specials[k].lit.size());
    }
    HIP_TRY(hipDeviceSynchronize());
    hipFree(t->d_sp_lits);
    t->d_sp_lits = nullptr;
    HIP_TRY(hipMalloc((void**)&t->d_sp_lits, recs.size()));
    HIP_TRY(hipMemcpy(t->d_sp_lits, recs.data(), recs.size(), hipMemcpyHostToDevice));
    t->sp_uploaded = true;
    return SPL_OK;
}

struct SlabOut { uint32_t* d_slab = nullptr; uint64_t cap_words = 0, max_docs = 0; };

int launch_all(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off, uint64_t n_docs,
               uint32_t flags, uint32_t* d_ids, uint64_t ids_cap, uint64_t* d_out_off, hipStream_t s,
               const SlabOut* so = nullptr) {
    if (((uintptr_t)d_utf8 & 15) != 0) return fail(SPL_EINVAL, "text buffer must be 16-byte aligned");
    const bool special = (flags & SPL_WITH_SPECIAL) && !t->specials.empty();
    if (special) { int rc0 = upload_specials(t); if (rc0) return rc0; }
    if (n_bytes > 0x7FFF0000ull) return fail(SPL_EINVAL, "n_bytes per device call must be < 2^31 - 65536");
    if (n_docs > 0xFFFFFFF0ull) return fail(SPL_EINVAL, "n_docs per device call must be < 2^32 - 16");
    int rc = reserve(t, n_bytes, n_docs);
    if (rc) return rc;
    if (t->prof && !t->ev_ready) {
        for (auto& e : t->ev) HIP_TRY(hipEventCreate(&e));
        t->ev_ready = true;
    }
    Batch b{};
    b.text = d_utf8; b.n_bytes = (uint32_t)n_bytes; b.doc_off = d_doc_off; b.n_docs = (uint32_t)n_docs;
    b.n_blk = (uint32_t)(n_bytes / RANK_BLK + 1);
    const size_t uw = (size_t)b.n_blk * 32 + 32;
    // bitmaps and queue counters packed back to back for THIS batch size: one memset clears them
    b.tbits = t->d_zero; b.tstart = t->d_zero + uw;
    b.skip = special ? t->d_zero + 2 * uw : nullptr;
    b.qcount = t->d_zero + (special ? 3 : 2) * uw;
    t->last_qcount = b.qcount;
    b.sp_lits = t->d_sp_lits; b.n_special = special ? (uint32_t)t->specials.size() : 0u;
    b.stage = t->d_stage; b.rank_scr = t->d_rank; b.aux = t->d_aux;
    b.q64 = t->d_q64; b.qlong = t->d_qlong; b.qdefer = t->d_qdefer;
    b.qcap64 = t->qcap64; b.qcaplong = t->qcaplong; b.qcapdefer = t->qcapdefer;
    b.dbg = (t->dbg_on || t->prof) ? t->d_dbg : nullptr;
    b.stop_phase = (uint32_t)t->stop_phase;
    if (t->prof) {
        const unsigned long long init[2] = {~0ull, 0ull};
        HIP_TRY(hipMemcpyAsync(t->d_dbg + 14, init, 16, hipMemcpyHostToDevice, s));
    }
    b.blk_base = t->d_blk;
    b.ids_out = d_ids; b.ids_cap = ids_cap; b.off_out = d_out_off;

    const bool pf = t->prof;
#define MARK(i) do { if (pf) HIP_TRY(hipEventRecord(t->ev[i], s)); } while (0)
    // small batches: small tiles (occupancy hides latency); large batches: 4 KiB tiles
    // queue mode: tile-owned tiles + global queues for what is long, for batches beyond the two-launch limit
    const bool queue_mode = !special && (t->force_tile == 4 || (t->force_tile == 0 && n_bytes > SPL_DIRECT_MAX_BYTES)) &&
                            n_bytes <= SPL_QUEUE_MAX_BYTES;
    const bool small_tiles = queue_mode || t->force_tile == 1 || t->force_tile == 3 ||
                             (t->force_tile == 0 && n_bytes <= SPL_DIRECT_MAX_BYTES);
    const uint32_t tile_bytes = small_tiles ? TileGeom<SPL_TILE_SMALL>::TBv : TileGeom<SPL_TILE_LARGE>::TBv;
    const uint32_t ntiles = (uint32_t)((n_bytes + tile_bytes - 1) / tile_bytes);
    // (A/B on the 1 MB bench batch: folding these launches together -- clean-after-use bitmaps, one
    //  tail kernel with a grid barrier and a last-workgroup scan -- was SLOWER than this plain
    //  sequence: back-to-back launches overlap their dispatch with the previous kernel, while
    //  single-workgroup tails and agent-scope fences sit on the critical path.)
    // Single pass: small batches without special tokens are finished by ONE kernel.
    bool fused_scan_used = false;
    const bool direct = !queue_mode && small_tiles && t->force_tile != 3 && n_bytes <= SPL_DIRECT_MAX_BYTES;
    if (queue_mode) {
        t->bitmap_dirty = true;
        HIP_TRY(hipMemsetAsync(t->d_zero, 0, (2 * uw + QCOUNT_WORDS) * 4, s));
        b.tdesc = t->d_tdesc; b.tile_ids = t->d_tile_ids; b.tctl = t->d_tctl; b.tile_bits = t->d_tile_bits; b.tcnt = t->d_tcnt;
        b.tgroups = t->tgroups; b.tpar = t->tpar; b.tslot = (uint32_t)TileGeom<SPL_TILE_SMALL>::Wv + 1u;
        t->tpar ^= 1u;
        MARK(KI_MARK);
        if (n_docs) hipLaunchKernelGGL(k_mark_docs, dim3((uint32_t)((n_docs + 255) / 256)), dim3(256), 0, s, b);
        MARK(KI_SPECIAL); MARK(KI_PRETOK);
        hipLaunchKernelGGL((k_pretok<SPL_TILE_SMALL, false, true>), dim3(ntiles), dim3(NT), 0, s, t->dt, b);
        MARK(KI_DEFER);
        hipLaunchKernelGGL(k_deferred_wave, dim3(256), dim3(64), 0, s, t->dt, b);
        MARK(KI_BPELANES);
        hipLaunchKernelGGL(k_bpe_segments, dim3(std::min<uint32_t>(2048, ntiles / 4 + 8)), dim3(NT), 0, s, t->dt, b);
        MARK(KI_BPELONG);
        hipLaunchKernelGGL(k_bpe_long, dim3(std::min<uint32_t>(2048, ntiles / 4 + 8)), dim3(NT), 0, s, t->dt, b, 1);
        MARK(KI_COUNT);
        hipLaunchKernelGGL((k_range_count<SPL_TILE_SMALL>), dim3(ntiles), dim3(64), 0, s, b);
        MARK(KI_SCAN); MARK(KI_COMPACT);
        hipLaunchKernelGGL((k_range_out<SPL_TILE_SMALL>), dim3(ntiles), dim3(64), 0, s, b);
        MARK(KI_N);
    } else if (direct) {
        if (special) {
            // the three bitmaps are cleared per call; documents and literals are marked by the
            // multi-pass kernels, the tile kernel reads the bitmaps on top of its document search
            HIP_TRY(hipMemsetAsync(t->d_zero, 0, (3 * uw + QCOUNT_WORDS) * 4, s));
            t->bitmap_dirty = true;
        } else if (t->bitmap_dirty) {
            HIP_TRY(hipMemsetAsync(t->d_zero, 0, t->zero_words * 4, s));
            t->bitmap_dirty = false;
        }
        b.tdesc = t->d_tdesc; b.tile_ids = t->d_tile_ids; b.tctl = t->d_tctl;
        b.tgroups = t->tgroups; b.tpar = t->tpar; b.tslot = (uint32_t)TileGeom<SPL_TILE_SMALL>::Wv + 1u;
        if (so && ntiles) { b.slab = so->d_slab; b.slab_cap = (uint32_t)so->cap_words; b.slab_max_docs = (uint32_t)so->max_docs; }
        if (ntiles) t->tpar ^= 1u;              // k_tile_out zeroes the other parity's sums for the next call
        if (!special) b.tstart = nullptr;
        b.qcount = nullptr;
        t->last_qcount = nullptr;
        MARK(KI_MARK);
        if (special && n_docs) hipLaunchKernelGGL(k_mark_docs, dim3((uint32_t)((n_docs + 255) / 256)), dim3(256), 0, s, b);
        MARK(KI_SPECIAL);
        if (special && n_bytes) {
            hipLaunchKernelGGL(k_special_scan, dim3((uint32_t)((n_bytes + 255) / 256)), dim3(256), 0, s, b);
        }
        MARK(KI_PRETOK);
        if (ntiles) hipLaunchKernelGGL((k_pretok<SPL_TILE_SMALL, false, true>), dim3(ntiles), dim3(NT), 0, s, t->dt, b);
        else HIP_TRY(hipMemsetAsync(d_out_off, 0, (n_docs + 1) * 8, s));
        MARK(KI_DEFER); MARK(KI_BPELANES); MARK(KI_BPELONG); MARK(KI_COUNT); MARK(KI_SCAN); MARK(KI_COMPACT);
        if (ntiles) hipLaunchKernelGGL(k_tile_out, dim3(ntiles), dim3(NT), 0, s, b);
        MARK(KI_N);
    } else {
    t->bitmap_dirty = true;
    MARK(KI_MARK);
    HIP_TRY(hipMemsetAsync(t->d_zero, 0, ((special ? 3 : 2) * uw + QCOUNT_WORDS) * 4, s));
    if (n_docs) hipLaunchKernelGGL(k_mark_docs, dim3((uint32_t)((n_docs + 255) / 256)), dim3(256), 0, s, b);
    MARK(KI_SPECIAL);
    if (special && n_bytes) {
        hipLaunchKernelGGL(k_special_scan, dim3((uint32_t)((n_bytes + 255) / 256)), dim3(256), 0, s, b);
    }
    MARK(KI_PRETOK);
    if (ntiles) {
        if (small_tiles) hipLaunchKernelGGL((k_pretok<SPL_TILE_SMALL, false>), dim3(ntiles), dim3(NT), 0, s, t->dt, b);
        else hipLaunchKernelGGL((k_pretok<SPL_TILE_LARGE, true>), dim3(ntiles), dim3(NT), 0, s, t->dt, b);
    }
    MARK(KI_DEFER);
    if (ntiles) hipLaunchKernelGGL(k_deferred_wave, dim3(256), dim3(64), 0, s, t->dt, b);
    MARK(KI_BPELANES);
    if (ntiles && !small_tiles) hipLaunchKernelGGL(k_bpe_lanes64, dim3(256 * 5), dim3(64), 0, s, t->dt, b);
    if (ntiles && small_tiles) hipLaunchKernelGGL(k_bpe_segments, dim3(std::min<uint32_t>(2048, ntiles / 4 + 8)), dim3(NT), 0, s, t->dt, b);
    MARK(KI_BPELONG);
    if (ntiles) hipLaunchKernelGGL(k_bpe_long, dim3(std::min<uint32_t>(2048, ntiles / 4 + 8)), dim3(NT), 0, s, t->dt, b, small_tiles ? 1 : 0);
    MARK(KI_COUNT);
    const bool fused_scan = b.n_blk <= 8192;
    fused_scan_used = fused_scan;
    if (!fused_scan) hipLaunchKernelGGL(k_count, dim3((b.n_blk + 255) / 256), dim3(256), 0, s, b);
    MARK(KI_SCAN);
    if (fused_scan) hipLaunchKernelGGL(k_scan<true>, dim3(1), dim3(1024), 0, s, b);
    else hipLaunchKernelGGL(k_scan<false>, dim3(1), dim3(1024), 0, s, b);
    MARK(KI_COMPACT);
    {
        const uint32_t n_compact = (b.n_blk * 32 + NT - 1) / NT;
        const uint32_t n_docblk = (uint32_t)((n_docs + 1 + NT - 1) / NT);
        hipLaunchKernelGGL(k_compact_docs, dim3(n_compact + n_docblk), dim3(NT), 0, s, b, n_compact);
    }
    MARK(KI_N);
    }
#undef MARK
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            return fail(SPL_EDEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
        }
    }
    if (so && !(direct && ntiles))          // the slab copy of the result, where k_tile_out did not write it
        hipLaunchKernelGGL(k_gatherv_pack, dim3(256), dim3(256), 0, s, d_ids, d_out_off, (uint32_t)n_docs, so->d_slab,
                           (uint32_t)so->cap_words, (uint32_t)so->max_docs);
    if (pf) {
        HIP_TRY(hipEventSynchronize(t->ev[KI_N]));
        for (int i = 0; i < KI_N; i++) {
            // slots whose kernels were not launched in this mode would only show the event overhead
            const bool launched = queue_mode ? (i != KI_SPECIAL && i != KI_SCAN) : direct ? (i == KI_PRETOK || i == KI_COMPACT || (special && (i == KI_MARK || i == KI_SPECIAL)))
                                         : !((i == KI_SPECIAL && !special) ||
                                             (i == KI_COUNT && fused_scan_used));
            if (!launched) continue;
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, t->ev[i], t->ev[i + 1]));
            if (i == KI_PRETOK && ntiles) {
                // the dominant kernel is timed on the device's wall clock instead (see k_pretok)
                unsigned long long span[2];
                HIP_TRY(hipMemcpy(span, t->d_dbg + 14, 16, hipMemcpyDeviceToHost));
                int khz = 0;
                HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, t->device));
                if (khz > 0 && span[1] > span[0]) ms = (float)((double)(span[1] - span[0]) / (double)khz);
            }
            t->prof_ms[i] += ms;
            t->prof_n[i] += 1;
        }
    }
    return SPL_OK;
}

}  // namespace

extern "C" {

const char* spl_last_error(void) { return g_err.c_str(); }

int spl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

spl_tokenizer* spl_create(const void* vocab_splv, size_t vocab_len, const void* uclass_tab, size_t uclass_len,
                          const spl_opts* opts) {
    if (!vocab_splv || !uclass_tab || !opts) { fail(SPL_EINVAL, "spl_create: null argument"); return nullptr; }
    spl_tokenizer* t = new spl_tokenizer();
    std::string err;
    if (build_tables((const uint8_t*)vocab_splv, vocab_len, (const uint8_t*)uclass_tab, uclass_len, opts->pattern, t->ht, err)) {
        fail(SPL_EINVAL, "spl_create: " + err);
        delete t;
        return nullptr;
    }
    t->device = opts->device;
    auto up = [&]() -> int {
        HIP_TRY(hipSetDevice(t->device));
        int rc;
        if ((rc = dev_upload(t->ht.ucls_stage1, &t->dt.ucls_stage1))) return rc;
        if ((rc = dev_upload(t->ht.ucls_stage2, &t->dt.ucls_stage2))) return rc;
        if ((rc = dev_upload(t->ht.short_tab, &t->dt.short_tab))) return rc;
        if ((rc = dev_upload(t->ht.tiny_tab, &t->dt.tiny_tab))) return rc;
        if ((rc = dev_upload(t->ht.t8_tab, &t->dt.t8_tab))) return rc;
        if ((rc = dev_upload(t->ht.long_tab, &t->dt.long_tab))) return rc;
        if ((rc = dev_upload(t->ht.key_blob, &t->dt.key_blob))) return rc;
        if ((rc = dev_upload(t->ht.pair_tab, &t->dt.pair_tab))) return rc;
        if ((rc = dev_upload(t->ht.byte_id, &t->dt.byte_id))) return rc;
        if ((rc = dev_upload(t->ht.p8_tab, reinterpret_cast<const uint32_t**>(&t->dt.p8_tab)))) return rc;
        if ((rc = dev_upload(t->ht.len_mask, &t->dt.len_mask))) return rc;
        if ((rc = dev_upload(t->ht.tok_off, &t->d_tok_off))) return rc;
        if ((rc = dev_upload(t->ht.tok_bytes, &t->d_tok_bytes))) return rc;
        return SPL_OK;
    };
    if (up() != SPL_OK) { delete t; return nullptr; }
    t->dt.ucls_shift = t->ht.ucls_shift;
    t->dt.cjk_fast = t->ht.cjk_fast ? 1u : 0u;
    t->dt.short_mask = (uint32_t)(t->ht.short_tab.size() / SPL_SHORT_BUCKET) - 1;
    t->dt.tiny_mask = (uint32_t)(t->ht.tiny_tab.size() / (SPL_TINY_BUCKET * 2)) - 1;
    t->dt.t8_mask = (uint32_t)(t->ht.t8_tab.size() / SPL_T8_WORDS) - 1;
    t->dt.long_mask = (uint32_t)t->ht.long_tab.size() - 1;
    t->dt.pair_mask = (uint32_t)(t->ht.pair_tab.size() / SPL_PAIR_BUCKET) - 1;
    t->dt.p8_mask = (uint32_t)(t->ht.p8_tab.size() / 2) - 1;
    t->dt.tiny_free = t->ht.tiny_free; t->dt.t8_free = t->ht.t8_free;
    t->dt.max_key_len = t->ht.max_key_len;
    t->dt.pattern = (uint32_t)t->ht.pattern;
    t->dt.all_bytes = t->ht.all_bytes ? 1u : 0u;
    return t;
}

int spl_add_special(spl_tokenizer* t, const uint8_t* literal, size_t len, uint32_t id) {
    if (!t || !literal || len == 0) return fail(SPL_EINVAL, "spl_add_special: bad argument");
    if (len > (size_t)SP_MAXLEN) return fail(SPL_EINVAL, "spl_add_special: literal longer than 32 bytes");
    const std::string lit((const char*)literal, len);
    // The device scan treats every occurrence as a match, which equals Aho-Corasick's
    // non-overlapping Standard semantics only if no two occurrences can ever overlap.
    auto overlaps = [](const std::string& a, const std::string& b) {
        if (a.find(b) != std::string::npos || b.find(a) != std::string::npos) return true;
        for (size_t k = 1; k < a.size() && k < b.size(); k++) {
            if (a.compare(a.size() - k, k, b, 0, k) == 0) return true;   // suffix of a == prefix of b
            if (b.compare(b.size() - k, k, a, 0, k) == 0) return true;
        }
        return false;
    };
    for (size_t k = 1; k < lit.size(); k++)
        if (lit.compare(lit.size() - k, k, lit, 0, k) == 0)
            return fail(SPL_EINVAL, "spl_add_special: literal can overlap itself");
    for (const auto& sp : t->specials)
        if (overlaps(sp.lit, lit)) return fail(SPL_EINVAL, "spl_add_special: literal can overlap '" + sp.lit + "'");
    t->specials.push_back(Special{lit, id});
    t->sp_uploaded = false;
    t->max_special_id = std::max(t->max_special_id, id);
    return SPL_OK;
}

uint32_t spl_vocab_size(const spl_tokenizer* t) {
    if (!t) return 0;
    return std::max(t->ht.max_id, t->max_special_id) + 1;
}

void spl_destroy(spl_tokenizer* t) {
    if (!t) return;
    hipSetDevice(t->device);
    hipDeviceSynchronize();
    free_workspace(t);
    hipFree((void*)t->dt.ucls_stage1); hipFree((void*)t->dt.ucls_stage2); hipFree((void*)t->dt.short_tab);
    hipFree((void*)t->dt.tiny_tab); hipFree((void*)t->dt.t8_tab);
    hipFree((void*)t->dt.long_tab); hipFree((void*)t->dt.key_blob); hipFree((void*)t->dt.pair_tab);
    hipFree((void*)t->dt.byte_id); hipFree((void*)t->dt.p8_tab); hipFree((void*)t->dt.len_mask); hipFree((void*)t->d_tok_off); hipFree((void*)t->d_tok_bytes);
    hipFree(t->d_in_text); hipFree(t->d_in_off); hipFree(t->d_out_ids); hipFree(t->d_out_off); hipFree(t->d_sp_lits);
    if (t->ev_ready) for (auto& e : t->ev) hipEventDestroy(e);
    delete t;
}

int spl_reserve(spl_tokenizer* t, uint64_t max_bytes, uint64_t max_docs) {
    if (!t) return fail(SPL_EINVAL, "spl_reserve: null handle");
    return reserve(t, max_bytes, max_docs);
}

int spl_encode_batch_device(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                            uint64_t n_docs, uint32_t flags, uint32_t* d_ids, uint64_t ids_capacity,
                            uint64_t* d_out_off, void* hip_stream) {
    if (!t || !d_doc_off || !d_out_off || (n_bytes && (!d_utf8 || !d_ids)))
        return fail(SPL_EINVAL, "spl_encode_batch_device: null argument");
    HIP_TRY(hipSetDevice(t->device));
    return launch_all(t, d_utf8, n_bytes, d_doc_off, n_docs, flags, d_ids, ids_capacity, d_out_off, (hipStream_t)hip_stream);
}

int spl_encode_batch_device_packed(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                                   uint64_t n_docs, uint32_t flags, uint32_t* d_ids, uint64_t ids_capacity,
                                   uint64_t* d_out_off, uint32_t* d_slab, uint64_t cap_words, uint64_t max_docs,
                                   void* hip_stream) {
    if (!t || !d_doc_off || !d_out_off || !d_slab || (n_bytes && (!d_utf8 || !d_ids)))
        return fail(SPL_EINVAL, "spl_encode_batch_device_packed: null argument");
    if (cap_words < max_docs + 4 || n_docs > max_docs) return fail(SPL_EINVAL, "spl_encode_batch_device_packed: slab too small");
    HIP_TRY(hipSetDevice(t->device));
    SlabOut so;
    so.d_slab = d_slab; so.cap_words = cap_words; so.max_docs = max_docs;
    return launch_all(t, d_utf8, n_bytes, d_doc_off, n_docs, flags, d_ids, ids_capacity, d_out_off, (hipStream_t)hip_stream, &so);
}

int spl_encode_batch(spl_tokenizer* t, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, uint32_t flags,
                     spl_result** out) {
    if (!t || !doc_off || !out) return fail(SPL_EINVAL, "spl_encode_batch: null argument");
    if (doc_off[0] != 0) return fail(SPL_EINVAL, "spl_encode_batch: doc_off[0] must be 0");
    for (uint64_t d = 0; d < n_docs; d++)
        if (doc_off[d + 1] < doc_off[d]) return fail(SPL_EINVAL, "spl_encode_batch: doc_off must be non-decreasing");
    const uint64_t n_bytes = doc_off[n_docs];
    if (n_bytes && !utf8) return fail(SPL_EINVAL, "spl_encode_batch: null text");
    HIP_TRY(hipSetDevice(t->device));

    spl_result* r = new spl_result();
    r->off.assign(n_docs + 1, 0);
    // a device call takes < 2^31 bytes: walk the documents in slabs of about 1 GiB
    const uint64_t SLAB = 1ull << 30;
    uint64_t d0 = 0;
    while (d0 < n_docs) {
        uint64_t d1 = d0;
        while (d1 < n_docs && (d1 == d0 || doc_off[d1 + 1] - doc_off[d0] <= SLAB)) d1++;
        const uint64_t nb = doc_off[d1] - doc_off[d0], nd = d1 - d0;
        if (nb > 0x7FFF0000ull) { delete r; return fail(SPL_EINVAL, "spl_encode_batch: a single document exceeds 2^31 bytes"); }
        if (nb > t->in_cap_bytes || nd > t->in_cap_docs || !t->d_in_off) {
            HIP_TRY(hipDeviceSynchronize());
            hipFree(t->d_in_text); hipFree(t->d_in_off); hipFree(t->d_out_ids); hipFree(t->d_out_off);
            t->in_cap_bytes = std::max<uint64_t>(nb, t->in_cap_bytes);
            t->in_cap_docs = std::max<uint64_t>(nd, t->in_cap_docs);
            HIP_TRY(hipMalloc((void**)&t->d_in_text, t->in_cap_bytes + 64));
            HIP_TRY(hipMalloc((void**)&t->d_in_off, (t->in_cap_docs + 1) * 8));
            HIP_TRY(hipMalloc((void**)&t->d_out_ids, (t->in_cap_bytes + 16) * 4));
            HIP_TRY(hipMalloc((void**)&t->d_out_off, (t->in_cap_docs + 1) * 8));
        }
        std::vector<uint64_t> rel(nd + 1);
        for (uint64_t k = 0; k <= nd; k++) rel[k] = doc_off[d0 + k] - doc_off[d0];
        if (nb) HIP_TRY(hipMemcpyAsync(t->d_in_text, utf8 + doc_off[d0], nb, hipMemcpyHostToDevice, 0));
        HIP_TRY(hipMemcpyAsync(t->d_in_off, rel.data(), (nd + 1) * 8, hipMemcpyHostToDevice, 0));
        int rc = launch_all(t, t->d_in_text, nb, t->d_in_off, nd, flags, t->d_out_ids, nb, t->d_out_off, 0);
        if (rc) { delete r; return rc; }
        std::vector<uint64_t> oo(nd + 1);
        HIP_TRY(hipMemcpy(oo.data(), t->d_out_off, (nd + 1) * 8, hipMemcpyDeviceToHost));
        const uint64_t total = oo[nd], base = r->ids.size();
        r->ids.resize(base + total);
        if (total) HIP_TRY(hipMemcpy(r->ids.data() + base, t->d_out_ids, total * 4, hipMemcpyDeviceToHost));
        for (uint64_t k = 0; k <= nd; k++) r->off[d0 + k] = base + oo[k];
        d0 = d1;
    }
    *out = r;
    return SPL_OK;
}

const uint32_t* spl_result_tokens(const spl_result* r) { return r ? r->ids.data() : nullptr; }
const uint64_t* spl_result_offsets(const spl_result* r) { return r ? r->off.data() : nullptr; }
uint64_t spl_result_n_tokens(const spl_result* r) { return r ? r->ids.size() : 0; }
uint64_t spl_result_n_docs(const spl_result* r) { return r ? r->off.size() - 1 : 0; }
void spl_result_free(spl_result* r) { delete r; }

int spl_decode_batch(spl_tokenizer* t, const uint32_t* ids, const uint64_t* ids_off, uint64_t n_docs, uint8_t** out_bytes,
                     uint64_t** out_off) {
    if (!t || !ids_off || !out_bytes || !out_off) return fail(SPL_EINVAL, "spl_decode_batch: null argument");
    HIP_TRY(hipSetDevice(t->device));
    const uint64_t n = ids_off[n_docs] - ids_off[0];
    const uint32_t* src = ids + ids_off[0];
    // special ids decode to their literal on the host side (few); vocabulary ids gather on the GPU
    uint32_t* d_ids = nullptr; uint64_t* d_len = nullptr; uint8_t* d_out = nullptr;
    std::vector<uint64_t> len(n + 1, 0);
    if (n) {
        HIP_TRY(hipMalloc((void**)&d_ids, n * 4));
        HIP_TRY(hipMalloc((void**)&d_len, n * 8));
        HIP_TRY(hipMemcpy(d_ids, src, n * 4, hipMemcpyHostToDevice));
        DecodeArgs a{d_ids, n, t->d_tok_off, t->d_tok_bytes, t->ht.max_id, d_len, nullptr};
        hipLaunchKernelGGL(k_decode_len, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, a);
        HIP_TRY(hipMemcpy(len.data(), d_len, n * 8, hipMemcpyDeviceToHost));
    }
    // splice special literals: they are not in the device table (id > max_id or unmapped)
    std::vector<const std::string*> sp_of(n, nullptr);
    for (uint64_t i = 0; i < n; i++) {
        const bool in_vocab = src[i] <= t->ht.max_id && t->ht.tok_off[src[i] + 1] > t->ht.tok_off[src[i]];
        if (!in_vocab)
            for (const auto& s : t->specials)
                if (s.id == src[i]) { sp_of[i] = &s.lit; len[i] = s.lit.size(); break; }
    }
    uint64_t acc = 0;
    for (uint64_t i = 0; i < n; i++) { const uint64_t l = len[i]; len[i] = acc; acc += l; }
    len[n] = acc;
    uint8_t* ob = (uint8_t*)malloc(acc ? acc : 1);
    uint64_t* oo = (uint64_t*)malloc((n_docs + 1) * 8);
    if (n) {
        HIP_TRY(hipMalloc((void**)&d_out, acc ? acc : 16));
        HIP_TRY(hipMemcpy(d_len, len.data(), n * 8, hipMemcpyHostToDevice));
        DecodeArgs a{d_ids, n, t->d_tok_off, t->d_tok_bytes, t->ht.max_id, d_len, d_out};
        hipLaunchKernelGGL(k_decode_copy, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, a);
        if (acc) HIP_TRY(hipMemcpy(ob, d_out, acc, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n; i++)
            if (sp_of[i]) memcpy(ob + len[i], sp_of[i]->data(), sp_of[i]->size());
        hipFree(d_ids); hipFree(d_len); hipFree(d_out);
    }
    for (uint64_t d = 0; d <= n_docs; d++) oo[d] = len[ids_off[d] - ids_off[0]];
    *out_bytes = ob;
    *out_off = oo;
    return SPL_OK;
}

void spl_free(void* p) { free(p); }

int spl_profile_enable(spl_tokenizer* t, int on) {
    if (!t) return fail(SPL_EINVAL, "null handle");
    t->prof = on != 0;
    return SPL_OK;
}
int spl_profile_reset(spl_tokenizer* t) {
    if (!t) return fail(SPL_EINVAL, "null handle");
    memset(t->prof_ms, 0, sizeof t->prof_ms);
    memset(t->prof_n, 0, sizeof t->prof_n);
    return SPL_OK;
}
int spl_profile_read(spl_tokenizer* t, double ms_out[SPL_MAX_KERNELS], uint64_t launches_out[SPL_MAX_KERNELS]) {
    if (!t) return fail(SPL_EINVAL, "null handle");
    memcpy(ms_out, t->prof_ms, sizeof t->prof_ms);
    memcpy(launches_out, t->prof_n, sizeof t->prof_n);
    return SPL_OK;
}
const char* spl_kernel_name(int index) { return (index >= 0 && index < KI_N) ? k_names[index] : nullptr; }

int spl_gatherv_pack(spl_tokenizer* t, const uint32_t* d_ids, const uint64_t* d_out_off, uint64_t n_docs, uint32_t* d_slab,
                     uint64_t cap_words, uint64_t max_docs, void* hip_stream) {
    if (!t || !d_ids || !d_out_off || !d_slab) return fail(SPL_EINVAL, "spl_gatherv_pack: null argument");
    if (n_docs > max_docs || cap_words < max_docs + 4 || cap_words > 0xFFFFFFFFull)
        return fail(SPL_EINVAL, "spl_gatherv_pack: slab too small for the document table");
    HIP_TRY(hipSetDevice(t->device));
    hipLaunchKernelGGL(k_gatherv_pack, dim3(256), dim3(256), 0, (hipStream_t)hip_stream, d_ids, d_out_off, (uint32_t)n_docs,
                       d_slab, (uint32_t)cap_words, (uint32_t)max_docs);
    HIP_TRY(hipGetLastError());
    return SPL_OK;
}

int spl_gatherv_unpack(spl_tokenizer* t, const uint32_t* d_slabs, uint32_t world, uint64_t cap_words, uint64_t max_docs,
                       uint32_t* d_all_ids, uint64_t all_ids_cap, uint64_t* d_all_off, uint32_t* d_status, void* hip_stream) {
    if (!t || !d_slabs || !d_all_ids || !d_all_off || !d_status || world == 0)
        return fail(SPL_EINVAL, "spl_gatherv_unpack: bad argument");
    HIP_TRY(hipSetDevice(t->device));
    hipLaunchKernelGGL(k_gatherv_unpack, dim3(128, world), dim3(256), 0, (hipStream_t)hip_stream, d_slabs, world,
                       (uint32_t)cap_words, (uint32_t)max_docs, d_all_ids, all_ids_cap, d_all_off, d_status,
                       (uint64_t)cap_words, (uint64_t)0);
    HIP_TRY(hipGetLastError());
    return SPL_OK;
}

int spl_gatherv_unpack_group(spl_tokenizer* t, const uint32_t* d_slabs, uint32_t world, uint32_t depth, uint32_t n_batches,
                             uint64_t cap_words, uint64_t max_docs, uint32_t* d_all_ids, uint64_t all_ids_cap,
                             uint64_t* d_all_off, uint64_t off_stride, uint32_t* d_status, void* hip_stream) {
    if (!t || !d_slabs || !d_all_ids || !d_all_off || !d_status || world == 0 || depth == 0 || n_batches > depth)
        return fail(SPL_EINVAL, "spl_gatherv_unpack_group: bad argument");
    if (n_batches == 0) return SPL_OK;
    HIP_TRY(hipSetDevice(t->device));
    hipLaunchKernelGGL(k_gatherv_unpack, dim3(128, world, n_batches), dim3(256), 0, (hipStream_t)hip_stream, d_slabs, world,
                       (uint32_t)cap_words, (uint32_t)max_docs, d_all_ids, all_ids_cap, d_all_off, d_status,
                       (uint64_t)depth * cap_words, off_stride);
    HIP_TRY(hipGetLastError());
    return SPL_OK;
}

int spl_debug_blocks(spl_tokenizer* t, unsigned long long* out, int max_blocks) {
    if (!t || !out) return fail(SPL_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!t->d_dbg) return 0;
    const int n = max_blocks < SPL_DEBUG_BLOCKS ? max_blocks : SPL_DEBUG_BLOCKS;
    HIP_TRY(hipMemcpy(out, t->d_dbg + 16, (size_t)n * 32, hipMemcpyDeviceToHost));
    return n;
}

int spl_debug_phases(spl_tokenizer* t, int enable, unsigned long long stamps_out[16]) {
    if (!t) return fail(SPL_EINVAL, "null handle");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    if (stamps_out && t->d_dbg) HIP_TRY(hipMemcpy(stamps_out, t->d_dbg, 16 * 8, hipMemcpyDeviceToHost));
    t->dbg_on = (enable & 1) != 0;
    t->stop_phase = (enable >> 4) & 7;
    t->force_tile = (enable >> 1) & 7;      // development: 1 = small tiles, 2 = large tiles, 3 = small tiles + multi-pass, 4 = queue mode
    return SPL_OK;
}

#ifdef SPL_MERGE_TIMING
int spl_debug_merge_timing(unsigned long long out[8], int reset) {
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(spl::g_mt), z, sizeof z); return 0; }
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(spl::g_mt), 64);
    return 0;
}
#endif

int spl_last_queue_counts(spl_tokenizer* t, uint32_t counts_out[4]) {
    if (!t || !t->d_zero) return fail(SPL_EINVAL, "no batch has run");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipDeviceSynchronize());
    if (!t->last_qcount) {       // single-pass call: no global queues exist
        for (int i = 0; i < 4; i++) counts_out[i] = 0;
        return SPL_OK;
    }
    uint32_t q[8];
    HIP_TRY(hipMemcpy(q, t->last_qcount, 32, hipMemcpyDeviceToHost));
    counts_out[0] = q[0]; counts_out[1] = q[1]; counts_out[3] = q[3];
    counts_out[2] = q[2] + q[4];             // the long-chunk queue is filled from both ends
    return SPL_OK;
}

}  // extern "C"
