// hostsim.cpp -- TEST TOOL (not part of the product, never shipped in the package).
//
// Compiles the product's host/device-shared headers (spl_scan.h, spl_lookup.h) and the host
// table builder with plain g++ and drives them serially, so that the scanner closed forms, the
// sync-point rules, the table formats and the lane-serial merge loop can be checked against the
// oracle on a machine without a GPU.  The kernels run the very same inline functions.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../splintr_amd/csrc/spl_lookup.h"
#include "../../splintr_amd/csrc/spl_scan.h"
#include "../../splintr_amd/csrc/spl_scan_masks.h"
#include "../../splintr_amd/csrc/spl_tables.h"

using namespace spl;

struct Sim {
    HostTables ht;
    DeviceTables dt;
};

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct WinAcc {   // class records of one window [0, n), sentinel at n; text readable to text_n
    const uint8_t* recs;   // (the kernels stage a few real bytes past the window end as well, so a
    const uint8_t* text;   //  character straddling the end still decodes)
    int n;
    uint32_t end_rec;   // record returned at/after n
    int text_n;
    uint32_t rec(int q) const { return q >= n ? end_rec : recs[q]; }
    uint32_t txt(int q) const { return q >= text_n ? 0u : text[q]; }
    uint32_t load32(int p) const {
        uint32_t w = 0;
        for (int i = 0; i < 4; i++) w |= txt(p + i) << (8 * i);
        return w;
    }
};

// class records for text[0,n) treated as ONE text that starts at 0: every byte through byte_record
// (spl_scan.h), exactly as the kernels classify -- in parallel, each byte on its own
static void classify(const Sim& s, const uint8_t* text, int n, std::vector<uint8_t>& recs) {
    recs.assign(n, 0);
    WinAcc tx{nullptr, text, n, 0, n};
    for (int q = 0; q < n; q++)
        recs[q] = (uint8_t)byte_record(s.dt, tx, [](int i) { return i == 0; }, [&](uint32_t c) { return cp_class(s.dt, c); }, q, 0, n);
    if (n) recs[0] |= CB_TSTART | CB_SYNC;
    uint32_t prev = C_EOT;
    for (int q = 0; q < n; q++) {
        if ((recs[q] & CB_CLASS) == C_CONT) continue;
        uint32_t cur = recs[q] & CB_CLASS;
        if (q && is_sync(s.ht.pattern, prev, cur)) recs[q] |= CB_SYNC;
        prev = cur;
    }
}

// the policy as a sequential definition (what oracle.c restates): for the cross-check of byte_record
static void classify_seq(const Sim& s, const uint8_t* text, int n, std::vector<uint8_t>& recs) {
    recs.assign(n, 0);
    WinAcc tx{nullptr, text, n, 0, n};
    for (int q = 0; q < n;) {
        const uint32_t b = text[q];
        uint32_t len = 1, cls;
        if (b < 0x80) cls = cp_class(s.dt, b);
        else if (b < 0xC0) cls = C_P;
        else {
            const uint32_t want = utf8_len(b);
            while (len < want && q + (int)len < n && (text[q + len] & 0xC0) == 0x80) len++;
            cls = len == want ? cp_class(s.dt, decode_at(tx, q, b)) : (uint32_t)C_P;
        }
        recs[q] = (uint8_t)(cls | ((len - 1) << CB_LEN_SHIFT));
        for (uint32_t i = 1; i < len; i++) recs[q + i] = C_CONT;
        q += len;
    }
}

extern "C" {

// 1 if the per-byte classification equals the sequential definition on this text
int hs_classify_check(void* p, const uint8_t* text, int n) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> a, b;
    classify(*s, text, n, a);
    classify_seq(*s, text, n, b);
    if (n) b[0] |= CB_TSTART | CB_SYNC;
    for (int q = 0; q < n; q++) if ((a[q] & ~CB_SYNC) != (b[q] & ~CB_SYNC)) return 0;
    return 1;
}

void* hs_create(const char* splv, const char* ucls, int pattern, char* errbuf, int errlen) {
    Sim* s = new Sim();
    auto v = slurp(splv), u = slurp(ucls);
    std::string err;
    if (build_tables(v.data(), v.size(), u.data(), u.size(), pattern, false, s->ht, err)) {
        snprintf(errbuf, errlen, "%s", err.c_str());
        delete s;
        return nullptr;
    }
    HostTables& h = s->ht;
    s->dt = DeviceTables{};
    DeviceTables& d = s->dt;
    d.ucls_stage1 = h.ucls_stage1.data(); d.ucls_stage2 = h.ucls_stage2.data(); d.ucls_shift = h.ucls_shift; d.cjk_fast = h.cjk_fast ? 1u : 0u;
    d.short_tab = h.short_tab.data(); d.short_mask = (uint32_t)(h.short_tab.size() / SPL_SHORT_BUCKET) - 1;
    d.tiny_tab = h.tiny_tab.data(); d.tiny_mask = (uint32_t)((h.tiny_tab.size() - 4) / SPL_TINY_WORDS) - 1;
    d.t8_tab = h.t8_tab.data(); d.t8_mask = (uint32_t)((h.t8_tab.size() - 4) / SPL_T8_WORDS) - 1;
    d.long_tab = h.long_tab.data(); d.long_mask = (uint32_t)h.long_tab.size() - 1; d.key_blob = h.key_blob.data();
    d.pair_tab = h.pair_tab.data(); d.pair_mask = (uint32_t)(h.pair_tab.size() / SPL_PAIR_BUCKET) - 1;
    d.byte_id = h.byte_id.data(); d.max_key_len = h.max_key_len; d.pattern = (uint32_t)h.pattern; d.all_bytes = h.all_bytes ? 1u : 0u; d.id_limit = h.id_limit;
    d.p8_tab = reinterpret_cast<const P8Bucket*>(h.p8_tab.data()); d.p8_mask = (uint32_t)(h.p8_tab.size() / 2) - 1;
    d.len_mask = h.len_mask.data(); d.tiny_free = h.tiny_free; d.t8_free = h.t8_free;
    d.ascii_base = (uint32_t)h.ucls_stage1[0] << h.ucls_shift;
    d.pfx = h.pfx.data(); d.filt4 = h.filt4.data(); d.filt4_shift = h.filt4_shift;
    return s;
}
void hs_destroy(void* p) { delete (Sim*)p; }
void hs_info(void* p, uint32_t* out) {
    Sim* s = (Sim*)p;
    out[0] = s->ht.n_keys; out[1] = s->ht.n_pairs; out[2] = s->ht.max_key_len; out[3] = s->ht.max_id;
    out[4] = (uint32_t)s->ht.short_tab.size(); out[5] = (uint32_t)s->ht.long_tab.size();
    out[6] = (uint32_t)s->ht.pair_tab.size(); out[7] = s->ht.cjk_fast;
}

// Sequential orbit of match_end from 0.  window>0: classify/scan through windows of that many
// bytes and restart a fresh window at every SPL_DEFER (exercises the deferral contract).
// Returns number of starts, or -1 on a stuck scan.
int hs_split(void* p, const uint8_t* text, int n, uint32_t* starts, int window) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> recs;
    classify(*s, text, n, recs);
    int k = 0, pos = 0;
    while (pos < n) {
        int wend = window > 0 ? std::min(n, pos + window) : n;
        WinAcc a{recs.data(), text, wend, wend == n ? (uint32_t)(C_EOT | CB_TSTART | CB_SYNC) : (uint32_t)C_WEND, n};
        int e = match_end(a, pos, s->ht.pattern);
        if (e == SPL_DEFER) {
            if (wend == n) return -1;
            // widen: retry with the rest of the text as window
            WinAcc b{recs.data(), text, n, (uint32_t)(C_EOT | CB_TSTART | CB_SYNC), n};
            e = match_end(b, pos, s->ht.pattern);
            if (e == SPL_DEFER) return -1;
        }
        if (e <= pos) return -2;
        starts[k++] = pos;
        pos = e;
    }
    return k;
}

// Sync-point ownership: every sync point runs its own chain to the next sync point; returns the
// union of marked starts (sorted, unique) -- must equal hs_split's answer.  Also reports the
// number of sync points found and the longest chain (in matches).
int hs_split_sync(void* p, const uint8_t* text, int n, uint32_t* starts, uint32_t* stats) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> recs;
    classify(*s, text, n, recs);
    std::vector<uint8_t> mark(n + 1, 0);
    WinAcc a{recs.data(), text, n, (uint32_t)(C_EOT | CB_TSTART | CB_SYNC), n};
    uint32_t nsync = 0, longest = 0;
    for (int sp = 0; sp < n; sp++) {
        if (!(recs[sp] & CB_SYNC)) continue;
        nsync++;
        int pos = sp;
        uint32_t len = 0;
        for (;;) {
            mark[pos] = 1;
            int e = match_end(a, pos, s->ht.pattern);
            if (e <= pos) return -1;
            pos = e;
            len++;
            if (pos >= n || (recs[pos] & CB_SYNC)) break;
        }
        longest = std::max(longest, len);
    }
    int k = 0;
    for (int q = 0; q < n; q++) if (mark[q]) starts[k++] = q;
    stats[0] = nsync; stats[1] = longest;
    return k;
}

struct VecStore {
    std::vector<uint32_t> i, r;
    uint32_t& id(int k) { return i[k]; }
    uint32_t& rk(int k) { return r[k]; }
};

// Full single-text encode through the shared code: split, whole-chunk probe, lane-serial merge.
int hs_encode(void* p, const uint8_t* text, int n, uint32_t* ids, uint32_t* n_probe_hits) {
    Sim* s = (Sim*)p;
    std::vector<uint32_t> starts(n + 1);
    int k = hs_split(p, text, n, starts.data(), 0);
    if (k < 0) return k;
    starts[k] = n;
    WinAcc tx{nullptr, text, n, 0, n};
    int out = 0;
    uint32_t hits = 0;
    VecStore st;
    for (int c = 0; c < k; c++) {
        int a = starts[c], len = starts[c + 1] - a;
        uint32_t id = probe_chunk(s->dt, tx, a, len);
        if (id != SPL_NO_RANK) { ids[out++] = id; hits++; continue; }
        if (len == 1) continue;
        st.i.assign(len, 0); st.r.assign(len, 0);
        bpe_serial(s->dt, st, tx, a, len);
        for (int i = 0; i < len; i++)
            if (st.i[i] != SPL_DEAD && st.i[i] < s->dt.id_limit) ids[out++] = st.i[i];     // (a pseudo id: a single byte the vocabulary lacks, dropped)
    }
    if (n_probe_hits) *n_probe_hits = hits;
    return out;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Mask-based scanner (spl_scan_masks.h) driven the way k_pretok drives it: tiles of `tb` bytes
// with a 32-byte left halo and `rh` bytes of right halo; masks are built per window from the class
// records; every sync point inside the tile runs its chain; chains that return SPL_DEFER or run
// off the window are finished with the byte-wise scanner over the whole text (what k_deferred
// does).  Returns the union of marked starts.
struct MaskWin {
    const uint8_t* recs;      // records of the window, index 0 = window start
    const uint8_t* text;
    std::vector<uint32_t> mk[MK_COUNT];
    int W;
    bool eot;
    int text_n_rel;           // readable text bytes relative to the window start
    uint32_t mw(int which, int w) const { return (w >= 0 && w < (int)mk[which].size()) ? mk[which][w] : 0u; }
    uint32_t rec(int q) const { return recs[q]; }
    uint32_t txt(int q) const { return q < text_n_rel ? text[q] : 0u; }
    int wbits() const { return W; }
    bool end_is_eot() const { return eot; }
};

extern "C" int hs_split_masks(void* p, const uint8_t* text, int n, uint32_t* starts, int tb, int rh) {
    Sim* s = (Sim*)p;
    std::vector<uint8_t> recs;
    classify(*s, text, n, recs);
    recs.resize(n + 64, 0);
    std::vector<uint8_t> mark(n + 1, 0);
    const int LHv = 32;
    for (int t0 = 0; t0 < n; t0 += tb) {
        const int w0 = t0 - LHv;
        const int W = LHv + tb + rh;
        const int iB = (n - w0 < W) ? n - w0 : W;
        // window-local records (positions before the text read as CONT, sentinel at iB)
        std::vector<uint8_t> wrec(W + 64, (uint8_t)C_WEND);
        std::vector<uint8_t> wtxt(W + 64, 0);
        for (int i = 0; i < W + 32; i++) {
            const int g = w0 + i;
            if (g >= 0 && g < n) wtxt[i] = text[g];
            if (i >= iB) continue;
            wrec[i] = g < 0 ? (uint8_t)C_CONT : (uint8_t)(recs[g] & ~CB_SYNC);
        }
        if (iB < W) wrec[iB] = (uint8_t)(C_EOT | CB_TSTART);
        MaskWin m;
        m.recs = wrec.data(); m.text = wtxt.data(); m.W = W; m.eot = (n - w0 <= W); m.text_n_rel = W + 32;
        const int nw = W / 32 + 1;
        for (int k = 0; k < MK_COUNT; k++) m.mk[k].assign(nw, 0);
        for (int i = 0; i <= W && i < (int)wrec.size(); i++) {
            if (i > iB || (i == W && iB == W)) break;
            const uint32_t r = wrec[i];
            uint32_t cls = r & CB_CLASS;
            if (r & CB_TSTART) m.mk[MK_TS][i >> 5] |= 1u << (i & 31);
            if (i == iB) break;
            uint32_t kc = cls;
            if (cls == C_CONT) {                       // inherit the kind of the lead byte
                int j = i - 1;
                while (j >= 0 && (wrec[j] & CB_CLASS) == C_CONT && j > i - 3) j--;
                kc = j >= 0 ? (wrec[j] & CB_CLASS) : (uint32_t)C_CONT;
            } else if (cls < C_EOT) {
                m.mk[MK_CS][i >> 5] |= 1u << (i & 31);
            }
            if (kc < C_EOT) {
                const uint32_t kb = kind_bits(kc);
                for (int k = 0; k < MK_CS; k++) if ((kb >> k) & 1u) m.mk[k][i >> 5] |= 1u << (i & 31);
            }
        }
        for (int w = 0; w < nw; w++) {
            uint32_t kw[MK_COUNT], kp[MK_COUNT];
            for (int k = 0; k < MK_COUNT; k++) { kw[k] = m.mk[k][w]; kp[k] = w ? m.mk[k][w - 1] : 0u; }
            m.mk[MK_SY][w] = sync_word(s->ht.pattern, kw, kp);
        }
        // chains from the sync points inside the tile
        for (int i = LHv; i < LHv + tb && i < iB; i++) {
            if (!bit_m(m, MK_SY, i)) continue;
            int pos = i;
            for (;;) {
                mark[w0 + pos] = 1;
                int e = match_end_m(m, pos, s->ht.pattern);
                bool deferred = (e == SPL_DEFER);
                if (!deferred && e >= W && w0 + e < n) { pos = e; deferred = true; mark[w0 + pos] = 1; }
                if (deferred) {                          // finish the segment over the whole text (k_deferred)
                    WinAcc g{recs.data(), text, n, (uint32_t)(C_EOT | CB_TSTART | CB_SYNC), n};
                    int gp = w0 + pos;
                    for (;;) {
                        mark[gp] = 1;
                        int ge = match_end(g, gp, s->ht.pattern);
                        if (ge <= gp) return -1;
                        gp = ge;
                        if (gp >= n || (recs[gp] & (CB_SYNC | CB_TSTART))) break;
                    }
                    break;
                }
                if (e <= pos) return -2;
                pos = e;
                if (pos >= iB || bit_m(m, MK_SY, pos) || bit_m(m, MK_TS, pos)) break;
            }
        }
    }
    int k = 0;
    for (int q = 0; q < n; q++) if (mark[q]) starts[k++] = q;
    return k;
}

// ---------------------------------------------------------------------------------------------
// Bit-vector start computation (spl_scan_starts.h) driven the way k_pretok drives it: several
// documents packed back to back, tiles of `tb` bytes with a 32-byte left halo and `rh` bytes of right
// halo.  A tile whose window qualifies (no MK_BAD byte from its first sync point to its end
// sync point, end sync point inside the window, loops converged) takes its starts from the masks;
// any other tile runs the chains over the whole text.  stats: [0] tiles, [1] tiles on the fast path.
#include "../../splintr_amd/csrc/spl_scan_starts.h"
#include "../../splintr_amd/csrc/spl_scan_words.h"

struct HostBV {
    std::vector<uint32_t> w;
    HostBV() {}
    explicit HostBV(size_t n) : w(n, 0u) {}
    HostBV operator&(const HostBV& o) const { HostBV r(w.size()); for (size_t i = 0; i < w.size(); i++) r.w[i] = w[i] & o.w[i]; return r; }
    HostBV operator|(const HostBV& o) const { HostBV r(w.size()); for (size_t i = 0; i < w.size(); i++) r.w[i] = w[i] | o.w[i]; return r; }
    HostBV operator~() const { HostBV r(w.size()); for (size_t i = 0; i < w.size(); i++) r.w[i] = ~w[i]; return r; }
    HostBV shl1() const { HostBV r(w.size()); for (size_t i = 0; i < w.size(); i++) r.w[i] = (w[i] << 1) | (i ? w[i - 1] >> 31 : 0u); return r; }
    HostBV shr1() const { HostBV r(w.size()); for (size_t i = 0; i < w.size(); i++) r.w[i] = (w[i] >> 1) | (i + 1 < w.size() ? w[i + 1] << 31 : 0u); return r; }
    bool any() const { for (uint32_t x : w) if (x) return true; return false; }
    bool bit(int i) const { return (w[i >> 5] >> (i & 31)) & 1u; }
};

extern "C" int hs_split_starts(void* p, const uint8_t* text, int n, const int* doc_off, int n_docs, uint32_t* starts,
                               int tb, int rh, int max_iter, uint32_t* stats) {
    Sim* s = (Sim*)p;
    const int pat = s->ht.pattern;
    std::vector<uint8_t> recs(n + 64, 0);
    for (int d = 0; d < n_docs; d++) {
        std::vector<uint8_t> r;
        classify(*s, text + doc_off[d], doc_off[d + 1] - doc_off[d], r);
        std::copy(r.begin(), r.end(), recs.begin() + doc_off[d]);
    }
    std::vector<uint8_t> mark(n + 1, 0);
    const int LHv = 32;
    uint32_t n_tiles = 0, n_fast = 0, n_nofe = 0, n_bad = 0, n_iter = 0, n_lean = 0, n_general = 0;
    for (int t0 = 0; t0 < n; t0 += tb) {
        n_tiles++;
        const int w0 = t0 - LHv;
        const int W = LHv + tb + rh;
        const int iB = (n - w0 < W) ? n - w0 : W;
        std::vector<uint8_t> wrec(W + 64, (uint8_t)C_WEND);
        std::vector<uint8_t> wtxt(W + 64, 0);
        for (int i = 0; i < W + 32; i++) {
            const int g = w0 + i;
            if (g >= 0 && g < n) wtxt[i] = text[g];
            if (i >= iB) continue;
            wrec[i] = g < 0 ? (uint8_t)C_CONT : (uint8_t)(recs[g] & ~CB_SYNC);
        }
        if (iB < W) wrec[iB] = (uint8_t)(C_EOT | CB_TSTART);
        MaskWin m;
        m.recs = wrec.data(); m.text = wtxt.data(); m.W = W; m.eot = (n - w0 <= W); m.text_n_rel = W + 32;
        const int nw = W / 32 + 1;
        for (int k = 0; k < MK_COUNT; k++) m.mk[k].assign(nw, 0);
        for (int i = 0; i <= iB && i <= W; i++) {
            if (i == W && iB == W) break;
            const uint32_t r = wrec[i];
            const uint32_t cls = r & CB_CLASS;
            if (r & CB_TSTART) m.mk[MK_TS][i >> 5] |= 1u << (i & 31);
            if (i == iB) break;
            uint32_t kc = cls;
            if (cls == C_CONT) {
                int j = i - 1;
                while (j >= 0 && (wrec[j] & CB_CLASS) == C_CONT && j > i - 3) j--;
                kc = j >= 0 ? (wrec[j] & CB_CLASS) : (uint32_t)C_CONT;
            } else if (cls < C_EOT) {
                m.mk[MK_CS][i >> 5] |= 1u << (i & 31);
            }
            uint32_t kb = kc < C_EOT ? kind_bits(kc) : 0u;
            if (bad_for_starts(pat, r, kc, true)) kb |= 1u << MK_BAD;
            if (wtxt[i] == '/') kb |= 1u << MK_SL;
            for (int k = 0; k < MK_CS; k++) if ((kb >> k) & 1u) m.mk[k][i >> 5] |= 1u << (i & 31);
        }
        for (int w = 0; w < nw; w++) {
            uint32_t kw[MK_COUNT], kp[MK_COUNT];
            for (int k = 0; k < MK_COUNT; k++) { kw[k] = m.mk[k][w]; kp[k] = w ? m.mk[k][w - 1] : 0u; }
            m.mk[MK_SY][w] = sync_word(s->ht.pattern, kw, kp);
        }
        // ---- the one-pass word classifier of the kernel (spl_scan_words.h) must give these very masks and records ----
        {
            KindEnt aent[128], kent[16];
            for (uint32_t c = 0; c < 128; c++) aent[c] = ascii_entry(pat, c, cp_class(s->dt, c));
            for (uint32_t c = 0; c < 16; c++) kent[c] = kind_entry(c);
            std::vector<uint32_t> nm[MK_COUNT];
            for (int k = 0; k < MK_COUNT; k++) nm[k].assign(nw, 0);
            const int lo = w0 < 0 ? -w0 : 0;
            const int iT = (n - w0 < W + 16) ? n - w0 : W + 16;
            struct RW { const uint8_t* t; uint32_t txt(int j) const { return j < 0 ? 0u : t[j]; } };
            const RW rw{wtxt.data()};
            // (the kernel's s_ts: the document starts of the window and of the word behind it, from doc_off)
            auto tsbit = [&](int j) { const int g = w0 + j; return g >= 0 && g < n && (recs[g] & CB_TSTART) != 0; };
            for (int wi = 0; wi < W / 4; wi++) {
                const int i0 = wi * 4;
                uint32_t tw = 0, ts4 = 0;
                for (int k = 0; k < 4; k++) {
                    tw |= (uint32_t)wtxt[i0 + k] << (8 * k);
                    // (the kernel's s_ts holds document starts only; the end-of-text mark at iB is the classifier's own)
                    if (i0 + k < iB && tsbit(i0 + k)) ts4 |= 1u << k;
                }
                WordKinds wk;
                if (!(tw & 0x80808080u) && i0 + 3 < iB && w0 + i0 >= 0) {
                    const KindEnt e[4] = {aent[tw & 0xFF], aent[(tw >> 8) & 0xFF], aent[(tw >> 16) & 0xFF], aent[tw >> 24]};
                    wk = classify_word_ascii(e, ts4);
                } else {
                    auto word = [&](int wj) { uint32_t x = 0; if (wj >= 0) for (int k = 0; k < 4; k++) x |= (uint32_t)wtxt[wj * 4 + k] << (8 * k); return x; };
                    uint32_t ts16 = 0;
                    for (int d2 = 0; d2 < 16; d2++) if (i0 - 4 + d2 >= 0 && tsbit(i0 - 4 + d2)) ts16 |= 1u << d2;
                    bool done = false;
                    if (i0 + 3 < iB && i0 >= lo)
                        done = classify_word_text(s->dt, pat, word(wi - 1), tw, word(wi + 1), ts16, [&](uint32_t c) { return aent[c]; },
                                                  [&](uint32_t c) { return kent[c]; }, i0, lo, iT, wk);
                    n_lean += done; n_general += (!done && (tw & 0x80808080u)) ? 1 : 0;
                    if (!done)
                        wk = classify_word(s->dt, pat, word(wi - 1), tw, word(wi + 1), ts16, [&](uint32_t c) { return kent[c]; },
                                           [&](uint32_t c) { return cp_class(s->dt, c); }, ts4, 0u, i0, iB, W, lo, iT);
                }
                for (int k = 0; k < 4; k++) {
                    const int i = i0 + k;
                    for (int j = 0; j < 8; j++) if ((wk.v0 >> (4 * j + k)) & 1u) nm[(V0_KINDS >> (4 * j)) & 15][i >> 5] |= 1u << (i & 31);
                    for (int j = 0; j < V1_NKINDS; j++) if ((wk.v1 >> (4 * j + k)) & 1u) nm[(V1_KINDS >> (4 * j)) & 15][i >> 5] |= 1u << (i & 31);
                    // records: the window's first bytes may begin inside a character whose lead the window does not hold
                    const uint32_t want = (i == iB && iB < W) ? (uint32_t)(C_EOT | CB_TSTART | CB_SYNC) : i > iB ? (uint32_t)C_WEND
                                        : (uint32_t)wrec[i] | ((wrec[i] & CB_TSTART) ? (uint32_t)CB_SYNC : 0u);
                    const uint32_t got = (wk.rec >> (8 * k)) & 0xFFu;
                    if ((i >= 4 || w0 <= 0) && got != want) { if (getenv("HS_DEBUG")) fprintf(stderr, "rec mismatch w0=%d i=%d iB=%d W=%d got=%02x want=%02x tw=%08x ts4=%x lo=%d\n", w0, i, iB, W, got, want, tw, ts4, lo); return -7; }
                }
            }
            for (int k = 0; k < MK_SY; k++) {
                if (pat == PAT_CL100K && (k == MK_M || k == MK_UP || k == MK_LB)) continue;     // (never asked for)
                if (pat != PAT_MISTRAL_V3 && k == MK_SL) continue;
                for (int w = 0; w < nw; w++) {
                    uint32_t a = m.mk[k][w], b2 = nm[k][w];
                    if (w == 0 && w0 > 0) { a &= ~0xFu; b2 &= ~0xFu; }
                    if (w * 32 > iB) { a = 0; b2 = 0; }
                    if (a != b2) return -8 - k;
                }
            }
        }
        // ---- fast path ----
        bool fast = false;
        {
            int fs = -1, fe = -1;
            for (int i = LHv; i < LHv + tb && i < iB; i++) if (bit_m(m, MK_SY, i)) { fs = i; break; }
            for (int i = LHv + tb; i <= iB && i <= W; i++) if (bit_m(m, MK_SY, i) || bit_m(m, MK_TS, i)) { fe = i; break; }
            if (LHv + tb >= iB) fe = iB;                   // the text ends inside the tile
            if (fs < 0) fast = true;                       // nothing owned
            else if (fe >= 0) {
                bool bad = false;
                for (int i = fs; i < fe; i++) if (bit_m(m, MK_BAD, i)) bad = true;
                if (bad) n_bad++;
                auto mkbv = [&](int k) { HostBV b(nw); b.w = m.mk[k]; return b; };
                if (!bad && pat != PAT_CL100K) {
                    O200kStartMasks<HostBV> om{mkbv(MK_L), mkbv(MK_UP), mkbv(MK_LB), mkbv(MK_N), mkbv(MK_S), mkbv(MK_NL), mkbv(MK_O), mkbv(MK_AP),
                                               mkbv(MK_SP), mkbv(MK_SL), mkbv(MK_CS), mkbv(MK_TS)};
                    bool ok1, ok2, ok3;
                    HostBV CAND;
                    HostBV Bv = o200k_starts_ln(om, pat == PAT_MISTRAL_V3, ok1, max_iter) | o200k_starts_o(om, pat == PAT_MISTRAL_V3, CAND, ok2, max_iter) |
                                o200k_starts_s(om, pat == PAT_MISTRAL_V3, ok3, max_iter);
                    bool ok = ok1 && ok2 && ok3;
                    std::vector<uint8_t> kill(W + 2, 0), add(W + 2, 0);
                    if (ok && pat == PAT_O200K)
                        for (int i = fs; i < fe && ok; i++) {
                            if (!CAND.bit(i)) continue;
                            int e;
                            if (!o200k_contraction_at(m, i, e)) { ok = false; break; }
                            if (e == SPL_DEFER) return -3;
                            if (e > 0) { for (int q = i; q < e; q++) kill[q] = 1; add[e] = 1; if (e > fe) return -4; }
                        }
                    if (ok) {
                        for (int i = fs; i < fe; i++) if ((Bv.bit(i) && !kill[i]) || add[i]) mark[w0 + i] = 1;
                        fast = true;
                    } else n_iter++;
                } else if (!bad) {
                    Cl100kStartMasks<HostBV> cm;
                    cm.L = mkbv(MK_L); cm.N = mkbv(MK_N); cm.S = mkbv(MK_S); cm.NL = mkbv(MK_NL); cm.O = mkbv(MK_O);
                    cm.AP = mkbv(MK_AP); cm.SP = mkbv(MK_SP); cm.CS = mkbv(MK_CS); cm.TS = mkbv(MK_TS);
                    HostBV CA; bool ok;
                    HostBV Bv = cl100k_starts(cm, CA, ok, max_iter);
                    if (ok) {
                        for (int i = fs; i < fe; i++) {
                            if (Bv.bit(i)) mark[w0 + i] = 1;
                            if (CA.bit(i)) {
                                const int e = contraction(m, i);
                                if (e == SPL_DEFER) return -3;
                                if (e > 0 && e < fe) mark[w0 + e] = 1;
                                if (e > fe) return -4;
                            }
                        }
                        fast = true;
                    } else n_iter++;
                }
            } else n_nofe++;
        }
        if (fast) { n_fast++; continue; }
        // ---- chains over the whole text (the kernel's chain phase + deferral) ----
        for (int i = LHv; i < LHv + tb && i < iB; i++) {
            if (!bit_m(m, MK_SY, i)) continue;
            WinAcc g{recs.data(), text, n, (uint32_t)(C_EOT | CB_TSTART | CB_SYNC), n};
            int gp = w0 + i;
            // end of this document
            int dend = n;
            for (int d = 0; d < n_docs; d++) if (doc_off[d] <= gp && gp < doc_off[d + 1]) dend = doc_off[d + 1];
            WinAcc gd{recs.data(), text, dend, (uint32_t)(C_EOT | CB_TSTART | CB_SYNC), dend};
            for (;;) {
                mark[gp] = 1;
                int ge = match_end(gd, gp, s->ht.pattern);
                if (ge <= gp) return -1;
                gp = ge;
                if (gp >= dend || (recs[gp] & (CB_SYNC | CB_TSTART))) break;
            }
        }
    }
    if (stats) { stats[0] = n_tiles; stats[1] = n_fast; stats[2] = n_nofe; stats[3] = n_bad; stats[4] = n_iter; stats[5] = n_lean; stats[6] = n_general; }
    int k = 0;
    for (int q = 0; q < n; q++) if (mark[q]) starts[k++] = q;
    return k;
}

// table sizes, and short-table buckets from which a key went on to the next one (a probe that misses there walks on).
// tiny / t8: slots, keys found by the single-slot probes (every key of the vocabulary must be: checked by the caller
// through hs_row_head_check and the encode tests), empty slots.
extern "C" void hs_bucket_stats(void* p, uint32_t* out) {
    Sim* s = (Sim*)p;
    const HostTables& h = s->ht;
    uint32_t ns = (uint32_t)((h.tiny_tab.size() - 4) / SPL_TINY_WORDS), used = 0;
    for (uint32_t i = 0; i < ns; i++) used += h.tiny_tab[(size_t)i * SPL_TINY_WORDS + 1] != SPL_EMPTY;
    out[0] = ns; out[1] = used;
    ns = (uint32_t)((h.t8_tab.size() - 4) / SPL_T8_WORDS); used = 0;
    for (uint32_t i = 0; i < ns; i++) used += h.t8_tab[(size_t)i * SPL_T8_WORDS + 2] != SPL_EMPTY;
    out[2] = ns; out[3] = used;
    uint32_t nb = (uint32_t)(h.short_tab.size() / SPL_SHORT_BUCKET), full = 0;
    const uint32_t* st = reinterpret_cast<const uint32_t*>(h.short_tab.data());
    const size_t wpb = sizeof(h.short_tab[0]) * SPL_SHORT_BUCKET / 4;
    for (uint32_t b = 0; b < nb; b++) if (bucket_overflowed(st[(size_t)b * wpb + wpb - 1])) full++;
    out[4] = nb; out[5] = full;
    out[6] = h.unsalted_groups;
}

// The tables a substring-tabulation row starts from (DeviceTables::pfx, ::filt4), checked against the key tables
// themselves: every key of the tiny / t8 / short / long tables must pass the four-byte-prefix filter at its length
// (no false negatives -- a cleared bit means "no probe"), every two-byte key must be its prefix entry's id2, and no
// prefix entry may name an id2 that is not a two-byte key.  out: [0] keys checked, [1] violations,
// [2] filter slots, [3] non-zero slots.
extern "C" void hs_row_head_check(void* p, uint32_t* out) {
    Sim* s = (Sim*)p;
    const HostTables& h = s->ht;
    uint32_t keys = 0, bad = 0, two = 0;
    auto check = [&](uint32_t k0, uint32_t n, uint32_t id) {
        keys++;
        const PfxEnt pe = h.pfx[k0 & 0xFFFFu];
        if ((pe.lm & 0xFFFFu) != h.len_mask[k0 & 0xFFFFu]) bad++;
        if (n >= 2 && !((pe.lm >> (n <= (uint32_t)SPL_T8_MAX ? n - 2 : 7)) & 1u)) bad++;
        if (n == 2) { two++; if (pe.id2 != id) bad++; }
        if (n >= 4) {
            const uint32_t f = h.filt4[hash_f4(k0) >> h.filt4_shift];
            if (!((f >> (n <= (uint32_t)SPL_T8_MAX ? n - 4 : 5)) & 1u)) bad++;
        }
    };
    // one entry per slot: every stored key must come back from the single-slot probes (its salt from the prefix entry /
    // the filter entry), i.e. sit exactly where its hash says
    for (size_t e = 0; e + 4 < h.tiny_tab.size(); e += SPL_TINY_WORDS)
        if (h.tiny_tab[e + 1] != SPL_EMPTY) {
            const uint32_t k0 = h.tiny_tab[e], n = (h.tiny_tab[e + 1] >> 24) & 0x7Fu, id = h.tiny_tab[e + 1] & SPL_ID_MASK;
            check(k0, n, id);
            if (probe_tiny(s->dt, k0, n, tiny_salt(s->dt, k0)) != id) bad++;
        }
    for (size_t e = 0; e + 4 < h.t8_tab.size(); e += SPL_T8_WORDS)
        if (h.t8_tab[e + 2] != SPL_EMPTY) {
            const uint32_t k0 = h.t8_tab[e], k1 = h.t8_tab[e + 1], n = (h.t8_tab[e + 2] >> 24) & 0x7Fu, id = h.t8_tab[e + 2] & SPL_ID_MASK;
            check(k0, n, id);
            if (probe_t8(s->dt, k0, k1, n, t8_salt(s->dt, k0)) != id) bad++;
        }
    for (const ShortEnt& e : h.short_tab) if (e.id_len != SPL_EMPTY) check(e.k0, (e.id_len >> 24) & 0x7Fu, e.id_len & SPL_ID_MASK);
    for (const LongEnt& e : h.long_tab)
        if (e.id != SPL_EMPTY) { uint32_t k0; memcpy(&k0, h.key_blob.data() + e.off, 4); check(k0, e.len, e.id); }
    uint32_t named = 0, nz = 0;
    for (const PfxEnt& pe : h.pfx) { named += pe.id2 != SPL_NO_RANK; if ((pe.id2 != SPL_NO_RANK) != ((pe.lm & 1u) != 0u)) bad++; }
    if (named != two) bad++;
    for (uint16_t f : h.filt4) nz += (f & 0x3Fu) != 0;
    out[0] = keys; out[1] = bad; out[2] = (uint32_t)h.filt4.size(); out[3] = nz;
}

// ---------------------------------------------------------------------------------------------
// The host splitter for custom patterns (spl_regex.h), driven directly: compile, split one text.
#include "../../splintr_amd/csrc/spl_regex.h"
extern "C" void* hs_regex_compile(void* p, const char* pattern, int plen, char* err, int errcap) {
    Sim* s = (Sim*)p;
    std::string e;
    spl::RegexPtr r = spl::regex_compile(std::string(pattern, (size_t)plen), s->ht, e);
    if (!r) { snprintf(err, (size_t)errcap, "%s", e.c_str()); return nullptr; }
    return r.release();
}
extern "C" void hs_regex_free(void* r) { spl::RegexDeleter()((spl::RegexProg*)r); }
// spans_out: up to cap (start, end) pairs; returns the count, -1 if the step budget ran out
extern "C" int hs_regex_split(void* r, const uint8_t* text, int n, uint32_t* spans_out, int cap) {
    std::vector<std::pair<uint32_t, uint32_t>> v;
    if (!spl::regex_split_spans(*(spl::RegexProg*)r, text, (size_t)n, v)) return -1;
    for (int k = 0; k < (int)v.size() && k < cap; k++) { spans_out[2 * k] = v[k].first; spans_out[2 * k + 1] = v[k].second; }
    return (int)v.size();
}
extern "C" int hs_regex_split_bits(void* r, const uint8_t* text, int n, uint64_t base, uint32_t* starts, uint32_t* gaps) {
    return spl::regex_split_bits(*(spl::RegexProg*)r, text, (size_t)n, base, starts, gaps) ? 0 : -1;
}
