// hostsim.cpp -- TEST TOOL (not part of the product, never shipped in the package).
//
// Compiles the product's host/device-shared headers (spl_scan.h, spl_lookup.h) and the host
// table builder with plain g++ and drives them serially, so that the scanner closed forms, the
// sync-point rules, the table formats and the lane-serial merge loop can be checked against the
// oracle on a machine without a GPU.  The kernels run the very same inline functions.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../splintr_amd/csrc/spl_lookup.h"
#include "../../splintr_amd/csrc/spl_scan.h"
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

// class records for text[0,n) treated as ONE text that starts at 0
static void classify(const Sim& s, const uint8_t* text, int n, std::vector<uint8_t>& recs) {
    recs.assign(n, 0);
    WinAcc tx{nullptr, text, n, 0, n};
    for (int q = 0; q < n;) {
        uint32_t b = text[q];
        uint32_t len = utf8_len(b);
        if (b >= 0x80 && b < 0xC0) len = 1;
        if (q + (int)len > n) len = 1;
        uint32_t cls = b < 0x80 ? cp_class(s.dt, b) : (len == 1 ? (uint32_t)C_P : cp_class(s.dt, decode_at(tx, q, b)));
        recs[q] = (uint8_t)(cls | ((len - 1) << CB_LEN_SHIFT));
        for (uint32_t i = 1; i < len; i++) recs[q + i] = C_CONT;
        q += len;
    }
    if (n) recs[0] |= CB_TSTART | CB_SYNC;
    uint32_t prev = C_EOT;
    for (int q = 0; q < n; q++) {
        if ((recs[q] & CB_CLASS) == C_CONT) continue;
        uint32_t cur = recs[q] & CB_CLASS;
        if (q && is_sync(s.ht.pattern, prev, cur)) recs[q] |= CB_SYNC;
        prev = cur;
    }
}

extern "C" {

void* hs_create(const char* splv, const char* ucls, int pattern, char* errbuf, int errlen) {
    Sim* s = new Sim();
    auto v = slurp(splv), u = slurp(ucls);
    std::string err;
    if (build_tables(v.data(), v.size(), u.data(), u.size(), pattern, s->ht, err)) {
        snprintf(errbuf, errlen, "%s", err.c_str());
        delete s;
        return nullptr;
    }
    HostTables& h = s->ht;
    s->dt = DeviceTables{h.ucls_stage1.data(), h.ucls_stage2.data(), h.ucls_shift, h.cjk_fast ? 1u : 0u,
                         h.short_tab.data(), (uint32_t)(h.short_tab.size() / SPL_SHORT_BUCKET) - 1, h.long_tab.data(),
                         (uint32_t)h.long_tab.size() - 1, h.key_blob.data(), h.pair_tab.data(),
                         (uint32_t)(h.pair_tab.size() / SPL_PAIR_BUCKET) - 1, h.byte_id.data(), h.max_key_len, (uint32_t)h.pattern,
                         h.all_bytes ? 1u : 0u};
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
            if (st.i[i] != SPL_DEAD && st.i[i] != SPL_NO_RANK) ids[out++] = st.i[i];
    }
    if (n_probe_hits) *n_probe_hits = hits;
    return out;
}

}  // extern "C"
