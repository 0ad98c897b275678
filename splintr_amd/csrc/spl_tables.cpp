// spl_tables.cpp -- see spl_tables.h.
#include "spl_tables.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>

namespace spl {

namespace {

uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }

uint32_t pow2_at_least(size_t n) {
    uint32_t c = 16;
    while (c < n) c <<= 1;
    return c;
}

// ByteLevel alphabet (reference src/core/byte_level.rs:46-74): byte -> code point.
void byte_level_alphabet(uint32_t cp_of_byte[256]) {
    bool direct[256] = {false};
    for (int b = 33; b <= 126; b++) direct[b] = true;
    for (int b = 161; b <= 172; b++) direct[b] = true;
    for (int b = 174; b <= 255; b++) direct[b] = true;
    uint32_t next = 256;
    for (int b = 0; b < 256; b++) cp_of_byte[b] = direct[b] ? (uint32_t)b : next++;
}

// Decode a ByteLevel key (UTF-8 of alphabet chars) into raw bytes; false if any char is foreign.
bool byte_level_to_raw(const std::string& key, const std::unordered_map<uint32_t, uint8_t>& byte_of_cp,
                       std::string& raw) {
    raw.clear();
    size_t i = 0;
    while (i < key.size()) {
        const uint8_t b = (uint8_t)key[i];
        uint32_t cp;
        size_t l;
        if (b < 0x80) { cp = b; l = 1; }
        else if (b >= 0xC2 && b < 0xE0 && i + 1 < key.size() && ((uint8_t)key[i + 1] & 0xC0u) == 0x80u) {
            cp = ((b & 0x1Fu) << 6) | ((uint8_t)key[i + 1] & 0x3Fu); l = 2;
        }
        else return false;                              // alphabet is U+0021..U+0143: at most 2 bytes
        auto it = byte_of_cp.find(cp);
        if (it == byte_of_cp.end()) return false;
        raw.push_back((char)it->second);
        i += l;
    }
    return true;
}

bool valid_utf8(const std::string& s) {
    size_t i = 0;
    while (i < s.size()) {
        const uint8_t b = (uint8_t)s[i];
        size_t l = b < 0x80 ? 1 : (b >= 0xC2 && b < 0xE0) ? 2 : (b >= 0xE0 && b < 0xF0) ? 3 : (b >= 0xF0 && b < 0xF5) ? 4 : 0;
        if (l == 0 || i + l > s.size()) return false;
        for (size_t k = 1; k < l; k++) if (((uint8_t)s[i + k] & 0xC0u) != 0x80u) return false;
        if (l == 3 && b == 0xE0 && (uint8_t)s[i + 1] < 0xA0) return false;
        if (l == 3 && b == 0xED && (uint8_t)s[i + 1] >= 0xA0) return false;
        if (l == 4 && b == 0xF0 && (uint8_t)s[i + 1] < 0x90) return false;
        if (l == 4 && b == 0xF4 && (uint8_t)s[i + 1] >= 0x90) return false;
        i += l;
    }
    return true;
}

uint32_t load_le(const std::string& s, size_t off) {      // up to 4 bytes, zero padded
    uint32_t w = 0;
    for (size_t i = 0; i < 4 && off + i < s.size(); i++) w |= (uint32_t)(uint8_t)s[off + i] << (8 * i);
    return w;
}

}  // namespace

uint32_t host_cp_class(const HostTables& t, uint32_t cp) {
    if (cp >= 0x110000u) return C_P;
    const uint32_t blk = t.ucls_stage1[cp >> t.ucls_shift];
    return t.ucls_stage2[(blk << t.ucls_shift) | (cp & ((1u << t.ucls_shift) - 1u))];
}

uint32_t host_cp_category(const HostTables& t, uint32_t cp) {
    if (cp >= 0x110000u || t.gc_stage1.empty()) return 0;
    const uint32_t blk = t.gc_stage1[cp >> t.ucls_shift];
    return t.gc_stage2[(blk << t.ucls_shift) | (cp & ((1u << t.ucls_shift) - 1u))];
}

int build_tables(const uint8_t* splv, size_t splv_len, const uint8_t* ucls, size_t ucls_len, int pattern,
                 bool force_byte_level, HostTables& out, std::string& err) {
    // ---- class table -------------------------------------------------------------------
    if (ucls_len < 32 || memcmp(ucls, "SPLU", 4) != 0 || rd32(ucls + 4) < 1 || rd32(ucls + 4) > 3) { err = "bad unicode class table"; return 1; }
    const uint32_t ucls_version = rd32(ucls + 4);
    out.ucls_shift = rd32(ucls + 8);
    const uint32_t nblocks = rd32(ucls + 12);
    const size_t n1 = 0x110000u >> out.ucls_shift, n2 = (size_t)nblocks << out.ucls_shift;
    if (ucls_len < 32 + n1 * 2 + n2) { err = "truncated unicode class table"; return 1; }
    out.ucls_stage1.resize(n1);
    memcpy(out.ucls_stage1.data(), ucls + 32, n1 * 2);
    out.ucls_stage2.assign(ucls + 32 + n1 * 2, ucls + 32 + n1 * 2 + n2);
    for (uint16_t b : out.ucls_stage1)
        if (b >= nblocks) { err = "unicode class table: block index out of range"; return 1; }
    // version 2: behind the caseless partners, the GENERAL CATEGORY of every code point (tools/gen_unicode_tables.py) --
    // host only: the splitter's \p{P} \p{S} \p{Z} \p{Nd} ... \d (spl_regex.cpp); the kernels never see it
    out.gc_stage1.clear(); out.gc_stage2.clear();
    if (ucls_version >= 2) {
        size_t at = 32 + n1 * 2 + n2;
        if (ucls_len < at + 4) { err = "truncated unicode class table"; return 1; }
        at += 4 + (size_t)rd32(ucls + at) * 8;
        if (ucls_len < at + 4) { err = "truncated unicode class table"; return 1; }
        const uint32_t gnb = rd32(ucls + at);
        at += 4;
        const size_t g2 = (size_t)gnb << out.ucls_shift;
        if (ucls_len < at + n1 * 2 + g2) { err = "truncated unicode class table (general categories)"; return 1; }
        out.gc_stage1.resize(n1);
        memcpy(out.gc_stage1.data(), ucls + at, n1 * 2);
        out.gc_stage2.assign(ucls + at + n1 * 2, ucls + at + n1 * 2 + g2);
        for (uint16_t b : out.gc_stage1)
            if (b >= gnb) { err = "unicode class table: category block index out of range"; return 1; }
        // version 3: behind the categories, the SCRIPT property as ranges per script name (\p{Han}, \p{Hiragana}, \p{Latin} ... in a
        // custom split pattern: spl_regex.cpp)
        out.scripts.clear();
        if (ucls_version >= 3) {
            at += n1 * 2 + g2;
            if (ucls_len < at + 4) { err = "truncated unicode class table (scripts)"; return 1; }
            const uint32_t ns = rd32(ucls + at);
            at += 4;
            for (uint32_t k = 0; k < ns; k++) {
                if (ucls_len < at + 36) { err = "truncated unicode class table (scripts)"; return 1; }
                HostTables::Script sc;
                sc.name.assign((const char*)ucls + at, strnlen((const char*)ucls + at, 32));
                const uint32_t nr = rd32(ucls + at + 32);
                at += 36;
                if (ucls_len < at + (size_t)nr * 8) { err = "truncated unicode class table (scripts)"; return 1; }
                for (uint32_t q = 0; q < nr; q++) sc.ranges.emplace_back(rd32(ucls + at + 8 * q), rd32(ucls + at + 8 * q + 4));
                at += (size_t)nr * 8;
                out.scripts.push_back(std::move(sc));
            }
        }
    }
    out.cjk_fast = true;
    for (uint32_t cp = 0x4E00; cp < 0xA000 && out.cjk_fast; cp++) out.cjk_fast = host_cp_class(out, cp) == C_LO;
    for (uint32_t cp = 0xAC00; cp < 0xD7A4 && out.cjk_fast; cp++) out.cjk_fast = host_cp_class(out, cp) == C_LO;
    if (pattern != PAT_CL100K && pattern != PAT_O200K && pattern != PAT_MISTRAL_V3) { err = "unknown pattern id"; return 1; }
    out.pattern = pattern;

    // ---- vocabulary (reference src/core/vocab.rs:57-89: later duplicate wins) ---------------
    std::unordered_map<std::string, uint32_t> enc;
    uint32_t max_rank_seen = 0;
    if (splv_len >= 4 && memcmp(splv, "SPLV", 4) == 0) {
        // this repo's packed container (tools/pack_vocab.py); the ByteLevel flag travels inside
        if (splv_len < 20 || rd32(splv + 4) != 1) { err = "bad vocabulary container"; return 1; }
        const uint32_t nrec = rd32(splv + 8);
        out.byte_level = (rd32(splv + 12) & 1u) != 0 || force_byte_level;
        enc.reserve(nrec * 2);
        size_t off = 20;
        for (uint32_t i = 0; i < nrec; i++) {
            if (off + 6 > splv_len) { err = "truncated vocabulary container"; return 1; }
            const uint32_t rank = rd32(splv + off);
            const uint16_t len = rd16(splv + off + 4);
            off += 6;
            if (off + len > splv_len) { err = "truncated vocabulary container"; return 1; }
            enc[std::string((const char*)splv + off, len)] = rank;
            off += len;
        }
    } else {
        // the reference's on-disk format: tiktoken text, one `base64(token bytes) rank` per line
        // (load_tiktoken_bpe, src/core/vocab.rs:57-89: lines split at \n, empty lines skipped, the LAST
        // space separates token and rank, the rank is trimmed and parsed as u32, a later duplicate
        // key replaces the earlier one)
        out.byte_level = force_byte_level;
        static int8_t b64[256];
        static bool b64_init = false;
        if (!b64_init) {
            for (int i = 0; i < 256; i++) b64[i] = -1;
            const char* al = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
            for (int i = 0; i < 64; i++) b64[(uint8_t)al[i]] = (int8_t)i;
            b64_init = true;
        }
        enc.reserve(splv_len / 12 + 16);
        std::string key;
        size_t ln = 1;
        for (size_t a = 0; a < splv_len; ln++) {
            size_t e = a;
            while (e < splv_len && splv[e] != '\n') e++;
            const uint8_t* line = splv + a;
            const size_t n = e - a;
            a = e + 1;
            if (n == 0) continue;
            size_t sp = n;
            while (sp > 0 && line[sp - 1] != ' ') sp--;
            if (sp == 0) { err = "Invalid line format: Missing space separator (line " + std::to_string(ln) + ")"; return 1; }
            const size_t nb = sp - 1;                           // base64 text = line[0, nb)
            // base64, standard alphabet, canonical padding (base64::engine::general_purpose::STANDARD)
            key.clear();
            if (nb % 4 != 0) { err = "Invalid base64 encoding (line " + std::to_string(ln) + ")"; return 1; }
            for (size_t i = 0; i < nb; i += 4) {
                int v[4], pad = 0;
                for (int k = 0; k < 4; k++) {
                    const uint8_t c = line[i + k];
                    if (c == '=' && i + 4 == nb && k >= 2) { v[k] = 0; pad++; }
                    else if (b64[c] < 0 || pad) { err = "Invalid base64 encoding (line " + std::to_string(ln) + ")"; return 1; }
                    else v[k] = b64[c];
                }
                const uint32_t w = (uint32_t)v[0] << 18 | (uint32_t)v[1] << 12 | (uint32_t)v[2] << 6 | (uint32_t)v[3];
                if ((pad == 1 && (v[2] & 3)) || (pad == 2 && (v[1] & 15))) {      // non-canonical trailing bits
                    err = "Invalid base64 encoding (line " + std::to_string(ln) + ")"; return 1;
                }
                key.push_back((char)(w >> 16));
                if (pad < 2) key.push_back((char)(w >> 8));
                if (pad < 1) key.push_back((char)w);
            }
            // rank: str::trim + parse::<u32> (an optional leading '+', decimal digits, no overflow)
            // (str::trim trims Unicode White_Space, not only ASCII: U+0085, U+00A0, U+1680, U+2000..200A, U+2028,
            //  U+2029, U+202F, U+205F, U+3000 as well -- a rank field "12\xc2\xa0" loads in the reference)
            size_t r0 = sp, r1 = n;
            auto ws_at = [&](size_t i, size_t end) -> size_t {       // bytes of the White_Space character at line[i], 0 if none
                const uint8_t c = line[i];
                if (c == ' ' || (c >= 9 && c <= 13)) return 1;
                if (c == 0xC2 && i + 1 < end && (line[i + 1] == 0x85 || line[i + 1] == 0xA0)) return 2;
                if (i + 2 < end) {
                    const uint8_t d = line[i + 1], f = line[i + 2];
                    if (c == 0xE1 && d == 0x9A && f == 0x80) return 3;
                    if (c == 0xE2 && d == 0x80 && ((f >= 0x80 && f <= 0x8A) || f == 0xA8 || f == 0xA9 || f == 0xAF)) return 3;
                    if (c == 0xE2 && d == 0x81 && f == 0x9F) return 3;
                    if (c == 0xE3 && d == 0x80 && f == 0x80) return 3;
                }
                return 0;
            };
            for (size_t w; r0 < r1 && (w = ws_at(r0, r1)) != 0;) r0 += w;
            for (;;) {
                bool cut = false;
                for (size_t w = 1; w <= 3 && !cut; w++)
                    if (r1 >= r0 + w && ws_at(r1 - w, r1) == w) { r1 -= w; cut = true; }
                if (!cut) break;
            }
            if (r0 < r1 && line[r0] == '+') r0++;
            uint64_t rank = 0;
            bool ok = r0 < r1;
            for (size_t i = r0; i < r1 && ok; i++) {
                ok = line[i] >= '0' && line[i] <= '9';
                rank = rank * 10 + (uint64_t)(line[i] - '0');
                if (rank > 0xFFFFFFFFull) ok = false;
            }
            if (!ok) { err = "Invalid line format: Invalid rank: " + std::string((const char*)line + sp, n - sp); return 1; }
            enc[key] = (uint32_t)rank;
        }
    }
    for (const auto& kv : enc) max_rank_seen = std::max(max_rank_seen, kv.second);   // vocab_size: max id of the map as loaded
    if (max_rank_seen > SPL_ID_MASK) { err = "token id does not fit 21 bits"; return 1; }
    enc.erase(std::string());          // an empty key can never match a chunk
    // decoder side (build_decoder, src/core/vocab.rs:146-148, + Tokenizer::decode_bytes,
    // src/core/tokenizer.rs:877-897): id -> the bytes decode_bytes emits for it.  ByteLevel: the key
    // decoded to raw bytes, or the key itself where it is not ByteLevel text (byte_level.rs:125-146).
    std::unordered_map<uint32_t, std::string> dec;
    std::vector<uint32_t> verbatim;               // ByteLevel: ids whose key is emitted as it is
    dec.reserve(enc.size() * 2);
    if (!out.byte_level) for (const auto& kv : enc) dec[kv.second] = kv.first;
    if (out.byte_level) {
        // Re-key into raw-byte space.  The reference merges over the UTF-8 bytes of the ByteLevel
        // text, where a byte whose alphabet char is U+0080 or above starts out as TWO nodes that only
        // the char's own token joins (neither half is a key).  Working on raw bytes -- those chars
        // joined from the start -- is exact iff every alphabet char is a token and every longer
        // token that CONTAINS such a two-byte char ranks above all of them: then no merge the
        // reference makes before a char is whole can involve it (DESIGN.md "ByteLevel equivalence").
        // Tokens made of one-byte chars only may rank anywhere (mistral_v3's control tokens, ids 0..999).
        uint32_t cp_of_byte[256];
        byte_level_alphabet(cp_of_byte);
        std::unordered_map<uint32_t, uint8_t> byte_of_cp;
        for (int b = 0; b < 256; b++) byte_of_cp[cp_of_byte[b]] = (uint8_t)b;
        std::unordered_map<std::string, uint32_t> raw_enc;
        raw_enc.reserve(enc.size() * 2);
        uint32_t max_char_rank = 0, min_multi_rank = 0xFFFFFFFFu;
        std::string raw;
        for (const auto& kv : enc) {
            if (!byte_level_to_raw(kv.first, byte_of_cp, raw)) {
                // Text with a char outside the alphabet can never equal a stretch of encoded text
                // (deepseek_v3 ids 0..2).  A key that is no UTF-8 text at all could: the reference's nodes
                // start as the single UTF-8 BYTES of the ByteLevel text, and a stretch of them need not be
                // whole chars -- such a vocabulary is outside what the raw-byte re-keying reproduces.
                if (!valid_utf8(kv.first)) { err = "ByteLevel vocabulary: a key is not UTF-8 text"; return 1; }
                dec[kv.second] = kv.first;
                verbatim.push_back(kv.second);
                continue;
            }
            dec[kv.second] = raw;
            if (raw.empty()) continue;
            raw_enc[raw] = kv.second;
            bool wide = false;
            for (char ch : raw) wide = wide || cp_of_byte[(uint8_t)ch] >= 0x80u;
            if (!wide) continue;
            if (raw.size() == 1) max_char_rank = std::max(max_char_rank, kv.second);
            else min_multi_rank = std::min(min_multi_rank, kv.second);
        }
        for (int b = 0; b < 256; b++)
            if (!raw_enc.count(std::string(1, (char)b))) { err = "ByteLevel vocabulary lacks an alphabet character"; return 1; }
        if (max_char_rank >= min_multi_rank) { err = "ByteLevel vocabulary: a token outranks an alphabet character it contains"; return 1; }
        enc.swap(raw_enc);
    }
    if (enc.empty()) { err = "empty vocabulary"; return 1; }
    {   // the merge kernels identify a node with its token id: an id must name ONE key
        std::vector<uint8_t> seen((size_t)max_rank_seen + 1, 0);
        for (const auto& kv : enc) {
            if (seen[kv.second]) { err = "two keys share the id " + std::to_string(kv.second); return 1; }
            seen[kv.second] = 1;
        }
    }
    out.max_id = max_rank_seen;
    out.max_key_len = 0;
    for (const auto& kv : enc) out.max_key_len = std::max<uint32_t>(out.max_key_len, (uint32_t)kv.first.size());
    out.n_keys = (uint32_t)enc.size();
    out.byte_id.assign(256, SPL_NO_RANK);
    out.all_bytes = true;
    for (int b = 0; b < 256; b++) {
        auto it = enc.find(std::string(1, (char)b));
        if (it != enc.end()) out.byte_id[b] = it->second; else out.all_bytes = false;
    }
    // A vocabulary that LACKS single bytes (the reference takes any: byte_pair_encode ranks pairs by their concatenated bytes and drops a
    // node whose bytes are no token when it collects the result, src/core/bpe.rs:73-75, 99-111, 182-191).  The pair-table merge loops
    // identify a node with an id, so every missing byte gets a PSEUDO id behind the vocabulary's (max_id + 1 + k): pairs through such a byte
    // are in the pair table like any other, the byte's own "token" is dropped where tokens are emitted (id >= id_limit).
    out.id_limit = 0xFFFFFFFFu;
    if (!out.all_bytes) {
        uint32_t next = max_rank_seen + 1;
        for (int b = 0; b < 256; b++) if (out.byte_id[b] == SPL_NO_RANK) out.byte_id[b] = next++;
        if (next - 1 > SPL_ID_MASK) { err = "token ids must be < 2^21 (with the pseudo ids of the single bytes the vocabulary lacks)"; return 1; }
        out.id_limit = max_rank_seen + 1;
    }

    // ---- short / long key tables ----------------------------------------------------------
    size_t n_short = 0, n_long = 0, n_tiny = 0, n_t8 = 0;
    for (const auto& kv : enc) {
        const size_t n = kv.first.size();
        (n <= (size_t)SPL_TINY_MAX ? n_tiny : n <= (size_t)SPL_T8_MAX ? n_t8 : n <= (size_t)SPL_SHORT_MAX ? n_short : n_long)++;
    }
    // short table (keys of 9..12 bytes): buckets of 4, at most ~40 % full, an 8-bit salt per two-byte prefix (below).
    // tiny / t8 tables: ONE entry per slot, placed by displacement (further below).
    const uint32_t sbuckets = pow2_at_least(n_short * 100 / (4 * 40) + 2);
    out.short_tab.assign((size_t)sbuckets * SPL_SHORT_BUCKET, ShortEnt{0, 0, 0, SPL_EMPTY});
    out.unsalted_groups = 0;
    const uint32_t lcap = pow2_at_least(n_long * 2 + 2);
    out.long_tab.assign(lcap, LongEnt{0, SPL_EMPTY, 0, 0});
    out.key_blob.clear();
    out.len_mask.assign(65536, 0);
    for (const auto& kv : enc) {
        const std::string& k = kv.first;
        if (k.size() < 2) continue;
        out.len_mask[(uint8_t)k[0] | (uint32_t)(uint8_t)k[1] << 8] |= (uint16_t)(1u << (k.size() > (size_t)SPL_T8_MAX ? 7 : k.size() - 2));
    }
    {   // p8: for every 8-byte prefix of a longer token, the longest such token
        size_t n9 = 0;
        for (const auto& kv : enc) n9 += kv.first.size() > (size_t)SPL_T8_MAX;
        const uint32_t buckets = std::max<uint32_t>(1u << 14, pow2_at_least(n9 * 2 + 2));   // sparse: a false hit costs parallelism
        out.p8_tab.assign((size_t)buckets * 2, 0);
        for (const auto& kv : enc) {
            const std::string& k = kv.first;
            if (k.size() <= (size_t)SPL_T8_MAX) continue;
            const uint32_t k0 = load_le(k, 0), k1 = load_le(k, 4);
            uint32_t* e = &out.p8_tab[(size_t)(hash_p8(k0, k1) & (buckets - 1)) * 2];
            const uint32_t len = (uint32_t)std::min<size_t>(k.size(), 255), tag = p8_tag(k0, k1);
            int slot = -1;
            for (int i = 0; i < 2 && slot < 0; i++) if (e[i] != 0 && (e[i] >> 8) == tag) slot = i;
            for (int i = 0; i < 2 && slot < 0; i++) if (e[i] == 0) slot = i;
            if (slot < 0) {                              // a third prefix: entry 1 turns into "every key"
                e[1] = 0xFFFFFFu << 8 | std::max<uint32_t>(e[1] & 0xFFu, len);
            } else if ((e[slot] >> 8) == 0xFFFFFFu) {    // (cannot happen: the wildcard tag is not a key's tag)
                e[slot] = 0xFFFFFFu << 8 | std::max<uint32_t>(e[slot] & 0xFFu, len);
            } else {
                e[slot] = tag << 8 | std::max<uint32_t>(e[slot] & 0xFFu, len);
            }
        }
    }
    // ---- short table: one 8-bit SALT per two-byte key prefix (DeviceTables::len_mask) ---------------------
    // Keys are placed group by group (a group = all keys of 9..12 bytes with the same first two bytes, largest group
    // first); a group takes the first salt under which each of its keys finds its home bucket with a free slot.  No key
    // then ever overflows, so no probe -- hit or miss -- goes on to a second bucket.  A group that finds no salt keeps salt
    // 0 and overflows into the next bucket, marking the full one (SPL_OVF_BIT); the probes walk on from a marked bucket.
    struct KeyRef { const std::string* k; uint32_t id; };
    {
        constexpr int SPL_BUCKET_FILL = 4;
        std::vector<std::vector<KeyRef>> groups(65536);
        for (const auto& kv : enc) {
            const std::string& k = kv.first;
            if (k.size() <= (size_t)SPL_T8_MAX || k.size() > (size_t)SPL_SHORT_MAX) continue;
            groups[(uint8_t)k[0] | (uint32_t)(uint8_t)k[1] << 8].push_back(KeyRef{&k, kv.second});
        }
        std::vector<uint32_t> order;
        for (uint32_t g = 0; g < 65536; g++) if (!groups[g].empty()) order.push_back(g);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return groups[a].size() > groups[b].size(); });
        std::vector<uint8_t> cnt_s(sbuckets, 0);
        auto home = [&](const std::string& k, uint32_t salt) {
            return hash_short(load_le(k, 0), load_le(k, 4), load_le(k, 8), (uint32_t)k.size(), salt) & (sbuckets - 1);
        };
        std::vector<uint32_t> touched;
        for (uint32_t g : order) {
            const auto& keys = groups[g];
            uint32_t salt = 0;
            bool found = false;
            for (; salt < 256 && !found; salt++) {
                touched.clear();
                bool ok = true;
                for (const KeyRef& kr : keys) {
                    const uint32_t bkt = home(*kr.k, salt);
                    if (cnt_s[bkt] >= SPL_BUCKET_FILL) { ok = false; break; }
                    cnt_s[bkt]++;
                    touched.push_back(bkt);
                }
                for (uint32_t t : touched) cnt_s[t]--;
                if (ok) { found = true; break; }
            }
            if (!found) { salt = 0; out.unsalted_groups++; }
            out.len_mask[g] = (uint16_t)((out.len_mask[g] & 0xFFu) | (salt << 8));
            for (const KeyRef& kr : keys) {
                uint32_t bkt = home(*kr.k, salt);
                for (;;) {
                    ShortEnt* e = &out.short_tab[(size_t)bkt * SPL_SHORT_BUCKET];
                    int f = 0;
                    while (f < SPL_SHORT_BUCKET && e[f].id_len != SPL_EMPTY) f++;
                    if (f < SPL_SHORT_BUCKET) {
                        e[f] = ShortEnt{load_le(*kr.k, 0), load_le(*kr.k, 4), load_le(*kr.k, 8), kr.id | ((uint32_t)kr.k->size() << 24)};
                        cnt_s[bkt]++;
                        break;
                    }
                    e[SPL_SHORT_BUCKET - 1].id_len |= SPL_OVF_BIT;
                    bkt = (bkt + 1) & (sbuckets - 1);
                }
            }
        }
    }
    // ---- prefix entries and the four-byte-prefix filter (the salts of the two tables below live in them) ---------
    {
        out.pfx.assign(65536, PfxEnt{0u, SPL_NO_RANK});
        for (uint32_t g = 0; g < 65536; g++) out.pfx[g].lm = out.len_mask[g];
        size_t n4 = 0;
        for (const auto& kv : enc) {
            const std::string& k = kv.first;
            if (k.size() == 2) out.pfx[(uint8_t)k[0] | (uint32_t)(uint8_t)k[1] << 8].id2 = kv.second;
            n4 += k.size() >= 4;
        }
        // about one key in ten slots (distinct prefixes are fewer still); 16-bit entries: six length bits, ten bits of salt
        uint32_t bits = 16;
        while (bits < 22 && ((size_t)1 << bits) < n4 * 4) bits++;
        out.filt4_shift = 32 - bits;
        out.filt4.assign((size_t)1 << bits, 0);
        for (const auto& kv : enc) {
            const std::string& k = kv.first;
            if (k.size() < 4) continue;
            out.filt4[hash_f4(load_le(k, 0)) >> out.filt4_shift] |= (uint16_t)(1u << (k.size() > (size_t)SPL_T8_MAX ? 5 : k.size() - 4));
        }
    }
    // ---- tiny / t8 tables: ONE ENTRY PER SLOT, by displacement ("hash and displace") ------------------------------
    // A group of keys shares a salt: the keys of 1..4 bytes by their first two bytes (salt in the prefix entry, 16 bits), the
    // keys of 5..8 bytes by the filter slot of their first four bytes (salt in the filter entry, 10 bits) -- both entries a
    // probe has read anyway before it can hash.  Groups are placed largest first; a group takes the first salt under which
    // every one of its keys lands in a slot that is still free, no two of them in the same one.  Large groups go in while
    // the table is nearly empty, the thousands of one- and two-key groups fill what is left: cl100k_base places 27 000
    // tiny keys in 2^16 slots and 48 600 t8 keys in 2^17, o200k_base 99 000 t8 keys in 2^17 slots (76 % full).  A table
    // that cannot be completed is doubled (up to 2^24 slots).
    auto displace = [&](std::vector<std::vector<KeyRef>>& groups, int words, int salt_bits, size_t n_keys, size_t min_slots,
                        std::vector<uint32_t>& tab, std::vector<uint32_t>& salts, const char* what) -> bool {
        std::vector<uint32_t> order;
        for (uint32_t g = 0; g < groups.size(); g++) if (!groups[g].empty()) order.push_back(g);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return groups[a].size() > groups[b].size(); });
        uint32_t slots = pow2_at_least(std::max(n_keys + n_keys / 4 + 2, min_slots));
        // (a table that cannot be completed is doubled, up to 2^24 slots: a group of k keys needs about k^2 / (2 ln(salts)) slots to find a salt
        //  under which its keys collide neither with each other nor with what is placed -- 4 000 keys of 3..4 bytes under ONE two-byte
        //  prefix take 2^20 slots, 8 MB; the shipped vocabularies complete at the first size)
        for (; slots <= (1u << 24); slots *= 2) {
            std::vector<uint8_t> used(slots, 0);
            salts.assign(groups.size(), 0);
            std::vector<uint32_t> at;
            bool all = true;
            for (uint32_t g : order) {
                const auto& keys = groups[g];
                bool found = false;
                for (uint32_t salt = 0; salt < (1u << salt_bits) && !found; salt++) {
                    at.clear();
                    bool ok = true;
                    for (const KeyRef& kr : keys) {
                        const uint32_t n = (uint32_t)kr.k->size();
                        const uint32_t h = (words == SPL_TINY_WORDS ? hash_tiny(load_le(*kr.k, 0), n, salt)
                                                                     : hash_t8(load_le(*kr.k, 0), load_le(*kr.k, 4), n, salt)) & (slots - 1);
                        if (used[h]) { ok = false; break; }
                        used[h] = 1;                     // (also catches two keys of the group in one slot)
                        at.push_back(h);
                    }
                    if (ok) { found = true; salts[g] = salt; }
                    else for (uint32_t h : at) used[h] = 0;
                }
                if (!found) { all = false; break; }
            }
            if (!all) continue;
            // entries (+ 4 words of padding: a probe may read one word past its entry)
            tab.assign((size_t)slots * words + 4, SPL_EMPTY);
            for (uint32_t g : order)
                for (const KeyRef& kr : groups[g]) {
                    const uint32_t n = (uint32_t)kr.k->size();
                    const uint32_t h = (words == SPL_TINY_WORDS ? hash_tiny(load_le(*kr.k, 0), n, salts[g])
                                                                 : hash_t8(load_le(*kr.k, 0), load_le(*kr.k, 4), n, salts[g])) & (slots - 1);
                    uint32_t* e = &tab[(size_t)h * words];
                    e[0] = load_le(*kr.k, 0);
                    if (words == SPL_T8_WORDS) e[1] = load_le(*kr.k, 4);
                    e[words - 1] = kr.id | (n << 24);
                }
            return true;
        }
        err = std::string("could not give every key of the ") + what + " table a slot of its own within 2^24 slots (a two-byte / four-byte prefix shared by more than about ten thousand keys)";
        return false;
    };
    {
        std::vector<std::vector<KeyRef>> groups(65536);
        for (const auto& kv : enc) {
            const std::string& k = kv.first;
            if (k.size() > (size_t)SPL_TINY_MAX) continue;
            groups[(uint8_t)k[0] | (k.size() > 1 ? (uint32_t)(uint8_t)k[1] << 8 : 0u)].push_back(KeyRef{&k, kv.second});
        }
        std::vector<uint32_t> salts;
        if (!displace(groups, SPL_TINY_WORDS, SPL_TINY_SALT_BITS, n_tiny, 1u << 12, out.tiny_tab, salts, "tiny")) return 1;
        for (uint32_t g = 0; g < 65536; g++) out.pfx[g].lm = (out.pfx[g].lm & 0xFFFFu) | (salts[g] << 16);
    }
    {
        std::vector<std::vector<KeyRef>> groups(out.filt4.size());
        for (const auto& kv : enc) {
            const std::string& k = kv.first;
            if (k.size() <= (size_t)SPL_TINY_MAX || k.size() > (size_t)SPL_T8_MAX) continue;
            groups[hash_f4(load_le(k, 0)) >> out.filt4_shift].push_back(KeyRef{&k, kv.second});
        }
        std::vector<uint32_t> salts;
        if (!displace(groups, SPL_T8_WORDS, SPL_T8_SALT_BITS, n_t8, 1u << 12, out.t8_tab, salts, "t8")) return 1;
        for (size_t g = 0; g < out.filt4.size(); g++) out.filt4[g] = (uint16_t)((out.filt4[g] & 0x3Fu) | (salts[g] << SPL_F4_MASK_BITS));
    }
    for (const auto& kv : enc) {
        const std::string& k = kv.first;
        const uint32_t n = (uint32_t)k.size();
        if (n <= (uint32_t)SPL_SHORT_MAX) {
            continue;                                    // placed above
        } else {
            uint32_t h = 0;
            for (uint32_t i = 0; i < n; i += 4) h = hash_long_step(h, load_le(k, i));
            h = hash_long_fin(h, n);
            uint32_t slot = h & (lcap - 1);
            while (out.long_tab[slot].id != SPL_EMPTY) slot = (slot + 1) & (lcap - 1);
            const uint32_t off = (uint32_t)out.key_blob.size();
            out.key_blob.insert(out.key_blob.end(), k.begin(), k.end());
            while (out.key_blob.size() & 3) out.key_blob.push_back(0);
            out.long_tab[slot] = LongEnt{hash_long_tag(h), kv.second, off, n};
        }
    }
    out.key_blob.resize(out.key_blob.size() + 16, 0);

    // ---- pair table: every 2-split of every token whose halves are tokens --------------------
    std::vector<std::pair<uint64_t, uint32_t>> pairs;
    pairs.reserve(enc.size() * 3);
    auto node_id = [&](const std::string& bytes) -> uint32_t {     // a token's id; a single byte the vocabulary lacks: its pseudo id
        auto it = enc.find(bytes);
        if (it != enc.end()) return it->second;
        return bytes.size() == 1 ? out.byte_id[(uint8_t)bytes[0]] : SPL_NO_RANK;
    };
    for (const auto& kv : enc) {
        const std::string& k = kv.first;
        for (size_t cut = 1; cut < k.size(); cut++) {
            const uint32_t l = node_id(k.substr(0, cut));
            if (l == SPL_NO_RANK) continue;
            const uint32_t r = node_id(k.substr(cut));
            if (r == SPL_NO_RANK) continue;
            pairs.emplace_back(pair_key(l, r), kv.second);
        }
    }
    out.n_pairs = (uint32_t)pairs.size();
    // buckets of 4, about 1.8 entries per bucket on average
    const uint32_t pbuckets = pow2_at_least(pairs.size() * 5 / 9 + 2);
    out.pair_tab.assign((size_t)pbuckets * SPL_PAIR_BUCKET, SPL_PAIR_EMPTY);
    for (const auto& pr : pairs) {
        const uint32_t l = (uint32_t)(pr.first & SPL_ID_MASK), r = (uint32_t)(pr.first >> SPL_ID_BITS);
        uint32_t bkt = hash_pair(l, r) & (pbuckets - 1);
        for (;;) {
            uint64_t* e = &out.pair_tab[(size_t)bkt * SPL_PAIR_BUCKET];
            int f = 0;
            bool dup = false;
            while (f < SPL_PAIR_BUCKET && e[f] != SPL_PAIR_EMPTY) {
                if ((e[f] & SPL_PAIR_KEY_MASK) == pr.first) { dup = true; break; }
                f++;
            }
            if (dup) {   // two different tokens with the same (left,right) split would be the same bytes
                if ((uint32_t)(e[f] >> (2 * SPL_ID_BITS)) != pr.second) { err = "pair table conflict"; return 1; }
                break;
            }
            if (f < SPL_PAIR_BUCKET) { e[f] = pr.first | ((uint64_t)pr.second << (2 * SPL_ID_BITS)); break; }
            bkt = (bkt + 1) & (pbuckets - 1);
        }
    }

    // ---- decoder CSR (id -> raw bytes) ---------------------------------------------------------
    std::vector<const std::string*> by_id(out.max_id + 1, nullptr);
    for (const auto& kv : dec) by_id[kv.first] = &kv.second;   // (two keys with one id: arbitrary, as build_decoder)
    out.tok_off.assign(out.max_id + 2, 0);
    out.tok_present.assign(out.max_id + 1, 0);
    out.tok_bytes.clear();
    for (uint32_t id = 0; id <= out.max_id; id++) {
        out.tok_off[id] = (uint32_t)out.tok_bytes.size();
        if (by_id[id]) { out.tok_present[id] = 1; out.tok_bytes.insert(out.tok_bytes.end(), by_id[id]->begin(), by_id[id]->end()); }
    }
    out.tok_off[out.max_id + 1] = (uint32_t)out.tok_bytes.size();
    for (uint32_t id : verbatim) if (id <= out.max_id && out.tok_present[id]) out.tok_present[id] = 2;
    // an EMPTY slot in each of the two small-key tables: where probes known to miss are sent
    {
        const size_t ts = (out.tiny_tab.size() - 4) / SPL_TINY_WORDS, es = (out.t8_tab.size() - 4) / SPL_T8_WORDS;
        bool ft = false, fe = false;
        for (size_t i = 0; i < ts && !ft; i++)
            if (out.tiny_tab[i * SPL_TINY_WORDS + 1] == SPL_EMPTY) { out.tiny_free = (uint32_t)i; ft = true; }
        for (size_t i = 0; i < es && !fe; i++)
            if (out.t8_tab[i * SPL_T8_WORDS + 2] == SPL_EMPTY) { out.t8_free = (uint32_t)i; fe = true; }
        if (!ft || !fe) { err = "small-key tables have no empty slot"; return 1; }
    }
    return 0;
}

}  // namespace spl
