// spl_common.h -- definitions shared by the host table builder and the gfx950 kernels.
//
// Everything here is integer/byte work: class codes of the split patterns
// (reference src/core/tokenizer.rs:39, :42), the packed lookup-table entry formats that
// replace the reference's FxHashMap<Vec<u8>,u32> (src/core/tokenizer.rs:302), and the
// hash functions the host builder and the device probes must agree on.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPL_HD __host__ __device__ __forceinline__
#else
#define SPL_HD inline
#endif

namespace spl {

// ----------------------------------------------------------------------------------------
// Code-point classes (same numbering as tools/gen_unicode_tables.py CLASS_NAMES).
// ----------------------------------------------------------------------------------------
enum : uint32_t {
    C_P = 0,    // anything else: punctuation, symbols, controls, format, unassigned, emoji
    C_AP = 1,   // U+0027 apostrophe (member of the "other" set, singled out for contractions)
    C_SP = 2,   // U+0020
    C_WS = 3,   // \s minus U+0020, CR, LF
    C_NL = 4,   // CR, LF
    C_N = 5,    // \p{N}
    C_LU = 6, C_LL = 7, C_LT = 8, C_LM = 9, C_LO = 10,   // \p{Lu} \p{Ll} \p{Lt} \p{Lm} \p{Lo}
    C_M = 11,   // \p{M}
    C_EOT = 12, // end of the current text (document end / special-token span)
    C_WEND = 13,// end of the staged window: the scanner must defer
    C_CONT = 15 // UTF-8 continuation byte (not a character start)
};
#define SPL_BIT(c) (1u << (c))
constexpr uint32_t M_L = SPL_BIT(C_LU) | SPL_BIT(C_LL) | SPL_BIT(C_LT) | SPL_BIT(C_LM) | SPL_BIT(C_LO);
constexpr uint32_t M_S = SPL_BIT(C_SP) | SPL_BIT(C_WS) | SPL_BIT(C_NL);
constexpr uint32_t M_OTHER = SPL_BIT(C_P) | SPL_BIT(C_AP) | SPL_BIT(C_M);           // [^\s\p{L}\p{N}]
constexpr uint32_t M_X = SPL_BIT(C_P) | SPL_BIT(C_AP) | SPL_BIT(C_SP) | SPL_BIT(C_WS) | SPL_BIT(C_M); // [^\r\n\p{L}\p{N}]
constexpr uint32_t M_U = SPL_BIT(C_LU) | SPL_BIT(C_LT) | SPL_BIT(C_LM) | SPL_BIT(C_LO) | SPL_BIT(C_M);
constexpr uint32_t M_W = SPL_BIT(C_LL) | SPL_BIT(C_LM) | SPL_BIT(C_LO) | SPL_BIT(C_M);
constexpr uint32_t M_UW = SPL_BIT(C_LM) | SPL_BIT(C_LO) | SPL_BIT(C_M);             // in both U and W

// Per-byte class record kept in LDS by the pre-tokeniser:
//   bits 0-3 class code, bit 4 SYNC (context-free match start), bit 5 TSTART (a text starts
//   here: look-ahead from the left must see end-of-text), bits 6-7 UTF-8 length - 1.
constexpr uint32_t CB_CLASS = 0x0F, CB_SYNC = 0x10, CB_TSTART = 0x20, CB_LEN_SHIFT = 6;

enum : int { PAT_CL100K = 0, PAT_O200K = 1, PAT_MISTRAL_V3 = 2 };

// ----------------------------------------------------------------------------------------
// Lookup tables in HBM (built on the host by spl_tables.cpp, probed by the kernels).
// ----------------------------------------------------------------------------------------
// Both big tables are BUCKETED: a probe fetches one whole bucket with back-to-back 16-byte loads
// (one memory round trip) and almost never needs a second bucket, because a wavefront waits for
// its slowest lane and linear probing by single entries made that lane take 3-5 dependent trips.
// Buckets fill left to right and nothing is ever deleted; a key that finds its bucket full goes to
// the next one and marks the full bucket (SPL_OVF_BIT), so a probe knows from ONE bucket whether it is
// settled.  The builder salts the hashes per two-byte key prefix (DeviceTables::len_mask) so that no
// key of the shipped vocabularies has to go on at all; the walk to the next bucket stays in the
// probes for vocabularies where it does not manage that.
struct alignas(16) Quad { uint32_t x, y, z, w; };       // one dwordx4 load
// Short-key table: vocabulary entries whose key is <= 12 bytes, key stored inline.
struct ShortEnt {            // 16 B
    uint32_t k0, k1, k2;     // key bytes, little endian, zero padded
    uint32_t id_len;         // id | len << 24 ; 0xFFFFFFFF = empty slot
};
constexpr int SPL_SHORT_BUCKET = 4;                     // entries per 64-byte bucket (one cache line)
// Keys of up to 8 bytes -- nearly every probe of the hot path (whole chunks of a few bytes, the substring tabulation of
// the merge loops) -- have two tables of their own with ONE ENTRY PER SLOT (round 4): the builder places every key
// in a slot of its own (a perfect hash by displacement: groups of keys share a salt that is chosen so that none of
// them collides with anything placed before), so a probe loads exactly one entry -- 8 or 12 bytes instead of a
// 32- / 48-byte bucket of four -- and compares once.  The tile kernel was bound, in its probe and merge phases, by the
// rate at which a CU takes scattered load addresses and by instruction issue; both fall with the bytes and compares
// per probe (profiles/r04_single_slot_tables.txt).
//   tiny table: keys of 1..4 bytes, entries {key, id | len << 24}; slot = hash_tiny(key, len, salt) & tiny_mask with
//               the salt of the key's first two bytes (16 bits, PfxEnt::lm >> 16)
//   t8 table  : keys of 5..8 bytes, entries {k0, k1, id | len << 24} (12 bytes, packed); slot = hash_t8(..., salt) with
//               the salt of the key's first FOUR bytes (10 bits, in the four-byte-prefix filter's entry: filt4 >> 6)
// An empty slot is all-ones (its length byte, 0xFF, equals no key length).  The short table keeps the keys of 9..12 bytes
// in buckets of four with the 8-bit salt of round 2.
constexpr int SPL_TINY_WORDS = 2;                       // u32 words per tiny entry
constexpr int SPL_TINY_MAX = 4;
constexpr int SPL_T8_WORDS = 3;                         // u32 words per t8 entry
constexpr int SPL_T8_MAX = 8;
constexpr int SPL_TINY_SALT_BITS = 16, SPL_T8_SALT_BITS = 10, SPL_F4_MASK_BITS = 6;
// Long-key table: 13..max_key_len bytes; key bytes live in a 4-byte-aligned blob.
struct LongEnt {             // 16 B
    uint32_t tag;            // second hash, filters almost every false candidate
    uint32_t id;             // 0xFFFFFFFF = empty slot
    uint32_t off;            // byte offset into key blob (multiple of 4)
    uint32_t len;
};
constexpr uint32_t SPL_EMPTY = 0xFFFFFFFFu;
constexpr int SPL_SHORT_MAX = 12;
// Pair table: (left id, right id) -> id of the concatenation, one u64 per entry:
//   bits 0-20 left, 21-41 right, 42-62 merged id; all-ones = empty.  Ids < 2^21.
// Buckets of SPL_PAIR_BUCKET entries (32 bytes).
constexpr uint64_t SPL_PAIR_EMPTY = ~0ull;
constexpr int SPL_PAIR_BUCKET = 4;
constexpr uint32_t SPL_ID_BITS = 21;
constexpr uint32_t SPL_ID_MASK = (1u << SPL_ID_BITS) - 1;
constexpr uint32_t SPL_NO_RANK = 0xFFFFFFFFu;
// short table, id word (id | len << 24) of a bucket's LAST slot: a key went on from this (full) bucket to the next
// one.  Without it a probe that misses is settled by this bucket alone.
constexpr uint32_t SPL_OVF_BIT = 1u << 23;

struct P8Bucket { uint32_t a, b; };
// Prefix table entry (one dwordx2 load): what the first TWO bytes of a key say about it.
//   lm : low byte = length mask, next byte = salt of the short table (exactly DeviceTables::len_mask's entry), upper half =
//        salt of the tiny table for the keys that begin with these two bytes;
//   id2: the id of the token that IS those two bytes, or SPL_NO_RANK -- a two-byte key needs no bucket probe at all.
struct alignas(8) PfxEnt { uint32_t lm, id2; };

// Chunk memo (round 6): what the reference's LRU of encoded chunks is to its CPU path (src/core/tokenizer.rs:707-722), as a direct-mapped
// table in HBM -- chunks of up to 32 bytes that the vocabulary does not hold as ONE token, with the tokens their merge produced.  Probed
// (read-only) right behind the whole-chunk probe; filled BETWEEN launches by k_memo_fill from the misses the tiles logged, so that kernel
// boundaries on one stream give the coherence an in-kernel insert lacked (round 3).  Keys are compared in full: result-transparent.
struct alignas(64) MemoEnt {
    uint32_t key[8];     // the chunk's bytes, little endian, zero padded
    uint32_t meta0;      // bits 0-5 length in bytes (2..32), bits 8-11 tokens (0: known, but not memoizable -- more than fourteen), bit 31 valid
    uint32_t meta1;      // token t (t = 0..4) ends at byte (meta1 >> 6 t) & 63 of the chunk; the last token ends with the chunk
    uint32_t ids[6];
};
static_assert(sizeof(MemoEnt) == 64, "one cache line");
// A chunk of seven to fourteen tokens -- a long identifier, a hex string: few, but they are the tile kernel's longest merges, and the tile that
// holds one ends last -- has the rest of its tokens in the slot's entry of a second array, which only such a hit touches.
struct alignas(64) MemoExt {
    uint32_t ids[8];     // tokens 6..13
    uint32_t ends[2];    // token t (t = 5..12) ends at byte (ends[(t - 5) / 5] >> 6 ((t - 5) % 5)) & 63
    uint32_t pad[6];
};
static_assert(sizeof(MemoExt) == 64, "one cache line");
// Chunks of 33..64 bytes -- long identifiers, URLs: few, but a wavefront merges each of them alone and the tile that holds one ends its launch
// last (profiles/r06_memo.txt, block 8) -- live in a second, smaller table of the same entries; bytes 32..63 of their keys in a parallel array.
struct alignas(32) MemoHi { uint32_t k[8]; };
constexpr int SPL_MEMO_MAX_LEN = 32, SPL_MEMO_MAX_LEN2 = 64, SPL_MEMO_TOK1 = 6, SPL_MEMO_MAX_TOK = 14;
constexpr uint32_t SPL_MEMO_LOG_WORDS = 12;              // one logged miss: length, eight key words, padding (48 bytes)
constexpr uint32_t SPL_MEMO_LOG_WORDS2 = 20;             // ... of 33..64 bytes: length, sixteen key words, padding (80 bytes)
constexpr uint32_t SPL_MEMO_LOG_REGIONS = 64;            // a tile appends to region tile % 64: one returning atomic per tile and region counter

struct DeviceTables {
    // code-point classes
    const uint16_t* ucls_stage1;
    const uint8_t* ucls_stage2;
    uint32_t ucls_shift;
    uint32_t cjk_fast;        // 1 if U+4E00..U+9FFF and U+AC00..U+D7A3 are uniformly C_LO
    // vocabulary
    const ShortEnt* short_tab; uint32_t short_mask;   // masks index BUCKETS
    const uint32_t* tiny_tab;  uint32_t tiny_mask;      // masks index SLOTS (one entry each)
    const uint32_t* t8_tab;    uint32_t t8_mask;
    const LongEnt* long_tab;   uint32_t long_mask;
    const uint8_t* key_blob;
    const uint64_t* pair_tab;  uint32_t pair_mask;
    const uint32_t* byte_id;   // [256] id of each single-byte token or SPL_NO_RANK
    uint32_t max_key_len;
    uint32_t pattern;         // PAT_*
    uint32_t all_bytes;       // 1 if all 256 single bytes are tokens
    uint32_t id_limit;        // byte_id[] of a byte the vocabulary lacks is a pseudo id >= id_limit: merged like any id, dropped where tokens are emitted (0xFFFFFFFF: none)
    // p8: an upper bound of the length of tokens longer than 8 bytes by their first 8 bytes.  Bucket =
    // hash of the 8 bytes, two entries of tag << 8 | longest such token (255 = "unbounded"), 0 = free;
    // tag 0xFFFFFF matches every key (a third prefix met in the bucket).  A miss is exact, a hit may
    // be too long (another prefix with the same 24-bit tag).
    const P8Bucket* p8_tab;    uint32_t p8_mask;
    // len_mask[b0 | b1 << 8], low byte: bit L-2 set iff some token of exactly L bytes (L = 2..8) starts
    // with these two bytes, bit 7 iff a longer one does.  High byte: the SALT of the bucket hashes of
    // every key of up to 12 bytes that starts with these two bytes (a one-byte key: b1 = 0) -- chosen
    // by the builder so that no key of the tiny / t8 / short tables ever overflows its home bucket (a bucket
    // may be full; SPL_OVF_BIT alone says whether anything went on from it): a probe then never
    // needs a second bucket, hit or miss (a wavefront waits for its slowest lane; with plain hashing
    // 2-7 % of the buckets were full and nearly every wavefront had a lane that went on to the next).
    // tiny_free / t8_free: an EMPTY slot of each table -- where probes known to miss are sent (one cache line for
    // all of them).
    const uint16_t* len_mask;  uint32_t tiny_free, t8_free;
    uint32_t ascii_base;      // ucls_stage1[0] << ucls_shift: where the classes of U+0000..U+007F start in ucls_stage2
    // akind[2 c], akind[2 c + 1]: the kind-nibble entry of the ASCII byte c for this handle's split pattern (spl_scan_words.h
    // ascii_entry: what the byte adds to the two nibble words, its class code in the top nibble) -- built once on the host;
    // k_pretok copies the 1 KB into LDS instead of deriving it from the class table in every workgroup
    const uint32_t* akind;
    // pfx[b0 | b1 << 8]: len_mask's entry and the id of the two-byte token in one 8-byte load (the substring
    // tabulation and the whole-chunk probe read this one; len_mask stays for the generic probes).
    // filt4[hash_f4(first four bytes) >> filt4_shift]: bit k (k = 0..4) set iff SOME token of exactly 4 + k bytes
    // begins with four bytes that hash there, bit 5 iff a longer one does -- a Bloom-style filter (no false
    // negatives; a false positive costs one probe that misses).  The length mask of a two-byte prefix says which
    // lengths exist behind " t" -- all of them; the filter says which exist behind " tzq" -- none: 60 % of the
    // 5..8-byte probes of the substring tabulation (C2's missed chunks) are never issued.
    const PfxEnt* pfx;
    // (round 4: 16-bit entries -- the six length bits below, the t8 table's salt for keys with these four bytes above)
    const uint16_t* filt4;    uint32_t filt4_shift;
    const MemoEnt* memo;      uint32_t memo_mask;        // chunk memo (nullptr: off); slots - 1
    const MemoExt* memo_ext;
    const MemoEnt* memo2;     uint32_t memo2_mask;       // ... for chunks of 33..64 bytes (meta0's length field: bytes - 32)
    const MemoExt* memo2_ext; const MemoHi* memo2_hi;
};
SPL_HD uint32_t hash_f4(uint32_t w0) { return w0 * 0x9E3779B1u; }      // (index = the upper bits: >> filt4_shift)

// ----------------------------------------------------------------------------------------
// Hashes (host builder and device probes use these very functions).
// ----------------------------------------------------------------------------------------
SPL_HD uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
// `salt` (tiny, t8 and short tables): a byte the table builder chose for all keys that begin with the
// same two bytes so that none of them lands in a bucket that is full (DeviceTables::len_mask).
SPL_HD uint32_t hash_t8(uint32_t k0, uint32_t k1, uint32_t len, uint32_t salt = 0) {
    return mix32(k0 * 0x9E3779B1u ^ (k1 * 0x85EBCA77u + 0x165667B1u) ^ (len * 0x27D4EB2Fu) ^ (salt * 0xC2B2AE3Du));
}
// (the t8 hash with no second word: a probe whose lanes hold keys of both tables computes ONE hash)
SPL_HD uint32_t hash_tiny(uint32_t k0, uint32_t len, uint32_t salt = 0) { return hash_t8(k0, 0u, len, salt); }
SPL_HD uint32_t hash_p8(uint32_t k0, uint32_t k1) { return hash_t8(k0, k1, 9u); }
SPL_HD uint32_t p8_tag(uint32_t k0, uint32_t k1) {                                   // 1 .. 0xFFFFFE
    return mix32(k0 * 0x85EBCA77u ^ (k1 * 0xC2B2AE3Du + 0x27D4EB2Fu)) % 0xFFFFFEu + 1u;
}
SPL_HD uint32_t p8_match(uint32_t e0, uint32_t e1, uint32_t tag) {                   // 0: no longer token starts so
    const uint32_t t0 = e0 >> 8, t1 = e1 >> 8;
    uint32_t l = 0;
    if (e0 != 0 && (t0 == tag || t0 == 0xFFFFFFu)) l = e0 & 0xFFu;
    if (e1 != 0 && (t1 == tag || t1 == 0xFFFFFFu) && (e1 & 0xFFu) > l) l = e1 & 0xFFu;
    return l;
}
SPL_HD uint32_t hash_short(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t len, uint32_t salt = 0) {
    uint32_t h = k0 * 0x9E3779B1u ^ (k1 * 0x85EBCA77u + 0x165667B1u) ^ (k2 * 0xC2B2AE3Du) ^ (len * 0x27D4EB2Fu) ^ (salt * 0x165667B1u);
    return mix32(h);
}
// long keys: word-at-a-time over the zero-padded little-endian words of the key
SPL_HD uint32_t hash_long_step(uint32_t h, uint32_t w) { h = (h ^ w) * 0x9E3779B1u; return (h << 13) | (h >> 19); }
SPL_HD uint32_t hash_long_fin(uint32_t h, uint32_t len) { return mix32(h ^ len); }
SPL_HD uint32_t hash_long_tag(uint32_t h) { return mix32(h * 0x85EBCA77u + 0x3C6EF372u); }
template <int KW> SPL_HD uint32_t hash_memo_w(const uint32_t (&k)[KW], uint32_t n) {
    uint32_t h = n * 0x27D4EB2Fu + 0x165667B1u;
    for (int i = 0; i < KW; i++) { h = (h ^ k[i]) * 0x9E3779B1u; h = (h << 13) | (h >> 19); }
    return mix32(h);
}
SPL_HD uint32_t hash_memo(const uint32_t (&k)[8], uint32_t n) { return hash_memo_w<8>(k, n); }
// (a chunk has TWO candidate slots: the second one is tried where the first is taken by another chunk)
SPL_HD uint32_t memo_slot2(uint32_t h, uint32_t mask) { const uint32_t g = ((h >> 17) | (h << 15)) * 0x85EBCA77u; return ((g ^ (g >> 15)) & mask) ^ 1u; }
SPL_HD uint32_t hash_pair(uint32_t l, uint32_t r) { return mix32(l * 0x9E3779B1u + r * 0x85EBCA77u + 0x27D4EB2Fu); }
SPL_HD uint64_t pair_key(uint32_t l, uint32_t r) { return (uint64_t)l | ((uint64_t)r << SPL_ID_BITS); }
constexpr uint64_t SPL_PAIR_KEY_MASK = (1ull << (2 * SPL_ID_BITS)) - 1;

}  // namespace spl
