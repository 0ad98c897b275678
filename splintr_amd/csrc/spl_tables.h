// spl_tables.h -- host-side construction of the packed lookup tables (pure C++17, no HIP).
//
// Takes the place of load_tiktoken_bpe + the FxHashMap encoder (reference
// src/core/vocab.rs:57-89, src/core/tokenizer.rs:302, 410-456): parses this repo's SPLV
// vocabulary container, re-keys ByteLevel vocabularies into raw-byte space
// (src/core/byte_level.rs:46-74; DESIGN.md "ByteLevel equivalence"), and builds the
// short-key / long-key / pair tables the kernels probe (spl_common.h).
#pragma once
#include <string>
#include <utility>
#include <vector>

#include "spl_common.h"

namespace spl {

struct HostTables {
    // code-point classes
    std::vector<uint16_t> ucls_stage1;
    std::vector<uint8_t> ucls_stage2;
    std::vector<uint16_t> gc_stage1;      // general categories (class table version 2; empty otherwise): host splitter only
    std::vector<uint8_t> gc_stage2;
    // the SCRIPT property (class table version 3; empty otherwise): inclusive code-point ranges per script name, host splitter only
    struct Script { std::string name; std::vector<std::pair<uint32_t, uint32_t>> ranges; };
    std::vector<Script> scripts;
    uint32_t ucls_shift = 7;
    bool cjk_fast = false;
    // vocabulary
    std::vector<ShortEnt> short_tab;     // keys of 9..12 bytes
    std::vector<uint32_t> tiny_tab;      // keys of 1..4 bytes: 2 words per entry, ONE entry per slot (+ 4 words of padding)
    std::vector<uint32_t> t8_tab;        // keys of 5..8 bytes: 3 words per entry, ONE entry per slot (+ 4 words of padding)
    std::vector<LongEnt> long_tab;
    std::vector<uint8_t> key_blob;
    std::vector<uint64_t> pair_tab;
    std::vector<uint32_t> byte_id;      // 256
    std::vector<uint32_t> p8_tab;       // DeviceTables::p8_tab (two words per bucket)
    std::vector<uint16_t> len_mask;     // DeviceTables::len_mask (65536): length mask | salt << 8
    std::vector<PfxEnt> pfx;            // DeviceTables::pfx (65536): len_mask's entry + the id of the two-byte token
    std::vector<uint16_t> filt4;        // DeviceTables::filt4: token lengths by (hashed) four-byte prefix | t8 salt << 6
    uint32_t filt4_shift = 0;
    uint32_t unsalted_groups = 0;       // short table: two-byte key prefixes for which no salt kept every bucket below full (0 for the shipped vocabularies)
    uint32_t tiny_free = 0, t8_free = 0;
    uint32_t max_key_len = 0;           // in the key space the kernels see (raw bytes)
    uint32_t n_keys = 0, n_pairs = 0;
    uint32_t max_id = 0;
    bool all_bytes = false;
    uint32_t id_limit = 0xFFFFFFFFu;     // ids from here on are pseudo ids of single bytes the vocabulary lacks: never emitted
    bool byte_level = false;
    int pattern = PAT_CL100K;
    // decoder side: id -> raw bytes (CSR); ids with no entry have empty spans
    std::vector<uint32_t> tok_off;      // max_id + 2
    std::vector<uint8_t> tok_bytes;
    std::vector<uint8_t> tok_present;   // max_id + 1: 0 no such id, 1 a token, 2 (ByteLevel) a token whose key is emitted verbatim
};

// Returns 0 on success; on failure fills err.
// `vocab` is either this repo's SPLV container or the reference's tiktoken text (autodetected by the
// magic); force_byte_level = the reference's from_bytes_byte_level (tokenizer.rs:562-569).
int build_tables(const uint8_t* vocab, size_t vocab_len, const uint8_t* ucls, size_t ucls_len, int pattern,
                 bool force_byte_level, HostTables& out, std::string& err);

// Host mirror of the device two-stage lookup (used by build checks and tests/hostsim).
uint32_t host_cp_class(const HostTables& t, uint32_t cp);
// General category code of a code point (tools/gen_unicode_tables.py GC_NAMES: 0 Cn, 1 Lu, 2 Ll, ...); 0 without the table.
uint32_t host_cp_category(const HostTables& t, uint32_t cp);

}  // namespace spl
