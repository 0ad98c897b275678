/* splintr_hip.h -- C ABI of the MI355X (gfx950) batch BPE encoder.
 *
 * The reference (ml-rust/splintr v0.8.0) has no C ABI: its only boundary is the PyO3 class
 * `Tokenizer` (src/python/bindings.rs:57-446) over `core::Tokenizer` (src/core/tokenizer.rs).
 * Each entry point below names the reference interface it stands in for; INTEGRATION.md shows the
 * binding a maintainer would add on the reference side (a `mod hip` FFI block in Rust, or the
 * ctypes class this repo ships as splintr_amd.Tokenizer).
 *
 * Conventions: plain pointers and sizes, no exceptions, no aborts.  Functions returning int
 * return SPL_OK (0) or a negative SPL_E* code; spl_last_error() gives the thread-local message.
 * Text is UTF-8, documents are packed back to back: doc d = utf8[doc_off[d] .. doc_off[d+1]).
 * Results are CSR: ids[T] (u32) + out_off[n_docs+1] (u64), in document order -- the flattened
 * form of the reference's Vec<Vec<u32>>.  A handle is bound to one GPU (spl_set_devices: to
 * several, for the host entry point); one batch in flight per handle; different handles may be
 * used from different threads.
 *
 * Text that is not valid UTF-8 (the reference's core only ever sees &str, so it defines nothing
 * here): every byte that is not part of a well-formed sequence -- a stray continuation byte, a lead
 * byte with too few continuation bytes behind it -- is ONE character of the class "other"
 * ([^\s\p{L}\p{N}]); a lead byte takes the continuation bytes actually present (at most as many as
 * it announces) and the decoded value is looked up as it is (overlong forms and surrogates are not
 * rejected).  Nothing is dropped or replaced: the ids always decode back to the input bytes
 * (tests/test_gpu_parity.py::test_invalid_utf8_policy pins this against the oracle).
 */
#ifndef SPLINTR_HIP_H
#define SPLINTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPL_OK 0
#define SPL_EINVAL (-1)   /* bad argument / malformed table blob */
#define SPL_EDEVICE (-2)  /* HIP runtime error (no GPU, out of memory, launch failure) */
#define SPL_ECAPACITY (-3)/* caller-provided output buffer too small */

/* Split pattern ids: CL100K_BASE_PATTERN (src/core/tokenizer.rs:39), O200K_BASE_PATTERN (:42;
 * also LLAMA3_PATTERN :45 and deepseek_v3, src/python/bindings.rs:116-129). */
#define SPL_PATTERN_CL100K 0
#define SPL_PATTERN_O200K 1
/* MISTRAL_V3_PATTERN (src/core/tokenizer.rs:64; mistral_v3 / Tekken, src/python/bindings.rs:152-158) */
#define SPL_PATTERN_MISTRAL_V3 2
/* These three are the patterns the GPU scanner implements as closed forms.  ANY OTHER pattern (Tokenizer::new compiles whatever it
 * is given, src/core/tokenizer.rs:410-456): SPL_PATTERN_CUSTOM with the pattern text in spl_opts.  It is compiled by csrc/spl_regex.h
 * (a backtracking matcher over the same code-point class table: literals, classes, \s \d \w, every general category and -- with a
 * version-3 class table -- every SCRIPT as \p{..} (\p{Han}, \p{Hiragana}, \p{Latin} ...), groups, (?i:), (?>), alternation, greedy /
 * lazy / possessive quantifiers, look-ahead, ^ $ \A \Z \z \b \B -- upstream tiktoken's cl100k_base / o200k_base strings, Qwen2's and
 * GPT-2's are accepted as they are) and its split runs ON THE GPU (csrc/spl_rx_split.h: the same matcher program at every text position,
 * the walk from each document's start by pointer doubling) in front of the same probe / merge kernels; documents that hold something the
 * device matcher gives up on -- a match longer than ~1 KB -- are split on the host cores instead, one by one (spl_device_split_fallbacks).
 * What the matcher cannot express (binary properties, script extensions, look-behind, back-references, \p{Lu} under (?i), a pattern that
 * can match the empty string) is refused by spl_create with the construct named. */
#define SPL_PATTERN_CUSTOM 3

/* spl_opts.flags */
#define SPL_OPT_BYTE_LEVEL 1u /* Tokenizer::from_bytes_byte_level (src/core/tokenizer.rs:562-569): the
                                 vocabulary's keys are ByteLevel text (src/core/byte_level.rs:46-74) */

/* encode flags */
#define SPL_WITH_SPECIAL 1u /* encode_with_special semantics (src/core/tokenizer.rs:842-874) */
#define SPL_INTRA_DOC 2u    /* encode_rayon (tokenizer.rs:815-837): accepted, no effect -- the GPU
                               path is always chunk-parallel inside a document, output identical */

typedef struct spl_tokenizer spl_tokenizer;
typedef struct spl_result spl_result;

typedef struct spl_opts {
    uint32_t struct_size; /* sizeof(spl_opts) as the CALLER was compiled: fields beyond it read as 0, so the struct
                             can grow without breaking callers built against an earlier header (ABI versioning) */
    int32_t pattern;      /* SPL_PATTERN_* */
    int32_t device;       /* HIP device ordinal */
    uint32_t flags;       /* SPL_OPT_* */
    const char* pattern_text; /* SPL_PATTERN_CUSTOM: the split pattern, UTF-8, pattern_len bytes (no terminator needed) */
    uint64_t pattern_len;
} spl_opts;

/* Thread-local text of the last failure in this thread. */
const char* spl_last_error(void);

/* Number of visible HIP devices (0 if none / no driver). */
int spl_device_count(void);

/* Tokenizer::from_bytes / from_bytes_byte_level (src/core/tokenizer.rs:552-569) + with_full_options
 * (:410-456): parse the vocabulary and the code-point class table (tools/gen_unicode_tables.py),
 * build the lookup tables and upload them to the device.  Returns NULL on failure.
 * `vocab` is either the reference's on-disk format -- tiktoken text, one `base64(token) rank` per
 * line, parsed as load_tiktoken_bpe does (src/core/vocab.rs:57-89: last space separates, rank
 * trimmed, a later duplicate key replaces the earlier one) -- or this repo's packed SPLV container
 * (tools/pack_vocab.py), told apart by the container's magic.
 * Restrictions (refused with SPL_EINVAL): ids must be < 2^21; two different keys must not share an id; a ByteLevel vocabulary must hold
 * all 256 alphabet characters, each ranking below every longer token.  A vocabulary that LACKS single bytes is taken as the reference
 * takes it (src/core/bpe.rs:73-75, 99-111, 182-191: pairs are ranked by their concatenated bytes, a node whose bytes are no token is
 * dropped from the result): the missing bytes get pseudo ids behind the vocabulary's for the merge loops and are never emitted.
 * Keys of up to 8 bytes live in single-slot tables built by hash-and-displace: keys that share a two-byte prefix (1..4-byte keys, 16-bit
 * salt) or a four-byte-prefix filter slot (5..8-byte keys, 10-bit salt) share a salt; a table in which some group finds no salt is
 * doubled, up to 2^24 slots (4 000 keys under one two-byte prefix: 2^20 slots).  Only a group of more than about ten thousand keys is
 * refused (SPL_EINVAL, "could not give every key of the ... table a slot of its own"); the reference's hash map has no
 * such limit. */
spl_tokenizer* spl_create(const void* vocab, size_t vocab_len, const void* uclass_tab, size_t uclass_len,
                          const spl_opts* opts);

/* The GPUs spl_encode_batch spreads a host batch over (the degenerate form of the multi-GPU path:
 * one process, documents sharded by bytes, every GPU copies its part of the CSR straight into the
 * one pinned result).  Replaces the handle's device list; the tables are uploaded to each.  The same
 * ordinal may be listed more than once (independent pipelines on one GPU).  The device-pointer
 * entry points keep using the first device of the list. */
int spl_set_devices(spl_tokenizer* t, const int32_t* devices, uint32_t n);
uint32_t spl_n_devices(const spl_tokenizer* t);

/* Tuning switches (name, value; measurements behind each default: DESIGN.md section 5 and profiles/).
 *   "chunk_bytes"            upper bound of one pipeline chunk of spl_encode_batch (default 5 MiB)
 *   "single_chunk_max_bytes" batches up to this size run as ONE chunk (default 4 MiB)
 *   "result_estimate_div"    first guess of the token count = bytes / div (default 2; a result that outgrows it moves to a larger buffer)
 *   "subdoc_split"           0/1 (1): balance the GPUs by cutting large documents at context-free boundaries
 *   "direct_write"           0/1 (1): one-chunk batches -- the last kernel writes the ids straight into the pinned result
 *   "direct_read"            0/1 (1): one-chunk batches whose text comes from spl_host_alloc are read where they lie (no H2D copy)
 *   "device_split"           0/1 (1): a custom split pattern's split runs on the GPU (spl_split_device); 0 keeps it on the host cores
 *   "small_path"             0/1 (1): batches of at most 4 KB and 256 documents take the latency path (spl_small_path_calls)
 *   "memo"                   0/1 (1): the chunk memo (spl_memo_stats); "memo_bits" 4..22 (20): log2 of its entries for chunks of up to 32 bytes
 *                            (128 bytes each); "memo_long_bits" 0..20 (16; 0: none): ... for chunks of 33..64 bytes (164 bytes each);
 *                            "memo_log_cap" 1..65536 (1024): missed chunks the tiles log per region (of 64) between two fills
 *   "memo_clear"             (any value) empties the memo of every context: Tokenizer::clear_cache (src/core/tokenizer.rs:995-1000)
 *   "group_scan_min"         0..2^24 (256; 0: never): a batch of more than this many groups of 64 tiles gets the groups' prefix sums from one small launch
 *                            (k_group_scan) between k_pretok and k_tile_out instead of every tile adding up the sums of the groups in front of it
 *   "range_tiles"            0..2^24 (0: one launch pair): a device-resident batch of more than 1.25 x this many tiles goes out as ranges of its tiles --
 *                            k_pretok and k_tile_out per range -- alternating between the caller's stream and a second one ("range_streams" 1/2 (2))
 *   "fuse"                   0/1 (1): batches of up to "fuse_max_tiles" tiles (default and maximum 1536: about 1.2 MB) are ONE launch --
 *                            every tile learns the number of tokens in front of it from the other tiles' published counts and writes its
 *                            part of the CSR itself; 0: the tile kernel and k_tile_out, as for larger batches
 *   "twin_streams"           0/1 (1): the kernels of consecutive pipeline chunks run on two compute streams with a workspace each
 *   "pick_streams"           0/1 (1): the pipeline's copy streams and second compute stream are chosen by measurement at first use
 *                            (spl_pick_stream) so that they run side by side; costs 5-80 ms once per context
 *   "copy_threads"           1..64 (4): threads that copy a chunk of pageable text into pinned staging
 *   "chunk_ramp"             0/1 (0): a lane's first and last chunk a quarter of the others
 *   "sdma_d2h"               0/1 (0): the ids of a pipeline chunk leave through hsa_amd_memory_async_copy (an SDMA engine) instead of
 *                            hipMemcpyAsync; hipMemcpyAsync where the HSA runtime cannot be bound
 *   "decode_chunk_ids"       >= 1024 (2^21): spl_decode_batch pipelines batches of at least three such chunks through two slots
 *   "slab_pack24"            0/1 (0): the ids of the all-gather slabs travel three bytes each; every rank alike; refused (SPL_EINVAL) when an
 *                            id of the tokenizer -- vocabulary or special token -- does not fit 24 bits
 * Unknown names and values out of range: SPL_EINVAL. */
int spl_set_option(spl_tokenizer* t, const char* name, int64_t value);

/* One entry of the special_tokens map (src/core/tokenizer.rs:304, 429-434).  Call before the first
 * encode.  Literals are non-empty and at most 255 bytes; adding a literal again replaces its id.
 * Matching follows the reference's matcher (Aho-Corasick, MatchKind::Standard, non-overlapping
 * find_iter, tokenizer.rs:849-869): from the end of the previous match, the occurrence that ends
 * first, the longest one on a tie.  Sets in which no two occurrences can overlap (no literal
 * contains another, no proper suffix of one is a prefix of another or of itself, all literals <= 32
 * bytes -- every pretrained table) take a one-launch scan in which every occurrence is a match; any
 * other set takes a two-launch matcher (candidate ends, then a per-document walk). */
int spl_add_special(spl_tokenizer* t, const uint8_t* literal, size_t len, uint32_t id);

/* Tokenizer::vocab_size (src/core/tokenizer.rs:964-972): max id over vocab and specials, plus 1. */
uint32_t spl_vocab_size(const spl_tokenizer* t);

void spl_destroy(spl_tokenizer* t);

/* Pre-size the device workspace for batches of up to max_bytes / max_docs, so that later encode
 * calls neither allocate nor synchronise. */
int spl_reserve(spl_tokenizer* t, uint64_t max_bytes, uint64_t max_docs);

/* Tokenizer::encode_batch / encode_batch_with_special (src/core/tokenizer.rs:932-942) on HOST
 * buffers -- what PyTokenizer::encode_batch (src/python/bindings.rs:337-339) calls.  The batch is
 * cut into chunks of whole documents that flow through pinned staging -> H2D -> kernels -> D2H on
 * private streams, three chunks in flight per GPU, and over the GPUs of spl_set_devices.  `utf8`
 * may be pageable (it is copied through pinned staging) or come from spl_host_alloc (DMA reads it
 * directly).  *out lives in pinned memory owned by the library; release with spl_result_free (the
 * buffers are recycled; a result stays valid after spl_destroy). */
int spl_encode_batch(spl_tokenizer* t, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs,
                     uint32_t flags, spl_result** out);
/* The latency path (Tokenizer::encode of ONE text, src/core/tokenizer.rs:729-808 -- "~50 MB/s", :266-268): a batch of at most 4096 bytes and
 * 256 documents on a handle with a built-in pattern does not go through the chunk pipeline.  The CPU copies text and offsets into one small
 * pinned buffer, the tile kernel reads them there (a few cache lines over PCIe) and writes ids and offsets into the pinned result; the last
 * tile to finish stores, behind a system-scope fence, a completion word the calling thread spins on: ONE launch ("fuse"), no copy engine,
 * no stream synchronisation.  spl_set_option("small_path", 0) turns the path off; spl_small_path_calls counts the calls that took it.
 * (Latencies per text size: DESIGN.md section 7.) */
uint64_t spl_small_path_calls(const spl_tokenizer* t);
const uint32_t* spl_result_tokens(const spl_result* r);   /* ids[T] */
const uint64_t* spl_result_offsets(const spl_result* r);  /* out_off[n_docs+1] */
uint64_t spl_result_n_tokens(const spl_result* r);
uint64_t spl_result_n_docs(const spl_result* r);
void spl_result_free(spl_result* r);

/* Pinned (page-locked, all-device) host memory for callers that build the packed corpus themselves. */
void* spl_host_alloc(size_t bytes);
void spl_host_free(void* p);

/* Same path with everything resident in HBM (device pointers).  Fully asynchronous on `hip_stream`
 * (a hipStream_t; NULL = the default stream) once spl_reserve has sized the workspace.
 *   d_utf8[n_bytes]       corpus, 16-byte aligned (SPL_EINVAL otherwise) and readable up to the next
 *                         multiple of 16; doc offsets d_doc_off[n_docs+1] with d_doc_off[0]==0,
 *                         d_doc_off[n_docs]==n_bytes
 *   d_ids[ids_capacity]   output ids; n_bytes entries always suffice
 *   d_out_off[n_docs+1]   output offsets; d_out_off[n_docs] is the total token count
 * Tokens beyond ids_capacity are dropped (compare d_out_off[n_docs] with the capacity).
 * Size limits of ONE device call (SPL_EINVAL beyond them; spl_encode_batch on host buffers has none, it feeds
 * chunks of at most 5 MiB): n_bytes < 2^31 - 65536 (2047 MiB) without SPL_WITH_SPECIAL, n_bytes <= 256 MB with it and for handles with
 * SPL_PATTERN_CUSTOM (the special-token scan and the device splitter exist for the two-launch mode only).  Split a larger corpus at
 * document boundaries.
 * What is asynchronous: a handle with one of the three built-in patterns enqueues its kernels on `hip_stream` and returns -- no host
 * synchronisation, with or without SPL_WITH_SPECIAL.  A handle with SPL_PATTERN_CUSTOM synchronises `hip_stream` ONCE before it returns
 * (to read which documents, if any, the device splitter gave up on; see below).  All calls on one handle -- this one, spl_split_device,
 * spl_encode_chunks_device, spl_encode_batch -- share the handle's workspace: they must be stream-ordered (issued on the same stream, or
 * with the earlier one complete); different handles are independent. */
int spl_encode_batch_device(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                            uint64_t n_docs, uint32_t flags, uint32_t* d_ids, uint64_t ids_capacity,
                            uint64_t* d_out_off, void* hip_stream);

/* Handles with SPL_PATTERN_CUSTOM: spl_encode_batch / spl_decode_batch work as for every other handle (the split of each pipeline chunk
 * runs on the device splitter, csrc/spl_rx_split.h, behind the GPU's special-token scan with SPL_WITH_SPECIAL).  What the device matcher
 * gives up on is handled per DOCUMENT: the 256-byte blocks of such positions come back in a small pinned list, the documents they lie in
 * are split on the host cores by the calling (producer) thread -- while the previous chunk's tile kernel runs -- and their stretch of the
 * two bitmaps is patched before the tile kernel reads it; every other document keeps the device split.  Only a list that overflows (more
 * than 1024 such blocks in one chunk: a pattern that gives up everywhere) sends the whole batch through the host splitter.
 * spl_encode_batch_device / _packed run splitter and tile kernel in one go and synchronise `hip_stream` ONCE before they return; if
 * documents need the host, THEIR text (nothing else) is copied back, split, patched, and the tile kernel runs again on the patched bitmaps.
 * The halves are available separately:
 *   spl_split_host          the matches of the handle's pattern over a packed HOST corpus as two bitmaps of
 *                           n_bytes / 32 + 2 words each (zeroed here): bit p of start_bits -- a chunk, or a stretch of
 *                           bytes no match covers, starts at byte p; bit p of gap_bits -- byte p is dropped
 *                           (find_iter semantics, tokenizer.rs:729-808).  Multi-threaded over documents.
 *   spl_encode_chunks_device  spl_encode_batch_device with the chunk boundaries GIVEN (device copies of the two
 *                           bitmaps; any handle: the handle's own pattern is not consulted).  Up to 256 MB per call. */
int spl_split_host(spl_tokenizer* t, const uint8_t* utf8, const uint64_t* doc_off, uint64_t n_docs, uint32_t* start_bits,
                   uint32_t* gap_bits);
/*   spl_split_device        the same two bitmaps for a packed corpus that is in HBM, by the device splitter (the same matcher
 *                           program, one text position per lane, then the walk from each document's start by pointer doubling:
 *                           csrc/spl_rx_split.h), asynchronously on `hip_stream`.  Bitmaps of n_bytes / 32 + 2 words (zeroed
 *                           here); *d_status (one word, device memory, cleared by the CALLER -- calls may share it) is non-zero
 *                           afterwards if the text held something the device matcher gives up on (a match or look-ahead
 *                           reaching more than ~1 KB beyond its start, a runaway attempt): the bitmaps are then incomplete
 *                           and the split belongs to spl_split_host.  SPL_EINVAL if the pattern's program does not fit the
 *                           device matcher.  spl_encode_batch uses it for every batch and falls back by itself, document by document
 *                           (spl_set_option "device_split" 0 keeps the split on the host cores; spl_device_split_fallbacks counts the
 *                           DOCUMENTS the host split instead -- all of a batch's if the whole batch went there).  Up to 256 MB per call.
 *                           Shares the handle's splitter workspace with the encode calls: stream-ordered with them (see above). */
int spl_split_device(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off, uint64_t n_docs,
                     uint32_t* d_start_bits, uint32_t* d_gap_bits, uint32_t* d_status, void* hip_stream);
uint64_t spl_device_split_fallbacks(const spl_tokenizer* t);
int spl_encode_chunks_device(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                             uint64_t n_docs, const uint32_t* d_start_bits, const uint32_t* d_gap_bits, uint32_t* d_ids,
                             uint64_t ids_capacity, uint64_t* d_out_off, void* hip_stream);

/* Ragged all-gather of the CSR result across the GPUs of a node (north_star: "RCCL all-gatherv
 * over xGMI"; the reference has nothing distributed).  RCCL has no all-gatherv, so every rank
 * packs {T, N, local offsets, ids} into a fixed-capacity slab of u32 words, ONE all-gather of
 * equal-sized slabs moves them (spl_allgather_slabs below, or the caller's own collective, e.g.
 * torch.distributed all_gather_into_tensor), and every rank unpacks the `world` slabs into the global CSR in
 * rank order.  Both kernels are asynchronous on `hip_stream`; no host synchronisation is needed
 * because the counts travel inside the slabs.
 *   slab: [0] T, [1] N, [2 .. 2+max_docs] local out_off, then ids; cap_words >= max_docs + 4.
 *   d_status[0] is set to 1 by the unpacker if any rank's ids exceeded its slab (re-run larger). */
int spl_gatherv_pack(spl_tokenizer* t, const uint32_t* d_ids, const uint64_t* d_out_off, uint64_t n_docs,
                     uint32_t* d_slab, uint64_t cap_words, uint64_t max_docs, void* hip_stream);
int spl_gatherv_unpack(spl_tokenizer* t, const uint32_t* d_slabs, uint32_t world, uint64_t cap_words,
                       uint64_t max_docs, uint32_t* d_all_ids, uint64_t all_ids_cap, uint64_t* d_all_off,
                       uint32_t* d_status, void* hip_stream);
/* spl_encode_batch_device that ALSO leaves the result in slab form (the layout spl_gatherv_pack
 * makes) in d_slab[cap_words]: in tile-owned mode the last kernel writes both copies in one pass,
 * so the per-batch pack launch of a multi-GPU pipeline goes away; otherwise the pack kernel is
 * queued behind the encode. */
int spl_encode_batch_device_packed(spl_tokenizer* t, const uint8_t* d_utf8, uint64_t n_bytes, const uint64_t* d_doc_off,
                                   uint64_t n_docs, uint32_t flags, uint32_t* d_ids, uint64_t ids_capacity,
                                   uint64_t* d_out_off, uint32_t* d_slab, uint64_t cap_words, uint64_t max_docs,
                                   void* hip_stream);
/* Bucketed form (fewer, larger collectives: xGMI rings are per-link bound and a collective has a
 * fixed launch cost): every rank packs up to `depth` consecutive batches into `depth` slabs laid
 * back to back, ONE all-gather moves world x depth slabs, and one launch unpacks the first
 * n_batches of them: batch j's global CSR goes to d_all_ids + j * all_ids_cap and
 * d_all_off + j * off_stride. */
int spl_gatherv_unpack_group(spl_tokenizer* t, const uint32_t* d_slabs, uint32_t world, uint32_t depth, uint32_t n_batches,
                             uint64_t cap_words, uint64_t max_docs, uint32_t* d_all_ids, uint64_t all_ids_cap,
                             uint64_t* d_all_off, uint64_t off_stride, uint32_t* d_status, void* hip_stream);

/* ONE batch exchanged in WAVES (strong scaling, pipelined: the ids of wave k travel while wave k + 1 encodes).  The batch's documents, in
 * their order, are cut into waves and every wave into one contiguous slice per rank; rank r encodes its slice of wave k into a slab
 * (spl_encode_batch_device_packed), one all-gather of equal slabs moves the wave (spl_allgather_slabs / _p2p), and this call unpacks the
 * `world` slabs BEHIND what the waves before it left: d_run[0] tokens and d_run[1] documents (device memory, zeroed by the caller before wave
 * 0, advanced here -- in stream order, so no host synchronisation sits between the waves).  After the last wave d_all_ids / d_all_off hold
 * the CSR of the whole batch in document order and d_run its totals.  d_status[0] = 1 if a slab or a result buffer was too small. */
int spl_gatherv_unpack_at(spl_tokenizer* t, const uint32_t* d_slabs, uint32_t world, uint64_t cap_words, uint64_t max_docs,
                          uint32_t* d_all_ids, uint64_t all_ids_cap, uint64_t* d_all_off, uint64_t all_off_cap, uint64_t* d_run,
                          uint32_t* d_status, void* hip_stream);

/* The collective itself behind the C ABI (north_star: "an RCCL all-gatherv over xGMI to reassemble the ragged
 * token-id output"; the reference has nothing distributed -- src/core/tokenizer.rs:932-934 is a Rayon par_iter on
 * one host -- so these are this build's own entry points).  One process per GPU.  RCCL is bound at run time
 * (dlopen of librccl.so.1: a process that already holds a copy, e.g. PyTorch's, keeps using that one; a
 * single-GPU caller needs no RCCL).
 *   spl_comm_unique_id   one rank makes the 128-byte id (ncclGetUniqueId); the caller hands it to the others
 *                        (MPI, a file, a key-value store -- no transport is imposed)
 *   spl_comm_create      ncclCommInitRank on `device`; collective: every rank of `world` must call it
 *   spl_allgather_slabs  ONE ncclAllGather of equal-sized u32 slabs (the padded, synchronisation-free form:
 *                        slabs from spl_gatherv_pack / spl_encode_batch_device_packed, results through
 *                        spl_gatherv_unpack[_group]); asynchronous on hip_stream
 *   spl_allgatherv_csr   the exact form: every rank's {T, N} and the capacities of its result buffers first
 *                        (32 bytes per rank, ONE host synchronisation), then exactly T_r ids and N_r offsets per rank land at their place of
 *                        the global CSR by grouped ncclSend / ncclRecv -- one message per peer and direction,
 *                        every xGMI link busy at once, nothing padded -- and the offsets are rebased on the
 *                        device.  d_all_ids[all_ids_cap], d_all_off[all_off_cap >= N_total + 1]; the totals are
 *                        returned; SPL_ECAPACITY if they do not fit the SMALLEST buffers any rank passed -- decided
 *                        from the gathered capacities, so every rank returns it alike and none is left waiting
 *                        in the exchange.  Rank order == document order when rank r holds the r-th contiguous
 *                        shard. */
/* A new HIP stream on `device` that really runs BESIDE the given ones (an exchange stream beside the encoder's, a second encode stream
 * beside the first): HIP maps streams to hardware queues and queues to the four pipes of the command processor by what else the process has
 * created -- two busy streams on one queue run one behind the other, on one pipe they take turns -- and no API tells which.  This call
 * measures it: up to twelve candidates over the three priorities, a 120 us spin kernel on a given stream and four empty kernels on the
 * candidate (and the other way round); the first candidate without a conflict, else the least bad one.  *conflict_us (may be NULL): what is
 * left, 0 when the streams run side by side.  Costs a few ms (synchronises the given streams).  The stream belongs to the caller
 * (hipStreamDestroy). */
int spl_pick_stream(int device, void* const* busy_hip_streams, uint32_t n_busy, void** hip_stream_out, double* conflict_us);

#define SPL_COMM_ID_BYTES 128
typedef struct spl_comm spl_comm;
int spl_comm_unique_id(uint8_t id_out[SPL_COMM_ID_BYTES]);
spl_comm* spl_comm_create(const uint8_t id[SPL_COMM_ID_BYTES], int rank, int world, int device);
void spl_comm_destroy(spl_comm* c);
int spl_comm_rank(const spl_comm* c);
int spl_comm_world(const spl_comm* c);
int spl_allgather_slabs(spl_comm* c, const uint32_t* d_send, uint32_t* d_recv, uint64_t words_per_rank, void* hip_stream);
/* the same exchange as grouped ncclSend / ncclRecv -- one message per peer and direction, each over its own xGMI link, no ring: the other
 * candidate of bench.py's start-up calibration (which of the two wins depends on the message size and on RCCL's algorithm choice) */
int spl_allgather_slabs_p2p(spl_comm* c, const uint32_t* d_send, uint32_t* d_recv, uint64_t words_per_rank, void* hip_stream);
int spl_allgatherv_csr(spl_comm* c, const uint32_t* d_ids, const uint64_t* d_out_off, uint64_t n_docs, uint32_t* d_all_ids,
                       uint64_t all_ids_cap, uint64_t* d_all_off, uint64_t all_off_cap, uint64_t* n_tokens_total,
                       uint64_t* n_docs_total, void* hip_stream);

/* Tokenizer::decode_bytes for a batch (src/core/tokenizer.rs:877-897, 944-958), HOST buffers:
 * ids CSR in, bytes CSR out.  An id of the vocabulary gives its token's bytes (ByteLevel: decoded
 * to raw bytes; a key that is not ByteLevel text gives the key itself, src/core/byte_level.rs:125-146),
 * otherwise an id of the special-token map gives its literal, any other id gives nothing.
 * *out_bytes / *out_off are library-owned (pinned host memory from the handle's pool: the copies back run at
 * PCIe speed); release each with spl_free. */
int spl_decode_batch(spl_tokenizer* t, const uint32_t* ids, const uint64_t* ids_off, uint64_t n_docs,
                     uint8_t** out_bytes, uint64_t** out_off);
void spl_free(void* p);

/* Host-side lookup of ONE id, for consumers that decode token by token on the CPU -- the streaming
 * decoders (src/core/streaming.rs, src/python/bindings.rs:465-834 take clones of Tokenizer::decoder
 * and special_tokens_decoder).  *bytes / *len = what decode_bytes emits for the id; the pointer
 * stays valid until spl_destroy or the next spl_add_special.  Returns 0 unknown id, 1 vocabulary
 * token, 2 vocabulary token of a ByteLevel vocabulary whose key is NOT ByteLevel text (the bytes
 * are the key itself), 3 special token. */
int spl_token_bytes(const spl_tokenizer* t, uint32_t id, const uint8_t** bytes, uint32_t* len);
/* 1 if the vocabulary's keys are ByteLevel text (from_bytes_byte_level / the container's flag). */
int spl_is_byte_level(const spl_tokenizer* t);

/* Per-kernel timing (HIP events on the launch stream).  While enabled every encode call records
 * events around each kernel; spl_profile_read returns the accumulated milliseconds and launch
 * counts per kernel since the last spl_profile_reset and synchronises the stream. */
#define SPL_MAX_KERNELS 16
int spl_profile_enable(spl_tokenizer* t, int on);
int spl_profile_reset(spl_tokenizer* t);
int spl_profile_read(spl_tokenizer* t, double ms_out[SPL_MAX_KERNELS], uint64_t launches_out[SPL_MAX_KERNELS]);
const char* spl_kernel_name(int index);   /* NULL past the last kernel */

/* The chunk memo of the handle's first context -- the GPU path's counterpart of the reference's LRU of encoded chunks
 * (src/core/tokenizer.rs:707-722; result-transparent there and here: keys are compared in full): out[0] fills run (k_memo_fill, between two
 * launches), out[1] chunks put in, out[2] chunks found to be beyond an entry (more than fourteen tokens: remembered as such), out[3] entries of
 * the table (0: off).  spl_set_option("memo", 0) turns it off; "memo_bits" (4..22, default 16) sizes it.  Synchronises the device. */
int spl_memo_stats(spl_tokenizer* t, uint64_t out[4]);

/* Counters of the last encode call on this handle (device -> host copy, synchronises):
 * [2] items of the global long-chunk queue (> 64 B, plus every miss of a deferred segment),
 * [3] deferred segments (scanner chains that outgrew a tile window); [0], [1] unused. */
int spl_last_queue_counts(spl_tokenizer* t, uint32_t counts_out[4]);

/* Development aid: when enabled (bit 0 of `enable`; -DSPL_DEBUG_STAMPS builds), one k_pretok workgroup
 * stamps the shader clock at its phase boundaries; the call returns the stamps of the previous batch
 * (synchronises).  Bits 1-3 force an execution mode for the tests (0 the size decides, 1 the small-tile geometry,
 * 4 queue mode, 5 tile-owned mode with the second geometry; 2 and 3 were the multi-pass pipeline, removed in round 4: SPL_EINVAL);
 * bits 4-6 cut the kernel off after a phase (profiling builds only). */
int spl_debug_phases(spl_tokenizer* t, int enable, unsigned long long stamps_out[16]);

/* Development aid: per-workgroup records of the last stamped k_pretok launch, 4 wall-clock ticks
 * each (start, end of the merge phase, counts done, end) for the first SPL_DEBUG_BLOCKS
 * workgroups.  Returns the count copied. */
#define SPL_DEBUG_BLOCKS 4096
int spl_debug_blocks(spl_tokenizer* t, unsigned long long* out, int max_blocks);

#ifdef __cplusplus
}
#endif
#endif /* SPLINTR_HIP_H */
