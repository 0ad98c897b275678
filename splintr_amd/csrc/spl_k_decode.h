// spl_k_decode.h -- part of spl_kernels.hip (included there, in this order; one translation unit): decode (k_decode_len / _scan / _copy / _docs), k_ext_specials, the gather-v slabs and the CSR counts / rebase of the collective.
#pragma once

namespace spl {

// ------------------------------------------------------------------------------------------
// decode_bytes (reference src/core/tokenizer.rs:877-897, batch form :945-958): gather token byte
// strings.  The id -> bytes table covers the vocabulary AND the special tokens (the reference looks
// an id up in `decoder` first, then in `special_tokens_decoder`; an id in neither contributes
// nothing).  Three launches, no host round trip in between:
//   k_decode_len    length of every id + sums per block of DEC_BLK ids
//   k_decode_scan   exclusive scan of the block sums (one workgroup)
//   k_decode_copy   offset of every id (block base + scan inside the block), byte copy, and the
//                   byte offset of every document (doc d starts at id ids_off[d])
constexpr int DEC_BLK = 1024;
struct DecodeArgs {
    const uint32_t* ids; uint64_t n_ids;
    const uint32_t* tok_off; const uint8_t* tok_bytes; uint32_t max_id;
    // special tokens whose ids lie beyond the vocabulary's largest id: sorted ids, byte spans sp_off[k] .. sp_off[k + 1]
    // of tok_bytes (a dense table up to the largest SPECIAL id would be O(that id): spl_add_special takes any id < 2^31)
    const uint32_t* sp_ids; const uint32_t* sp_off; uint32_t n_sp;
    uint64_t* blk;          // [n_blk + 1] block sums, then exclusive offsets (+ total)
    uint64_t* id_off;       // [n_ids + 1] byte offset of every id (+ total)
    uint8_t* out;
    const uint64_t* doc_first; uint64_t n_docs; uint64_t* doc_off;   // doc d = ids [doc_first[d] - doc_first[0], ...)
    uint64_t out_base;      // added to every document offset (a chunk of the host pipeline: where its bytes start in the whole result)
};
__device__ __forceinline__ uint32_t dec_span(const DecodeArgs& a, uint32_t id, uint32_t& off) {
    if (id <= a.max_id) { off = a.tok_off[id]; return a.tok_off[id + 1] - off; }
    uint32_t lo = 0, hi = a.n_sp;                        // (a few dozen entries at most; ids of real text never get here)
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.sp_ids[mid] < id) lo = mid + 1; else hi = mid; }
    if (lo < a.n_sp && a.sp_ids[lo] == id) { off = a.sp_off[lo]; return a.sp_off[lo + 1] - off; }
    off = 0;
    return 0u;
}
__device__ __forceinline__ uint32_t dec_len(const DecodeArgs& a, uint64_t i) {
    if (i >= a.n_ids) return 0u;
    uint32_t off;
    return dec_span(a, a.ids[i], off);
}
__global__ __launch_bounds__(NT) void k_decode_len(DecodeArgs a) {
    __shared__ uint32_t s_w[NT / 64];
    uint32_t sum = 0;
    const uint64_t base = (uint64_t)blockIdx.x * DEC_BLK;
    for (int k = 0; k < DEC_BLK / NT; k++) sum += dec_len(a, base + (uint64_t)k * NT + threadIdx.x);
    sum = wave_scan_incl(sum);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int w = 0; w < NT / 64; w++) t += s_w[w];
        a.blk[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_decode_scan(uint64_t* blk, uint64_t n_blk) {
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < n_blk; base += 1024) {
        const uint64_t i = base + tid;
        const uint64_t v = i < n_blk ? blk[i] : 0ull;
        uint64_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d); if ((tid & 63) >= d) x += y; }
        if ((tid & 63) == 63) s_w[tid >> 6] = x;
        __syncthreads();
        uint64_t pre = s_carry;
        for (int w = 0; w < (tid >> 6); w++) pre += s_w[w];
        if (i < n_blk) blk[i] = pre + x - v;
        __syncthreads();
        if (tid == 1023) s_carry = pre + x;
        __syncthreads();
    }
    if (tid == 0) blk[n_blk] = s_carry;
}
__global__ __launch_bounds__(NT) void k_decode_copy(DecodeArgs a) {
    __shared__ uint32_t s_w[NT / 64];
    const uint64_t base = (uint64_t)blockIdx.x * DEC_BLK;
    uint64_t run = a.blk[blockIdx.x];
    for (int k = 0; k < DEC_BLK / NT; k++) {
        const uint64_t i = base + (uint64_t)k * NT + threadIdx.x;
        const uint32_t len = dec_len(a, i);
        const uint32_t x = wave_scan_incl(len);
        __syncthreads();
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        uint64_t o = run + (x - len);
        uint32_t all = 0;
        for (int w = 0; w < NT / 64; w++) { if (w < (int)(threadIdx.x >> 6)) o += s_w[w]; all += s_w[w]; }
        if (i < a.n_ids) {
            a.id_off[i] = o;
            uint32_t soff;
            (void)dec_span(a, a.ids[i], soff);
            const uint8_t* src = a.tok_bytes + soff;
            for (uint32_t q = 0; q < len; q++) a.out[o + q] = src[q];
        }
        run += all;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.id_off[a.n_ids] = run;
}
__global__ void k_decode_docs(DecodeArgs a) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > a.n_docs) return;
    a.doc_off[d] = a.id_off[a.doc_first[d] - a.doc_first[0]] + a.out_base;
}

// External chunk boundaries with special tokens: the host splitter found the literals too; their ids go where
// k_special_scan would have put them (the tile that owns a literal's first byte takes it as its token).
__global__ void k_ext_specials(Batch b, const uint32_t* pos, const uint32_t* id, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pos[i];
    b.stage[p] = id[i];
    atomicOr(&b.tbits[p >> 5], 1u << (p & 31));
}

// ------------------------------------------------------------------------------------------
// Ragged all-gather support (multi-GPU reassembly of the CSR result).  RCCL has no all-gatherv:
// every rank packs {T, N, local offsets[N+1], ids[T]} into a fixed-capacity slab, ONE
// all_gather_into_tensor moves the slabs over xGMI, and every rank unpacks them into the global
// CSR.  No host synchronisation: the token counts travel inside the slabs.
//   slab (u32 words): [0] T  [1] N  [2 .. 2+max_docs] local out_off (N+1 used)  [2+max_docs+1 ..] ids
__global__ void k_gatherv_pack(const uint32_t* ids, const uint64_t* out_off, uint32_t n_docs, uint32_t* slab,
                               uint32_t cap_words, uint32_t max_docs, uint32_t p24) {
    const uint32_t T = (uint32_t)out_off[n_docs];
    const uint32_t ids_at = 3 + max_docs, ids_cap = slab_id_cap(cap_words - ids_at, p24 != 0u);
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i == 0) { slab[0] = T; slab[1] = n_docs; }
    for (uint32_t d = i; d <= n_docs; d += stride) slab[2 + d] = (uint32_t)out_off[d];
    const uint32_t ncopy = T < ids_cap ? T : ids_cap;        // T > ids_cap is reported by the unpacker
    for (uint32_t k = i; k < ncopy; k += stride) slab_put_id(slab + ids_at, k, ids[k], p24 != 0u);
}
// grid.y = source rank, grid.z = batch of the group (a rank sends `depth` slabs back to back per
// collective: rank_stride = depth * cap_words; batch j's slabs start at j * cap_words and its outputs
// at j * all_ids_cap / j * off_stride).  status[0] is set to 1 if any slab overflowed its id capacity.
__global__ void k_gatherv_unpack(const uint32_t* slabs_all, uint32_t world, uint32_t cap_words, uint32_t max_docs,
                                 uint32_t* all_ids_all, uint64_t all_ids_cap, uint64_t* all_off_all, uint32_t* status,
                                 uint64_t rank_stride, uint64_t off_stride, uint32_t p24) {
    const uint32_t r = blockIdx.y, j = blockIdx.z;
    const uint32_t* slabs = slabs_all + (size_t)j * cap_words;
    uint32_t* all_ids = all_ids_all + (size_t)j * all_ids_cap;
    uint64_t* all_off = all_off_all + (size_t)j * off_stride;
    const uint32_t ids_at = 3 + max_docs, ids_cap = slab_id_cap(cap_words - ids_at, p24 != 0u);
    uint64_t tbase = 0, dbase = 0;
    for (uint32_t q = 0; q < r; q++) { tbase += slabs[(size_t)q * rank_stride]; dbase += slabs[(size_t)q * rank_stride + 1]; }
    const uint32_t* slab = slabs + (size_t)r * rank_stride;
    const uint32_t T = slab[0], N = slab[1];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i == 0 && T > ids_cap) status[0] = 1;
    for (uint32_t d = i; d < N; d += stride) all_off[dbase + d] = tbase + slab[2 + d];
    if (r == world - 1 && i == 0) all_off[dbase + N] = tbase + T;
    const uint32_t ncopy = T < ids_cap ? T : ids_cap;
    for (uint32_t k = i; k < ncopy; k += stride)
        if (tbase + k < all_ids_cap) all_ids[tbase + k] = slab_get_id(slab + ids_at, k, p24 != 0u);
}

// The same for ONE batch that is exchanged in WAVES (strong scaling, pipelined: wave k is on the links while wave k + 1 encodes): the
// documents of the batch, in their order, are cut into waves, every wave into one contiguous slice per rank; wave k's slabs land BEHIND
// what the waves before it left -- run[0] tokens, run[1] documents, in device memory, advanced by k_gatherv_advance behind the unpack
// (stream order) -- so that all waves together are ONE CSR in document order and no host synchronisation sits between them.
__global__ void k_gatherv_unpack_at(const uint32_t* slabs, uint32_t world, uint32_t cap_words, uint32_t max_docs, uint32_t* all_ids,
                                    uint64_t all_ids_cap, uint64_t* all_off, uint64_t all_off_cap, const uint64_t* run, uint32_t* status, uint32_t p24) {
    const uint32_t r = blockIdx.y;
    const uint32_t ids_at = 3 + max_docs, ids_cap = slab_id_cap(cap_words - ids_at, p24 != 0u);
    uint64_t tbase = run[0], dbase = run[1];
    for (uint32_t q = 0; q < r; q++) { tbase += slabs[(size_t)q * cap_words]; dbase += slabs[(size_t)q * cap_words + 1]; }
    const uint32_t* slab = slabs + (size_t)r * cap_words;
    const uint32_t T = slab[0], N = slab[1];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    if (i == 0 && (T > ids_cap || tbase + T > all_ids_cap || dbase + N + 1 > all_off_cap)) status[0] = 1;
    for (uint32_t d = i; d < N; d += stride)
        if (dbase + d < all_off_cap) all_off[dbase + d] = tbase + slab[2 + d];
    if (r == world - 1 && i == 0 && dbase + N < all_off_cap) all_off[dbase + N] = tbase + T;      // the closing entry (the next wave's first)
    const uint32_t ncopy = T < ids_cap ? T : ids_cap;
    for (uint32_t k = i; k < ncopy; k += stride)
        if (tbase + k < all_ids_cap) all_ids[tbase + k] = slab_get_id(slab + ids_at, k, p24 != 0u);
}
__global__ void k_gatherv_advance(const uint32_t* slabs, uint32_t world, uint32_t cap_words, uint64_t* run) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint64_t t = 0, n = 0;
        for (uint32_t q = 0; q < world; q++) { t += slabs[(size_t)q * cap_words]; n += slabs[(size_t)q * cap_words + 1]; }
        run[0] += t; run[1] += n;
    }
}

// Exact ragged all-gather (spl_allgatherv_csr): every rank's {T, N} travel first, then exactly T ids and N
// offsets per rank land at their place of the global CSR by grouped send / recv.  These two kernels are the
// device side: the counts as the collective's input, and the received LOCAL offsets rebased by the tokens of the
// ranks before (+ the closing entry).
constexpr int COMM_MAX_WORLD = 64;
struct RankTable { uint64_t n_pre[COMM_MAX_WORLD + 1], t_pre[COMM_MAX_WORLD + 1]; };
__global__ void k_csr_counts(const uint64_t* out_off, uint64_t n_docs, uint64_t ids_cap, uint64_t off_cap, uint64_t* cnt) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { cnt[0] = out_off[n_docs]; cnt[1] = n_docs; cnt[2] = ids_cap; cnt[3] = off_cap; }
}
__global__ void k_rebase_offsets(uint64_t* all_off, RankTable tab, uint32_t world) {
    const uint32_t r = blockIdx.y;
    const uint64_t lo = tab.n_pre[r], hi = tab.n_pre[r + 1], add = tab.t_pre[r];
    for (uint64_t d = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; d < hi; d += (uint64_t)gridDim.x * blockDim.x) all_off[d] += add;
    if (r == world - 1 && blockIdx.x == 0 && threadIdx.x == 0) all_off[tab.n_pre[world]] = tab.t_pre[world];
}

}  // namespace spl
