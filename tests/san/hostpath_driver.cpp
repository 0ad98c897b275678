// hostpath_driver.cpp -- TEST TOOL (GPU box): the host pipeline of spl_encode_batch under ThreadSanitizer.
// The library's host side is rebuilt with -fsanitize=thread (tools/build_sanitizers.sh); this driver runs
//   * a handle with TWO pipelines on one GPU (spl_set_devices {0, 0}: a producer thread per lane plus the placing
//     thread) in many small chunks,
//   * a handle with a custom split pattern (the host splitter's helper threads fill shared bitmaps per chunk),
// from two caller threads at once, several rounds each, and checks every result against the handle's first one, the
// multi-chunk result against a one-chunk handle's, and decode(encode(text)) == text.
//   hostpath_driver <vocab.splv> <unicode_classes.bin>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <vector>

#include "../../include/splintr_hip.h"

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

struct Corpus { std::vector<uint8_t> text; std::vector<uint64_t> off; };
static Corpus make_corpus(uint32_t seed, int docs) {
    static const char* W[] = {"the", "of", "and", "tokenizer", "wavefront", "internationalization", "x", "42", "3.14159", "don't", "HTTPServer",
                              "snake_case", "\n", "\n\n", "  ", "\t", "{", "}", "(", ")", ";", "==", "\xe4\xbd\xa0\xe5\xa5\xbd", "\xc3\xa9t\xc3\xa9", "\xf0\x9f\x99\x82"};
    Corpus c;
    c.off.push_back(0);
    uint32_t s = seed;
    for (int d = 0; d < docs; d++) {
        const int n = 20 + (int)((s = s * 1664525u + 1013904223u) >> 24);
        for (int k = 0; k < n; k++) {
            s = s * 1664525u + 1013904223u;
            const char* w = W[(s >> 16) % (sizeof W / sizeof *W)];
            c.text.insert(c.text.end(), w, w + strlen(w));
            if ((s >> 8) & 3) c.text.push_back(' ');
        }
        c.off.push_back(c.text.size());
    }
    c.text.resize(c.text.size() + 64, 0);
    return c;
}

struct Res { std::vector<uint32_t> ids; std::vector<uint64_t> off; };
static bool encode(spl_tokenizer* t, const Corpus& c, Res& r) {
    spl_result* res = nullptr;
    if (spl_encode_batch(t, c.text.data(), c.off.data(), c.off.size() - 1, 0, &res) != SPL_OK) { fprintf(stderr, "encode: %s\n", spl_last_error()); return false; }
    r.ids.assign(spl_result_tokens(res), spl_result_tokens(res) + spl_result_n_tokens(res));
    r.off.assign(spl_result_offsets(res), spl_result_offsets(res) + spl_result_n_docs(res) + 1);
    spl_result_free(res);
    return true;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const auto vocab = slurp(argv[1]), ucls = slurp(argv[2]);
    const Corpus c = make_corpus(7, 6000);                       // ~1.6 MB
    auto make = [&](const char* pattern) {
        spl_opts o{};
        o.struct_size = sizeof o; o.pattern = pattern ? SPL_PATTERN_CUSTOM : SPL_PATTERN_CL100K; o.device = 0;
        o.pattern_text = pattern; o.pattern_len = pattern ? strlen(pattern) : 0;
        spl_tokenizer* t = spl_create(vocab.data(), vocab.size(), ucls.data(), ucls.size(), &o);
        if (!t) fprintf(stderr, "spl_create: %s\n", spl_last_error());
        return t;
    };
    spl_tokenizer* one = make(nullptr);                           // one chunk, one pipeline (direct write)
    spl_tokenizer* multi = make(nullptr);
    spl_tokenizer* custom = make("'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+");
    if (!one || !multi || !custom) return 3;
    const int32_t devs[2] = {0, 0};
    if (spl_set_devices(multi, devs, 2) != SPL_OK) return 3;
    for (spl_tokenizer* t : {multi, custom}) {
        if (spl_set_option(t, "chunk_bytes", 128 << 10) != SPL_OK || spl_set_option(t, "single_chunk_max_bytes", 0) != SPL_OK) return 3;
    }
    Res ref;
    if (!encode(one, c, ref)) return 4;
    std::atomic<int> bad{0};
    auto rounds = [&](spl_tokenizer* t, bool must_equal_ref) {
        Res first;
        for (int k = 0; k < 4; k++) {
            Res r;
            if (!encode(t, c, r)) { bad = 1; return; }
            if (k == 0) first = r;
            else if (r.ids != first.ids || r.off != first.off) { fprintf(stderr, "a repeated call differs from the first one\n"); bad = 1; }
            if (must_equal_ref && (r.ids != ref.ids || r.off != ref.off)) { fprintf(stderr, "the multi-chunk result differs from the one-chunk result\n"); bad = 1; }
        }
    };
    std::thread a([&] { rounds(multi, true); }), b([&] { rounds(custom, false); });
    a.join();
    b.join();
    // decode(encode(text)) == text
    uint8_t* ob = nullptr; uint64_t* oo = nullptr;
    if (spl_decode_batch(one, ref.ids.data(), ref.off.data(), ref.off.size() - 1, &ob, &oo) != SPL_OK) { fprintf(stderr, "decode: %s\n", spl_last_error()); return 5; }
    const uint64_t nb = c.off.back();
    if (oo[ref.off.size() - 1] != nb || memcmp(ob, c.text.data(), nb) != 0) { fprintf(stderr, "decode(encode(text)) != text\n"); bad = 1; }
    spl_free(ob); spl_free(oo);
    spl_destroy(one); spl_destroy(multi); spl_destroy(custom);
    printf("host pipeline: %zu tokens, %s\n", ref.ids.size(), bad ? "MISMATCH" : "consistent");
    return bad ? 1 : 0;
}
