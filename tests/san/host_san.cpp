// host_san.cpp -- TEST TOOL: the product's host-only code under sanitizers (no GPU): the table builder
// (spl_tables.cpp: tiktoken / SPLV parser, salted placement, pair table) and the host splitter (spl_regex.cpp)
// splitting many documents from several threads into SHARED bitmaps, as spl_api.hip's host_split_docs does.
//   host_san <vocab.splv> <unicode_classes.bin> <corpus.bin>      corpus.bin = u64 n_docs | u64 off[n + 1] | bytes
// Built by tests/test_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <thread>
#include <vector>

#include "../../splintr_amd/csrc/spl_regex.h"
#include "../../splintr_amd/csrc/spl_tables.h"

static std::vector<uint8_t> slurp(const char* p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const auto vocab = slurp(argv[1]), ucls = slurp(argv[2]), corpus = slurp(argv[3]);
    if (vocab.empty() || ucls.empty() || corpus.size() < 16) return 2;
    spl::HostTables ht;
    std::string err;
    if (spl::build_tables(vocab.data(), vocab.size(), ucls.data(), ucls.size(), spl::PAT_CL100K, false, ht, err)) { fprintf(stderr, "build_tables: %s\n", err.c_str()); return 3; }
    printf("tables: %u keys, %u pairs, unsalted groups %u\n", ht.n_keys, ht.n_pairs, ht.unsalted_groups);
    uint64_t nd;
    memcpy(&nd, corpus.data(), 8);
    const uint64_t* off = reinterpret_cast<const uint64_t*>(corpus.data() + 8);
    const uint8_t* text = corpus.data() + 8 + 8 * (nd + 1);
    const char* pats[] = {"'s|'t|'re|'ve|'m|'ll|'d| ?\\p{L}+| ?\\p{N}+| ?[^\\s\\p{L}\\p{N}]+|\\s+(?!\\S)|\\s+", "\\p{L}+|[0-9]{1,3}",
                          "\\p{Han}+|[\\p{Hiragana}\\p{Katakana}]+|\\P{Latin}|\\p{Latin}+"};
    for (const char* pat : pats) {
        spl::RegexPtr re = spl::regex_compile(pat, ht, err);
        if (!re) { fprintf(stderr, "regex_compile: %s\n", err.c_str()); return 4; }
        const uint64_t words = off[nd] / 32 + 2;
        std::vector<uint32_t> st1(words, 0), gp1(words, 0), stN(words, 0), gpN(words, 0);
        for (uint64_t d = 0; d < nd; d++)
            if (!spl::regex_split_bits(*re, text + off[d], (size_t)(off[d + 1] - off[d]), off[d], st1.data(), gp1.data())) return 5;
        std::atomic<uint64_t> next{0};
        std::atomic<int> bad{0};
        auto work = [&] {
            for (;;) {
                const uint64_t d = next.fetch_add(1);
                if (d >= nd) return;
                if (!spl::regex_split_bits(*re, text + off[d], (size_t)(off[d + 1] - off[d]), off[d], stN.data(), gpN.data())) bad = 1;
            }
        };
        std::vector<std::thread> ths;
        for (int k = 0; k < 6; k++) ths.emplace_back(work);
        for (auto& t : ths) t.join();
        if (bad || st1 != stN || gp1 != gpN) { fprintf(stderr, "threaded split differs from the sequential one\n"); return 6; }
        uint64_t chunks = 0;
        for (uint32_t w : st1) chunks += (uint64_t)__builtin_popcount(w);
        printf("pattern ok: %llu start bits over %llu bytes\n", (unsigned long long)chunks, (unsigned long long)off[nd]);
    }
    // refused patterns must fail cleanly
    for (const char* pat : {"\\p{Alphabetic}+", "(a", "a*", "[z-a]", "\\p{Foo}", "(?<=x)y", "(?i:\\p{Lu})", "\\b+"}) {
        spl::RegexPtr re = spl::regex_compile(pat, ht, err);
        if (re) { fprintf(stderr, "pattern %s should have been refused\n", pat); return 7; }
    }
    // a truncated / corrupt vocabulary must be an error, not a crash
    for (size_t cut : {(size_t)3, (size_t)19, (size_t)1000, vocab.size() / 2}) {
        spl::HostTables h2;
        if (!spl::build_tables(vocab.data(), cut, ucls.data(), ucls.size(), spl::PAT_CL100K, false, h2, err)) { fprintf(stderr, "truncated vocabulary accepted\n"); return 8; }
    }
    const char* tk = "SGVsbG8= 0\nd29ybGQ= 1 \n\nYQ== +2\n";
    spl::HostTables h3;
    (void)spl::build_tables(reinterpret_cast<const uint8_t*>(tk), strlen(tk), ucls.data(), ucls.size(), spl::PAT_CL100K, false, h3, err);
    return 0;
}
