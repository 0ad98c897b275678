/* san_driver.c -- TEST INFRASTRUCTURE: runs the C oracle's batch path (pthread pool, optional mutex-guarded memo)
 * under a sanitizer build (make -C oracle asan tsan).  Usage: san_driver <vocab.splv> <unicode_classes.bin> <corpus.bin>
 * where corpus.bin = u64 n_docs | u64 off[n_docs + 1] | bytes.  Prints token count and a checksum per configuration;
 * exits 1 if two configurations disagree. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc orc_t;
orc_t *orc_create(const char *splv_path, const char *uclass_path, int pattern_id, int byte_level);
void orc_set_memo(orc_t *t, int on);
void orc_destroy(orc_t *t);
void orc_free(void *p);
uint64_t orc_encode_batch(orc_t *t, const uint8_t *text, const uint64_t *off, uint64_t ndocs, int with_special, int threads,
                          uint32_t **ids, uint64_t *out_off);

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *f = fopen(argv[3], "rb");
    if (!f) return 2;
    uint64_t nd;
    if (fread(&nd, 8, 1, f) != 1) return 2;
    uint64_t *off = malloc((nd + 1) * 8);
    if (fread(off, 8, nd + 1, f) != nd + 1) return 2;
    uint8_t *text = malloc(off[nd] + 16);
    if (fread(text, 1, off[nd], f) != off[nd]) return 2;
    fclose(f);
    orc_t *t = orc_create(argv[1], argv[2], 0, 0);
    if (!t) return 3;
    uint64_t first_sum = 0, first_n = 0;
    int bad = 0;
    for (int cfg = 0; cfg < 4; cfg++) {
        const int threads = (cfg & 1) ? 8 : 3, memo = cfg >> 1;
        orc_set_memo(t, memo);
        uint32_t *ids = NULL;
        uint64_t *oo = malloc((nd + 1) * 8);
        const uint64_t n = orc_encode_batch(t, text, off, nd, 0, threads, &ids, oo);
        uint64_t sum = 1469598103934665603ull;
        for (uint64_t i = 0; i < n; i++) sum = (sum ^ ids[i]) * 1099511628211ull;
        for (uint64_t d = 0; d <= nd; d++) sum = (sum ^ oo[d]) * 1099511628211ull;
        printf("threads %d memo %d: %llu tokens, checksum %016llx\n", threads, memo, (unsigned long long)n, (unsigned long long)sum);
        if (cfg == 0) { first_sum = sum; first_n = n; } else if (sum != first_sum || n != first_n) bad = 1;
        orc_free(ids);
        free(oo);
    }
    orc_destroy(t);
    free(text);
    free(off);
    return bad;
}
