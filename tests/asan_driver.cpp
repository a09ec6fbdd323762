// asan_driver.cpp -- test infrastructure: the host-only translation units of the library (block allocator,
// tokenizer) and the CPU checker, built with -fsanitize=address,undefined and driven hard (make asan;
// tests/test_sanitizers.py).  No GPU, no HIP.  Exit 0 = every check held and no sanitizer report.
//   asan_driver bm                       randomized allocator traces, product vs the checker's linear scan
//   asan_driver tok <tokenizer.bin> V    encode/decode sweeps over a good file
//   asan_driver tokbad <file> V          a short or damaged file: loading must fail or succeed, never fault
//   asan_driver attn                     the checker's attention on small ragged cases
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nvk_paged_attn.h"

extern "C" {
typedef struct oracle_bm oracle_bm;
oracle_bm *oracle_bm_create(int num_blocks, int block_size);
void       oracle_bm_destroy(oracle_bm *b);
int        oracle_bm_allocate_block(oracle_bm *b);
int        oracle_bm_free_block(oracle_bm *b, int id);
int        oracle_bm_num_free(const oracle_bm *b);
int        oracle_bm_allocate_block_for_request(oracle_bm *b, int request_id);
void       oracle_bm_free_request(oracle_bm *b, int request_id);
int        oracle_bm_get_request_block_count(oracle_bm *b, int request_id);
void       oracle_paged_attention(float *out, const float *q, const float *key_cache, const float *value_cache,
                                  const int *block_table, float *att_scores, int num_tokens, int block_size, int head_dim,
                                  int n_heads, int n_kv_heads);
}

static uint64_t rng_state = 0x1234567ULL;
static uint64_t rnd()
{
    uint64_t x = (rng_state += 0x9E3779B97F4A7C15ULL);
    x          = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x          = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
#define CHECK(c)                                                                                                       \
    do {                                                                                                               \
        if (!(c)) {                                                                                                    \
            fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__);                                     \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

static int run_bm()
{
    for (int nb : {1, 7, 64, 65, 4096, 4097, 70000}) {
        nvk_block_manager *bm = nullptr;
        CHECK(nvk_bm_create(nb, 16, &bm) == NVK_OK);
        oracle_bm       *ob = oracle_bm_create(nb, 16);
        std::vector<int> held;
        const int        ops = nb > 10000 ? 40000 : 200000;
        for (int i = 0; i < ops; i++) {
            const unsigned r = (unsigned)(rnd() % 100);
            if (r < 45) {
                const int a = nvk_bm_allocate_block(bm), b = oracle_bm_allocate_block(ob);
                CHECK(a == b);
                if (a >= 0)
                    held.push_back(a);
            }
            else if (r < 80 && !held.empty()) {
                const size_t k = (size_t)(rnd() % held.size());
                CHECK(nvk_bm_free_block(bm, held[k]) == NVK_OK);
                oracle_bm_free_block(ob, held[k]);
                held[k] = held.back();
                held.pop_back();
            }
            else if (r < 85) {
                CHECK(nvk_bm_free_block(bm, nb + (int)(rnd() % 5)) == NVK_ERR_BAD_BLOCK); // where the reference throws
                CHECK(nvk_bm_free_block(bm, -1 - (int)(rnd() % 5)) == NVK_ERR_BAD_BLOCK);
            }
            else if (r < 93) {
                const int req = (int)(rnd() % 9);
                const int a = nvk_bm_allocate_block_for_request(bm, req), b = oracle_bm_allocate_block_for_request(ob, req);
                CHECK(a == b);
            }
            else {
                const int req = (int)(rnd() % 9);
                CHECK(nvk_bm_get_request_block_count(bm, req) == oracle_bm_get_request_block_count(ob, req));
                nvk_bm_free_request(bm, req);
                oracle_bm_free_request(ob, req);
            }
            CHECK(nvk_bm_num_free_blocks(bm) == oracle_bm_num_free(ob));
        }
        // the release the product adds for its roll-backs
        const int id = nvk_bm_allocate_block_for_request(bm, 100);
        if (id >= 0) {
            CHECK(nvk_bm_release_block_for_request(bm, 100, id) == NVK_OK);
            CHECK(nvk_bm_release_block_for_request(bm, 100, id) != NVK_OK);
        }
        int ids[4];
        (void)nvk_bm_allocate_for_request(bm, 200, 0x7fffffff, ids, 4);
        (void)nvk_bm_utilization(nullptr);
        nvk_bm_destroy(bm);
        oracle_bm_destroy(ob);
    }
    return 0;
}

static int run_tok(const char *path, int vocab, bool may_fail)
{
    nvk_tokenizer *t  = nullptr;
    const int      rc = nvk_tokenizer_load(path, vocab, &t);
    if (rc != NVK_OK) {
        if (!may_fail)
            fprintf(stderr, "load failed: %s\n", nvk_last_error());
        return may_fail ? 0 : 1;
    }
    const int        V = nvk_tokenizer_vocab_size(t), L = nvk_tokenizer_max_token_length(t);
    std::vector<char> piece((size_t)(L > 0 ? L : 1) + 8);
    for (int i = -3; i < V + 3; i++)
        (void)nvk_tokenizer_decode(t, i, piece.data(), (int)piece.size());
    (void)nvk_tokenizer_decode(t, V / 2, piece.data(), 1); // a buffer with room for the NUL only
    const char alphabet[] = " etaoinshrdlu .,?!\n\xC3\xA9\x7F";
    for (int n = 0; n < 400; n++) {
        std::string text;
        const int   len = (int)(rnd() % 90);
        for (int i = 0; i < len; i++)
            text.push_back(alphabet[rnd() % (sizeof alphabet - 1)]);
        if (n % 7 == 0)
            text.push_back((char)(rnd() & 0xff)); // any byte, known to the vocabulary or not
        std::vector<int> ids((size_t)len + 8);
        for (int bos = 0; bos < 2; bos++) {
            const int cnt = nvk_tokenizer_encode(t, text.data(), (int)text.size(), bos, 1 - bos, ids.data(), (int)ids.size());
            CHECK(cnt >= 0 && cnt <= len + 4);
            const int small = nvk_tokenizer_encode(t, text.data(), (int)text.size(), bos, 1 - bos, ids.data(), 2);
            CHECK(small == cnt); // the count does not depend on the room given
        }
    }
    nvk_tokenizer_destroy(t);
    return 0;
}

static int run_attn()
{
    for (int it = 0; it < 40; it++) {
        const int KV = 1 + (int)(rnd() % 4), H = KV * (1 + (int)(rnd() % 4)), hd = 4 * (1 + (int)(rnd() % 12));
        const int bs = 1 + (int)(rnd() % 16), T = (int)(rnd() % 90), nblk = (T + bs - 1) / bs, nb = nblk + 3;
        std::vector<float> k((size_t)nb * bs * KV * hd), v(k.size()), q((size_t)H * hd), out(q.size()), att((size_t)H * (T + 1));
        for (auto *a : {&k, &v, &q})
            for (float &x : *a)
                x = (float)(rnd() >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
        std::vector<int> table((size_t)(nblk > 0 ? nblk : 1));
        for (int i = 0; i < nblk; i++)
            table[(size_t)i] = (i * 7 + 2) % nb;
        oracle_paged_attention(out.data(), q.data(), k.data(), v.data(), table.data(), att.data(), T, bs, hd, H, KV);
        for (float x : out)
            CHECK(x == x && x >= -1.0001f && x <= 1.0001f); // a convex combination of V's rows (zero for T == 0)
    }
    return 0;
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "bm")
        return run_bm();
    if (mode == "tok" && argc > 3)
        return run_tok(argv[2], atoi(argv[3]), false);
    if (mode == "tokbad" && argc > 3)
        return run_tok(argv[2], atoi(argv[3]), true);
    if (mode == "attn")
        return run_attn();
    fprintf(stderr, "usage: asan_driver bm | tok FILE V | tokbad FILE V | attn\n");
    return 2;
}
