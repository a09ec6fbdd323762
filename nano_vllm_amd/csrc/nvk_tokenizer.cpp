// nvk_tokenizer.cpp -- the reference's tokenizer.bin format and its BPE encode / decode behind the C ABI
// (SURVEY.md 8f row f4).  Host-only code: bytes and integers, same results as the reference bit for bit,
// including the simplifications it documents itself.
//
// ref: class Tokenizer, include/core/tokenizer.hpp
//   load    :31-61   int32 max_token_length, then vocab_size x { float score, int32 len, len bytes }
//   decode  :63-75   the piece; "<0xXX>" (6 characters) becomes the single byte XX
//   encode  :77-131  BOS (1) if asked; the " " piece as a dummy prefix when the text is not empty and " " is in
//                    the vocabulary; then one lookup per BYTE of the text (bytes without a piece are dropped:
//                    the reference's own "skip complex fallback logic"); then repeatedly merge the adjacent
//                    pair whose concatenation is in the vocabulary with the highest score (first such pair on
//                    ties, scores compared with >, floor -1e10); EOS (2) if asked.
//   lookup  :140-148 exact match in the vocabulary sorted by string (std::sort + lower_bound; with duplicate
//                    strings the reference finds whichever equal element lower_bound reaches -- here the
//                    sort is stable on (string, id), and duplicates resolve to the lowest id, which is what
//                    lower_bound over a std::sort'ed range yields whenever the equal elements ended up in id
//                    order; synthetic vocabularies in the tests have no duplicates).

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "nvk_internal.h"

struct nvk_tokenizer
{
    int                      vocab_size = 0, max_token_length = 0;
    std::vector<std::string> vocab;
    std::vector<float>       scores;
    std::vector<int>         sorted; // ids ordered by (string, id)

    int lookup(const std::string &s) const
    {
        auto it = std::lower_bound(sorted.begin(), sorted.end(), s,
                                   [&](int id, const std::string &key) { return vocab[(size_t)id] < key; });
        if (it != sorted.end() && vocab[(size_t)*it] == s)
            return *it;
        return -1;
    }
};

namespace {

// The merge phase of the reference's encode (tokenizer.hpp:106-126) -- "while some adjacent pair's concatenation is a
// piece, replace the pair whose piece scores highest (the leftmost such pair on ties; scores above -1e10 only)" -- as
// a linked list of symbols with one cached candidate per symbol and a heap over the candidates.  A merge touches two
// candidates (the pairs to the left and to the right of the new symbol) instead of rescanning and re-looking-up every
// pair of the text as the reference does: O(n log n) lookups per text instead of O(n^2), same ids.
//   symbol i   token id, links to its neighbours, `stamp` bumped whenever its token or its right neighbour changes
//   candidate  (score, left symbol, stamp at the time, merged id): stale when the stamp has moved on
// Symbols keep the index they start with and a merge keeps the LEFT symbol, so index order is text order and
// "leftmost on ties" is "lowest index on ties".
struct Candidate
{
    float score;
    int   left, stamp, merged;
};
struct WorseCandidate // heap order: best score on top, lowest index among equals
{
    bool operator()(const Candidate &a, const Candidate &b) const
    {
        return a.score < b.score || (a.score == b.score && a.left > b.left);
    }
};

void merge_pairs(const nvk_tokenizer &t, std::vector<int> &tokens)
{
    const int n = (int)tokens.size();
    if (n < 2)
        return; // (the reference's loop bound underflows on an empty list: nothing to merge there)
    std::vector<int> prev((size_t)n), next((size_t)n), stamp((size_t)n, 0);
    for (int i = 0; i < n; i++)
        prev[(size_t)i] = i - 1, next[(size_t)i] = i + 1 < n ? i + 1 : -1;
    std::vector<Candidate> heap;
    heap.reserve((size_t)n);
    const WorseCandidate worse;
    auto propose = [&](int left) { // the pair (left, its right neighbour), if its concatenation is a piece
        const int right = left < 0 ? -1 : next[(size_t)left];
        if (right < 0)
            return;
        const int id = t.lookup(t.vocab[(size_t)tokens[(size_t)left]] + t.vocab[(size_t)tokens[(size_t)right]]);
        if (id != -1 && t.scores[(size_t)id] > -1e10f) {
            heap.push_back(Candidate{t.scores[(size_t)id], left, stamp[(size_t)left], id});
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    };
    for (int i = 0; i + 1 < n; i++)
        propose(i);
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), worse);
        const Candidate c = heap.back();
        heap.pop_back();
        if (c.stamp != stamp[(size_t)c.left])
            continue; // one of the pair's symbols has changed since
        const int gone = next[(size_t)c.left], after = next[(size_t)gone];
        tokens[(size_t)c.left] = c.merged;
        next[(size_t)c.left]   = after;
        if (after >= 0)
            prev[(size_t)after] = c.left;
        stamp[(size_t)c.left]++;
        stamp[(size_t)gone] = -1; // (no live candidate carries -1)
        const int before = prev[(size_t)c.left];
        if (before >= 0)
            stamp[(size_t)before]++; // its right neighbour's token changed
        propose(before);
        propose(c.left);
    }
    std::vector<int> merged;
    for (int i = 0; i >= 0; i = next[(size_t)i])
        merged.push_back(tokens[(size_t)i]);
    tokens.swap(merged);
}

} // namespace

extern "C" {

int nvk_tokenizer_load(const char *path, int vocab_size, nvk_tokenizer **out_tokenizer)
{
    if (!out_tokenizer)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_tokenizer_load: out_tokenizer is null");
    *out_tokenizer = nullptr;
    if (!path || vocab_size <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_tokenizer_load: bad arguments");
    FILE *f = fopen(path, "rb");
    if (!f)
        return nvk::fail(NVK_ERR_INVALID_ARG, "Failed to open tokenizer: %s", path); // tokenizer.hpp:37
    nvk_tokenizer *t = new (std::nothrow) nvk_tokenizer();
    if (!t) {
        fclose(f);
        return nvk::fail(NVK_ERR_OOM, "nvk_tokenizer_load: host allocation failed");
    }
    t->vocab_size = vocab_size;
    bool ok       = fread(&t->max_token_length, sizeof(int), 1, f) == 1;
    t->vocab.resize((size_t)vocab_size);
    t->scores.resize((size_t)vocab_size);
    for (int i = 0; ok && i < vocab_size; i++) {
        int len = 0;
        ok      = fread(&t->scores[(size_t)i], sizeof(float), 1, f) == 1 && fread(&len, sizeof(int), 1, f) == 1 && len >= 0
             && len < (1 << 20);
        if (ok && len > 0) {
            t->vocab[(size_t)i].resize((size_t)len);
            ok = fread(&t->vocab[(size_t)i][0], 1, (size_t)len, f) == (size_t)len;
        }
    }
    fclose(f);
    if (!ok) { // the reference reads on without checking; a short file is an error here
        delete t;
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_tokenizer_load: %s ends before %d pieces", path, vocab_size);
    }
    t->sorted.resize((size_t)vocab_size);
    for (int i = 0; i < vocab_size; i++)
        t->sorted[(size_t)i] = i;
    std::stable_sort(t->sorted.begin(), t->sorted.end(),
                     [&](int a, int b) { return t->vocab[(size_t)a] < t->vocab[(size_t)b]; });
    *out_tokenizer = t;
    return NVK_OK;
}

void nvk_tokenizer_destroy(nvk_tokenizer *tokenizer) { delete tokenizer; }

int nvk_tokenizer_vocab_size(const nvk_tokenizer *t) { return t ? t->vocab_size : 0; }
int nvk_tokenizer_max_token_length(const nvk_tokenizer *t) { return t ? t->max_token_length : 0; }

int nvk_tokenizer_decode(const nvk_tokenizer *t, int token, char *out, int cap)
{
    if (!t || cap < 0 || (cap > 0 && !out))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_tokenizer_decode: bad arguments");
    std::string piece;
    if (token >= 0 && token < t->vocab_size) { // out of range: the empty string (tokenizer.hpp:65-66)
        piece = t->vocab[(size_t)token];
        if (piece.size() == 6 && piece.compare(0, 3, "<0x") == 0) { // a raw byte written as six characters, <0x0A> (:70-73)
            const char  digits[3] = {piece[3], piece[4], '\0'};
            char       *end       = nullptr;
            const long  byte      = strtol(digits, &end, 16);
            if (end != digits) // (the reference's std::stoi throws when there is no digit; "<0xZZ>" stays a piece here)
                piece.assign(1, (char)byte);
        }
    }
    const int n = (int)piece.size();
    if (cap > 0) {
        const int c = std::min(n, cap - 1);
        std::memcpy(out, piece.data(), (size_t)c);
        out[c] = '\0';
    }
    return n; // bytes in the piece (it may hold a NUL byte: use the count, not strlen)
}

int nvk_tokenizer_encode(const nvk_tokenizer *t, const char *text, int text_len, int bos, int eos, int *ids, int cap)
{
    if (!t || (!text && text_len != 0) || cap < 0 || (cap > 0 && !ids))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_tokenizer_encode: bad arguments");
    if (text_len < 0)
        text_len = (int)strlen(text);
    std::vector<int> tokens;
    if (bos)
        tokens.push_back(1);
    if (text_len > 0) {
        const int dummy = t->lookup(" ");
        if (dummy != -1)
            tokens.push_back(dummy);
    }
    for (int i = 0; i < text_len; i++) {
        const int id = t->lookup(std::string(1, text[i]));
        if (id != -1)
            tokens.push_back(id);
    }
    merge_pairs(*t, tokens);
    if (eos)
        tokens.push_back(2);
    for (int i = 0; i < (int)tokens.size() && i < cap; i++)
        ids[i] = tokens[(size_t)i];
    return (int)tokens.size();
}

} // extern "C"
