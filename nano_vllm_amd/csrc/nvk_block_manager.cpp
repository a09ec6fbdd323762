// nvk_block_manager.cpp -- host-side KV block allocator behind the C ABI.
//
// Same observable behaviour as the reference BlockManager
// (include/scheduler/block_manager.hpp:14-283): the lowest free block id is
// handed out first, double frees are ignored, out-of-range frees are errors on
// the plain API and ignored on the per-request API.  The reference finds the
// lowest free id with a linear scan over std::vector<bool> (:36-42), which is
// fine for its 256 blocks and quadratic for the 8K-540K block pools of the
// MI355X configs; here the free set is a 64-ary bitmap tree, so allocate and
// free are O(log64 n) and still return exactly the reference's ids.

#include <cstdint>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "nvk_internal.h"

struct nvk_block_manager
{
    int num_blocks = 0;
    int block_size = 0;
    int num_free   = 0;
    // levels[0] has one bit per block (1 = free); levels[k+1] has one bit per
    // 64-bit word of levels[k] (1 = that word is non-zero).  The top level is
    // a single word.
    std::vector<std::vector<uint64_t>> levels;

    mutable std::mutex                        mutex; // guards the per-request API only (ref :225)
    std::unordered_map<int, std::vector<int>> request_blocks;

    nvk_block_manager(int nb, int bs)
        : num_blocks(nb)
        , block_size(bs)
        , num_free(nb)
    {
        size_t bits = (size_t)(nb > 0 ? nb : 1);
        for (;;) {
            size_t words = (bits + 63) / 64;
            levels.emplace_back(words, 0ULL);
            if (words == 1)
                break;
            bits = words;
        }
        for (int i = 0; i < nb; i++)
            levels[0][(size_t)i >> 6] |= 1ULL << (i & 63);
        for (size_t lv = 1; lv < levels.size(); lv++)
            for (size_t w = 0; w < levels[lv - 1].size(); w++)
                if (levels[lv - 1][w])
                    levels[lv][w >> 6] |= 1ULL << (w & 63);
    }

    bool is_free(int id) const { return (levels[0][(size_t)id >> 6] >> (id & 63)) & 1ULL; }

    void mark_used(int id)
    {
        size_t idx = (size_t)id;
        for (size_t lv = 0; lv < levels.size(); lv++) {
            uint64_t &word = levels[lv][idx >> 6];
            word &= ~(1ULL << (idx & 63));
            if (word != 0)
                break; // siblings still free: summaries above stay set
            idx >>= 6;
        }
    }

    void mark_free(int id)
    {
        size_t idx = (size_t)id;
        for (size_t lv = 0; lv < levels.size(); lv++) {
            uint64_t &word     = levels[lv][idx >> 6];
            const bool was_set = word != 0;
            word |= 1ULL << (idx & 63);
            if (was_set)
                break;
            idx >>= 6;
        }
    }

    // lowest free id, or -1 (ref :28-44 / :231-244)
    int take_lowest()
    {
        if (num_free == 0)
            return -1;
        size_t idx = 0;
        for (size_t lv = levels.size(); lv-- > 0;) {
            const uint64_t word = levels[lv][idx];
            if (word == 0)
                return -1; // unreachable while num_free is consistent
            idx = (idx << 6) | (size_t)__builtin_ctzll(word);
        }
        mark_used((int)idx);
        num_free--;
        return (int)idx;
    }

    // ref :48-62 without the throw; returns false for an out-of-range id
    bool give_back(int id)
    {
        if (id < 0 || id >= num_blocks)
            return false;
        if (!is_free(id)) {
            mark_free(id);
            num_free++;
        }
        return true;
    }

    int blocks_for(int num_tokens) const // ref :69, in 64 bits (num_tokens near INT_MAX)
    {
        return (int)(((long long)num_tokens + block_size - 1) / block_size);
    }
};

extern "C" {

int nvk_bm_create(int num_blocks, int block_size, nvk_block_manager **out_bm)
{
    if (!out_bm || num_blocks < 0 || block_size <= 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_create: need out_bm, num_blocks >= 0, block_size > 0");
    *out_bm = new (std::nothrow) nvk_block_manager(num_blocks, block_size);
    if (!*out_bm)
        return nvk::fail(NVK_ERR_OOM, "nvk_bm_create: host allocation failed");
    return NVK_OK;
}

void nvk_bm_destroy(nvk_block_manager *bm) { delete bm; }

int nvk_bm_allocate_block(nvk_block_manager *bm) { return bm ? bm->take_lowest() : -1; }

int nvk_bm_free_block(nvk_block_manager *bm, int block_id)
{
    if (!bm)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_free_block: null manager");
    if (!bm->give_back(block_id))
        return nvk::fail(NVK_ERR_BAD_BLOCK, "Invalid block_id"); // message of ref :52
    return NVK_OK;
}

int nvk_bm_allocate_sequence(nvk_block_manager *bm, int num_tokens, int *ids, int cap)
{
    if (!bm || (!ids && cap > 0) || num_tokens < 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_allocate_sequence: bad arguments");
    const int need = bm->blocks_for(num_tokens);
    if (need > bm->num_free)
        return nvk::fail(NVK_ERR_OOM, "Out of memory"); // ref :71-74
    if (need > cap)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_allocate_sequence: ids capacity too small");
    for (int i = 0; i < need; i++)
        ids[i] = bm->take_lowest(); // cannot fail: need <= num_free (ref's rollback :81-86 is unreachable too)
    return need;
}

int nvk_bm_free_sequence(nvk_block_manager *bm, const int *ids, int n)
{
    if (!bm || (!ids && n > 0))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_free_sequence: bad arguments");
    for (int i = 0; i < n; i++)
        if (!bm->give_back(ids[i]))
            return nvk::fail(NVK_ERR_BAD_BLOCK, "Invalid block_id"); // ref stops at the first throw
    return NVK_OK;
}

int nvk_bm_num_free_blocks(const nvk_block_manager *bm) { return bm ? bm->num_free : 0; }
int nvk_bm_num_blocks(const nvk_block_manager *bm) { return bm ? bm->num_blocks : 0; }
int nvk_bm_block_size(const nvk_block_manager *bm) { return bm ? bm->block_size : 0; }

int nvk_bm_is_free(const nvk_block_manager *bm, int block_id)
{
    if (!bm || block_id < 0 || block_id >= bm->num_blocks)
        return 0;
    return bm->is_free(block_id) ? 1 : 0;
}

float nvk_bm_utilization(const nvk_block_manager *bm)
{
    if (!bm || bm->num_blocks <= 0)
        return 0.f; // (the reference divides 0 by 0 for an empty manager)
    return 1.0f - (static_cast<float>(bm->num_free) / bm->num_blocks); // ref :124, same expression
}

int nvk_bm_allocate_for_request(nvk_block_manager *bm, int request_id, int num_tokens, int *ids, int cap)
{
    if (!bm || (!ids && cap > 0) || num_tokens < 0)
        return 0;
    std::lock_guard<std::mutex> lock(bm->mutex);
    const int                   need = bm->blocks_for(num_tokens);
    if (need > cap) { // the caller's array, not the pool, is too small: say so (the count is 0 either way)
        nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_allocate_for_request: %d blocks needed, ids holds %d", need, cap);
        return 0;
    }
    if (need > bm->num_free)
        return 0; // ref :263-265 returns an empty vector
    for (int i = 0; i < need; i++)
        ids[i] = bm->take_lowest();
    if (need > 0) { // ref :137-139: no map entry for an empty result
        std::vector<int> &mine = bm->request_blocks[request_id];
        mine.insert(mine.end(), ids, ids + need);
    }
    return need;
}

int nvk_bm_allocate_block_for_request(nvk_block_manager *bm, int request_id)
{
    if (!bm)
        return -1;
    std::lock_guard<std::mutex> lock(bm->mutex);
    const int                   id = bm->take_lowest();
    if (id >= 0)
        bm->request_blocks[request_id].push_back(id);
    return id;
}

int nvk_bm_release_block_for_request(nvk_block_manager *bm, int request_id, int block_id)
{
    if (!bm)
        return NVK_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lock(bm->mutex);
    auto                        it = bm->request_blocks.find(request_id);
    if (it == bm->request_blocks.end())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_bm_release_block_for_request: request %d holds no blocks", request_id);
    std::vector<int> &mine = it->second;
    for (size_t i = mine.size(); i-- > 0;)
        if (mine[i] == block_id) {
            mine.erase(mine.begin() + (long)i);
            if (mine.empty())
                bm->request_blocks.erase(it);
            bm->give_back(block_id);
            return NVK_OK;
        }
    return nvk::fail(NVK_ERR_BAD_BLOCK, "nvk_bm_release_block_for_request: request %d does not hold block %d", request_id,
                     block_id);
}

void nvk_bm_free_request(nvk_block_manager *bm, int request_id)
{
    if (!bm)
        return;
    std::lock_guard<std::mutex> lock(bm->mutex);
    auto                        it = bm->request_blocks.find(request_id);
    if (it == bm->request_blocks.end())
        return;
    for (int id : it->second)
        bm->give_back(id);
    bm->request_blocks.erase(it);
}

int nvk_bm_get_request_blocks(nvk_block_manager *bm, int request_id, int *ids, int cap)
{
    if (!bm)
        return 0;
    std::lock_guard<std::mutex> lock(bm->mutex);
    auto                        it = bm->request_blocks.find(request_id);
    if (it == bm->request_blocks.end())
        return 0;
    const int n = (int)it->second.size();
    for (int i = 0; i < n && i < cap; i++)
        ids[i] = it->second[i];
    return n;
}

int nvk_bm_get_request_block_count(nvk_block_manager *bm, int request_id)
{
    if (!bm)
        return 0;
    std::lock_guard<std::mutex> lock(bm->mutex);
    auto                        it = bm->request_blocks.find(request_id);
    return it == bm->request_blocks.end() ? 0 : (int)it->second.size();
}

int nvk_bm_num_active_requests(nvk_block_manager *bm)
{
    if (!bm)
        return 0;
    std::lock_guard<std::mutex> lock(bm->mutex);
    return (int)bm->request_blocks.size();
}

int nvk_bm_allocate_block_safe(nvk_block_manager *bm)
{
    if (!bm)
        return -1;
    std::lock_guard<std::mutex> lock(bm->mutex);
    return bm->take_lowest();
}

void nvk_bm_free_block_safe(nvk_block_manager *bm, int block_id)
{
    if (!bm)
        return;
    std::lock_guard<std::mutex> lock(bm->mutex);
    bm->give_back(block_id); // out-of-range ignored (ref :249-251)
}

} // extern "C"
