// nvk_dropin/scheduler/block_manager.hpp -- C++ drop-in for the reference's
// include/scheduler/block_manager.hpp:14-283: the same class name, public methods, return
// values and exception messages, over the C ABI's O(log n) allocator (nvk_bm_*).
#pragma once

#include <stdexcept>
#include <vector>

#include "nvk_paged_attn.h"

class BlockManager
{
public:
    BlockManager(int num_blocks, int block_size)
    {
        if (nvk_bm_create(num_blocks, block_size, &bm_) != NVK_OK)
            throw std::runtime_error(nvk_last_error());
    }
    ~BlockManager() { nvk_bm_destroy(bm_); }
    BlockManager(const BlockManager &)            = delete;
    BlockManager &operator=(const BlockManager &) = delete;

    int allocate_block() { return nvk_bm_allocate_block(bm_); } // -1 when full (ref :28-44)

    void free_block(int block_id) // ref :48-62
    {
        if (nvk_bm_free_block(bm_, block_id) != NVK_OK)
            throw std::runtime_error("Invalid block_id");
    }

    std::vector<int> allocate_sequence(int num_tokens) // ref :67-92
    {
        std::vector<int> ids(static_cast<size_t>(get_num_blocks() > 0 ? get_num_blocks() : 1));
        const int        n = nvk_bm_allocate_sequence(bm_, num_tokens, ids.data(), static_cast<int>(ids.size()));
        if (n < 0)
            throw std::runtime_error("Out of memory");
        ids.resize(static_cast<size_t>(n));
        return ids;
    }

    void free_sequence(const std::vector<int> &block_ids) // ref :96-101
    {
        if (nvk_bm_free_sequence(bm_, block_ids.data(), static_cast<int>(block_ids.size())) != NVK_OK)
            throw std::runtime_error("Invalid block_id");
    }

    int   get_num_free_blocks() const { return nvk_bm_num_free_blocks(bm_); }
    int   get_num_blocks() const { return nvk_bm_num_blocks(bm_); }
    int   get_block_size() const { return nvk_bm_block_size(bm_); }
    bool  is_free(int block_id) const { return nvk_bm_is_free(bm_, block_id) != 0; }
    float get_utilization() const { return nvk_bm_utilization(bm_); }

    // per-request, mutex-guarded, non-throwing (ref :132-216)
    std::vector<int> allocate_for_request(int request_id, int num_tokens)
    {
        std::vector<int> ids(static_cast<size_t>(get_num_blocks() > 0 ? get_num_blocks() : 1));
        const int n = nvk_bm_allocate_for_request(bm_, request_id, num_tokens, ids.data(), static_cast<int>(ids.size()));
        ids.resize(static_cast<size_t>(n > 0 ? n : 0));
        return ids;
    }
    int  allocate_block_for_request(int request_id) { return nvk_bm_allocate_block_for_request(bm_, request_id); }
    void free_request(int request_id) { nvk_bm_free_request(bm_, request_id); }
    std::vector<int> get_request_blocks(int request_id) const
    {
        std::vector<int> ids(static_cast<size_t>(nvk_bm_get_request_block_count(bm_, request_id)) + 1);
        const int        n = nvk_bm_get_request_blocks(bm_, request_id, ids.data(), static_cast<int>(ids.size()));
        ids.resize(static_cast<size_t>(n < static_cast<int>(ids.size()) ? n : static_cast<int>(ids.size())));
        return ids;
    }
    int  get_request_block_count(int request_id) const { return nvk_bm_get_request_block_count(bm_, request_id); }
    int  get_num_active_requests() const { return nvk_bm_num_active_requests(bm_); }
    int  allocate_block_safe() { return nvk_bm_allocate_block_safe(bm_); }
    void free_block_safe(int block_id) { nvk_bm_free_block_safe(bm_, block_id); }

private:
    nvk_block_manager *bm_ = nullptr;
};
