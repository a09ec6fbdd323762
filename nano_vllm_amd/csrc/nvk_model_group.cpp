// nvk_model_group.cpp -- the engine sequence-sharded over several GPUs (primary mode, at the level of
// nvk_model_* / nvk_run): one nvk_model per device, each with its own weights, pool, allocator and batch, each driven
// by a host thread of its own; a request lives on ONE member from admission to finish, so the members share nothing
// and no collective exists on the data path.
//
// ref: what is being split is the scheduler's running list (include/scheduler/scheduler.hpp:66-103: FIFO admission
// up to max_batch_size) and Request::block_ids (include/scheduler/request.hpp:74: a request's blocks belong to it
// alone); sequences are independent in the attention itself (include/core/attention.hpp:82-159 touches one table, one
// q, one out).  Admission is balanced by tokens: a new request goes to the member whose running requests promise the
// fewest tokens (prompt + max_tokens), lowest index on ties -- sharding.partition_sequences' rule, applied on line.
//
// Host-only C++ over the public C ABI (nvk_model_*): nothing here touches a device itself.
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "nvk_internal.h"

namespace {

// One member: a model and the thread that makes every call on it.
struct Member
{
    nvk_model  *model  = nullptr;
    int         device = 0;
    long long   load   = 0; // tokens promised by its running requests
    int         active = 0; // running requests
    std::thread worker;
    std::mutex  mu;
    std::condition_variable cv;
    std::function<void()>   task; // at most one in flight
    bool        has_task = false, done = false, quit = false;

    void run()
    {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [&] { return has_task || quit; });
                if (!has_task && quit)
                    return;
                fn = std::move(task);
            }
            fn();
            {
                std::lock_guard<std::mutex> lock(mu);
                has_task = false, done = true;
            }
            cv.notify_all();
        }
    }
    void post(std::function<void()> fn)
    {
        {
            std::lock_guard<std::mutex> lock(mu);
            task = std::move(fn), has_task = true, done = false;
        }
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return done; });
    }
};

struct Placement
{
    int       member;
    long long tokens;
};

} // namespace

struct nvk_model_group
{
    std::vector<Member *>              members;
    std::unordered_map<int, Placement> where; // request id -> member, promised tokens
    int                                max_seqs = 0;
};

namespace {

// Runs fn(i) on member i's thread for every i in `which` at once; returns the first failing status (and makes its
// message the calling thread's last error).
int on_members(nvk_model_group *g, const std::vector<int> &which, const std::function<int(int)> &fn)
{
    std::vector<int>         rc(which.size(), NVK_OK);
    std::vector<std::string> msg(which.size());
    for (size_t k = 0; k < which.size(); k++) {
        const int i = which[k];
        g->members[(size_t)i]->post([&, k, i] {
            rc[k] = fn(i);
            if (rc[k] != NVK_OK)
                msg[k] = nvk_last_error(); // this thread's; handed to the caller below
        });
    }
    for (int i : which)
        g->members[(size_t)i]->wait();
    for (size_t k = 0; k < which.size(); k++)
        if (rc[k] != NVK_OK)
            return nvk::fail(rc[k], "member %d (device %d): %s", which[k], g->members[(size_t)which[k]]->device,
                             msg[k].c_str());
    return NVK_OK;
}

std::vector<int> all_members(const nvk_model_group *g)
{
    std::vector<int> v(g->members.size());
    for (size_t i = 0; i < v.size(); i++)
        v[i] = (int)i;
    return v;
}

} // namespace

extern "C" {

int nvk_model_group_destroy(nvk_model_group *g)
{
    if (!g)
        return NVK_OK;
    std::vector<int> live;
    for (size_t i = 0; i < g->members.size(); i++)
        if (g->members[i]->worker.joinable())
            live.push_back((int)i);
    (void)on_members(g, live, [&](int i) {
        Member *m = g->members[(size_t)i];
        const int rc = m->model ? nvk_model_destroy(m->model) : NVK_OK;
        m->model     = nullptr;
        return rc;
    });
    for (Member *m : g->members) {
        if (m->worker.joinable()) {
            {
                std::lock_guard<std::mutex> lock(m->mu);
                m->quit = true;
            }
            m->cv.notify_all();
            m->worker.join();
        }
        delete m;
    }
    delete g;
    return NVK_OK;
}

int nvk_model_group_load(const char *checkpoint_path, int block_size, int num_blocks, int max_seqs_per_member,
                         int n_members, const int *devices, nvk_model_group **out_group)
{
    if (!out_group)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_load: out_group is null");
    *out_group = nullptr;
    if (!checkpoint_path || n_members < 1 || n_members > 64 || !devices || max_seqs_per_member < 1)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_load: bad arguments (1..64 members, a device each)");
    nvk_model_group *g = new (std::nothrow) nvk_model_group();
    if (!g)
        return nvk::fail(NVK_ERR_OOM, "nvk_model_group_load: host allocation failed");
    g->max_seqs = max_seqs_per_member;
    for (int i = 0; i < n_members; i++) {
        Member *m = new (std::nothrow) Member();
        if (!m) {
            nvk_model_group_destroy(g);
            return nvk::fail(NVK_ERR_OOM, "nvk_model_group_load: host allocation failed");
        }
        m->device = devices[i];
        g->members.push_back(m);
        m->worker = std::thread([m] { m->run(); });
    }
    // every member reads the checkpoint and fills its own HBM, all at once
    const std::string path = checkpoint_path;
    const int rc = on_members(g, all_members(g), [&](int i) {
        Member *m = g->members[(size_t)i];
        return nvk_model_load(path.c_str(), block_size, num_blocks, max_seqs_per_member, m->device, &m->model);
    });
    if (rc != NVK_OK) {
        const std::string why = nvk_last_error();
        nvk_model_group_destroy(g);
        return nvk::fail(rc, "%s", why.c_str());
    }
    *out_group = g;
    return NVK_OK;
}

int nvk_model_group_size(const nvk_model_group *g) { return g ? (int)g->members.size() : 0; }

int nvk_model_group_config(const nvk_model_group *g, int *out8)
{
    if (!g || g->members.empty() || !out8)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_config: null argument");
    return nvk_model_config(g->members[0]->model, out8);
}

int nvk_model_group_member_device(const nvk_model_group *g, int member)
{
    if (!g || member < 0 || member >= (int)g->members.size())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_member_device: no member %d", member);
    return g->members[(size_t)member]->device;
}

int nvk_model_group_member_of(const nvk_model_group *g, int request_id)
{
    if (!g)
        return -1;
    auto it = g->where.find(request_id);
    return it == g->where.end() ? -1 : it->second.member;
}

int nvk_model_group_add_request(nvk_model_group *g, int request_id, int expected_tokens)
{
    if (!g || expected_tokens < 0)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_add_request: bad arguments");
    if (g->where.count(request_id))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_add_request: request %d is already running", request_id);
    int best = -1;
    for (size_t i = 0; i < g->members.size(); i++) {
        const Member *m = g->members[i];
        if (m->active >= g->max_seqs)
            continue;
        if (best < 0 || m->load < g->members[(size_t)best]->load)
            best = (int)i;
    }
    if (best < 0)
        return nvk::fail(NVK_ERR_OOM, "nvk_model_group_add_request: every member already runs %d requests", g->max_seqs);
    const int rc = on_members(g, {best}, [&](int i) { return nvk_model_add_request(g->members[(size_t)i]->model, request_id); });
    if (rc != NVK_OK)
        return rc;
    g->where[request_id] = Placement{best, expected_tokens};
    g->members[(size_t)best]->load += expected_tokens;
    g->members[(size_t)best]->active++;
    return best;
}

int nvk_model_group_finish_request(nvk_model_group *g, int request_id)
{
    if (!g)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_finish_request: null group");
    auto it = g->where.find(request_id);
    if (it == g->where.end())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_finish_request: no request %d", request_id);
    const Placement p  = it->second;
    const int       rc = on_members(g, {p.member}, [&](int i) {
        return nvk_model_finish_request(g->members[(size_t)i]->model, request_id);
    });
    if (rc != NVK_OK)
        return rc;
    g->members[(size_t)p.member]->load -= p.tokens;
    g->members[(size_t)p.member]->active--;
    g->where.erase(it);
    return NVK_OK;
}

int nvk_model_group_request_len(const nvk_model_group *g, int request_id)
{
    if (!g)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_request_len: null group");
    auto it = g->where.find(request_id);
    if (it == g->where.end())
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_request_len: no request %d", request_id);
    // (a read of host-side bookkeeping; the member's thread is idle between the group's calls)
    return nvk_model_request_len(g->members[(size_t)it->second.member]->model, request_id);
}

int nvk_model_group_step_chunks(nvk_model_group *g, const int *request_ids, const int *chunk_lens, const int *tokens,
                                int n, int *next_tokens, float *logits)
{
    if (!g || n < 0 || (n > 0 && (!request_ids || !chunk_lens || !tokens)))
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_step_chunks: bad arguments");
    if (n == 0)
        return NVK_OK;
    const size_t G = g->members.size();
    int cfg[8];
    int rc = nvk_model_group_config(g, cfg);
    if (rc != NVK_OK)
        return rc;
    const size_t vocab = (size_t)cfg[5];
    // the step's list, split by member in list order (a member sees its requests in the order the caller gave)
    struct Share
    {
        std::vector<int>   ids, lens, toks, slot, next;
        std::vector<float> logits;
    };
    std::vector<Share> share(G);
    size_t             tok_at = 0;
    for (int k = 0; k < n; k++) {
        auto it = g->where.find(request_ids[k]);
        if (it == g->where.end())
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_step_chunks: no request %d", request_ids[k]);
        if (chunk_lens[k] < 1)
            return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_step_chunks: chunk_lens[%d]=%d", k, chunk_lens[k]);
        Share &s = share[(size_t)it->second.member];
        s.ids.push_back(request_ids[k]);
        s.lens.push_back(chunk_lens[k]);
        s.toks.insert(s.toks.end(), tokens + tok_at, tokens + tok_at + chunk_lens[k]);
        s.slot.push_back(k);
        tok_at += (size_t)chunk_lens[k];
    }
    std::vector<int> busy;
    for (size_t i = 0; i < G; i++)
        if (!share[i].ids.empty()) {
            busy.push_back((int)i);
            share[i].next.resize(share[i].ids.size());
            if (logits)
                share[i].logits.resize(share[i].ids.size() * vocab);
        }
    rc = on_members(g, busy, [&](int i) {
        Share &s = share[(size_t)i];
        return nvk_model_step_chunks(g->members[(size_t)i]->model, s.ids.data(), s.lens.data(), s.toks.data(),
                                     (int)s.ids.size(), s.next.data(), logits ? s.logits.data() : nullptr);
    });
    if (rc != NVK_OK)
        return rc;
    for (int i : busy) {
        const Share &s = share[(size_t)i];
        for (size_t j = 0; j < s.slot.size(); j++) {
            if (next_tokens)
                next_tokens[s.slot[j]] = s.next[j];
            if (logits)
                std::copy(s.logits.begin() + (long)(j * vocab), s.logits.begin() + (long)((j + 1) * vocab),
                          logits + (size_t)s.slot[j] * vocab);
        }
    }
    return NVK_OK;
}

int nvk_model_group_use_graphs(nvk_model_group *g, int on)
{
    if (!g)
        return nvk::fail(NVK_ERR_INVALID_ARG, "nvk_model_group_use_graphs: null group");
    return on_members(g, all_members(g), [&](int i) { return nvk_model_use_graphs(g->members[(size_t)i]->model, on); });
}

} // extern "C"
