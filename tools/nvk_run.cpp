// nvk_run -- the reference's command line over libnvk_paged_attn.so.
//
// ref: src/main.cpp:19-26 (flags), include/utils/path.hpp:14-62 (model / tokenizer path resolution),
//      include/core/runner.hpp:24-70 (single prompt), :122-145 + include/utils/json_parser.hpp:303-326
//      (--input-json: {"requests": [{"prompt", "temperature", "top_p", "max_tokens"}]}),
//      include/scheduler/request_processor.hpp:26-88 (stop at max_tokens, max_seq_len or EOS = 2).
//
//   nvk_run <model dir | model.bin> (-i PROMPT | --input-json FILE) [-n STEPS] [-b MAX_BATCH] [-t TEMP] [-p TOPP]
//
// Differences, on purpose: the whole forward pass runs on the GPU (nvk_model_*); a prompt is prefilled as one
// chunk; --input-json with -b > 1 really batches (the reference forms the batch and then runs each request alone,
// batched_runner.hpp:84-85); sampling is greedy whatever -t / -p say (the reference seeds its sampler from the
// clock, so only greedy output is reproducible) -- a temperature other than 0 is an error unless --greedy is given.
// Host-only C++ over the C ABI; no CPU path for the model.

#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "nvk_paged_attn.h"

namespace {

struct RequestSpec
{
    std::string prompt;
    float       temperature = 1.0f, top_p = 0.9f;
    int         max_tokens  = 256;
};

// ---- the few pieces of JSON the examples use: objects, arrays, strings with escapes, numbers, literals ----
struct Json
{
    const std::string &s;
    size_t             i = 0;
    explicit Json(const std::string &text) : s(text) {}
    void ws()
    {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r'))
            i++;
    }
    char peek()
    {
        ws();
        if (i >= s.size())
            throw std::runtime_error("unexpected end of JSON");
        return s[i];
    }
    void expect(char c)
    {
        if (peek() != c)
            throw std::runtime_error(std::string("expected '") + c + "' at offset " + std::to_string(i));
        i++;
    }
    std::string string()
    {
        expect('"');
        std::string out;
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c == '\\' && i < s.size()) {
                c = s[i++];
                switch (c) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': { // BMP code point -> UTF-8
                    unsigned cp = (unsigned)strtoul(s.substr(i, 4).c_str(), nullptr, 16);
                    i += 4;
                    if (cp < 0x80)
                        out += (char)cp;
                    else if (cp < 0x800)
                        out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
                    else
                        out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)),
                            out += (char)(0x80 | (cp & 0x3F));
                    break;
                }
                default: out += c; // \" \\ \/
                }
            }
            else
                out += c;
        }
        if (i >= s.size())
            throw std::runtime_error("unterminated string");
        i++;
        return out;
    }
    double number()
    {
        ws();
        char        *end = nullptr;
        const double v   = strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i)
            throw std::runtime_error("expected a number at offset " + std::to_string(i));
        i = (size_t)(end - s.c_str());
        return v;
    }
    void skip_value()
    {
        const char c = peek();
        if (c == '"')
            (void)string();
        else if (c == '{') {
            i++;
            if (peek() == '}') {
                i++;
                return;
            }
            do {
                (void)string();
                expect(':');
                skip_value();
            } while (peek() == ',' && ++i);
            expect('}');
        }
        else if (c == '[') {
            i++;
            if (peek() == ']') {
                i++;
                return;
            }
            do
                skip_value();
            while (peek() == ',' && ++i);
            expect(']');
        }
        else if (c == 't' || c == 'f' || c == 'n')
            while (i < s.size() && s[i] >= 'a' && s[i] <= 'z')
                i++;
        else
            (void)number();
    }
};

std::vector<RequestSpec> parse_requests(const std::string &path)
{
    std::ifstream f(path);
    if (!f)
        throw std::runtime_error("Failed to open JSON file: " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string        text = ss.str();
    Json                     j(text);
    std::vector<RequestSpec> out;
    bool                     found = false;
    j.expect('{');
    if (j.peek() != '}')
        do {
            const std::string key = j.string();
            j.expect(':');
            if (key != "requests") {
                j.skip_value();
                continue;
            }
            found = true;
            j.expect('[');
            if (j.peek() != ']')
                do {
                    RequestSpec r;
                    j.expect('{');
                    if (j.peek() != '}')
                        do {
                            const std::string k = j.string();
                            j.expect(':');
                            if (k == "prompt")
                                r.prompt = j.string();
                            else if (k == "temperature")
                                r.temperature = (float)j.number();
                            else if (k == "top_p")
                                r.top_p = (float)j.number();
                            else if (k == "max_tokens")
                                r.max_tokens = (int)j.number();
                            else
                                j.skip_value();
                        } while (j.peek() == ',' && ++j.i);
                    j.expect('}');
                    if (r.prompt.empty()) // json_parser.hpp:318-320
                        throw std::runtime_error("Request " + std::to_string(out.size()) + " has empty prompt");
                    out.push_back(r);
                } while (j.peek() == ',' && ++j.i);
            j.expect(']');
        } while (j.peek() == ',' && ++j.i);
    j.expect('}');
    if (!found)
        throw std::runtime_error("no \"requests\" array in " + path);
    return out;
}

bool exists(const std::string &p, bool *is_dir = nullptr)
{
    struct stat st;
    if (stat(p.c_str(), &st) != 0)
        return false;
    if (is_dir)
        *is_dir = S_ISDIR(st.st_mode);
    return true;
}

// include/utils/path.hpp:14-62
void resolve_paths(const std::string &input, std::string *model, std::string *tokenizer)
{
    bool dir = false;
    if (!exists(input, &dir))
        throw std::runtime_error("Path does not exist: " + input);
    if (dir) {
        *model     = input + "/model.bin";
        *tokenizer = input + "/tokenizer.bin";
        if (!exists(*model))
            throw std::runtime_error("model.bin not found in: " + input);
        if (!exists(*tokenizer))
            throw std::runtime_error("tokenizer.bin not found in: " + input);
        return;
    }
    *model             = input;
    const size_t slash = input.find_last_of('/');
    *tokenizer         = (slash == std::string::npos ? std::string("") : input.substr(0, slash + 1)) + "tokenizer.bin";
    if (!exists(*tokenizer))
        *tokenizer = "tokenizer.bin";
}

#define NVK_OR_DIE(call)                                                                                               \
    do {                                                                                                               \
        if ((call) < 0) {                                                                                              \
            fprintf(stderr, "[ERROR] %s\n", nvk_last_error());                                                         \
            return 1;                                                                                                  \
        }                                                                                                              \
    } while (0)

std::string piece_of(const nvk_tokenizer *tok, int id)
{
    char      buf[512];
    const int n = nvk_tokenizer_decode(tok, id, buf, (int)sizeof buf);
    return std::string(buf, (size_t)std::max(0, std::min(n, (int)sizeof buf - 1)));
}

std::vector<int> encode(const nvk_tokenizer *tok, const std::string &text)
{
    std::vector<int> ids(text.size() + 8);
    const int        n = nvk_tokenizer_encode(tok, text.data(), (int)text.size(), 1, 0, ids.data(), (int)ids.size());
    ids.resize((size_t)std::max(n, 0));
    return ids;
}

void usage()
{
    fprintf(stderr, "usage: nvk_run <model dir | model.bin> (-i PROMPT | --input-json FILE) [-n STEPS] "
                    "[-b MAX_BATCH] [-t TEMPERATURE (0, or any with --greedy)] [-p TOP_P] [--greedy] [--block-size N] [--num-blocks N] [--device N]\n"
                    "               [--gpus N | --devices a,b,...]   sequence-sharded over several GPUs: a model per device, -b requests each,\n"
                    "               a request placed where the fewest tokens are promised; nothing exchanged between devices\n");
}

} // namespace

int main(int argc, char **argv)
{
    std::string path, prompt, input_json;
    int         max_batch = 1, steps = 256, block_size = 16, num_blocks = 0, device = 0; // main.cpp:19-26, model.hpp:35-36
    std::vector<int> devices; // --gpus N / --devices: the members of the sequence-sharded group (empty: one model on --device)
    float       temperature = 1.0f, top_p = 0.9f;
    bool        greedy = false; // --greedy: decode greedily whatever temperature a request asks for
    for (int a = 1; a < argc; a++) {
        const std::string k    = argv[a];
        auto              next = [&]() -> const char * {
            if (a + 1 >= argc) {
                usage();
                exit(1);
            }
            return argv[++a];
        };
        if (k == "-i" || k == "--prompt")
            prompt = next();
        else if (k == "--input-json")
            input_json = next();
        else if (k == "-b" || k == "--max-batch-size")
            max_batch = atoi(next());
        else if (k == "-t" || k == "--temperature")
            temperature = (float)atof(next());
        else if (k == "-p" || k == "--top-p")
            top_p = (float)atof(next());
        else if (k == "-n" || k == "--steps")
            steps = atoi(next());
        else if (k == "--greedy")
            greedy = true;
        else if (k == "--block-size")
            block_size = atoi(next());
        else if (k == "--num-blocks")
            num_blocks = atoi(next());
        else if (k == "--device")
            device = atoi(next());
        else if (k == "--gpus") {
            const int n = atoi(next());
            devices.clear();
            for (int d = 0; d < n; d++)
                devices.push_back(d);
        }
        else if (k == "--devices") { // explicit list; a device may appear twice (two members on one GPU)
            devices.clear();
            const std::string list = next();
            for (size_t at = 0; at <= list.size();) {
                const size_t comma = std::min(list.find(',', at), list.size());
                if (comma > at)
                    devices.push_back(atoi(list.substr(at, comma - at).c_str()));
                at = comma + 1;
            }
        }
        else if (k == "--without-paged-attn") {
            fprintf(stderr, "[ERROR] --without-paged-attn: this build has only the paged path\n");
            return 1;
        }
        else if (!k.empty() && k[0] == '-') {
            usage();
            return 1;
        }
        else
            path = k;
    }
    (void)top_p;
    if (path.empty() || (prompt.empty() == input_json.empty())) {
        fprintf(stderr, "[ERROR] Either --prompt or --input-json must be provided (and not both)\n"); // main.cpp:47-57
        usage();
        return 1;
    }
    if (max_batch < 1)
        max_batch = 1;

    std::string model_path, tokenizer_path;
    std::vector<RequestSpec> requests;
    try {
        resolve_paths(path, &model_path, &tokenizer_path);
        if (!input_json.empty())
            requests = parse_requests(input_json);
        else {
            RequestSpec r;
            r.prompt = prompt, r.temperature = temperature, r.max_tokens = steps;
            requests.push_back(r);
        }
    }
    catch (const std::exception &e) {
        fprintf(stderr, "[ERROR] %s\n", e.what());
        return 1;
    }
    // The reference samples by default (-t 1.0, -p 0.9; sampler.hpp:31-80, seeded from the clock).  This build decodes
    // greedily only, and never silently: a request with a temperature is an error unless --greedy says it is meant.
    for (const RequestSpec &r : requests)
        if (r.temperature != 0.0f) {
            if (!greedy) {
                fprintf(stderr,
                        "[ERROR] temperature %.2f asked: this build has no sampler (greedy decoding only).  Pass -t 0, or "
                        "--greedy to decode every request greedily whatever it asks for\n",
                        r.temperature);
                return 2;
            }
            fprintf(stderr, "[WARN] temperature %.2f asked; --greedy: decoding with temperature 0\n", r.temperature);
            break;
        }

    // the header first, for the pool size: room for every running request's whole context
    int hdr[7] = {0};
    {
        FILE *f = fopen(model_path.c_str(), "rb");
        if (!f || fread(hdr, sizeof(int), 7, f) != 7) {
            fprintf(stderr, "[ERROR] Failed to open model file: %s\n", model_path.c_str());
            return 1;
        }
        fclose(f);
    }
    if (num_blocks <= 0)
        num_blocks = std::max(256, max_batch * ((hdr[6] + block_size - 1) / block_size)); // Config::num_blocks = 256
    // One code path for one device and for several: a group of one member is a model with a thread of its own.
    if (devices.empty())
        devices.push_back(device);
    nvk_model_group *model = nullptr;
    nvk_tokenizer   *tok   = nullptr;
    NVK_OR_DIE(nvk_model_group_load(model_path.c_str(), block_size, num_blocks, max_batch, (int)devices.size(),
                                    devices.data(), &model));
    int cfg[8];
    NVK_OR_DIE(nvk_model_group_config(model, cfg));
    NVK_OR_DIE(nvk_tokenizer_load(tokenizer_path.c_str(), cfg[5], &tok));
    fprintf(stderr, "[INFO] %s: dim=%d layers=%d heads=%d/%d vocab=%d max_seq_len=%d; PagedAttention block_size=%d "
                    "num_blocks=%d; %s\n", model_path.c_str(), cfg[0], cfg[2], cfg[3], cfg[4], cfg[5], cfg[6], block_size,
            num_blocks, nvk_version());
    if (devices.size() > 1) {
        std::string list;
        for (size_t i = 0; i < devices.size(); i++)
            list += (i ? "," : "") + std::to_string(devices[i]);
        fprintf(stderr, "[INFO] sequence-sharded over %zu members (devices %s), up to %d requests each, no collective\n",
                devices.size(), list.c_str(), max_batch);
    }
    const int capacity = max_batch * (int)devices.size(); // running requests in all

    struct Running
    {
        int              id;
        std::vector<int> prompt_ids;
        int              last = 0, generated = 0;
        bool             prefilled = false;
        std::string      text;
    };
    std::vector<Running> running;
    size_t               next_request = 0;
    long long            total_generated = 0, total_prompt = 0, steps_run = 0;
    const bool           single = input_json.empty();
    if (single)
        fputc('\n', stdout), fwrite(prompt.data(), 1, prompt.size(), stdout), fflush(stdout); // runner.hpp:37
    const auto t0 = std::chrono::high_resolution_clock::now();
    while (!running.empty() || next_request < requests.size()) {
        // admission (Scheduler::schedule: fill up to max_batch_size, scheduler.hpp:66-103)
        while ((int)running.size() < capacity && next_request < requests.size()) {
            Running r;
            r.id         = (int)next_request;
            r.prompt_ids = encode(tok, requests[next_request].prompt);
            if ((int)r.prompt_ids.size() >= cfg[6]) // keep room for at least one generated token
                r.prompt_ids.resize((size_t)cfg[6] - 1);
            // placed by promised tokens (prompt + max_tokens): the member with the least takes it
            const int member = nvk_model_group_add_request(model, r.id, (int)r.prompt_ids.size() + requests[next_request].max_tokens);
            if (member < 0)
                NVK_OR_DIE(member);
            total_prompt += (long long)r.prompt_ids.size();
            running.push_back(std::move(r));
            next_request++;
        }
        // one model step: decoding requests first (one token each), then the new ones with their whole prompt
        std::vector<int> ids, lens, toks, order;
        for (int pass = 0; pass < 2; pass++)
            for (size_t i = 0; i < running.size(); i++) {
                Running &r = running[i];
                if (r.prefilled != (pass == 0))
                    continue;
                ids.push_back(r.id);
                order.push_back((int)i);
                if (r.prefilled) {
                    lens.push_back(1);
                    toks.push_back(r.last);
                }
                else {
                    lens.push_back((int)r.prompt_ids.size());
                    toks.insert(toks.end(), r.prompt_ids.begin(), r.prompt_ids.end());
                }
            }
        std::vector<int> next(ids.size());
        NVK_OR_DIE(nvk_model_group_step_chunks(model, ids.data(), lens.data(), toks.data(), (int)ids.size(), next.data(), nullptr));
        steps_run++;
        std::vector<int> done;
        for (size_t k = 0; k < ids.size(); k++) {
            Running &r  = running[(size_t)order[k]];
            r.prefilled = true;
            r.last      = next[k];
            r.generated++;
            total_generated++;
            const std::string piece = piece_of(tok, next[k]);
            r.text += piece;
            if (single)
                fwrite(piece.data(), 1, piece.size(), stdout), fflush(stdout); // (as std::cout << std::string, runner.hpp:51:
                                                                               //  a piece may be the byte 0x00, which a C string ends at)
            const int len = nvk_model_group_request_len(model, r.id);
            // request_processor.hpp:55-77: max_tokens, max_seq_len, EOS (2)
            if (r.generated >= requests[(size_t)r.id].max_tokens || len >= cfg[6] || (!single && next[k] == 2)) // runner.hpp:48-57 has no EOS stop
                done.push_back(order[k]);
        }
        std::sort(done.rbegin(), done.rend());
        for (int i : done) {
            Running &r = running[(size_t)i];
            if (!single)
            {
                const std::string &prompt = requests[(size_t)r.id].prompt;
                printf("[request %d] ", r.id);
                fwrite(prompt.data(), 1, prompt.size(), stdout);
                fwrite(r.text.data(), 1, r.text.size(), stdout); // (whole: generated bytes include 0x00)
                fputc('\n', stdout);
            }
            NVK_OR_DIE(nvk_model_group_finish_request(model, r.id));
            running.erase(running.begin() + i);
        }
    }
    const double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    if (single)
        printf("\n");
    fprintf(stderr, "[INFO] %zu request(s), %lld prompt tokens, %lld generated tokens in %lld model steps, %.3f s: "
                    "%.1f tokens/s\n", requests.size(), total_prompt, total_generated, steps_run, sec,
            sec > 0 ? total_generated / sec : 0.0);
    nvk_tokenizer_destroy(tok);
    nvk_model_group_destroy(model);
    return 0;
}
